"""GPU parity of dictionaries of 8-byte rows around the sizes where the way the
alphabet is held changes (DictionaryEncoding::materialize,
encodings/DictionaryEncoding.h:193-236): up to 2,048 entries it is staged by
the fused launch, 2,049 .. 4,096 entries run on the kernel that keeps the
alphabet in LDS (one workgroup per eight 2,048-row tiles), more are gathered
from memory -- except alphabets of strings no longer than 255 bytes under
bit-packed indices of up to 16 bits, which from 2,049 to 65,536 entries run
on the kernel that keeps their views in LDS in compact form.  Numbers and strings; alone, under a Nullable (bitmap and
SparseBool nulls) and under a MainlyConstant.  check_parity also decodes every
batch with the LDS kernel switched off."""
import struct

import numpy as np
import pytest

from nimble_amd import capi
from nimble_amd import encoder as E
from nimble_amd import wire as W
from nimble_amd import workloads as WL
from nimble_amd.decoder import BatchDecoder, Plan

from util import check_parity

pytestmark = pytest.mark.gpu

SIZES = (2048, 2049, 4096, 4097)
ROWS = (1, 2047, 16_384, 16_385, 50_001)  # one group of eight tiles, one row more, several groups


def _names(enc):
    p = Plan.from_encodings([enc], per_kernel=True, lds_dictionary_min_tiles=1)
    out = [l[0] for l in p.launches()]
    p.close()
    return out


def _dict_num(rng, A, n, dt, npt, indices=None):
    alpha = (rng.permutation(1 << 22)[:A].astype(np.uint64) << np.uint64(21)).astype(npt)
    idx = rng.integers(0, A, n) if indices is None else indices
    return E.encode_dictionary_from_indices(E.Dictionary(alphabet=E.Trivial(), indices=E.FixedBitWidth()), alpha, idx, dt)


def _alphabet(rng, A):
    """Exactly A distinct strings (random_alphabet drops repeats of the short ones)."""
    items = WL.random_alphabet(rng, A + A // 4 + 300).to_list()
    assert len(items) >= A
    return E.StringColumn.from_list(items[:A])


def _dict_str(rng, alpha, n, index_spec=None):
    """n rows over `alpha`; every entry is used when n allows it (the encoder drops unused ones)."""
    spec = E.Dictionary(alphabet=E.Trivial(lengths=E.FixedBitWidth()), indices=index_spec or E.FixedBitWidth())
    idx = rng.integers(0, len(alpha), n)
    if n >= len(alpha):
        idx[rng.permutation(n)[: len(alpha)]] = np.arange(len(alpha))
    return E.encode_dictionary_from_indices(spec, alpha, idx, W.STRING)


def test_numeric_dictionaries_at_the_size_boundaries(gpu, oracle):
    rng = np.random.default_rng(71)
    encs, labels = [], []
    for A in SIZES:
        for n in ROWS:
            for dt, npt in ((W.INT64, np.int64), (W.DOUBLE, np.float64)):
                encs.append(_dict_num(rng, A, n, dt, npt))
                labels.append(f"{W.DATA_TYPE_NAMES[dt]} alphabet {A} rows {n}")
    assert "k_dict_lds<8>" in _names(encs[labels.index("int64 alphabet 2049 rows 50001")])
    assert "k_dict_lds<8>" in _names(encs[labels.index("int64 alphabet 4096 rows 50001")])
    assert "k_dict_lds<8>" not in _names(encs[labels.index("int64 alphabet 2048 rows 50001")])
    assert "k_dict_lds<8>" not in _names(encs[labels.index("int64 alphabet 4097 rows 50001")])
    check_parity(oracle, encs, labels=labels)


def test_string_dictionaries_at_the_size_boundaries(gpu, oracle):
    rng = np.random.default_rng(72)
    encs, labels = [], []
    for A in SIZES:
        alpha = _alphabet(rng, A)
        for n in ROWS:
            encs.append(_dict_str(rng, alpha, n))
            labels.append(f"strings alphabet {A} rows {n}")
        encs.append(_dict_str(rng, alpha, 30_000, E.Trivial()))  # indices as plain uint32
        labels.append(f"strings alphabet {A}, raw indices")
    # short strings with bit-packed indices take the compact-views kernel from 2,049 entries on (below);
    # raw uint32 indices keep these columns on the kernel that holds the 8-byte views
    assert "k_dict_str<8>" in _names(encs[labels.index("strings alphabet 4096 rows 50001")])
    assert "k_dict_lds<8>" in _names(encs[labels.index("strings alphabet 4096, raw indices")])
    assert "k_dict_str<8>" not in _names(encs[labels.index("strings alphabet 2048 rows 50001")])
    check_parity(oracle, encs, labels=labels)


def test_dictionaries_gathered_inside_a_rank_scatter(gpu, oracle):
    """Nullable{Dictionary} and MainlyConstant{.., Dictionary}: the selected rows'
    indices are staged per tile, every density from nothing selected to all."""
    rng = np.random.default_rng(73)
    encs, labels = [], []
    for A in SIZES:
        alpha = _alphabet(rng, A)
        for n in (5, 16_384, 16_385, 70_000):
            for frac, nulls in ((0.0, E.Trivial()), (0.05, E.SparseBool()), (0.5, E.Trivial()), (0.95, E.SparseBool()), (1.0, E.Trivial())):
                mask = rng.random(n) < frac
                nn = int(mask.sum())
                data = _dict_str(rng, alpha, nn)
                encs.append(E.nullable_wrap(data, mask, nulls, W.STRING))
                labels.append(f"nullable strings alphabet {A} rows {n} non-null {frac}")
                data = _dict_num(rng, A, nn, W.INT64, np.int64)
                encs.append(E.nullable_wrap(data, mask, nulls, W.INT64))
                labels.append(f"nullable int64 alphabet {A} rows {n} non-null {frac}")
    names = _names(encs[labels.index("nullable strings alphabet 4096 rows 70000 non-null 0.5")])
    assert "k_dict_str<8>" in names and not any(k.startswith("k_select_scatter") for k in names)
    names = _names(encs[labels.index("nullable int64 alphabet 4096 rows 70000 non-null 0.5")])
    assert "k_dict_lds<8>" in names and not any(k.startswith("k_select_scatter") for k in names)
    # MainlyConstant whose other values are a dictionary
    n = 60_000
    is_common = rng.random(n) < 0.7
    others = int((~is_common).sum())
    alpha = (rng.permutation(1 << 20)[:3000].astype(np.int64) + 1000) << 13
    other_enc_vals = alpha[rng.integers(0, 3000, others)]
    spec = E.MainlyConstant(is_common=E.Trivial(), other=E.Dictionary(alphabet=E.Trivial(), indices=E.FixedBitWidth()))
    encs.append(E.mainly_constant_from_parts(spec, is_common, other_enc_vals, np.int64(7), W.INT64))
    labels.append("mainly constant over a 3000-entry dictionary")
    check_parity(oracle, encs, labels=labels)


def test_index_out_of_range_is_reported_by_the_lds_kernel(gpu):
    rng = np.random.default_rng(74)
    A, n = 3000, 40_000
    idx = rng.integers(0, A, n).astype(np.uint32)
    idx[[7, 20_000, n - 1]] = (A, 4_000_000_000, A + 5)
    alpha = E.encode(E.Trivial(), np.arange(A, dtype=np.uint64), W.INT64)
    ind = E.encode(E.Trivial(), idx, W.UINT32)
    enc = E.prefix(W.DICTIONARY, W.INT64, n) + struct.pack("<I", len(alpha)) + alpha + ind
    assert "k_dict_lds<8>" in _names(enc)
    mask = np.ones(n, dtype=bool)
    mask[::3] = False
    nn = int(mask.sum())
    inner = E.prefix(W.DICTIONARY, W.INT64, nn) + struct.pack("<I", len(alpha)) + alpha + E.encode(E.Trivial(), idx[:nn], W.UINT32)
    under = E.nullable_wrap(inner, mask, E.Trivial(), W.INT64)
    for e in (enc, under):
        plan = Plan.from_encodings([e], lds_dictionary_min_tiles=1)
        dec = BatchDecoder(plan)
        dec.stage()
        dec.decode()
        with pytest.raises(capi.NimbleError, match="dictionary index") as ei:
            dec.status()
        assert ei.value.status == capi.NBL_ERR_DEVICE_DATA
        dec.fetch()
        plan.close()


# ---- string alphabets of 2,049 .. 65,536 short entries, bit-packed indices: the compact-views kernel (k_dict_str)

STR_SIZES = (2049, 4097, 5000, 65_536)
STR_ROWS = (1, 2047, 65_536, 65_537, 150_001)  # one group of 32 tiles, one row more, several groups


def test_string_dictionaries_with_compact_views(gpu, oracle):
    rng = np.random.default_rng(75)
    encs, labels = [], []
    for A in STR_SIZES + (65_537,):
        alpha = _alphabet(rng, A)
        for n in STR_ROWS:
            encs.append(_dict_str(rng, alpha, n))
            labels.append(f"strings alphabet {A} rows {n}")
        encs.append(_dict_str(rng, alpha, 70_000, E.Trivial()))  # indices as plain uint32
        labels.append(f"strings alphabet {A}, raw indices")
    for A in STR_SIZES:
        assert "k_dict_str<8>" in _names(encs[labels.index(f"strings alphabet {A} rows 150001")]), A
    assert "k_dict_str<8>" not in _names(encs[labels.index("strings alphabet 65537 rows 150001")])
    check_parity(oracle, encs, labels=labels)


def test_compact_views_at_the_length_limit(gpu, oracle):
    """Strings of up to 255 bytes keep a block of 256 entries within 16 bits; one
    longer string and the column is gathered from memory."""
    rng = np.random.default_rng(76)
    A, n = 4400, 70_000
    def column(lengths):  # distinct strings of exactly these lengths (4 bytes or more, or empty)
        items = [bytes(rng.integers(0, 256, int(l) - 4, dtype=np.uint8)) + struct.pack("<I", i) if l else b"" for i, l in enumerate(lengths)]
        return E.StringColumn.from_list(items)
    full = column(np.r_[np.full(A - 1, 255), 0])            # 255 bytes each: blocks of 256 entries span 65,280 bytes
    mixed = column(np.r_[0, rng.integers(4, 256, A - 1)])
    too_long = column(np.r_[0, rng.integers(4, 50, A - 2), 256])
    encs = [_dict_str(rng, c, n) for c in (full, mixed, too_long)]
    labels = ["entries of 255 bytes", "entries of 0 .. 255 bytes", "one entry of 256 bytes"]
    assert "k_dict_str<8>" in _names(encs[0]) and "k_dict_str<8>" in _names(encs[1])
    assert "k_dict_str<8>" not in _names(encs[2])
    mask = rng.random(n) < 0.6
    encs.append(E.nullable_wrap(_dict_str(rng, full, int(mask.sum())), mask, E.Trivial(), W.STRING))
    labels.append("nullable, entries of 255 bytes")
    check_parity(oracle, encs, labels=labels)


def test_compact_views_gathered_inside_a_rank_scatter(gpu, oracle):
    rng = np.random.default_rng(77)
    encs, labels = [], []
    for A in (4097, 65_536):
        alpha = _alphabet(rng, A)
        for n in (5, 65_536, 65_537, 140_000):
            for frac, nulls in ((0.0, E.Trivial()), (0.05, E.SparseBool()), (0.5, E.Trivial()), (0.95, E.SparseBool()), (1.0, E.Trivial())):
                mask = rng.random(n) < frac
                encs.append(E.nullable_wrap(_dict_str(rng, alpha, int(mask.sum())), mask, nulls, W.STRING))
                labels.append(f"nullable strings alphabet {A} rows {n} non-null {frac}")
    names = _names(encs[labels.index("nullable strings alphabet 65536 rows 140000 non-null 0.5")])
    assert "k_dict_str<8>" in names and not any(k.startswith("k_select_scatter") for k in names)
    # MainlyConstant<string> whose other values are such a dictionary: the selection is read inverted, the fill is a view
    n = 90_000
    alpha = _alphabet(rng, 3000).to_list()
    common = b"the common one"
    pick = rng.integers(0, 3000, n)
    items = [common if c else alpha[i] for c, i in zip(rng.random(n) < 0.7, pick)]
    spec = E.MainlyConstant(is_common=E.Trivial(), other=E.Dictionary(alphabet=E.Trivial(lengths=E.FixedBitWidth()), indices=E.FixedBitWidth()))
    encs.append(E.encode(spec, E.StringColumn.from_list(items), W.STRING))
    labels.append("mainly constant strings over a 3000-entry dictionary")
    assert "k_dict_str<8>" in _names(encs[-1])
    check_parity(oracle, encs, labels=labels)


def test_index_out_of_range_is_reported_by_the_compact_views_kernel(gpu):
    rng = np.random.default_rng(78)
    A, n = 6000, 80_000
    alpha = _alphabet(rng, A)
    idx = rng.integers(0, A, n).astype(np.uint32)
    idx[[3, 40_000, n - 1]] = (A, 65_535, A + 9)  # the kernel takes bit-packed indices of up to 16 bits
    alpha_enc = E.encode(E.Trivial(lengths=E.FixedBitWidth()), alpha, W.STRING)
    enc = E.prefix(W.DICTIONARY, W.STRING, n) + struct.pack("<I", len(alpha_enc)) + alpha_enc + E.encode(E.FixedBitWidth(), idx, W.UINT32)
    assert "k_dict_str<8>" in _names(enc)
    plan = Plan.from_encodings([enc], lds_dictionary_min_tiles=1)
    dec = BatchDecoder(plan)
    dec.stage()
    dec.decode()
    with pytest.raises(capi.NimbleError, match="dictionary index") as ei:
        dec.status()
    assert ei.value.status == capi.NBL_ERR_DEVICE_DATA
    dec.fetch()
    plan.close()
