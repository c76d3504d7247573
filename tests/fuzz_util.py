"""Random cascades for the fuzz tests: a random spec tree (every encoding may
nest every other one that is legal for the child's type, as the reference's
selection policies allow: encodings/selection/EncodingSelectionPolicy.h:251-261)
over random data of a random shape."""
from __future__ import annotations

import numpy as np

from nimble_amd import encoder as E
from nimble_amd import wire as W

from util import NP_OF, random_strings

NUMERIC = (W.INT8, W.UINT8, W.INT16, W.UINT16, W.INT32, W.UINT32, W.INT64, W.UINT64, W.FLOAT, W.DOUBLE)


def random_numeric_spec(rng, dtype: int, depth: int):
    """Spec for a numeric column; children recurse with depth - 1."""
    es = W.elem_bytes(dtype)
    leaves = ["trivial", "fbw"] + (["varint"] if es >= 4 else [])
    if depth <= 0:
        kind = leaves[int(rng.integers(0, len(leaves)))]
    else:
        inner = ["rle", "dictionary", "mainly_constant", "rle", "dictionary", "block"] + (["for"] if dtype in W.INTEGER_TYPES else [])
        kind = (leaves + inner)[int(rng.integers(0, len(leaves) + len(inner)))]
    if kind == "trivial":
        return E.Trivial()
    if kind == "fbw":
        return E.FixedBitWidth(exact_bits=bool(rng.integers(0, 2)))
    if kind == "varint":
        return E.Varint()
    if kind == "rle":
        return E.RLE(lengths=random_numeric_spec(rng, W.UINT32, depth - 1), values=random_numeric_spec(rng, dtype, depth - 1))
    if kind == "dictionary":
        return E.Dictionary(alphabet=random_numeric_spec(rng, dtype, depth - 1), indices=random_numeric_spec(rng, W.UINT32, depth - 1))
    if kind in ("block", "for"):  # per-segment arrays: same-width unsigned baselines, uint8 widths, uint32 / uint64 offsets
        unsigned = {1: W.UINT8, 2: W.UINT16, 4: W.UINT32, 8: W.UINT64}[es]
        size = int(rng.integers(1, 300))
        first = int(rng.integers(1, size + 1)) if rng.random() < 0.3 else None
        meta = [random_numeric_spec(rng, t, depth - 1) for t in (unsigned, W.UINT8, W.UINT32 if kind == "block" else W.UINT64)]
        if kind == "block":
            return E.BlockBitPacking(baselines=meta[0], bit_widths=meta[1], offsets=meta[2], block_size=size, first_block_rows=first)
        return E.FOR(references=meta[0], bit_widths=meta[1], bit_offsets=meta[2], frame_size=size, first_frame_rows=first,
                     exact_bits=bool(rng.integers(0, 2)))
    return E.MainlyConstant(is_common=random_bool_spec(rng, depth - 1), other=random_numeric_spec(rng, dtype, depth - 1))


def random_bool_spec(rng, depth: int):
    kind = ("trivial", "sparse", "rle")[int(rng.integers(0, 3))] if depth > 0 else "trivial"
    if kind == "trivial":
        return E.Trivial()
    if kind == "sparse":
        return E.SparseBool(indices=random_numeric_spec(rng, W.UINT32, depth - 1))
    return E.RLE(lengths=random_numeric_spec(rng, W.UINT32, depth - 1))


def random_string_spec(rng, depth: int):
    kind = ("trivial", "rle", "dictionary", "mainly_constant")[int(rng.integers(0, 4))] if depth > 0 else "trivial"
    if kind == "trivial":
        return E.Trivial(lengths=random_numeric_spec(rng, W.UINT32, depth - 1))
    if kind == "rle":
        return E.RLE(lengths=random_numeric_spec(rng, W.UINT32, depth - 1), values=random_string_spec(rng, depth - 1))
    if kind == "dictionary":
        return E.Dictionary(alphabet=random_string_spec(rng, depth - 1), indices=random_numeric_spec(rng, W.UINT32, depth - 1))
    return E.MainlyConstant(is_common=random_bool_spec(rng, depth - 1), other=random_string_spec(rng, depth - 1))


def random_numeric_data(rng, dtype: int, n: int) -> np.ndarray:
    """Shapes that exercise every encoding: runs, few distinct values, one dominant value, full range."""
    npt = NP_OF[dtype]
    bits = 8 * np.dtype(npt).itemsize
    shape = int(rng.integers(0, 5))
    if shape == 0:  # full range bit patterns (floats: arbitrary payloads incl. NaNs)
        raw = rng.integers(0, 2**63, n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, n, dtype=np.uint64)
        return raw.astype(W.physical_dtype(dtype)).view(npt)
    card = int(rng.integers(1, 40))
    pool = rng.integers(0, 1 << min(bits - 1, int(rng.integers(1, 40))), card, dtype=np.uint64).astype(W.physical_dtype(dtype))
    if shape == 1:  # runs
        lens = rng.integers(1, int(rng.integers(2, 200)), n + 1)
        v = np.repeat(pool[rng.integers(0, card, n + 1)], lens)[:n]
    elif shape == 2:  # one dominant value
        v = np.where(rng.random(n) < 0.9, pool[0], pool[rng.integers(0, card, n)])
    elif shape == 3:  # low cardinality, no order
        v = pool[rng.integers(0, card, n)]
    else:  # small magnitudes near a large baseline
        base = np.uint64(rng.integers(0, 1 << (bits - 2)))
        v = (rng.integers(0, 1 << int(rng.integers(1, 20)), n, dtype=np.uint64) + base).astype(W.physical_dtype(dtype))
    return np.ascontiguousarray(v).view(npt)


def random_bool_data(rng, n: int) -> np.ndarray:
    shape = int(rng.integers(0, 3))
    if shape == 0:
        return rng.random(n) < rng.random()
    if shape == 1:
        return rng.random(n) < 0.02
    lens = rng.integers(1, 300, n + 1)
    return np.repeat(np.arange(n + 1) & 1, lens)[:n].astype(bool)


def random_string_data(rng, n: int) -> E.StringColumn:
    pool = random_strings(rng, int(rng.integers(1, 30)), max_len=int(rng.integers(1, 40)))
    shape = int(rng.integers(0, 3))
    if shape == 0:
        return random_strings(rng, n, max_len=int(rng.integers(1, 60)))
    if shape == 1:
        idx = np.repeat(rng.integers(0, len(pool), n + 1), rng.integers(1, 50, n + 1))[:n]
    else:
        idx = np.where(rng.random(n) < 0.85, 0, rng.integers(0, len(pool), n))
    return pool.take(idx)


def random_chunk(rng, max_rows: int = 5000, max_depth: int = 3):
    """(label, encoding bytes).  Roots may be wrapped in Nullable.  Draws whose
    nesting does not fit the data (a MainlyConstant over an empty child column)
    are redrawn."""
    while True:
        try:
            return _random_chunk(rng, max_rows, max_depth)
        except AssertionError:
            continue


def _random_chunk(rng, max_rows: int, max_depth: int):
    family = int(rng.integers(0, 10))
    n = int(rng.integers(1, max_rows + 1))
    depth = int(rng.integers(0, max_depth + 1))
    nullable = depth > 0 and rng.random() < 0.35
    mask = None
    if nullable:
        mask = rng.random(n) < rng.random()
        if not mask.any():
            mask[int(rng.integers(0, n))] = True
        nn = int(mask.sum())
        nulls_spec = random_bool_spec(rng, depth - 1)
        depth -= 1
    else:
        nn = n
    if family < 7:
        dtype = NUMERIC[int(rng.integers(0, len(NUMERIC)))]
        spec, data = random_numeric_spec(rng, dtype, depth), random_numeric_data(rng, dtype, nn)
    elif family < 8:
        dtype = W.BOOL
        spec, data = random_bool_spec(rng, depth), random_bool_data(rng, nn)
    else:
        dtype = W.STRING
        spec, data = random_string_spec(rng, depth), random_string_data(rng, nn)
    if nullable:
        enc = E.encode_nullable(E.Nullable(data=spec, nulls=nulls_spec), data, mask, dtype)
        return f"nullable/{dtype}/{n}/{spec!r}", enc
    return f"{dtype}/{n}/{spec!r}", E.encode(spec, data, dtype)
