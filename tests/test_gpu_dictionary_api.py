"""GPU: dictionary pass-through.  With NBL_PLAN_KEEP_DICTIONARY the plan also
emits the alphabet and the raw indices of dictionary-encoded chunks, and the
C++ Encoding answers dictionaryEnabled / dictionarySize / dictionaryEntries /
materializeIndices as DictionaryEncoding and NullableEncoding do
(encodings/DictionaryEncoding.h:81-83,147-161, NullableEncoding.h:107-122)."""
import numpy as np
import pytest

from nimble_amd import encoder as E
from nimble_amd import wire as W
from nimble_amd.decoder import BatchDecoder, Plan

from util import phys, random_strings

pytestmark = pytest.mark.gpu


def _first_occurrence_alphabet(values):
    _, first = np.unique(values, return_index=True)
    return values[np.sort(first)]


@pytest.fixture(scope="module")
def dict_factory(gpu):
    import host_shim

    return host_shim.Factory(keep_dictionary=True)


@pytest.mark.parametrize("dtype", [W.INT8, W.INT16, W.INT32, W.INT64, W.DOUBLE])
@pytest.mark.parametrize("alphabet_spec", [E.Trivial(), E.FixedBitWidth()])
def test_numeric_dictionary(dict_factory, dtype, alphabet_spec):
    rng = np.random.default_rng(dtype)
    npt = W.physical_dtype(dtype)
    for n, card in ((1, 1), (500, 7), (70_000, 100)):
        pool = rng.integers(0, 120, card).astype(npt)
        values = pool[rng.integers(0, card, n)]
        e = dict_factory.create(E.encode(E.Dictionary(alphabet=alphabet_spec, indices=E.FixedBitWidth()), values, dtype))
        assert e.dictionary_enabled
        alpha = e.dictionary_entries()
        assert np.array_equal(alpha, _first_occurrence_alphabet(values))
        assert e.dictionary_size() == alpha.size
        # indices and values share one row cursor
        k = n // 3
        idx0 = e.materialize_indices(k)
        vals = e.materialize(n - k)
        assert np.array_equal(alpha[idx0], values[:k])
        assert np.array_equal(vals, values[k:])
        e.reset()
        assert np.array_equal(alpha[e.materialize_indices(n)], values)


def test_string_dictionary_and_nullable_wrapper(dict_factory):
    import host_shim

    rng = np.random.default_rng(3)
    pool = random_strings(rng, 40)
    n = 20_000
    col = pool.take(rng.integers(0, 40, n))
    e = dict_factory.create(E.encode(E.Dictionary(), col, W.STRING))
    assert e.dictionary_enabled
    lengths, blob = e.dictionary_entries()
    alpha = E.StringColumn(lengths, blob).to_list()
    idx = e.materialize_indices(n)
    assert [alpha[i] for i in idx[:2000]] == col.to_list()[:2000]
    assert len(alpha) == len(set(col.to_list()))
    # Nullable{Dictionary}: the alphabet is exposed, materializeIndices is not (as in the reference)
    mask = rng.random(n) < 0.7
    nn = int(mask.sum())
    vals = rng.integers(0, 9, nn).astype(np.uint32) * 11
    ne = dict_factory.create(E.encode_nullable(E.Nullable(data=E.Dictionary(), nulls=E.SparseBool()), vals, mask, W.INT32))
    assert ne.dictionary_enabled and np.array_equal(ne.dictionary_entries(), _first_occurrence_alphabet(vals))
    with pytest.raises(host_shim.HostError, match="materializeIndices on non-dictionary"):
        ne.materialize_indices(1)
    want = np.zeros(n, dtype=np.uint32)
    want[mask] = vals
    assert np.array_equal(ne.materialize(n), want)


def test_non_dictionary_chunks_and_default_factory(dict_factory, gpu):
    import host_shim

    v = np.arange(100, dtype=np.int32)
    e = dict_factory.create(E.encode(E.RLE(), v // 10, W.INT32))
    assert not e.dictionary_enabled
    with pytest.raises(host_shim.HostError, match="dictionarySize on non-dictionary"):
        e.dictionary_size()
    plain = host_shim.Factory().create(E.encode(E.Dictionary(), v % 7, W.INT32))  # dictionaries not kept by default
    assert not plain.dictionary_enabled
    assert np.array_equal(plain.materialize(100), phys(v % 7, W.INT32))


def test_layout_reports_dictionary_regions(gpu):
    v = (np.arange(5000) % 13).astype(np.int64)
    plan = Plan.from_encodings([E.encode(E.Dictionary(), v, W.INT64), E.encode(E.Trivial(), v, W.INT64)], keep_dictionary=True)
    lay = plan.layouts
    assert lay[0].has_dictionary == 1 and lay[0].dict_size == 13 and lay[0].dict_indices_rows == 5000
    assert lay[1].has_dictionary == 0
    dec = BatchDecoder(plan)
    dec.stage(); dec.decode(); dec.status()
    host = dec.out.cpu().numpy()
    alpha = host[lay[0].dict_alphabet_offset: lay[0].dict_alphabet_offset + 13 * 8].view(np.uint64)
    idx = host[lay[0].dict_indices_offset: lay[0].dict_indices_offset + 5000 * 4].view(np.uint32)
    assert np.array_equal(alpha[idx], v.view(np.uint64))
    plan.close()
