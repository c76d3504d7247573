"""CPU: the oracle's restatement of the selective reader's ChunkedDecoder bulk
entry points against the contract computed from the plain columns."""
import numpy as np

from nimble_amd import wire as W

import chunked_contract as CC
from util import unpack_bits


def test_decode_nullable_places_values_and_nulls(oracle):
    rng = np.random.default_rng(31)
    for dtype in (W.INT16, W.UINT32, W.INT64):
        for _ in range(10):
            stream, values, nn = CC.value_stream(rng, dtype)
            reads = CC.script(rng, values.size)
            got = CC.run_decode_nullable(oracle.ChunkedDecoder(stream), dtype, reads)
            for (nulls, data), (valid, slots, vals) in zip(got, CC.expected_decode_nullable(values, nn, reads)):
                assert np.array_equal(unpack_bits(nulls, valid.size), valid)
                assert np.array_equal(data[slots], vals)


def test_next_bools_and_indices(oracle):
    rng = np.random.default_rng(32)
    for _ in range(10):
        stream, bools = CC.bool_stream(rng)
        reads = CC.script(rng, bools.size)
        pos = 0
        for (bits, _), (count, mask) in zip(CC.run_next_bools(oracle.ChunkedDecoder(stream), reads), reads):
            m = np.ones(count, dtype=bool) if mask is None else mask
            want = np.zeros(count, dtype=np.uint8)
            want[m] = bools[pos:pos + int(m.sum())]
            pos += int(m.sum())
            assert np.array_equal(bits, want)
        stream, values, _ = CC.value_stream(rng, W.UINT32, nullable_chunks=False)
        reads = CC.script(rng, values.size)
        pos = 0
        for data, (count, mask) in zip(CC.run_next_indices(oracle.ChunkedDecoder(stream), reads), reads):
            m = np.ones(count, dtype=bool) if mask is None else mask
            assert np.array_equal(data.view(np.uint32)[m], values[pos:pos + int(m.sum())])
            pos += int(m.sum())


def test_lengths_with_nulls_and_skip(oracle):
    """Array / map length streams: nextBools yields the null bitmap and keeps the
    values (getPreloadedValues); skip drops whole chunks and lands mid-chunk."""
    rng = np.random.default_rng(33)
    for _ in range(10):
        stream, values, nn = CC.value_stream(rng, W.UINT32)
        reads = CC.script(rng, values.size)
        got = CC.run_next_bools(oracle.ChunkedDecoder(stream, decode_values_with_nulls=True), reads)
        for (bits, pre), (valid, slots, vals) in zip(got, CC.expected_decode_nullable(values, nn, reads)):
            assert np.array_equal(bits.astype(bool), valid)
            assert np.array_equal(pre[slots], vals)
        skip = int(rng.integers(0, values.size))
        d = oracle.ChunkedDecoder(stream)
        d.skip_values(skip)
        rest = values.size - skip
        nulls = np.zeros(rest // 8 + 16, dtype=np.uint8)
        data = np.zeros(rest, dtype=np.uint32)
        d.decode_nullable(nulls, data, rest)
        assert np.array_equal(unpack_bits(nulls, rest), nn[skip:])
        assert np.array_equal(data[nn[skip:]], values[skip:][nn[skip:]])
