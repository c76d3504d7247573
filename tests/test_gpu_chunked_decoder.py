"""GPU: the host layer's ChunkedDecoder (selective reader interface over a
GPU-decoded stripe) leaves exactly the bytes the oracle's restatement of
velox/selective/ChunkedDecoder.{h,cpp} leaves, read for read."""
import numpy as np
import pytest

from nimble_amd import encoder as E
from nimble_amd import wire as W

import chunked_contract as CC
from util import random_strings, unpack_bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def factory(gpu):
    import host_shim

    return host_shim.Factory()


def test_bulk_entry_points_match_the_oracle(factory, oracle):
    import host_shim

    rng = np.random.default_rng(71)
    cases = []  # (kind, dtype, stream, total, with_nulls)
    for dtype in (W.INT16, W.UINT32, W.INT64, W.DOUBLE):
        for _ in range(4):
            stream, values, _ = CC.value_stream(rng, dtype)
            cases.append(("nullable", dtype, stream, values.size, False))
    for _ in range(6):
        stream, bools = CC.bool_stream(rng)
        cases.append(("bools", W.BOOL, stream, bools.size, False))
    for _ in range(4):
        stream, values, _ = CC.value_stream(rng, W.UINT32, nullable_chunks=False)
        cases.append(("indices", W.UINT32, stream, values.size, False))
    for _ in range(4):
        stream, values, _ = CC.value_stream(rng, W.UINT32)
        cases.append(("lengths", W.UINT32, stream, values.size, True))
    stripe = host_shim.StripeDecoder(factory, [c[2] for c in cases])  # every stream decoded in ONE GPU batch
    for i, (kind, dtype, stream, total, with_nulls) in enumerate(cases):
        reads = CC.script(rng, total)
        gpu_dec = stripe.chunked(i, with_nulls)
        cpu_dec = oracle.ChunkedDecoder(stream, decode_values_with_nulls=with_nulls)
        if kind == "nullable":
            for (gn, gd), (on, od) in zip(CC.run_decode_nullable(gpu_dec, dtype, reads), CC.run_decode_nullable(cpu_dec, dtype, reads)):
                # null slots hold whatever the dense-then-scatter order of the CPU decode left there: not part of the contract
                valid = unpack_bits(on, gd.size)
                assert np.array_equal(gn, on) and np.array_equal(gd[valid], od[valid]), (kind, dtype)
        elif kind == "indices":
            for g, o, (count, mask) in zip(CC.run_next_indices(gpu_dec, reads), CC.run_next_indices(cpu_dec, reads), reads):
                m = np.ones(count, dtype=bool) if mask is None else mask
                assert np.array_equal(g[m], o[m])
        else:
            for (gb, gp), (ob, op) in zip(CC.run_next_bools(gpu_dec, reads), CC.run_next_bools(cpu_dec, reads)):
                assert np.array_equal(gb, ob), kind
                if with_nulls:
                    valid = ob.astype(bool)
                    assert np.array_equal(gp[valid], op[valid])


def test_skip_and_chunk_inspection(factory, oracle):
    import host_shim

    rng = np.random.default_rng(72)
    stream, values, nn = CC.value_stream(rng, W.INT64, max_chunks=8)
    alpha = random_strings(rng, 11)
    strings = alpha.take(rng.integers(0, 11, 300))
    dict_stream = E.stream([E.encode(E.Dictionary(), strings, W.STRING), E.encode(E.Trivial(), strings, W.STRING)])
    stripe = host_shim.StripeDecoder(host_shim.Factory(keep_dictionary=True), [stream, dict_stream] + [stream] * 5)
    for k in range(5):
        skip = int(rng.integers(0, values.size))
        g, o = stripe.chunked(2 + k), oracle.ChunkedDecoder(stream)
        g.skip_values(skip)
        o.skip_values(skip)
        rest = values.size - skip
        outs = []
        for d in (g, o):
            nulls = np.zeros(rest // 8 + 16, dtype=np.uint8)
            data = np.zeros(rest, dtype=np.uint64)
            d.decode_nullable(nulls, data, rest)
            outs.append((nulls, data))
        valid = unpack_bits(outs[1][0], rest)
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1][valid], outs[1][1][valid])
        with pytest.raises(host_shim.HostError, match="chunk header"):
            g.skip_values(1)
            g.decode_nullable(np.zeros(16, np.uint8), np.zeros(1, np.uint64), 1)
    # currentEncoding / dictionaryConvertible / remainingValues as the string reader uses them
    d = stripe.chunked(1)
    assert d.state() == (0, -1, False)
    assert d.state(ensure_loaded=True) == (300, W.DICTIONARY, True)
    d.skip_values(300)  # exactly the first chunk: dropped without loading the next
    assert d.state()[0] == 0
    assert d.state(ensure_loaded=True) == (300, W.TRIVIAL, False)
