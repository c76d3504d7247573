"""GPU: the benchmark configurations at their full sizes, every chunk compared
bit-for-bit with the CPU oracle (the oracle decodes these sizes in seconds),
plus size-independent properties of the whole batch."""
import numpy as np
import pytest

from nimble_amd import wire as W
from nimble_amd import workloads as WL

from util import assert_chunk_equal, oracle_chunk

pytestmark = pytest.mark.gpu


def decode_and_check(O, chunks, stride=1):
    """One GPU batch over all chunks; compares every `stride`-th chunk with the
    oracle, streaming chunk by chunk so host memory stays bounded.  Returns the
    plan info and (rows, non-null rows) totals from the layouts."""
    from nimble_amd.decoder import BatchDecoder, Plan, slice_chunk

    buf, offs = WL.pack_chunks(chunks)
    plan = Plan(buf, offs)
    dec = BatchDecoder(plan)
    dec.stage()
    dec.decode()
    dec.status()
    # idempotence: a second decode into the same buffers leaves the same bytes
    first = dec.out.clone()
    dec.decode()
    dec.status()
    import torch

    assert torch.equal(first, dec.out)
    del first
    host = dec.out.cpu().numpy()
    rows = 0
    for i, (label, enc) in enumerate(chunks):
        lay = plan.layouts[i]
        rows += int(lay.rows)
        if i % stride == 0:
            assert_chunk_equal(slice_chunk(host, lay), oracle_chunk(O, enc), label)
    info = plan.info
    assert int(info.total_rows) == rows
    res = dict(n_chunks=int(info.n_chunks), n_launches=int(info.n_launches), n_nodes=int(info.n_nodes), rows=rows,
               kernels=[l[0] for l in plan.launches()])
    plan.close()
    return res


def test_fbw_rle_table_full_size(gpu, oracle):
    """10M-row int32/int64 columns, every FixedBitArray width branch + RLE cascades."""
    chunks = WL.fbw_rle_table(10_000_000, seed=42)
    r = decode_and_check(oracle, chunks)
    assert r["rows"] == 28 * 10_000_000 and r["n_launches"] <= 12


def test_cpu_reference_column(gpu, oracle):
    """1M-row int64 column, bit-packed + RLE: round trip equals the generated values."""
    from nimble_amd.decoder import decode_encodings

    chunks, vals = WL.cpu_reference_column(1_000_000, seed=42)
    for (label, enc), res in zip(chunks, decode_encodings([e for _, e in chunks])):
        assert np.array_equal(res.values, vals), label
        assert_chunk_equal(res, oracle_chunk(oracle, enc), label)


def test_dict_string_nullable_full_size(gpu, oracle):
    """10M-row dictionary string columns (three alphabet sizes) under Nullable
    with SparseBool / bitmap nulls."""
    chunks = WL.dict_string_table(10_000_000, seed=7, alphabets=(64, 4096, 65536), null_fracs=(0.0, 0.05, 0.5))
    r = decode_and_check(oracle, chunks)
    assert r["rows"] == 9 * 10_000_000


def test_mixed_table_stripe_full_size(gpu, oracle):
    """One stripe (1.25M rows x 100 columns) of the mixed-type 3-level-cascade table."""
    chunks = WL.mixed_table_stripe(stripe=3, rows=1_250_000)
    assert len(chunks) == 100
    r = decode_and_check(oracle, chunks)
    assert r["rows"] == 100 * 1_250_000


def test_wide_table_stripe_full_size(gpu, oracle):
    """One stripe of the 5,000-column sparse/nullable float table: 5,000 chunks,
    ~25K cascade nodes, still a few dozen launches."""
    chunks = WL.wide_table_stripe(stripe=1, columns=5000, rows=100_000)
    r = decode_and_check(oracle, chunks)
    assert r["n_chunks"] == 5000 and r["rows"] == 5000 * 100_000
    assert r["n_launches"] <= 40
    # a stripe of this shape has ~10^5 sparse rank-scatter tiles: they run on the one-wave kernel,
    # and its columns of 100,000 rows need no scan level (selection bitmaps, then the scatters)
    assert "k_select_scatter_wave<4>" in r["kernels"] and r["n_launches"] == 3
