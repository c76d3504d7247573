"""Multi-rank path on CPU (gloo, world_size 2): stripe assignment, per-rank
decode of the owned stripes, host-side reassembly in stripe order and the
barrier-bracketed max-over-ranks timing that bench.py reports.  The decode
itself is the CPU oracle here; on the GPU box each rank runs the same code with
the HIP decoder on its own device."""
import os
import socket
import sys
import zlib

import numpy as np
import pytest

from nimble_amd import encoder as E
from nimble_amd import shard
from nimble_amd import wire as W

HERE = os.path.dirname(os.path.abspath(__file__))


def test_lpt_assignment_properties():
    rng = np.random.default_rng(1)
    for world in (1, 2, 3, 8):
        for n in (0, 1, 5, 8, 37):
            sizes = rng.integers(1, 1000, n).tolist()
            owned = shard.assign_stripes(sizes, world)
            assert len(owned) == world
            flat = sorted(s for o in owned for s in o)
            assert flat == list(range(n))  # every stripe exactly once
            assert all(o == sorted(o) for o in owned)
            if n >= world and n:
                loads = [sum(sizes[s] for s in o) for o in owned]
                assert max(loads) - min(loads) <= max(sizes)  # LPT bound
            assert owned == shard.assign_stripes(sizes, world)  # deterministic
    # equal stripes: rank r gets stripes r, r+world, ...
    assert shard.assign_stripes([7] * 8, 4) == [[0, 4], [1, 5], [2, 6], [3, 7]]
    assert shard.imbalance([7] * 8, shard.assign_stripes([7] * 8, 4)) == 1.0
    # empty stripes still spread over the ranks (one each), they do not pile up on rank 0
    assert shard.assign_stripes([0, 0, 0], 3) == [[0], [1], [2]]


def test_units_fall_back_to_streams_when_few_stripes():
    units = shard.assign_units([[10, 20, 30, 40]], 2)  # one stripe, two GPUs
    assert sorted(u for r in units for u in r) == [(0, 0), (0, 1), (0, 2), (0, 3)]
    loads = [sum([10, 20, 30, 40][c] for _, c in r) for r in units]
    assert loads == [50, 50]
    units = shard.assign_units([[1, 2], [3, 4], [5]], 2)
    assert sorted(u for r in units for u in r) == [(0, None), (1, None), (2, None)]


def test_stitch_detects_missing_and_duplicate():
    assert shard.stitch([{0: "a", 2: "c"}, {1: "b"}], 3) == ["a", "b", "c"]
    with pytest.raises(RuntimeError, match="not decoded"):
        shard.stitch([{0: "a"}], 2)
    with pytest.raises(RuntimeError, match="more than one"):
        shard.stitch([{0: "a"}, {0: "a"}], 1)


def make_stripes(n_stripes=5, seed=3):
    """Each stripe: a few column chunks of different cascades and sizes."""
    rng = np.random.default_rng(seed)
    stripes = []
    for s in range(n_stripes):
        rows = int(rng.integers(200, 3000))
        a = rng.integers(0, 1 << 13, rows).astype(np.uint32)
        b = np.repeat(rng.integers(0, 99, rows // 7 + 1), 7)[:rows].astype(np.uint64)
        mask = rng.random(rows) < 0.7
        c = rng.integers(0, 1 << 20, int(mask.sum())).astype(np.uint32)
        stripes.append([
            E.encode(E.FixedBitWidth(exact_bits=True), a, W.UINT32),
            E.encode(E.RLE(), b, W.INT64),
            E.encode_nullable(E.Nullable(data=E.FixedBitWidth(), nulls=E.SparseBool()), c, mask, W.INT32),
        ])
    return stripes


def _decode_stripe_cpu(O, chunks):
    """(crc of values, crc of null bitmap, rows) per chunk, via the oracle."""
    out = []
    for enc in chunks:
        r = O.decode_chunk(enc)
        nulls = r["nulls"] if r["nulls"] is not None else np.zeros(0, dtype=np.uint8)
        out.append((zlib.crc32(np.ascontiguousarray(r["values"]).tobytes()), zlib.crc32(np.ascontiguousarray(nulls).tobytes()), int(r["rows"])))
    return out


def _worker(rank, world, port, q):
    import time

    import torch.distributed as dist

    sys.path.insert(0, os.path.dirname(HERE))
    from oracle import oracle as O

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        stripes = make_stripes()
        sizes = [sum(len(c) for c in s) for s in stripes]
        mine = shard.decode_owned(stripes, rank, world, lambda i, s: _decode_stripe_cpu(O, s), sizes)
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        whole = shard.stitch(gathered, len(stripes))

        # timing: rank 1 is slower; every rank must report the slower time
        elapsed = shard.timed_steps(lambda s: time.sleep(0.02 * (rank + 1)), 3, lambda: None, dist)
        q.put((rank, sorted(mine), whole, elapsed))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_stripe_sharded_decode_matches_single_process(oracle):
    import torch.multiprocessing as mp

    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    stripes = make_stripes()
    want = [_decode_stripe_cpu(oracle, s) for s in stripes]
    sizes = [sum(len(c) for c in s) for s in stripes]
    owned = shard.assign_stripes(sizes, world)
    by_rank = {r: (mine, whole, el) for r, mine, whole, el in got}
    for r in range(world):
        mine, whole, elapsed = by_rank[r]
        assert mine == owned[r]
        assert whole == want  # reassembled result identical to the one-process decode
        assert elapsed >= 3 * 0.02 * world * 0.9  # the MAX over ranks (rank 1 sleeps 0.04 s per step)
    assert abs(by_rank[0][2] - by_rank[1][2]) < 1e-9  # all ranks agree on the reduced time
