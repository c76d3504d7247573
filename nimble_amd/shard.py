"""Stripe -> GPU sharding for the multi-GPU decode.

A stripe is a self-contained set of streams (dwio/nimble/tablet/TabletReader.h:210-219;
stream offsets are relative to the stripe start, tablet/Footer.fbs:33-42) and an
Encoding reads nothing outside its own bytes (encodings/common/Encoding.h:52-54),
so ranks never exchange data: every rank plans, stages and decodes the stripes it
owns on its own GPU, and row order is restored by stripe index on the host, as
the reference reader walks stripes in order (velox/VeloxReader.cpp:299-353).
The only cross-rank traffic is the completion barrier and the reduction of the
timing / checksum scalars.
"""
from __future__ import annotations

import time
from typing import Callable, Dict, List, Sequence


def assign_stripes(stripe_sizes: Sequence[int], world: int) -> List[List[int]]:
    """Longest-processing-time greedy: stripes in decreasing size, each to the
    rank with the least bytes so far (ties: fewest stripes, then lowest rank).  Deterministic, so
    every rank computes the same assignment from the footer's stripe sizes
    without communicating.  Returns, per rank, its stripe indices ascending."""
    if world < 1:
        raise ValueError("world must be >= 1")
    load = [0] * world
    owned: List[List[int]] = [[] for _ in range(world)]
    for s in sorted(range(len(stripe_sizes)), key=lambda i: (-int(stripe_sizes[i]), i)):
        r = min(range(world), key=lambda k: (load[k], len(owned[k]), k))
        owned[r].append(s)
        load[r] += int(stripe_sizes[s])
    return [sorted(o) for o in owned]


def assign_units(stripe_stream_sizes: Sequence[Sequence[int]], world: int):
    """Work units for a file: whole stripes when there are at least `world` of
    them, otherwise (stripe, stream) pairs so that a file with few stripes still
    fills every GPU.  Returns, per rank, a list of (stripe, stream-or-None)."""
    n_stripes = len(stripe_stream_sizes)
    if n_stripes >= world:
        per_rank = assign_stripes([sum(s) for s in stripe_stream_sizes], world)
        return [[(s, None) for s in stripes] for stripes in per_rank]
    units = [(si, ci) for si, streams in enumerate(stripe_stream_sizes) for ci in range(len(streams))]
    sizes = [stripe_stream_sizes[si][ci] for si, ci in units]
    per_rank = assign_stripes(sizes, world)
    return [[units[u] for u in owned] for owned in per_rank]


def imbalance(stripe_sizes: Sequence[int], owned: Sequence[Sequence[int]]) -> float:
    """max rank load / mean rank load (1.0 = perfectly balanced)."""
    loads = [sum(int(stripe_sizes[s]) for s in o) for o in owned]
    mean = sum(loads) / max(len(loads), 1)
    return max(loads) / mean if mean > 0 else 1.0


def decode_owned(stripes: Sequence, rank: int, world: int, decode_stripe: Callable, sizes: Sequence[int] | None = None) -> Dict[int, object]:
    """Runs `decode_stripe(index, stripe)` for the stripes `rank` owns."""
    sizes = sizes if sizes is not None else [len(s) for s in stripes]
    return {s: decode_stripe(s, stripes[s]) for s in assign_stripes(sizes, world)[rank]}


def stitch(per_rank_results: Sequence[Dict[int, object]], n_stripes: int) -> list:
    """Host-side reassembly: results of all ranks back in stripe (= row) order.
    Raises if a stripe is missing or was decoded twice."""
    out = [None] * n_stripes
    seen = [False] * n_stripes
    for res in per_rank_results:
        for s, r in res.items():
            if seen[s]:
                raise RuntimeError(f"stripe {s} decoded by more than one rank")
            seen[s] = True
            out[s] = r
    missing = [i for i, ok in enumerate(seen) if not ok]
    if missing:
        raise RuntimeError(f"stripes not decoded by any rank: {missing}")
    return out


def timed_steps(step: Callable[[int], None], steps: int, sync: Callable[[], None], dist=None) -> float:
    """Times exactly `steps` calls of `step` between two barriers (`dist.barrier()`
    when a process group is given, then `sync()` = device synchronise, on both
    sides) and returns the MAX over ranks of the elapsed seconds."""

    def fence():
        if dist is not None:
            dist.barrier()
        sync()

    fence()
    t0 = time.perf_counter()
    for s in range(steps):
        step(s)
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch

        t = torch.tensor([elapsed], dtype=torch.float64)
        backend = dist.get_backend()
        if backend == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed
