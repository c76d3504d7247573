#!/usr/bin/env python3
"""Headline benchmark: device-resident decode of the "10M-row int32/int64
columns, fixed-bit-width + RLE cascades" table (BASELINE.json configs[1]).

One step = one pass of the decode path over the whole batch (every chunk of every
column of a 10M-row stripe), inputs and descriptor tables already resident in
HBM.  Stripes are independent, so with N GPUs every rank decodes its own stripe
(weak scaling, no data-path collective; a barrier brackets the timed region).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-chunks", type=int, default=0, help="0 = first chunk of every column")
    ap.add_argument("--check", action="store_true", help="verify a sample of the output against the oracle")
    return ap.parse_args()


def cpu_baseline(chunks, min_seconds=12.0):
    """Times the CPU oracle (a port of the reference decode) on ONE thread over a
    bounded sample: whole passes over the same chunks the GPU decodes, repeated
    until at least `min_seconds` of CPU work (the output buffer is reused)."""
    from oracle import oracle as O

    lib = O.lib()
    arrays = [np.frombuffer(enc, dtype=np.uint8) for _, enc in chunks]
    meta = [(int(np.frombuffer(enc[2:6], dtype=np.uint32)[0]), {5: 4, 6: 4, 7: 8, 8: 8}[enc[1]]) for _, enc in chunks]
    vals = np.zeros(max(n * es for n, es in meta) + 64, dtype=np.uint8)
    out_bytes = rows = passes = 0
    t0 = time.perf_counter()
    while True:
        for a, (n, es) in zip(arrays, meta):
            r = lib.orc_decode_all(a.ctypes.data, a.size, 0, vals.ctypes.data, None)
            if r == 0xFFFFFFFF:
                raise RuntimeError("oracle failed to decode a benchmark chunk")
            out_bytes += n * es
            rows += n
        passes += 1
        if time.perf_counter() - t0 >= min_seconds:
            break
    dt = time.perf_counter() - t0
    return {
        "value": round(out_bytes / dt / 1e9, 3), "unit": "GB/s", "cores": 1, "kind": "port",
        "sample": f"{passes} full pass(es) over the same {len(chunks)} chunks ({rows} rows, {out_bytes} decoded bytes, "
                  f"{dt:.1f} s, single thread, oracle/nimble_oracle.c -O3 -march=native)",
        "rows_per_s": round(rows / dt),
    }


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))
    n_gpus = max(world, 1)
    device = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(device)

    from nimble_amd import workloads
    from nimble_amd.decoder import BatchDecoder, Plan

    t_gen = time.perf_counter()
    chunks = workloads.fbw_rle_table(args.rows, seed=42 + rank)  # each rank owns one stripe
    buf, offs = workloads.pack_chunks(chunks)
    t_gen = time.perf_counter() - t_gen
    plan = Plan(buf, offs)
    dec = BatchDecoder(plan, str(device))
    dec.stage()
    info = plan.info
    launches = plan.launches()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        dec.decode()
    dec.status()  # also synchronises; raises if a kernel flagged bad data

    # timed region: K steps, per-launch HIP events recorded on the decode stream
    dec.profile_begin(args.steps)
    barrier()
    t0 = time.perf_counter()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    ev0.record()
    for s in range(args.steps):
        dec.decode_profiled(s)
    ev1.record()
    barrier()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    launch_ms = dec.profile_end()
    elapsed = torch.tensor([wall], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())

    decoded = int(info.decoded_bytes)
    rows = int(info.total_rows)
    total_decoded = decoded * n_gpus  # every rank decodes a stripe of identical shape
    ms_per_step = elapsed / args.steps * 1e3
    value = total_decoded * args.steps / elapsed / 1e9

    # roofline of the dominant kernel (largest share of device time)
    by_kernel = {}
    for (name, es, nt, lvl, alg), ms in zip(launches, launch_ms):
        k = by_kernel.setdefault((name, es), {"ms": 0.0, "alg": 0, "tiles": 0, "launches": 0})
        k["ms"] += ms
        k["alg"] += alg
        k["tiles"] += nt
        k["launches"] += 1
    dom_key = max(by_kernel, key=lambda k: by_kernel[k]["ms"])
    dom = by_kernel[dom_key]
    achieved = dom["alg"] / (dom["ms"] * 1e-3) / 1e9 if dom["ms"] > 0 else 0.0
    roofline = {
        "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
        "kernel": f"{dom_key[0]}<{dom_key[1]}>", "kernel_ms": round(dom["ms"] / max(dom["launches"], 1), 4),
        "algorithmic_bytes_per_launch": dom["alg"] // max(dom["launches"], 1),
    }
    whole = (int(info.encoded_bytes) + decoded) / (sum(launch_ms) * 1e-3) / 1e9 if sum(launch_ms) > 0 else 0.0

    result = {
        "metric": "device-resident decoded GB/s (output bytes) per GPU and per 8xMI355X node",
        "value": round(value, 2), "unit": "GB/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "config": {
            "workload": f"{args.rows // 1_000_000}M-row int32/int64 columns, fixed-bit-width + RLE cascades "
                        f"(8 int32 + 16 int64 FBW widths, 4 RLE columns; <=20MiB raw chunks; one stripe per GPU)",
            "rows_per_gpu": args.rows, "columns": 28, "chunks_per_gpu": int(info.n_chunks),
            "launches_per_step": int(info.n_launches), "parallelism": f"stripe-sharded x{n_gpus}",
        },
        "rows_per_s": round(rows * n_gpus * args.steps / elapsed),
        "decoded_bytes_per_gpu": decoded, "encoded_bytes_per_gpu": int(info.encoded_bytes),
        "device_ms_per_step": round(dev_ms / args.steps, 4),
        "whole_batch_algorithmic_gbs": round(whole, 1),
        "whole_batch_hbm_frac": round(whole / HBM_PEAK_GBS, 4),
        "roofline": roofline,
        "kernels": [
            {"kernel": f"{k[0]}<{k[1]}>", "ms_per_step": round(v["ms"], 4), "launches": v["launches"], "tiles": v["tiles"],
             "algorithmic_gbs": round(v["alg"] / (v["ms"] * 1e-3) / 1e9, 1) if v["ms"] > 0 and v["alg"] else None}
            for k, v in sorted(by_kernel.items(), key=lambda kv: -kv[1]["ms"])
        ],
        "gen_seconds": round(t_gen, 1),
    }

    if args.check and rank == 0:
        from oracle import oracle as O
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from util import assert_chunk_equal, oracle_chunk
        from nimble_amd.decoder import slice_chunk
        host = dec.out.cpu().numpy()
        for i in range(0, len(chunks), 7):
            assert_chunk_equal(slice_chunk(host, plan.layouts[i]), oracle_chunk(O, chunks[i][1]), chunks[i][0])
        result["checked_chunks"] = len(range(0, len(chunks), 7))

    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(chunks)
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
