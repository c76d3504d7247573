#!/usr/bin/env python3
"""Headline benchmark: device-resident decode of the "10M-row int32/int64
columns, fixed-bit-width + RLE cascades" table.

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
    ap.add_argument("--rows", type=int, default=0, help="rows per column (default: the workload's own size)")
    ap.add_argument("--workload", default="fbw_rle", choices=["fbw_rle", "dict_string", "mixed", "wide"],
                    help="fbw_rle is the headline configuration; the others are the remaining benchmark tables")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pcie", action="store_true", help="skip the H2D+decode+D2H measurement")
    ap.add_argument("--no-breakdown", action="store_true", help="skip the per-operation profile of the unfused plan")
    ap.add_argument("--cpu-sample-chunks", type=int, default=0, help="0 = first chunk of every column")
    ap.add_argument("--check", action="store_true", help="verify a sample of the output against the oracle")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N>1 (nccl = RCCL; gloo only to rehearse several ranks on one GPU)")
    return ap.parse_args()


def cpu_baseline(chunks, min_seconds=12.0):
    """Times the CPU oracle (a port of the reference decode) on ONE thread over a
    bounded sample: whole passes over the same chunks the GPU decodes, repeated
    until at least `min_seconds` of CPU work (the output buffer is reused)."""
    from oracle import oracle as O

    lib = O.lib()
    arrays = [np.frombuffer(enc, dtype=np.uint8) for _, enc in chunks]
    es_of = {1: 1, 2: 1, 3: 2, 4: 2, 5: 4, 6: 4, 7: 8, 8: 8, 9: 4, 10: 8, 11: 1, 12: 16}
    meta = [(int(np.frombuffer(enc[2:6], dtype=np.uint32)[0]), es_of[enc[1]]) for _, enc in chunks]
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


def cpu_baseline_threads(chunks, threads, min_seconds=8.0):
    """The same oracle on `threads` host threads, one chunk per task (chunks are
    independent, as columns and stripes are for the reference's reader).  The
    oracle calls run outside the GIL."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O

    lib = O.lib()
    arrays = [np.frombuffer(enc, dtype=np.uint8) for _, enc in chunks]
    es_of = {1: 1, 2: 1, 3: 2, 4: 2, 5: 4, 6: 4, 7: 8, 8: 8, 9: 4, 10: 8, 11: 1, 12: 16}
    meta = [(int(np.frombuffer(enc[2:6], dtype=np.uint32)[0]), es_of[enc[1]]) for _, enc in chunks]
    biggest = max(n * es for n, es in meta) + 64
    deadline = [0.0]

    def worker(t):
        vals = np.zeros(biggest, dtype=np.uint8)
        out = rows = 0
        while True:
            for i in range(t, len(arrays), threads):
                if lib.orc_decode_all(arrays[i].ctypes.data, arrays[i].size, 0, vals.ctypes.data, None) == 0xFFFFFFFF:
                    raise RuntimeError("oracle failed to decode a benchmark chunk")
                out += meta[i][0] * meta[i][1]
                rows += meta[i][0]
            if time.perf_counter() >= deadline[0]:
                return out, rows

    t0 = time.perf_counter()
    deadline[0] = t0 + min_seconds
    with ThreadPoolExecutor(max_workers=threads) as pool:
        parts = list(pool.map(worker, range(threads)))
    dt = time.perf_counter() - t0
    out_bytes, rows = sum(p[0] for p in parts), sum(p[1] for p in parts)
    return {"value": round(out_bytes / dt / 1e9, 3), "unit": "GB/s", "cores": threads, "kind": "port",
            "sample": f"{threads} threads, each looping over its share of the same {len(chunks)} chunks for {dt:.1f} s "
                      f"({rows} rows, {out_bytes} decoded bytes)",
            "rows_per_s": round(rows / dt)}


def copy_ceiling(torch, device, nbytes=1 << 31, reps=10):
    """In-run achievable HBM rate: a device-to-device copy of `nbytes` (bytes read + bytes written per second)."""
    src = torch.empty(nbytes, dtype=torch.uint8, device=device)
    dst = torch.empty_like(src)
    dst.copy_(src)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize(device)
    return 2 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def pcie_inclusive(dec, plan, torch, device, steps=5):
    """The same step with the host at both ends: H2D of the encoded bytes from
    pinned memory, decode, D2H of the decoded output into pinned memory.
    Reported beside the headline value, never as it."""
    enc = int(plan.buffer.size)
    h_in = torch.empty(enc, dtype=torch.uint8).pin_memory()
    h_in.numpy()[:] = plan.buffer
    h_out = torch.empty(int(plan.info.out_bytes), dtype=torch.uint8).pin_memory()

    def one():
        dec.arena[:enc].copy_(h_in, non_blocking=True)
        dec.decode()
        h_out.copy_(dec.out, non_blocking=True)

    one()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    torch.cuda.synchronize(device)
    dt = (time.perf_counter() - t0) / steps
    decoded = int(plan.info.decoded_bytes)
    return {"decoded_gbs": round(decoded / dt / 1e9, 2), "ms_per_step": round(dt * 1e3, 3), "h2d_bytes": enc,
            "d2h_bytes": int(plan.info.out_bytes), "steps": steps,
            "link_gbs": round((enc + int(plan.info.out_bytes)) / dt / 1e9, 2)}


def pcie_pipelined(plan, torch, device, n_buffers=2, stripes=8):
    """Host-to-host rate with stripes in flight on `n_buffers` streams: the H2D of
    one stripe overlaps the decode and the D2H of the previous one (PCIe is full
    duplex), which is how a reader would drive the path over many stripes."""
    from nimble_amd.decoder import BatchDecoder

    enc = int(plan.buffer.size)
    streams = [torch.cuda.Stream(device=device) for _ in range(n_buffers)]
    decs, h_in, h_out = [], [], []
    for s in streams:
        d = BatchDecoder(plan, str(device), stream=s)
        with torch.cuda.stream(s):
            d.stage()
        decs.append(d)
        hi = torch.empty(enc, dtype=torch.uint8).pin_memory()
        hi.numpy()[:] = plan.buffer
        h_in.append(hi)
        h_out.append(torch.empty(int(plan.info.out_bytes), dtype=torch.uint8).pin_memory())

    def run(n):
        for i in range(n):
            k = i % n_buffers
            with torch.cuda.stream(streams[k]):
                decs[k].arena[:enc].copy_(h_in[k], non_blocking=True)
                decs[k].decode()
                h_out[k].copy_(decs[k].out, non_blocking=True)
        torch.cuda.synchronize(device)

    run(n_buffers)
    t0 = time.perf_counter()
    run(stripes)
    dt = (time.perf_counter() - t0) / stripes
    for d in decs:
        d.status()
    decoded = int(plan.info.decoded_bytes)
    return {"decoded_gbs": round(decoded / dt / 1e9, 2), "ms_per_stripe": round(dt * 1e3, 3), "streams": n_buffers,
            "stripes": stripes, "h2d_gbs": round(enc / dt / 1e9, 2), "d2h_gbs": round(int(plan.info.out_bytes) / dt / 1e9, 2)}


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend="gloo")
            local_rank %= torch.cuda.device_count()  # rehearsal: ranks share the visible GPUs
    n_gpus = max(world, 1)
    device = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(device)

    from nimble_amd import shard, workloads
    from nimble_amd.decoder import BatchDecoder, Plan

    t_gen = time.perf_counter()
    # one stripe of identical shape per GPU (weak scaling); ownership by the same
    # assignment a file reader uses, computed on every rank without communication
    default_rows = {"fbw_rle": 10_000_000, "dict_string": 10_000_000, "mixed": 1_250_000, "wide": 100_000}
    args.rows = args.rows or default_rows[args.workload]
    stripe = shard.assign_stripes([args.rows] * n_gpus, n_gpus)[rank][0]
    chunks = workloads.WORKLOADS[args.workload](args.rows, 42 + stripe)
    buf, offs = workloads.pack_chunks(chunks)
    t_gen = time.perf_counter() - t_gen
    plan = Plan(buf, offs)
    dec = BatchDecoder(plan, str(device))
    dec.stage()
    info = plan.info
    launches = plan.launches()

    for _ in range(args.warmup):
        dec.decode()
    dec.status()  # also synchronises; raises if a kernel flagged bad data

    # timed region: exactly K steps between barrier + device synchronise on both
    # sides, max over ranks; per-launch HIP events recorded on the decode stream
    dec.profile_begin(args.steps)
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)

    def step(s):
        if s == 0:
            ev0.record()
        dec.decode_profiled(s)
        if s == args.steps - 1:
            ev1.record()

    elapsed = shard.timed_steps(step, args.steps, lambda: torch.cuda.synchronize(device), dist if world > 1 else None)
    dev_ms = ev0.elapsed_time(ev1)
    launch_ms = dec.profile_end()
    dec.status()

    decoded = int(info.decoded_bytes)
    rows = int(info.total_rows)
    total_decoded = decoded * n_gpus  # every rank decodes a stripe of identical shape
    ms_per_step = elapsed / args.steps * 1e3
    value = total_decoded * args.steps / elapsed / 1e9

    # roofline of the dominant kernel (largest share of device time)
    def group(launch_list, ms_list):
        by = {}
        for (name, es, nt, lvl, alg), ms in zip(launch_list, ms_list):
            k = by.setdefault(name, {"ms": 0.0, "alg": 0, "tiles": 0, "launches": 0})
            k["ms"] += ms
            k["alg"] += alg
            k["tiles"] += nt
            k["launches"] += 1
        return by

    def table(by):
        return [
            {"kernel": k, "ms_per_step": round(v["ms"], 4), "launches": v["launches"], "tiles": v["tiles"],
             "algorithmic_gbs": round(v["alg"] / (v["ms"] * 1e-3) / 1e9, 1) if v["ms"] > 0 and v["alg"] else None}
            for k, v in sorted(by.items(), key=lambda kv: -kv[1]["ms"])
        ]

    by_kernel = group(launches, launch_ms)
    dom_key = max(by_kernel, key=lambda k: by_kernel[k]["ms"])
    dom = by_kernel[dom_key]
    achieved = dom["alg"] / (dom["ms"] * 1e-3) / 1e9 if dom["ms"] > 0 else 0.0
    roofline = {
        "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
        "kernel": dom_key, "kernel_ms": round(dom["ms"] / max(dom["launches"], 1), 4),
        "algorithmic_bytes_per_launch": dom["alg"] // max(dom["launches"], 1),
    }
    # HBM bytes per launch from the PMC passes of this same command (FETCH_SIZE and
    # WRITE_SIZE collected in separate rocprofv3 runs, corrected by tools/pmc_traffic.py)
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        pmc = json.load(open(path))
        k = pmc.get("kernels", {}).get(roofline["kernel"])
        if pmc.get("rows") == args.rows and pmc.get("workload", "fbw_rle") == args.workload and k:
            roofline["traffic"] = k["traffic_bytes"]
            roofline["traffic_over_algorithmic"] = round(k["traffic_bytes"] / max(roofline["algorithmic_bytes_per_launch"], 1), 4)
            roofline["traffic_source"] = os.path.relpath(path, ROOT)
            break
    whole = (int(info.encoded_bytes) + decoded) / (sum(launch_ms) * 1e-3) / 1e9 if sum(launch_ms) > 0 else 0.0
    if rank == 0:
        # second denominator: what a plain device-to-device copy sustains on this GPU in this run
        ceiling = copy_ceiling(torch, device)
        roofline["copy_ceiling_gbs"] = round(ceiling, 1)
        roofline["frac_of_copy_ceiling"] = round(achieved / ceiling, 4) if ceiling > 0 else None

    result = {
        "metric": "device-resident decoded GB/s (output bytes) per GPU and per 8xMI355X node",
        "value": round(value, 2), "unit": "GB/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "config": {
            "workload": {
                "fbw_rle": f"{args.rows // 1_000_000}M-row int32/int64 columns, fixed-bit-width + RLE cascades "
                           f"(8 int32 + 16 int64 FBW widths, 4 RLE columns; <=20MiB raw chunks; one stripe per GPU)",
                "dict_string": f"{args.rows}-row dictionary-encoded string columns (3 alphabet sizes) with nullable wrapper",
                "mixed": f"mixed-type 100-column table, 3-level cascades, one {args.rows}-row stripe per GPU",
                "wide": f"wide ML table, 5000 sparse/nullable float columns, one {args.rows}-row stripe per GPU",
            }[args.workload],
            "rows_per_gpu": args.rows, "chunks_per_gpu": int(info.n_chunks), "nodes_per_gpu": int(info.n_nodes),
            "launches_per_step": int(info.n_launches), "parallelism": f"stripe-sharded x{n_gpus}",
        },
        "rows_per_s": round(rows * n_gpus * args.steps / elapsed),
        "decoded_bytes_per_gpu": decoded, "encoded_bytes_per_gpu": int(info.encoded_bytes),
        "device_ms_per_step": round(dev_ms / args.steps, 4),
        "whole_batch_algorithmic_gbs": round(whole, 1),
        "whole_batch_hbm_frac": round(whole / HBM_PEAK_GBS, 4),
        "roofline": roofline,
        "kernels": table(by_kernel),
        "gen_seconds": round(t_gen, 1),
    }

    if rank == 0 and n_gpus == 1 and not args.no_breakdown:
        # diagnostic, outside the timed region: the same batch with one launch per
        # (level, operation), which attributes device time to operations
        plan2 = Plan(buf, offs, per_kernel=True)
        dec2 = BatchDecoder(plan2, str(device))
        dec2.stage()
        for _ in range(2):
            dec2.decode()
        dec2.status()
        dec2.profile_begin(5)
        for s in range(5):
            dec2.decode_profiled(s)
        ms2 = dec2.profile_end()
        result["per_operation"] = {"launches_per_step": int(plan2.info.n_launches), "sum_ms_per_step": round(sum(ms2), 4),
                                   "kernels": table(group(plan2.launches(), ms2))}
        del dec2
        plan2.close()

    if args.check and rank == 0:
        from oracle import oracle as O
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from util import assert_chunk_equal, oracle_chunk
        from nimble_amd.decoder import slice_chunk
        host = dec.out.cpu().numpy()
        for i in range(0, len(chunks), 7):
            assert_chunk_equal(slice_chunk(host, plan.layouts[i]), oracle_chunk(O, chunks[i][1]), chunks[i][0])
        result["checked_chunks"] = len(range(0, len(chunks), 7))

    if rank == 0 and n_gpus == 1 and not args.no_pcie:
        result["pcie_inclusive"] = pcie_inclusive(dec, plan, torch, device)
        result["pcie_pipelined"] = pcie_pipelined(plan, torch, device)
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(chunks, min_seconds=10.0)
        # the GPU box gives this job 16 host threads (OMP_NUM_THREADS is set to that share)
        threads = max(1, min(int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or (os.cpu_count() or 1), 16, len(chunks)))
        result["cpu_baseline_all_cores"] = cpu_baseline_threads(chunks, threads)
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
