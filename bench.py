#!/usr/bin/env python3
"""Benchmark of the device-resident cascade decode.

Headline (default): the "10M-row int32/int64 columns, fixed-bit-width + RLE
cascades" table.  One step = one pass of the decode path over every chunk of
every column of the rank's stripe(s), inputs and descriptor tables already
resident in HBM.

Multi-GPU: stripes are independent, so ranks never exchange data; a barrier
brackets the timed region and the elapsed time is the MAX over ranks.
  --scaling weak   (default) every rank decodes one stripe of identical shape
  --scaling strong a fixed table (mixed: 8 stripes, wide: 16 stripes) is dealt
                   over the ranks with shard.assign_stripes; a step is one pass
                   over the whole table
`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts N rank
processes itself (fresh children, before this process touches the GPU) and
relays rank 0's line; under torchrun the ranks are taken from the environment.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak
DEFAULT_ROWS = {"fbw_rle": 10_000_000, "dict_string": 10_000_000, "mixed": 1_250_000, "wide": 100_000}
STRONG_STRIPES = {"mixed": 8, "wide": 16}  # stripes of the fixed tables
WORKLOAD_TEXT = {
    "fbw_rle": "{rows}-row int32/int64 columns, fixed-bit-width + RLE cascades "
               "(8 int32 + 16 int64 FBW widths, 4 RLE columns; <=20MiB raw chunks)",
    "dict_string": "{rows}-row dictionary-encoded string columns (3 alphabet sizes x 3 null fractions) with nullable wrapper",
    "mixed": "mixed-type 100-column table, 3-level cascades, {rows}-row stripes",
    "wide": "wide ML table, 5000 sparse/nullable float columns, {rows}-row stripes",
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=0, help="rows per stripe (default: the workload's own size)")
    ap.add_argument("--workload", default="fbw_rle", choices=list(DEFAULT_ROWS),
                    help="fbw_rle is the headline configuration; the others are the remaining benchmark tables")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="strong: the fixed 8-stripe (mixed) / 16-stripe (wide) table dealt over the ranks")
    ap.add_argument("--stripes", type=int, default=0, help="strong scaling: stripes of the table (default 8 mixed / 16 wide)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pcie", action="store_true", help="skip the H2D+decode+D2H measurement")
    ap.add_argument("--event-every", type=int, default=4, help="record the per-launch HIP events on every N-th timed step")
    ap.add_argument("--no-breakdown", action="store_true", help="skip the per-operation profile of the unfused plan")
    ap.add_argument("--no-other-workloads", action="store_true", help="skip the short runs of the other benchmark tables")
    ap.add_argument("--no-cold", action="store_true", help="skip the cold-cache steps (profiling runs: keeps the per-kernel averages to the timed loop)")
    ap.add_argument("--check", action="store_true", help="verify a sample of the output against the oracle")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N>1 (nccl = RCCL; gloo only to rehearse several ranks on one GPU)")
    ap.add_argument("--rank-probe", action="store_true",
                    help="launcher self-check: every rank joins the process group, nothing touches a GPU")
    return ap.parse_args(argv)


# ----------------------------------------------------------------- launcher
def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_commands(n: int, argv, port: int, base_env=None):
    """(argv, env) of the N rank processes `--gpus N` starts: this script again,
    one process per GPU, rendezvous on 127.0.0.1."""
    out = []
    for r in range(n):
        env = dict(base_env if base_env is not None else os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        out.append(([sys.executable, "-u", os.path.abspath(__file__)] + list(argv), env))
    return out


def launch_ranks(n: int, argv) -> int:
    """Parent of a `--gpus N` run: starts the ranks as fresh child processes (this
    process has made no GPU call and makes none), relays rank 0's JSON line and
    returns non-zero when any rank failed.  The children are ended by PID if one
    of them dies."""
    procs = []
    for r, (cmd, env) in enumerate(rank_commands(n, argv, free_port())):
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr))
    out0 = b""
    rc = 0
    try:
        out0, _ = procs[0].communicate()
        for p in procs:
            p.wait()
            rc = rc or p.returncode
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    lines = [l for l in out0.decode(errors="replace").splitlines() if l.strip()]
    for l in lines[:-1]:
        print(l, file=sys.stderr)
    if lines:
        print(lines[-1])
    return rc


def rank_probe(world: int, rank: int) -> None:
    import torch
    import torch.distributed as dist

    dist.init_process_group(backend="gloo")
    t = torch.tensor([rank + 1], dtype=torch.int64)
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"n_gpus": world, "rank_sum": int(t.item()), "probe": True}))
    dist.barrier()
    dist.destroy_process_group()


# ------------------------------------------------------------- CPU baseline
_ES_OF = {1: 1, 2: 1, 3: 2, 4: 2, 5: 4, 6: 4, 7: 8, 8: 8, 9: 4, 10: 8, 11: 1, 12: 16}


def _chunk_meta(chunks):
    arrays = [np.frombuffer(enc, dtype=np.uint8) for _, enc in chunks]
    meta = [(int(np.frombuffer(enc[2:6], dtype=np.uint32)[0]), _ES_OF[enc[1]]) for _, enc in chunks]
    return arrays, meta


def cpu_baseline(chunks, min_seconds=10.0, fast=True):
    """Times the CPU oracle (a port of the reference decode) on ONE thread over a
    bounded sample: whole passes over the same chunks the GPU decodes, repeated
    until at least `min_seconds` of CPU work.  `fast` selects the oracle's
    bench-only bulk path (64-value unrolled unpack groups and run fills, the
    shape of FixedBitArray::bulkGet32Internal / bulkGet64WithBaseline and
    RLEEncoding::materialize's simdFill); without it every value goes through
    the element-wise get, which understates the reference."""
    from oracle import oracle as O

    lib = O.lib()
    decode = lib.orc_decode_all_fast if fast else lib.orc_decode_all
    arrays, meta = _chunk_meta(chunks)
    vals = np.zeros(max(n * es for n, es in meta) + 64, dtype=np.uint8)
    out_bytes = rows = passes = 0
    t0 = time.perf_counter()
    while True:
        for a, (n, es) in zip(arrays, meta):
            r = decode(a.ctypes.data, a.size, 0, vals.ctypes.data, None)
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
        "path": "bulk (64-value unrolled groups)" if fast else "element-wise get",
        "sample": f"{passes} full pass(es) over the same {len(chunks)} chunks ({rows} rows, {out_bytes} decoded bytes, "
                  f"{dt:.1f} s, single thread, oracle/nimble_oracle.c -O3 -march=native; one output buffer reused, "
                  f"so stores mostly stay in cache: leans in the CPU's favour)",
        "rows_per_s": round(rows / dt),
    }


def cpu_baseline_threads(chunks, threads, min_seconds=8.0):
    """The oracle's bulk path on `threads` host threads, one chunk per task
    (chunks are independent, as columns and stripes are for the reference's
    reader).  Every chunk is decoded into an output region of its own, as a
    reader materialising the table would, so the stores go to memory.  The
    oracle calls run outside the GIL."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O

    lib = O.lib()
    arrays, meta = _chunk_meta(chunks)
    offs = np.zeros(len(meta) + 1, dtype=np.int64)
    np.cumsum([((n * es + 64 + 63) // 64) * 64 for n, es in meta], out=offs[1:])
    out = np.zeros(int(offs[-1]), dtype=np.uint8)
    deadline = [0.0]

    def worker(t):
        done = rows = 0
        while True:
            for i in range(t, len(arrays), threads):
                if lib.orc_decode_all_fast(arrays[i].ctypes.data, arrays[i].size, 0, out.ctypes.data + int(offs[i]), None) == 0xFFFFFFFF:
                    raise RuntimeError("oracle failed to decode a benchmark chunk")
                done += meta[i][0] * meta[i][1]
                rows += meta[i][0]
            if time.perf_counter() >= deadline[0]:
                return done, rows

    t0 = time.perf_counter()
    deadline[0] = t0 + min_seconds
    with ThreadPoolExecutor(max_workers=threads) as pool:
        parts = list(pool.map(worker, range(threads)))
    dt = time.perf_counter() - t0
    out_bytes, rows = sum(p[0] for p in parts), sum(p[1] for p in parts)
    return {"value": round(out_bytes / dt / 1e9, 3), "unit": "GB/s", "cores": threads, "kind": "port",
            "path": "bulk (64-value unrolled groups)",
            "sample": f"{threads} threads, each looping over its share of the same {len(chunks)} chunks for {dt:.1f} s "
                      f"({rows} rows, {out_bytes} decoded bytes; every chunk has its own output region, {out.size} bytes in all)",
            "rows_per_s": round(rows / dt)}


# ------------------------------------------------------------ GPU-side legs
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


def pcie_pipelined(plan, torch, device, n_buffers=2, stripes=16):
    """Host-to-host rate with `n_buffers` stripes in flight: the H2D of one stripe
    overlaps the D2H of the previous one (PCIe is full duplex), which is how a
    reader would drive the path over many stripes.  Two streams: one carries the
    copies in and the decodes, the other the copies back, which are kernels
    (nbl_copy_to_host_async) as in the C++ host layer.  With copy engines both
    ways, which engine a stream gets is the runtime's choice, and the copy in of
    one stripe can land behind the copy back of another on the same engine; more
    streams than the process has hardware queues (4) make streams wait for each
    other as well."""
    import ctypes as C

    from nimble_amd import capi
    from nimble_amd.decoder import BatchDecoder

    enc = int(plan.buffer.size)
    work, copy_out = torch.cuda.Stream(device=device), torch.cuda.Stream(device=device)
    decs, h_in, h_out = [], [], []
    for _ in range(n_buffers):
        d = BatchDecoder(plan, str(device), stream=work)
        with torch.cuda.stream(work):
            d.stage()
        decs.append(d)
        hi = torch.empty(enc, dtype=torch.uint8).pin_memory()
        hi.numpy()[:] = plan.buffer
        h_in.append(hi)
        h_out.append(torch.empty(int(plan.info.out_bytes), dtype=torch.uint8).pin_memory())
    torch.cuda.synchronize(device)
    decoded_ev = [torch.cuda.Event() for _ in decs]   # the buffer's output has been written
    out_free = [torch.cuda.Event() for _ in decs]     # the buffer's output has been copied back

    def run(n):
        for i in range(n):
            k = i % n_buffers
            work.wait_event(out_free[k])
            with torch.cuda.stream(work):
                decs[k].arena[:enc].copy_(h_in[k], non_blocking=True)
                decs[k].decode()
                decoded_ev[k].record()
            copy_out.wait_event(decoded_ev[k])
            capi.check(capi.lib().nbl_copy_to_host_async(C.c_void_p(h_out[k].data_ptr()), C.c_void_p(decs[k].out.data_ptr()),
                                                         int(plan.info.out_bytes), C.c_void_p(copy_out.cuda_stream)))
            out_free[k].record(copy_out)
        torch.cuda.synchronize(device)

    run(n_buffers)
    t0 = time.perf_counter()
    run(stripes)
    dt = (time.perf_counter() - t0) / stripes
    for d in decs:
        d.status()
    sample = min(int(plan.info.out_bytes), 16 << 20)
    for k in range(n_buffers):  # what arrived is what was decoded
        if not (torch.equal(h_out[k][:sample], decs[k].out[:sample].cpu()) and torch.equal(h_out[k][-sample:], decs[k].out[-sample:].cpu())):
            raise RuntimeError("pcie_pipelined: the host copy differs from the decoded output")
    decoded = int(plan.info.decoded_bytes)
    return {"decoded_gbs": round(decoded / dt / 1e9, 2), "ms_per_stripe": round(dt * 1e3, 3), "buffers": n_buffers,
            "stripes": stripes, "h2d_gbs": round(enc / dt / 1e9, 2), "d2h_gbs": round(int(plan.info.out_bytes) / dt / 1e9, 2)}


class Job:
    """The stripes one rank owns, planned and resident on its GPU."""

    def __init__(self, workload, rows, stripe_ids, device, per_kernel=False, keep_chunks=False, packed=None, **plan_options):
        """packed: [(buffer, offsets)] of stripes generated before (then nothing is generated here)."""
        from nimble_amd import workloads
        from nimble_amd.decoder import BatchDecoder, Plan

        self.plans, self.decs, self.chunks, self.packed = [], [], [], []
        t0 = time.perf_counter()
        for i, s in enumerate(stripe_ids):
            if packed is not None:
                chunks, (buf, offs) = None, packed[i]
            else:
                chunks = workloads.WORKLOADS[workload](rows, 42 + s)
                buf, offs = workloads.pack_chunks(chunks)
            plan = Plan(buf, offs, per_kernel=per_kernel, **plan_options)
            dec = BatchDecoder(plan, str(device))
            dec.stage()
            self.plans.append(plan)
            self.decs.append(dec)
            self.chunks.append(chunks if keep_chunks else None)
            self.packed.append((buf, offs))
        self.gen_seconds = time.perf_counter() - t0
        # host cost in front of a stripe's first launch: the header walk + scheduling
        # (nbl_plan_batch) and the descriptor upload (nbl_plan_upload), per stripe
        self.plan_ms = sum(p.plan_seconds for p in self.plans) * 1e3 / max(len(self.plans), 1)
        self.upload_ms = sum(d.upload_seconds for d in self.decs) * 1e3 / max(len(self.decs), 1)
        self.decoded = sum(int(p.info.decoded_bytes) for p in self.plans)
        self.encoded = sum(int(p.info.encoded_bytes) for p in self.plans)
        self.rows = sum(int(p.info.total_rows) for p in self.plans)
        self.launch_list = [l for p in self.plans for l in p.launches()]

    def decode(self):
        for d in self.decs:
            d.decode()

    def status(self):
        for d in self.decs:
            d.status()  # also synchronises; raises if a kernel flagged bad data

    def profile_begin(self, steps):
        for d in self.decs:
            d.profile_begin(steps)

    def decode_profiled(self, s):
        for d in self.decs:
            d.decode_profiled(s)

    def profile_end(self):
        return [ms for d in self.decs for ms in d.profile_end()]

    def close(self):
        self.decs.clear()
        for p in self.plans:
            p.close()


def by_kernel(launch_list, ms_list):
    by = {}
    for (name, es, nt, lvl, alg), ms in zip(launch_list, ms_list):
        k = by.setdefault(name, {"ms": 0.0, "alg": 0, "tiles": 0, "launches": 0})
        k["ms"] += ms
        k["alg"] += alg
        k["tiles"] += nt
        k["launches"] += 1
    return by


def kernel_table(by):
    return [
        {"kernel": k, "ms_per_step": round(v["ms"], 4), "launches": v["launches"], "tiles": v["tiles"],
         "algorithmic_gbs": round(v["alg"] / (v["ms"] * 1e-3) / 1e9, 1) if v["ms"] > 0 and v["alg"] else None}
        for k, v in sorted(by.items(), key=lambda kv: -kv[1]["ms"])
    ]


def roofline_of(by, workload, rows, launches_per_step_of=1):
    """Roofline object of the dominant kernel (largest share of device time):
    achieved = algorithmic bytes per launch / mean launch duration (HIP events on
    the decode stream); traffic = HBM bytes per launch from the committed PMC
    passes of the same workload (FETCH_SIZE and WRITE_SIZE collected in separate
    rocprofv3 runs, corrected by tools/pmc_traffic.py), or null."""
    dom_key = max(by, key=lambda k: by[k]["ms"])
    dom = by[dom_key]
    achieved = dom["alg"] / (dom["ms"] * 1e-3) / 1e9 if dom["ms"] > 0 else 0.0
    roofline = {
        "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
        "kernel": dom_key, "kernel_ms": round(dom["ms"] / max(dom["launches"], 1), 4),
        "launches_per_step": dom["launches"],
        "algorithmic_bytes_per_launch": dom["alg"] // max(dom["launches"], 1),
    }
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        pmc = json.load(open(path))
        kernels = pmc.get("kernels", {})
        k = kernels.get(dom_key) or kernels.get(dom_key.split("<")[0])  # the profiler prints non-template kernels without "<es>"
        if pmc.get("rows") == rows and pmc.get("workload", "fbw_rle") == workload and k:
            roofline["traffic"] = k["traffic_bytes"]
            roofline["traffic_over_algorithmic"] = round(k["traffic_bytes"] / max(roofline["algorithmic_bytes_per_launch"], 1), 4)
            roofline["traffic_source"] = os.path.relpath(path, ROOT)
            break
    # A launch is credited with B_enc + B_out of the chunks it finishes.  For string chunks that includes the string
    # bytes, which the level before copied: where the measured traffic is smaller than the credit, the launch cannot
    # have moved the credited bytes, and its rate is taken from the traffic instead.
    if roofline["traffic"] and roofline["traffic"] < roofline["algorithmic_bytes_per_launch"] and dom["ms"] > 0:
        moved = roofline["traffic"] * max(dom["launches"], 1) / (dom["ms"] * 1e-3) / 1e9
        roofline["achieved"] = round(moved, 1)
        roofline["frac"] = round(moved / HBM_PEAK_GBS, 4)
        roofline["achieved_from"] = "traffic (smaller than the bytes credited to the launch, which include string bytes the level before moved)"
    return roofline


def cold_steps(job, torch, device, steps=5, flush_bytes=1 << 30, per_kernel_out=None):
    """Mean device time of a decode whose inputs are not in any cache: between
    steps 1 GiB is written (four times the 256 MB memory-side cache) and then
    1 GiB of other memory is read, outside the events that bracket each decode.
    The read pass leaves the caches full of clean foreign lines: without it the
    decode also pays for writing the flush's dirty lines back (10-40 us on these
    tables).  A reader decodes every stripe once, so this is its case; the
    steady-state loop re-decodes one resident stripe."""
    flush = torch.empty(flush_bytes, dtype=torch.uint8, device=device)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    flush.fill_(1)
    other = torch.zeros(flush_bytes // 8, dtype=torch.int64, device=device)
    for a, b in evs:
        other.sum()
        a.record()
        job.decode()
        b.record()
        flush.fill_(2)
    torch.cuda.synchronize(device)
    ms = sum(a.elapsed_time(b) for a, b in evs) / steps
    if per_kernel_out is not None:  # the same again with the per-launch events, for the cold time of every launch
        job.profile_begin(steps)
        for s in range(steps):
            flush.fill_(3)
            other.sum()
            job.decode_profiled(s)
        per_kernel_out.extend(kernel_table(by_kernel(job.launch_list, job.profile_end())))
    del flush, other
    return ms


def plan_fields(job, ms_per_step, cold_ms=None):
    """Host planning cost per stripe and the decoded rate of a consumer that pays
    it in front of every decode (single host thread, nothing overlapped)."""
    stripes = max(len(job.plans), 1)
    per_stripe_decode_ms = (cold_ms if cold_ms is not None else ms_per_step) / stripes
    total_ms = job.plan_ms + job.upload_ms + per_stripe_decode_ms
    return {"plan_ms": round(job.plan_ms + job.upload_ms, 3), "plan_batch_ms": round(job.plan_ms, 3),
            "plan_upload_ms": round(job.upload_ms, 3),
            "decoded_gbs_incl_plan": round(job.decoded / stripes / (total_ms * 1e-3) / 1e9, 2)}


def short_run(workload, rows, device, torch, steps=5, warmup=2):
    """One of the other benchmark tables, outside the headline's timed region:
    `steps` passes over one stripe, wall clock between device synchronisations,
    then as many passes with per-launch HIP events for the roofline of its
    dominant kernel."""
    job = Job(workload, rows, [0], device)
    for _ in range(warmup):
        job.decode()
    job.status()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for s in range(steps):
        job.decode()
    torch.cuda.synchronize(device)
    dt = (time.perf_counter() - t0) / steps
    job.profile_begin(steps)  # a second pass with the per-launch events (each costs a few microseconds of stream time)
    for s in range(steps):
        job.decode_profiled(s)
    ms = job.profile_end()
    job.status()
    by = by_kernel(job.launch_list, ms)
    whole = (job.encoded + job.decoded) / (sum(ms) * 1e-3) / 1e9 if sum(ms) > 0 else 0.0
    cold_kernels = []
    cold_ms = cold_steps(job, torch, device, steps, per_kernel_out=cold_kernels)
    job.status()
    res = {
        "workload": WORKLOAD_TEXT[workload].format(rows=rows), "value": round(job.decoded / dt / 1e9, 2), "unit": "GB/s",
        "ms_per_step": round(dt * 1e3, 4), "steps": steps, "rows_per_s": round(job.rows / dt),
        "decoded_bytes": job.decoded, "encoded_bytes": job.encoded, "chunks": len(job.plans[0].layouts),
        "launches_per_step": len(job.launch_list),
        "whole_batch_algorithmic_gbs": round(whole, 1), "whole_batch_hbm_frac": round(whole / HBM_PEAK_GBS, 4),
        "cold_ms_per_step": round(cold_ms, 4), "cold_value": round(job.decoded / (cold_ms * 1e-3) / 1e9, 2),
        "cold_whole_batch_hbm_frac": round((job.encoded + job.decoded) / (cold_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        "roofline": roofline_of(by, workload, rows), "kernels": kernel_table(by), "cold_kernels": cold_kernels,
        "gen_seconds": round(job.gen_seconds, 1),
    }
    res.update(plan_fields(job, dt * 1e3, cold_ms))
    if workload == "wide":
        # the one-wave rank scatter's launch bound (nbl_options.wave_scatter_min_tiles) sides with
        # the cold case: the same stripe with every rank scatter on the four-wave tiles, cold and warm
        packed = job.packed
        job.close()
        job = Job(workload, rows, [0], device, packed=packed, four_wave_scatter=True)
        for _ in range(warmup):
            job.decode()
        job.status()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for s in range(steps):
            job.decode()
        torch.cuda.synchronize(device)
        res["four_wave_only"] = {"ms_per_step": round((time.perf_counter() - t0) / steps * 1e3, 4),
                                 "cold_ms_per_step": round(cold_steps(job, torch, device, steps), 4)}
        job.status()
    job.close()
    return res


def main():
    args = parse_args()
    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and world_env is None:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(world_env or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if args.rank_probe:
        return rank_probe(world, rank)

    import torch
    import torch.distributed as dist

    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend="gloo")
            local_rank %= torch.cuda.device_count()  # rehearsal: ranks share the visible GPUs
    n_gpus = max(world, 1)
    device = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(device)

    from nimble_amd import shard

    args.rows = args.rows or DEFAULT_ROWS[args.workload]
    if args.scaling == "strong":
        if args.workload not in STRONG_STRIPES:
            raise SystemExit("--scaling strong is defined for the fixed tables: --workload mixed | wide")
        n_stripes = args.stripes or STRONG_STRIPES[args.workload]
    else:
        n_stripes = n_gpus  # one stripe of identical shape per GPU
    # ownership by the assignment a file reader uses (longest-processing-time over the
    # footer's stripe sizes; the synthetic stripes have one shape, so one size),
    # computed on every rank without communication
    owned = shard.assign_stripes([args.rows] * n_stripes, n_gpus)[rank]
    job = Job(args.workload, args.rows, owned, device, keep_chunks=(rank == 0))

    for _ in range(args.warmup):
        job.decode()
    job.status()

    # timed region: exactly K steps between barrier + device synchronise on both
    # sides, max over ranks; per-launch HIP events recorded on the decode stream
    # on every `event_every`-th step (an event between two launches holds the stream
    # for about 5 us, which a step of 0.2-0.5 ms would otherwise pay three times)
    every = max(1, args.event_every)
    job.profile_begin((args.steps + every - 1) // every)
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)

    def step(s):
        if s == 0:
            ev0.record()
        if s % every == 0:
            job.decode_profiled(s // every)
        else:
            job.decode()
        if s == args.steps - 1:
            ev1.record()

    elapsed = shard.timed_steps(step, args.steps, lambda: torch.cuda.synchronize(device), dist if world > 1 else None)
    dev_ms = ev0.elapsed_time(ev1) if owned else 0.0
    launch_ms = job.profile_end()
    job.status()

    totals = [job.decoded, job.rows, job.encoded]
    if world > 1:  # whole-job totals (ranks may own different stripes): one scalar reduction, outside the timed region
        t = torch.tensor(totals, dtype=torch.int64)
        if args.backend == "nccl":
            t = t.to(device)
        dist.all_reduce(t)
        totals = [int(x) for x in t.tolist()]
    total_decoded, total_rows, total_encoded = totals
    ms_per_step = elapsed / args.steps * 1e3
    value = total_decoded * args.steps / elapsed / 1e9

    by = by_kernel(job.launch_list, launch_ms)
    roofline = roofline_of(by, args.workload, args.rows) if by else None
    whole = (job.encoded + job.decoded) / (sum(launch_ms) * 1e-3) / 1e9 if sum(launch_ms) > 0 else 0.0
    if rank == 0 and roofline is not None:
        # second denominator: what a plain device-to-device copy sustains on this GPU in this run
        ceiling = copy_ceiling(torch, device)
        roofline["copy_ceiling_gbs"] = round(ceiling, 1)
        roofline["frac_of_copy_ceiling"] = round(roofline["achieved"] / ceiling, 4) if ceiling > 0 else None

    text = WORKLOAD_TEXT[args.workload].format(rows=args.rows)
    text += f"; {n_stripes} stripes dealt over {n_gpus} GPU(s)" if args.scaling == "strong" else "; one stripe per GPU"
    result = {
        "metric": "device-resident decoded GB/s (output bytes) per GPU and per 8xMI355X node",
        "value": round(value, 2), "unit": "GB/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "config": {
            "workload": text, "rows_per_stripe": args.rows, "stripes": n_stripes, "stripes_on_rank0": len(owned),
            "chunks_on_rank0": sum(len(p.layouts) for p in job.plans),
            "nodes_on_rank0": sum(int(p.info.n_nodes) for p in job.plans),
            "launches_per_step_on_rank0": len(job.launch_list), "parallelism": f"stripe-sharded x{n_gpus}",
            "backend": (args.backend if world > 1 else None),
        },
        "rows_per_s": round(total_rows * args.steps / elapsed),
        "decoded_bytes_total": total_decoded, "encoded_bytes_total": total_encoded,
        "decoded_bytes_rank0": job.decoded, "encoded_bytes_rank0": job.encoded,
        "device_ms_per_step": round(dev_ms / args.steps, 4),
        "launch_events_on_steps": (args.steps + every - 1) // every,
        "whole_batch_algorithmic_gbs": round(whole, 1),
        "whole_batch_hbm_frac": round(whole / HBM_PEAK_GBS, 4),
        "roofline": roofline,
        "kernels": kernel_table(by),
        "gen_seconds": round(job.gen_seconds, 1),
    }
    if owned:
        result.update(plan_fields(job, ms_per_step))

    solo = rank == 0 and n_gpus == 1 and len(job.plans) == 1
    if solo and not args.no_cold:
        cold_ms = cold_steps(job, torch, device, 5)
        job.status()
        result["cold_ms_per_step"] = round(cold_ms, 4)
        result["cold_value"] = round(job.decoded / (cold_ms * 1e-3) / 1e9, 2)
        result["cold_whole_batch_hbm_frac"] = round((job.encoded + job.decoded) / (cold_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    if solo and not args.no_breakdown:
        # diagnostic, outside the timed region: the same batch with one launch per
        # (level, operation), which attributes device time to operations
        from nimble_amd.decoder import BatchDecoder, Plan

        plan2 = Plan(job.plans[0].buffer, job.plans[0].chunks, per_kernel=True)
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
                                   "kernels": kernel_table(by_kernel(plan2.launches(), ms2))}
        del dec2
        plan2.close()

    if args.check and rank == 0 and job.plans:
        from oracle import oracle as O
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from util import assert_chunk_equal, oracle_chunk
        from nimble_amd.decoder import slice_chunk
        host = job.decs[0].out.cpu().numpy()
        chunks = job.chunks[0]
        for i in range(0, len(chunks), 7):
            assert_chunk_equal(slice_chunk(host, job.plans[0].layouts[i]), oracle_chunk(O, chunks[i][1]), chunks[i][0])
        result["checked_chunks"] = len(range(0, len(chunks), 7))

    if solo and not args.no_pcie:
        result["pcie_inclusive"] = pcie_inclusive(job.decs[0], job.plans[0], torch, device)
        result["pcie_pipelined"] = pcie_pipelined(job.plans[0], torch, device)
    if solo and not args.no_cpu_baseline:
        chunks = job.chunks[0]
        result["cpu_baseline"] = cpu_baseline(chunks, min_seconds=10.0)
        result["cpu_baseline_elementwise"] = cpu_baseline(chunks, min_seconds=4.0, fast=False)
        # the GPU box gives this job 16 host threads (OMP_NUM_THREADS is set to that share)
        threads = max(1, min(int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or (os.cpu_count() or 1), 16, len(chunks)))
        result["cpu_baseline_all_cores"] = cpu_baseline_threads(chunks, threads)
    if solo and args.workload == "fbw_rle" and not args.no_other_workloads:
        job.close()
        del job
        torch.cuda.empty_cache()
        result["other_workloads"] = {w: short_run(w, DEFAULT_ROWS[w], device, torch) for w in ("dict_string", "mixed", "wide")}
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
