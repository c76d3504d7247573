"""Python harness over the C ABI: plan a batch of chunks, stage it on the GPU,
decode, fetch results.  PyTorch is used only for device memory and streams.
"""
from __future__ import annotations

import ctypes as C
import time
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from . import capi
from . import wire as W


@dataclass
class ChunkResult:
    """Decoded output of one chunk, as the reference's Encoding::materialize /
    materializeNullable define it (null rows hold zero / an empty string)."""

    rows: int
    data_type: int
    non_null: int
    values: object  # numeric: ndarray of the physical type; bool: uint8 0/1;
    #                 string: (lengths uint32[rows], blob uint8[...]) in row order
    nulls: Optional[np.ndarray]  # bitmap bytes (1 = non-null), None when the chunk is not Nullable
    views: Optional[np.ndarray] = None  # string: raw {offset,len} pairs as uint64
    string_bytes: Optional[np.ndarray] = None  # string: the chunk's string region


class Plan:
    """nbl_plan_batch over chunks stored in one host buffer."""

    def __init__(self, buffer, chunks: Sequence[tuple[int, int]], varint_rows: bool = False, per_kernel: bool = False,
                 keep_dictionary: bool = False, four_wave_scatter: bool = False, wave_scatter_min_tiles: int = 0,
                 no_lds_dictionary: bool = False, lds_dictionary_min_tiles: int = 0, plan_threads: int = 0):
        """per_kernel: one launch per (level, operation) instead of one fused
        launch per cascade level -- for profiling, same results.
        four_wave_scatter: no rank scatter on the one-wave kernel -- same results.
        wave_scatter_min_tiles: tiles of a level from which the one-wave kernel is
        launched (0: the library's default).
        no_lds_dictionary: every dictionary gathered from memory -- same results.
        lds_dictionary_min_tiles: tiles of a level from which the LDS-alphabet dictionary
        kernel is launched (0: the library's default).
        plan_threads: host threads walking the chunk headers (0: the library's choice, 1: this thread)."""
        self.buffer = capi.as_u8(buffer)
        self.chunks = [(int(o), int(s)) for o, s in chunks]
        n = len(self.chunks)
        offs = np.array([c[0] for c in self.chunks], dtype=np.uint64)
        sizes = np.array([c[1] for c in self.chunks], dtype=np.uint64)
        for o, s in self.chunks:
            if o + s > self.buffer.size:
                raise ValueError("chunk outside the buffer")
        opts = capi.NblOptions(C.sizeof(capi.NblOptions), int(varint_rows),
                               (capi.PLAN_PER_KERNEL_LAUNCHES if per_kernel else 0) |
                               (capi.PLAN_KEEP_DICTIONARY if keep_dictionary else 0) |
                               (capi.PLAN_FOUR_WAVE_SCATTER if four_wave_scatter else 0) |
                               (capi.PLAN_NO_LDS_DICTIONARY if no_lds_dictionary else 0), int(wave_scatter_min_tiles),
                               int(lds_dictionary_min_tiles), int(plan_threads))
        handle = C.c_void_p()
        lib = capi.lib()
        t0 = time.perf_counter()
        status = lib.nbl_plan_batch(self.buffer.ctypes.data, offs.ctypes.data, sizes.ctypes.data, n, C.byref(opts), C.byref(handle))
        self.plan_seconds = time.perf_counter() - t0  # the header walk + scheduling of the whole batch (host only)
        capi.check(status)
        self._h = handle
        info = capi.NblPlanInfo()
        capi.check(capi.lib().nbl_plan_get_info(self._h, C.byref(info)))
        self.info = info
        self.layouts = []
        for i in range(n):
            lay = capi.NblChunkLayout()
            capi.check(capi.lib().nbl_plan_get_chunk_layout(self._h, i, C.byref(lay)))
            self.layouts.append(lay)

    @staticmethod
    def from_encodings(encodings: Sequence[bytes], varint_rows: bool = False, per_kernel: bool = False,
                       keep_dictionary: bool = False, four_wave_scatter: bool = False,
                       wave_scatter_min_tiles: int = 0, no_lds_dictionary: bool = False,
                       lds_dictionary_min_tiles: int = 0, plan_threads: int = 0) -> "Plan":
        """Concatenates standalone encodings (16-byte aligned starts are NOT
        required; they are packed back to back on purpose)."""
        offs, pos = [], 0
        for e in encodings:
            offs.append((pos, len(e)))
            pos += len(e)
        buf = np.frombuffer(b"".join(bytes(e) for e in encodings), dtype=np.uint8) if encodings else np.zeros(0, np.uint8)
        return Plan(buf, offs, varint_rows, per_kernel, keep_dictionary, four_wave_scatter, wave_scatter_min_tiles,
                    no_lds_dictionary, lds_dictionary_min_tiles, plan_threads)

    def fingerprint(self) -> int:
        """Hash of the descriptor tables, launch list, layouts and sizes (nbl_plan_fingerprint)."""
        h = C.c_uint64(0)
        capi.check(capi.lib().nbl_plan_fingerprint(self._h, C.byref(h)))
        return int(h.value)

    def launches(self):
        """[(kernel, elem_bytes, n_tiles, level, alg_bytes)] in launch order."""
        out = []
        for i in range(self.info.n_launches):
            li = capi.NblLaunchInfo()
            capi.check(capi.lib().nbl_plan_get_launch(self._h, i, C.byref(li)))
            out.append((li.kernel.decode(), int(li.elem_bytes), int(li.n_tiles), int(li.level), int(li.alg_bytes)))
        return out

    def close(self):
        if getattr(self, "_h", None):
            capi.lib().nbl_plan_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown: the ctypes module may be gone already
            pass


class BatchDecoder:
    """Device buffers + decode for one Plan (one GPU, one stream)."""

    def __init__(self, plan: Plan, device: str = "cuda:0", stream=None):
        import torch

        if not torch.cuda.is_available():
            raise RuntimeError("BatchDecoder needs a GPU (there is no CPU fallback)")
        self.torch = torch
        self.plan = plan
        self.device = torch.device(device)
        info = plan.info
        with torch.cuda.device(self.device):
            self.arena = torch.zeros(int(info.arena_bytes), dtype=torch.uint8, device=self.device)
            self.out = torch.zeros(int(info.out_bytes), dtype=torch.uint8, device=self.device)
            self.work = torch.zeros(int(info.work_bytes), dtype=torch.uint8, device=self.device)
        self.stream = stream
        self._uploaded = False

    def _stream_ptr(self):
        s = self.stream if self.stream is not None else self.torch.cuda.current_stream(self.device)
        return C.c_void_p(s.cuda_stream)

    def stage(self):
        """H2D of the encoded bytes and of the plan's descriptor tables."""
        torch = self.torch
        host = torch.from_numpy(self.plan.buffer.copy() if not self.plan.buffer.flags.writeable else self.plan.buffer)
        with torch.cuda.device(self.device):
            self.arena[: host.numel()].copy_(host, non_blocking=False)
            t0 = time.perf_counter()
            capi.check(capi.lib().nbl_plan_upload(self.plan._h, C.c_void_p(self.work.data_ptr()), self._stream_ptr()))
            torch.cuda.current_stream(self.device).synchronize() if self.stream is None else self.stream.synchronize()
            self.upload_seconds = time.perf_counter() - t0  # descriptor tables to the device
        self._uploaded = True

    def decode(self):
        """Asynchronous: enqueues every cascade level on the stream."""
        if not self._uploaded:
            self.stage()
        with self.torch.cuda.device(self.device):
            capi.check(capi.lib().nbl_decode_batch_device(
                self.plan._h, C.c_void_p(self.arena.data_ptr()), C.c_void_p(self.out.data_ptr()),
                C.c_void_p(self.work.data_ptr()), self._stream_ptr()))

    def profile_begin(self, max_steps: int):
        self._prof = C.c_void_p()
        capi.check(capi.lib().nbl_profile_create(self.plan._h, max_steps, C.byref(self._prof)))

    def decode_profiled(self, step: int):
        """decode() with a HIP event before and after every launch."""
        with self.torch.cuda.device(self.device):
            capi.check(capi.lib().nbl_decode_batch_profiled(
                self.plan._h, self._prof, step, C.c_void_p(self.arena.data_ptr()), C.c_void_p(self.out.data_ptr()),
                C.c_void_p(self.work.data_ptr()), self._stream_ptr()))

    def profile_end(self):
        """Mean milliseconds per launch over the profiled steps."""
        n = int(self.plan.info.n_launches)
        ms = (C.c_float * max(n, 1))()
        capi.check(capi.lib().nbl_profile_collect(self._prof, ms, n))
        capi.lib().nbl_profile_free(self._prof)
        self._prof = None
        return [float(ms[i]) for i in range(n)]

    def status(self):
        with self.torch.cuda.device(self.device):
            capi.check(capi.lib().nbl_decode_status(self.plan._h, C.c_void_p(self.work.data_ptr()), self._stream_ptr()))

    def fetch(self) -> list[ChunkResult]:
        """D2H of the outputs, split per chunk."""
        self.torch.cuda.synchronize(self.device)
        host = self.out.cpu().numpy()
        return [slice_chunk(host, lay) for lay in self.plan.layouts]


def slice_chunk(host_out: np.ndarray, lay) -> ChunkResult:
    rows = int(lay.rows)
    vals = host_out[int(lay.values_offset): int(lay.values_offset) + int(lay.values_bytes)]
    nulls = None
    if lay.nullable:
        nulls = host_out[int(lay.nulls_offset): int(lay.nulls_offset) + (rows + 7) // 8].copy()
    if lay.data_type == W.STRING:
        views = vals.view(np.uint64).copy()
        region = host_out[int(lay.string_offset): int(lay.string_offset) + int(lay.string_bytes)]
        lengths = (views >> np.uint64(32)).astype(np.uint32)
        offs = (views & np.uint64(0xFFFFFFFF)).astype(np.int64)
        total = int(lengths.sum(dtype=np.uint64))
        if total:
            row_of = np.repeat(np.arange(rows, dtype=np.int64), lengths.astype(np.int64))
            starts = np.zeros(rows + 1, dtype=np.int64)
            np.cumsum(lengths, dtype=np.int64, out=starts[1:])
            src = offs[row_of] + (np.arange(total, dtype=np.int64) - starts[row_of])
            blob = region[src]
        else:
            blob = np.zeros(0, dtype=np.uint8)
        return ChunkResult(rows, int(lay.data_type), int(lay.non_null_rows), (lengths, blob), nulls, views, region.copy())
    if lay.data_type == W.BOOL:
        return ChunkResult(rows, int(lay.data_type), int(lay.non_null_rows), vals.copy(), nulls)
    return ChunkResult(rows, int(lay.data_type), int(lay.non_null_rows),
                       vals.view(W.physical_dtype(lay.data_type)).copy(), nulls)


def decode_encodings(encodings: Sequence[bytes], varint_rows: bool = False, device: str = "cuda:0",
                     per_kernel: bool = False, four_wave_scatter: bool = False,
                     wave_scatter_min_tiles: int = 0, no_lds_dictionary: bool = False,
                     lds_dictionary_min_tiles: int = 0) -> list[ChunkResult]:
    """Convenience: plan + stage + decode + status + fetch for standalone encodings."""
    plan = Plan.from_encodings(encodings, varint_rows, per_kernel, four_wave_scatter=four_wave_scatter,
                               wave_scatter_min_tiles=wave_scatter_min_tiles, no_lds_dictionary=no_lds_dictionary,
                               lds_dictionary_min_tiles=lds_dictionary_min_tiles)
    dec = BatchDecoder(plan, device)
    dec.stage()
    dec.decode()
    dec.status()
    res = dec.fetch()
    plan.close()
    return res
