"""ctypes binding of the C ABI declared in include/nimble_amd.h.

The product path has no CPU fallback: if the native library is missing or a
symbol cannot be resolved, importing this module's `lib()` raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import build as _build

NBL_OK = 0
NBL_ERR_CORRUPT = 1
NBL_ERR_UNSUPPORTED = 2
NBL_ERR_HIP = 3
NBL_ERR_ARGUMENT = 4
NBL_ERR_DEVICE_DATA = 5

# every entry point include/nimble_amd.h declares
EXPORTED_SYMBOLS = (
    "nbl_plan_batch", "nbl_plan_free", "nbl_plan_get_info", "nbl_plan_get_chunk_layout",
    "nbl_split_stream", "nbl_plan_upload", "nbl_decode_batch_device", "nbl_decode_status",
    "nbl_plan_get_launch", "nbl_last_error", "nbl_build_arch", "nbl_build_source_id",
    "nbl_profile_create", "nbl_profile_free", "nbl_decode_batch_profiled", "nbl_profile_collect",
    "nbl_plan_fingerprint", "nbl_split_stream_ex", "nbl_uncompressed_size", "nbl_uncompress", "nbl_chunk_inflated_size", "nbl_chunk_inflate",
    "nbl_copy_to_host_async",
)


class NblOptions(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("use_varint_row_count", C.c_uint32), ("flags", C.c_uint32),
                ("wave_scatter_min_tiles", C.c_uint32), ("lds_dictionary_min_tiles", C.c_uint32),
                ("plan_threads", C.c_uint32)]


PLAN_PER_KERNEL_LAUNCHES = 1
PLAN_KEEP_DICTIONARY = 2
PLAN_FOUR_WAVE_SCATTER = 4
PLAN_NO_LDS_DICTIONARY = 8


class NblPlanInfo(C.Structure):
    _fields_ = [
        ("n_chunks", C.c_uint32), ("n_nodes", C.c_uint32), ("n_launches", C.c_uint32), ("max_depth", C.c_uint32),
        ("arena_bytes", C.c_uint64), ("out_bytes", C.c_uint64), ("work_bytes", C.c_uint64),
        ("descriptor_bytes", C.c_uint64), ("total_rows", C.c_uint64), ("encoded_bytes", C.c_uint64),
        ("decoded_bytes", C.c_uint64),
    ]


class NblChunkLayout(C.Structure):
    _fields_ = [
        ("rows", C.c_uint32), ("encoding_type", C.c_uint8), ("data_type", C.c_uint8), ("nullable", C.c_uint8),
        ("elem_bytes", C.c_uint8), ("non_null_rows", C.c_uint32), ("reserved", C.c_uint32),
        ("values_offset", C.c_uint64), ("values_bytes", C.c_uint64), ("nulls_offset", C.c_uint64),
        ("nulls_bytes", C.c_uint64), ("string_offset", C.c_uint64), ("string_bytes", C.c_uint64),
        ("has_dictionary", C.c_uint32), ("dict_size", C.c_uint32), ("dict_alphabet_offset", C.c_uint64),
        ("dict_indices_offset", C.c_uint64), ("dict_indices_rows", C.c_uint64),
    ]


class NblLaunchInfo(C.Structure):
    _fields_ = [("kernel", C.c_char_p), ("elem_bytes", C.c_uint32), ("n_tiles", C.c_uint32),
                ("level", C.c_uint32), ("reserved", C.c_uint32), ("alg_bytes", C.c_uint64)]


class NimbleError(RuntimeError):
    """Raised for every non-zero nbl_status; `.status` holds the code."""

    def __init__(self, status: int, message: str):
        super().__init__(f"[nbl_status {status}] {message}")
        self.status = status
        self.message = message


class NimbleCorruptError(NimbleError):
    """Malformed bytes (the reference's NIMBLE_CHECK / NIMBLE_UNREACHABLE failures)."""


class NimbleUnsupportedError(NimbleError):
    """Valid Nimble the GPU path does not decode; delegate to the CPU factory."""


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.environ.get("NIMBLE_AMD_LIB", _build.LIB_PATH)  # override: A/B builds while tuning
        if not os.path.exists(path):
            raise ImportError(
                f"{path} is missing: run `python -m nimble_amd.build` (the decoder has no CPU fallback)")
        # PyTorch bundles its own HIP runtime; it must be the one already in the
        # process when this library resolves libamdhip64, or the process ends up
        # with two runtimes and ours sees no device.
        import torch  # noqa: F401

        L = C.CDLL(path)
        vp, u32, u64 = C.c_void_p, C.c_uint32, C.c_uint64
        L.nbl_last_error.restype = C.c_char_p
        L.nbl_build_arch.restype = C.c_char_p
        L.nbl_build_source_id.restype = C.c_char_p
        L.nbl_plan_batch.restype = C.c_int
        L.nbl_plan_batch.argtypes = [vp, vp, vp, u32, C.POINTER(NblOptions), C.POINTER(vp)]
        L.nbl_plan_free.argtypes = [vp]
        L.nbl_plan_free.restype = None
        L.nbl_plan_get_info.restype = C.c_int
        L.nbl_plan_get_info.argtypes = [vp, C.POINTER(NblPlanInfo)]
        L.nbl_plan_get_chunk_layout.restype = C.c_int
        L.nbl_plan_get_chunk_layout.argtypes = [vp, u32, C.POINTER(NblChunkLayout)]
        L.nbl_split_stream.restype = C.c_int
        L.nbl_split_stream.argtypes = [vp, u64, vp, vp, u32, C.POINTER(u32)]
        L.nbl_split_stream_ex.restype = C.c_int
        L.nbl_split_stream_ex.argtypes = [vp, u64, vp, vp, vp, u32, C.POINTER(u32)]
        L.nbl_uncompressed_size.restype = C.c_int
        L.nbl_uncompressed_size.argtypes = [C.c_uint8, vp, u64, C.POINTER(u64)]
        L.nbl_uncompress.restype = C.c_int
        L.nbl_uncompress.argtypes = [C.c_uint8, vp, u64, vp, u64]
        L.nbl_chunk_inflated_size.restype = C.c_int
        L.nbl_chunk_inflated_size.argtypes = [vp, u64, C.POINTER(NblOptions), C.POINTER(u64)]
        L.nbl_chunk_inflate.restype = C.c_int
        L.nbl_chunk_inflate.argtypes = [vp, u64, C.POINTER(NblOptions), vp, u64, C.POINTER(u64)]
        L.nbl_plan_upload.restype = C.c_int
        L.nbl_plan_upload.argtypes = [vp, vp, vp]
        L.nbl_decode_batch_device.restype = C.c_int
        L.nbl_decode_batch_device.argtypes = [vp, vp, vp, vp, vp]
        L.nbl_decode_status.restype = C.c_int
        L.nbl_decode_status.argtypes = [vp, vp, vp]
        L.nbl_copy_to_host_async.restype = C.c_int
        L.nbl_copy_to_host_async.argtypes = [vp, vp, u64, vp]
        L.nbl_plan_get_launch.restype = C.c_int
        L.nbl_plan_get_launch.argtypes = [vp, u32, C.POINTER(NblLaunchInfo)]
        L.nbl_plan_fingerprint.restype = C.c_int
        L.nbl_plan_fingerprint.argtypes = [vp, C.POINTER(u64)]
        L.nbl_profile_create.restype = C.c_int
        L.nbl_profile_create.argtypes = [vp, u32, C.POINTER(vp)]
        L.nbl_profile_free.restype = None
        L.nbl_profile_free.argtypes = [vp]
        L.nbl_decode_batch_profiled.restype = C.c_int
        L.nbl_decode_batch_profiled.argtypes = [vp, vp, u32, vp, vp, vp, vp]
        L.nbl_profile_collect.restype = C.c_int
        L.nbl_profile_collect.argtypes = [vp, C.POINTER(C.c_float), u32]
        _lib = L
    return _lib


def check(status: int):
    if status == NBL_OK:
        return
    msg = lib().nbl_last_error().decode(errors="replace")
    if status == NBL_ERR_CORRUPT:
        raise NimbleCorruptError(status, msg)
    if status == NBL_ERR_UNSUPPORTED:
        raise NimbleUnsupportedError(status, msg)
    raise NimbleError(status, msg)


def as_u8(data) -> np.ndarray:
    """Contiguous uint8 view/copy of bytes-like input."""
    if isinstance(data, np.ndarray):
        a = np.ascontiguousarray(data)
        return a.view(np.uint8).reshape(-1)
    return np.frombuffer(data, dtype=np.uint8)


def split_stream(stream) -> list[tuple[int, int]]:
    """(offset, size) of every chunk payload of a stream (nbl_split_stream)."""
    a = as_u8(stream)
    n = C.c_uint32(0)
    check(lib().nbl_split_stream(a.ctypes.data, a.size, None, None, 0, C.byref(n)))
    offs = np.zeros(max(n.value, 1), dtype=np.uint64)
    sizes = np.zeros(max(n.value, 1), dtype=np.uint64)
    check(lib().nbl_split_stream(a.ctypes.data, a.size, offs.ctypes.data, sizes.ctypes.data, n.value, C.byref(n)))
    return [(int(offs[i]), int(sizes[i])) for i in range(n.value)]


def inflate_chunk(data, varint_rows: bool = False) -> bytes:
    """The chunk with every compressed leaf payload inflated (nbl_chunk_inflate);
    returns the input unchanged when nothing in it is compressed."""
    a = as_u8(data)
    opts = NblOptions(C.sizeof(NblOptions), int(varint_rows), 0)
    need = C.c_uint64(0)
    check(lib().nbl_chunk_inflated_size(a.ctypes.data, a.size, C.byref(opts), C.byref(need)))
    if need.value == 0:
        return bytes(data)
    out = np.empty(need.value, dtype=np.uint8)
    got = C.c_uint64(0)
    check(lib().nbl_chunk_inflate(a.ctypes.data, a.size, C.byref(opts), out.ctypes.data, out.size, C.byref(got)))
    return out[: got.value].tobytes()
