"""In-tree build of the native library (hipcc, gfx950 only).

`python -m nimble_amd.build` or `nimble_amd.build.build()` compiles
csrc/*.hip + csrc/*.cpp into nimble_amd/_lib/libnimble_amd.so.  hipcc
cross-compiles without a GPU, so this also runs on a CPU-only machine.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "_lib")
LIB_PATH = os.path.join(LIB_DIR, "libnimble_amd.so")
HOST_LIB_PATH = os.path.join(LIB_DIR, "libnimble_amd_host.so")
ARCH = "gfx950"

_DEVICE_SOURCES = ["nbl_kernels.hip", "nbl_plan.cpp", "nbl_inflate.cpp"]
_HEADERS = ["nbl_desc.h", os.path.join("..", "..", "include", "nimble_amd.h")]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the native library cannot be built")
    return exe


def _stale(target: str, sources) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources if os.path.exists(s))


def source_id() -> str:
    """Hash of every source and header the library is built from; compiled into
    it (nbl_build_source_id), so a test can tell a stale binary."""
    import hashlib

    h = hashlib.sha256()
    for name in _DEVICE_SOURCES + _HEADERS:
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def _obj(src: str) -> str:
    return os.path.join(LIB_DIR, "obj", os.path.basename(src) + ".o")


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP/C++ source for gfx950 and link the shared library."""
    os.makedirs(os.path.join(LIB_DIR, "obj"), exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in _HEADERS]
    objs = []
    relink = force
    sid = source_id()
    sid_path = os.path.join(LIB_DIR, "obj", "source_id.txt")
    sid_changed = not os.path.exists(sid_path) or open(sid_path).read().strip() != sid
    for name in _DEVICE_SOURCES:
        src = os.path.join(CSRC, name)
        obj = _obj(src)
        objs.append(obj)
        carries_id = name == "nbl_plan.cpp"  # the translation unit that holds nbl_build_source_id()
        if force or _stale(obj, [src] + headers) or (carries_id and sid_changed):
            cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall",
                   "-Wno-unused-function", "-Wno-pass-failed", "-c", src, "-o", obj]
            if carries_id:
                cmd.insert(-4, f'-DNBL_SOURCE_ID="{sid}"')
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
            relink = True
    if relink or _stale(LIB_PATH, objs):
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_PATH] + objs + [
            "-ldl", "-Wl,-rpath,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    with open(sid_path, "w") as f:
        f.write(sid + "\n")
    return LIB_PATH


def build_host(force: bool = False, verbose: bool = False):
    """Compile the C++ host shim (Encoding / EncodingFactory / Decoder mirror)
    when its sources are present; returns the library path or None."""
    src_dir = os.path.join(_HERE, "host")
    srcs = sorted(os.path.join(src_dir, f) for f in os.listdir(src_dir) if f.endswith(".cpp")) if os.path.isdir(src_dir) else []
    if not srcs:
        return None
    hdrs = [os.path.join(src_dir, f) for f in os.listdir(src_dir) if f.endswith(".h")]
    hdrs.append(os.path.join(_HERE, "..", "include", "nimble_amd.h"))
    build(force=False, verbose=verbose)
    if force or _stale(HOST_LIB_PATH, srcs + hdrs + [LIB_PATH]):
        cmd = [_hipcc(), "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", HOST_LIB_PATH] + srcs + [
            f"-I{os.path.join(_HERE, '..', 'include')}", f"-L{LIB_DIR}", "-lnimble_amd",
            "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return HOST_LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
