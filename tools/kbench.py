#!/usr/bin/env python3
"""Per-kernel throughput of single cascades (tuning aid, needs a GPU).
Each case is one chunk replicated until ~`target` rows are decoded in one batch.
Usage: python tools/kbench.py [case-substring] [target_rows]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

from nimble_amd import encoder as E, wire as W, workloads as WL  # noqa: E402
from nimble_amd.decoder import BatchDecoder, Plan  # noqa: E402

rng = np.random.default_rng(0)
N = 2_000_000
PER_KERNEL = os.environ.get("NBL_PER_KERNEL", "1") != "0"  # NBL_PER_KERNEL=0: fused launches


def nullable(dtype, npt, frac_null, data_spec, nulls_spec, n=N):
    mask = rng.random(n) >= frac_null
    v = rng.integers(0, 1 << 20, int(mask.sum())).astype(npt)
    return E.encode_nullable(E.Nullable(data=data_spec, nulls=nulls_spec), v, mask, dtype)


def mainly(dtype, npt, frac_common, is_common_spec, n=N):
    ic = rng.random(n) < frac_common
    other = rng.integers(0, 1 << 30, int((~ic).sum())).astype(npt)
    return E.mainly_constant_from_parts(E.MainlyConstant(is_common=is_common_spec, other=E.Trivial()), ic, other, npt(5), dtype)


def dict_num(dtype, npt, A, n=N):
    alpha = rng.permutation(1 << 22)[:A].astype(npt)
    return E.encode_dictionary_from_indices(E.Dictionary(alphabet=E.Trivial(), indices=E.FixedBitWidth()), alpha,
                                            rng.integers(0, A, n), dtype)


def dict_str(A, n=700_000):
    alpha = WL.random_alphabet(rng, A)
    return E.encode_dictionary_from_indices(E.Dictionary(alphabet=E.Trivial(lengths=E.FixedBitWidth()), indices=E.FixedBitWidth()),
                                            alpha, rng.integers(0, len(alpha), n), W.STRING)


def rle(dtype, npt, lo, hi, n=N):
    return E.encode(E.RLE(), WL.run_length_values(rng, n, lo, hi).astype(npt), dtype)


CASES = {
    "nullable/i64/5%/trivial+sparse": lambda: nullable(W.INT64, np.uint64, 0.05, E.Trivial(), E.SparseBool()),
    "nullable/i64/50%/trivial+bitmap": lambda: nullable(W.INT64, np.uint64, 0.5, E.Trivial(), E.Trivial()),
    "nullable/i64/50%/fbw+bitmap": lambda: nullable(W.INT64, np.uint64, 0.5, E.FixedBitWidth(), E.Trivial()),
    "nullable/i32/5%/fbw+sparse": lambda: nullable(W.INT32, np.uint32, 0.05, E.FixedBitWidth(), E.SparseBool()),
    "nullable/f32/99%/trivial+sparse": lambda: nullable(W.FLOAT, np.uint32, 0.99, E.Trivial(), E.SparseBool()),
    "nullable/f32/20%/trivial+bitmap": lambda: nullable(W.FLOAT, np.uint32, 0.2, E.Trivial(), E.Trivial()),
    "small/nullable/f32/99%/100K": lambda: nullable(W.FLOAT, np.uint32, 0.99, E.Trivial(), E.SparseBool(), n=100_000),
    "small/nullable/f32/80%/100K": lambda: nullable(W.FLOAT, np.uint32, 0.80, E.Trivial(), E.SparseBool(), n=100_000),
    "small/nullable/f32/20%/100K": lambda: nullable(W.FLOAT, np.uint32, 0.20, E.Trivial(), E.Trivial(), n=100_000),
    "small/fbw/i32/w13/100K": lambda: E.encode(E.FixedBitWidth(bit_width=13), rng.integers(0, 1 << 13, 100_000).astype(np.uint32), W.INT32),
    "small/fbw/i32/w13/10K": lambda: E.encode(E.FixedBitWidth(bit_width=13), rng.integers(0, 1 << 13, 10_000).astype(np.uint32), W.INT32),
    "mainly/i64/95%/sparse": lambda: mainly(W.INT64, np.uint64, 0.95, E.SparseBool()),
    "mainly/f32/70%/bitmap": lambda: mainly(W.FLOAT, np.uint32, 0.7, E.Trivial()),
    "dict/i64/A64": lambda: dict_num(W.INT64, np.uint64, 64),
    "dict/i64/A4096": lambda: dict_num(W.INT64, np.uint64, 4096),
    "dict/i64/A65536": lambda: dict_num(W.INT64, np.uint64, 65536),
    "dict/i32/A1024": lambda: dict_num(W.INT32, np.uint32, 1024),
    "dict/str/A64": lambda: dict_str(64),
    "dict/str/A4096": lambda: dict_str(4096),
    "dict/str/A65536": lambda: dict_str(65536),
    "rle/i64/10-60": lambda: rle(W.INT64, np.uint64, 10, 60),
    "rle/i32/10-60": lambda: rle(W.INT32, np.uint32, 10, 60),
    "rle/i64/2-4": lambda: rle(W.INT64, np.uint64, 2, 4),
    "rle/i64/1000-5000": lambda: rle(W.INT64, np.uint64, 1000, 5000),
    "bool/trivial": lambda: E.encode(E.Trivial(), rng.random(N * 4) < 0.5, W.BOOL),
    "bool/sparse3%": lambda: E.encode(E.SparseBool(), rng.random(N * 4) < 0.03, W.BOOL),
    "bool/rle": lambda: E.encode(E.RLE(), np.repeat(np.arange(N // 5) & 1, 20).astype(bool), W.BOOL),
    "varint/i64": lambda: E.encode(E.Varint(), rng.integers(0, 1 << 20, N).astype(np.uint64), W.INT64),
    "varint/i32": lambda: E.encode(E.Varint(), rng.integers(0, 1 << 14, N).astype(np.uint32), W.INT32),
    "const/i64": lambda: E.encode(E.Constant(), np.full(N, 7, dtype=np.uint64), W.INT64),
    "block/bbp/i64/narrow": lambda: E.encode(E.BlockBitPacking(), (rng.integers(0, 1 << 13, N) + (1 << 40)).astype(np.uint64), W.INT64),
    "block/bbp/i32/narrow": lambda: E.encode(E.BlockBitPacking(), (rng.integers(0, 1 << 13, N) + (1 << 20)).astype(np.uint32), W.INT32),
    "block/bbp/i64/raw": lambda: E.encode(E.BlockBitPacking(), rng.integers(0, 1 << 63, N, dtype=np.uint64) * np.uint64(2), W.INT64),
    "block/for/i64/narrow": lambda: E.encode(E.FOR(), (rng.integers(0, 1 << 13, N) + (1 << 40)).astype(np.uint64), W.INT64),
    "block/for/i32/narrow": lambda: E.encode(E.FOR(), (rng.integers(0, 1 << 13, N) + (1 << 20)).astype(np.uint32), W.INT32),
    "fbw/i64/w13": lambda: E.encode(E.FixedBitWidth(bit_width=13), rng.integers(0, 1 << 13, N).astype(np.uint64), W.INT64),
    "fbw/i32/w13": lambda: E.encode(E.FixedBitWidth(bit_width=13), rng.integers(0, 1 << 13, N).astype(np.uint32), W.INT32),
    "fbw/i32/w16": lambda: E.encode(E.FixedBitWidth(bit_width=16), rng.integers(0, 1 << 16, N).astype(np.uint32), W.INT32),
    "string/trivial": lambda: E.encode(E.Trivial(lengths=E.FixedBitWidth()), WL.random_alphabet(rng, 400_000), W.STRING),
}


def main():
    sel = sys.argv[1] if len(sys.argv) > 1 else ""
    target = int(sys.argv[2]) if len(sys.argv) > 2 else 40_000_000
    for name, make in CASES.items():
        if sel not in name:
            continue
        enc = make()
        n = int(np.frombuffer(enc[2:6], dtype=np.uint32)[0])
        k = max(1, target // n)
        shift = int(os.environ.get("NBL_SHIFT", "3"))  # chunk starts at 16*n + shift: misaligned on purpose
        blob = enc + bytes((-len(enc)) % 16)
        buf = np.frombuffer(bytes(shift) + blob * k, dtype=np.uint8)
        plan = Plan(buf, [(shift + i * len(blob), len(enc)) for i in range(k)], per_kernel=PER_KERNEL)
        dec = BatchDecoder(plan)
        dec.stage()
        for _ in range(3):
            dec.decode()
        dec.status()
        steps = 10
        dec.profile_begin(steps)
        for s in range(steps):
            dec.decode_profiled(s)
        ms = dec.profile_end()
        info = plan.info
        t = sum(ms) * 1e-3
        alg = int(info.encoded_bytes) + int(info.decoded_bytes)
        print(f"{name:34s} rows={n * k:>9d} alg {alg / t / 1e9:7.1f} GB/s  out {int(info.decoded_bytes) / t / 1e9:7.1f} GB/s  {t * 1e6:8.1f} us  "
              + " ".join(f"{l[0][2:]}={m * 1e3:.0f}" for l, m in zip(plan.launches(), ms)), flush=True)
        plan.close()
        del dec


if __name__ == "__main__":
    main()
