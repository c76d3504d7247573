"""GPU parity for nodes whose running counts span many scan tiles: every consumer
of the three-phase scan (run ends, rank scatter of values and of bools, varint
decode, string views) decoded with about a thousand and about three thousand
tile partials per node, so that the one-workgroup scan of the partials loops
several times over its 256 threads."""
import numpy as np
import pytest

from nimble_amd import encoder as E
from nimble_amd import wire as W

from util import check_parity, random_strings

pytestmark = pytest.mark.gpu


def _cases(rng, scale):
    """about 1024 * scale scan tiles per node"""
    n_sel = 2048 * 1024 * scale - 5          # rank scatter of 8-byte rows: 2048 rows per tile
    mask = rng.random(n_sel) < 0.4
    yield "nullable/i64", E.encode_nullable(E.Nullable(data=E.FixedBitWidth(), nulls=E.Trivial()),
                                            rng.integers(0, 1000, int(mask.sum())).astype(np.int64), mask, W.INT64)
    common = rng.random(n_sel) < 0.9
    yield "mainly/f32", E.mainly_constant_from_parts(E.MainlyConstant(is_common=E.Trivial(), other=E.Trivial()), common,
                                                     rng.integers(0, 2**31, int((~common).sum())).astype(np.uint32), np.uint32(7), W.FLOAT)
    runs = 2048 * 1024 * scale - 3           # run ends: 2048 runs per tile
    lengths = rng.integers(1, 4, runs)
    yield "rle/i32", E.encode(E.RLE(), np.repeat(rng.integers(0, 1 << 20, runs), lengths).astype(np.int32), W.INT32)
    yield "rle/bool", E.encode(E.RLE(), np.repeat(np.arange(runs) & 1, lengths).astype(bool), W.BOOL)
    n_var = 2_200_000 * scale                # varint: 4096 payload bytes per tile, about 1.9 bytes per value
    yield "varint/i64", E.encode(E.Varint(), rng.integers(0, 1 << 10, n_var).astype(np.uint64), W.INT64)
    n_str = 2048 * 1024 * scale - 1          # string views: 2048 rows per tile
    alpha = random_strings(rng, 50, max_len=6)
    yield "string/trivial", E.encode(E.Trivial(), alpha.take(rng.integers(0, 50, n_str)), W.STRING)
    n_bool = 16384 * 1024 * scale // (2 if scale > 1 else 1) - 9   # rank scatter of bools: 16384 rows per tile
    bmask = rng.random(n_bool) < 0.7
    yield "nullable/bool", E.encode_nullable(E.Nullable(data=E.Trivial(), nulls=E.Trivial()), rng.random(int(bmask.sum())) < 0.5, bmask, W.BOOL)


@pytest.mark.parametrize("scale", [1, 3])
def test_consumers_of_long_scans(gpu, oracle, scale):
    rng = np.random.default_rng(90 + scale)
    made = list(_cases(rng, scale))
    for label, enc in made:  # one chunk per batch keeps device memory modest
        check_parity(oracle, [enc], labels=[f"{label} x{scale}"])
