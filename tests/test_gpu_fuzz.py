"""GPU: randomly composed cascades (every legal nesting, random data shapes)
decoded in one batch and compared with the oracle; and the same encodings with
random byte damage, which must be rejected by the planner or decoded without a
fault (inconsistent data is reported through the status word)."""
import numpy as np
import pytest

from nimble_amd import capi
from nimble_amd.decoder import BatchDecoder, Plan

from fuzz_util import random_chunk
from util import check_parity

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(8))
def test_random_cascades_match_the_oracle(gpu, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    made = [random_chunk(rng, max_rows=(200_000 if seed == 7 else 6000)) for _ in range(12 if seed == 7 else 70)]
    check_parity(oracle, [e for _, e in made], labels=[l for l, _ in made])


def test_damaged_bytes_never_fault(gpu):
    rng = np.random.default_rng(77)
    survivors, rejected = [], 0
    for _ in range(400):
        _, enc = random_chunk(rng, max_rows=3000)
        b = bytearray(enc)
        for _ in range(int(rng.integers(1, 4))):
            pos = int(rng.integers(len(b) // 3, len(b)))  # mostly payload, sometimes a nested header
            b[pos] = int(rng.integers(0, 256))
        try:
            p = Plan.from_encodings([bytes(b)])
            # a damaged row count can ask for gigabytes: keep the batch small
            if int(p.info.out_bytes) + int(p.info.work_bytes) < (64 << 20):
                survivors.append(bytes(b))
            p.close()
        except (capi.NimbleCorruptError, capi.NimbleUnsupportedError):
            rejected += 1
    assert len(survivors) > 100 and rejected > 0
    for per_kernel, wave_min in ((False, 0), (True, 0), (False, 1)):  # the last: rank scatters on the one-wave kernel
        plan = Plan.from_encodings(survivors, per_kernel=per_kernel, wave_scatter_min_tiles=wave_min)
        dec = BatchDecoder(plan)
        dec.stage()
        dec.decode()
        try:
            dec.status()
        except capi.NimbleError as e:
            assert e.status == capi.NBL_ERR_DEVICE_DATA
        import torch

        torch.cuda.synchronize()
        assert dec.out.numel() == int(plan.info.out_bytes)
        plan.close()
