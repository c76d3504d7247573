"""CPU: the planner never crashes on damaged or truncated input: every outcome is
a plan, NBL_ERR_CORRUPT or NBL_ERR_UNSUPPORTED (the reference throws from
NIMBLE_CHECK in the same places)."""
import numpy as np

from nimble_amd import capi
from nimble_amd.decoder import Plan

from fuzz_util import random_chunk


def test_random_cascades_plan_and_decode_on_the_oracle(oracle):
    rng = np.random.default_rng(5)
    for _ in range(150):
        label, enc = random_chunk(rng, max_rows=800)
        p = Plan.from_encodings([enc])
        r = oracle.decode_chunk(enc)
        assert r["rows"] == p.layouts[0].rows, label
        assert r["non_null"] == p.layouts[0].non_null_rows, label
        p.close()


def test_damaged_and_truncated_input_is_rejected_cleanly():
    rng = np.random.default_rng(6)
    outcomes = {"ok": 0, "corrupt": 0, "unsupported": 0}
    for i in range(1500):
        _, enc = random_chunk(rng, max_rows=300)
        b = bytearray(enc)
        if i % 3 == 0:
            b = b[: int(rng.integers(0, len(b)))]
        else:
            for _ in range(int(rng.integers(1, 6))):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        try:
            Plan.from_encodings([bytes(b)]).close()
            outcomes["ok"] += 1
        except capi.NimbleCorruptError:
            outcomes["corrupt"] += 1
        except capi.NimbleUnsupportedError:
            outcomes["unsupported"] += 1
    assert outcomes["corrupt"] > 300 and outcomes["ok"] > 100, outcomes
