"""Loader for tests/golden/golden_v1.json (hand-assembled known-answer vectors)."""
import json
import os

import numpy as np

from nimble_amd import wire as W

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_v1.json")


def load_cases():
    with open(_PATH) as f:
        doc = json.load(f)
    cases = []
    for c in doc["cases"]:
        dt = c["data_type"]
        enc = bytes.fromhex(c["hex"])
        if dt == W.STRING:
            strs = [bytes.fromhex(s) for s in c["expected_strings_hex"]]
            lengths = np.array([len(s) for s in strs], dtype=np.uint32)
            blob = np.frombuffer(b"".join(strs), dtype=np.uint8)
            expected = (lengths, blob)
        elif dt == W.BOOL:
            expected = np.array(c["expected"], dtype=np.uint8)
        else:
            expected = np.array(c["expected"], dtype=np.uint64).astype(W.physical_dtype(dt))
        nulls = None
        if "expected_non_null" in c:
            nulls = np.array(c["expected_non_null"], dtype=bool)
        cases.append(dict(name=c["name"], enc=enc, data_type=dt, varint_rows=c["varint_rows"], expected=expected,
                          non_null=nulls, source=c["source"], origin=c["origin"], cursor=c.get("cursor", []),
                          dictionary_size=c.get("dictionary_size"), dictionary_entries=c.get("dictionary_entries")))
    return cases


def load_doc():
    with open(_PATH) as f:
        return json.load(f)


def assert_matches(case, values, nulls_bitmap, non_null_count, label):
    dt = case["data_type"]
    exp = case["expected"]
    if dt == W.STRING:
        assert np.array_equal(values[0], exp[0]), f"{label}: string lengths differ"
        assert np.array_equal(values[1], exp[1]), f"{label}: string bytes differ"
    else:
        assert np.array_equal(np.asarray(values), exp), f"{label}: values differ: {values} vs {exp}"
    if case["non_null"] is not None:
        n = case["non_null"].size
        bits = np.unpackbits(np.asarray(nulls_bitmap, dtype=np.uint8), bitorder="little")[:n].astype(bool)
        assert np.array_equal(bits, case["non_null"]), f"{label}: null bitmap differs"
        assert non_null_count == int(case["non_null"].sum()), f"{label}: non-null count"
