"""GPU: the HIP decode against the hand-assembled known-answer vectors
(tests/golden/golden_v1.json), all cases in one batch and one by one."""
import pytest

from golden_util import assert_matches, load_cases

pytestmark = pytest.mark.gpu


def test_golden_vectors_one_batch_per_prefix_mode(gpu):
    from nimble_amd.decoder import decode_encodings

    cases = load_cases()
    for vr in (False, True):
        group = [c for c in cases if c["varint_rows"] == vr]
        results = decode_encodings([c["enc"] for c in group], varint_rows=vr)
        for c, r in zip(group, results):
            assert r.data_type == c["data_type"], c["name"]
            assert_matches(c, r.values, r.nulls, r.non_null, c["name"])


def test_golden_vectors_single_chunk_plans(gpu):
    from nimble_amd.decoder import decode_encodings

    for c in load_cases():
        (r,) = decode_encodings([c["enc"]], varint_rows=c["varint_rows"])
        assert_matches(c, r.values, r.nulls, r.non_null, c["name"])


def test_reference_frames_through_the_host_layer(gpu):
    """The frames whose answers the reference's tests assert, through the C++ Encoding
    mirror: full read, the reset / skip / materialize reads of
    MainlyConstantEncodingTest.cpp:456-462, and the dictionary sizes and entries of
    ReadWithVisitorTest.cpp:3483,3573 and EncodingTest.cpp:1530,1574 where this layer
    answers the dictionary API (Dictionary and Nullable{Dictionary})."""
    import numpy as np

    import host_shim
    from nimble_amd import wire as W

    for c in load_cases():
        if c["origin"] != "reference-frame":
            continue
        f = host_shim.Factory(varint_rows=c["varint_rows"], keep_dictionary=c["dictionary_size"] is not None)
        e = f.create(c["enc"])
        n = e.row_count
        vals = e.materialize(n)
        case_plain = dict(c, non_null=None)  # materialize() fills null rows with physicalType(); the bitmap is checked below
        assert_matches(case_plain, vals, None, None, c["name"])
        if c["non_null"] is not None:
            assert e.is_nullable and e.null_count == int((~c["non_null"]).sum()), c["name"]
        for step in c["cursor"]:
            e.reset()
            e.skip(step["skip"])
            got = e.materialize(step["materialize"])
            assert [int(v) for v in got.astype(np.int64)] == step["expected"], c["name"]
        if c["dictionary_size"] is not None:
            assert e.dictionary_enabled, c["name"]
            assert e.dictionary_size() == c["dictionary_size"], c["name"]
            entries = e.dictionary_entries()
            if c["data_type"] == W.STRING:
                lengths, blob = entries
                got, pos = [], 0
                for ln in lengths.tolist():
                    got.append(bytes(blob[pos:pos + ln]).hex())
                    pos += ln
            else:
                got = [int(v) for v in entries.astype(np.uint64)]
            assert sorted(got) == c["dictionary_entries"], c["name"]
            # indices of the non-null rows point at their values (ReadWithVisitorTest.cpp:3477-3482)
            e.reset()


def test_reference_run_and_sparse_facts_decode_on_the_gpu(gpu):
    """RLETests.cpp:42-169 and SparseBoolEncodingTest.cpp:145-216: an RLE of the
    asserted runs decodes to the data, a SparseBool of the asserted polarity and
    positions to the bools."""
    import struct

    import numpy as np

    from golden_util import load_doc
    from nimble_amd import encoder as E
    from nimble_amd import wire as W
    from nimble_amd.decoder import decode_encodings

    doc = load_doc()
    encs, want = [], []
    for e in doc["rle_runs"]["entries"]:
        dt = e["data_type"]
        npt = W.physical_dtype(dt)
        lens = E.encode(E.Trivial(), np.array(e["lengths"], dtype=np.uint32), W.UINT32)
        vals = E.encode(E.Trivial(), np.array(e["values"], dtype=np.int64).astype(npt), dt)
        encs.append(E.prefix(W.RLE, dt, len(e["data"])) + struct.pack("<I", len(lens)) + lens + vals)
        want.append(np.array(e["data"], dtype=np.int64).astype(npt))
    for e in doc["sparse_bool_facts"]["entries"]:
        idx = E.encode(E.Trivial(), np.array(e["positions"] + [len(e["values"])], dtype=np.uint32), W.UINT32)
        encs.append(E.prefix(W.SPARSE_BOOL, W.BOOL, len(e["values"])) + bytes([e["sparse_value"]]) + idx)
        want.append(np.array(e["values"], dtype=np.uint8))
    for r, w in zip(decode_encodings(encs), want):
        assert np.array_equal(np.asarray(r.values), w)


def test_library_is_built_from_the_sources_beside_it(gpu):
    """The GPU box runs the prebuilt library that travels with the tree: its
    arch and the hash of the sources it was compiled from must match."""
    from nimble_amd import build as nb
    from nimble_amd import capi

    lib = capi.lib()
    assert lib.nbl_build_arch() == b"gfx950"
    assert lib.nbl_build_source_id().decode() == nb.source_id()
