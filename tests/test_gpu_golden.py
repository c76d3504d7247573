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


def test_library_is_built_from_the_sources_beside_it(gpu):
    """The GPU box runs the prebuilt library that travels with the tree: its
    arch and the hash of the sources it was compiled from must match."""
    from nimble_amd import build as nb
    from nimble_amd import capi

    lib = capi.lib()
    assert lib.nbl_build_arch() == b"gfx950"
    assert lib.nbl_build_source_id().decode() == nb.source_id()
