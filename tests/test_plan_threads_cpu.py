"""The header walk of a large batch runs on several host threads, each over a
range of chunks, and the per-thread plans are put together in chunk order
(the reference builds its Encoding objects one stream after the other,
velox/VeloxReader.cpp:330-351, encodings/common/EncodingFactory.cpp:62-173).
How many threads made a plan must not show in it: the fingerprint covers the
descriptor tables as uploaded, the launch list, every layout and all sizes."""
import numpy as np
import pytest

from nimble_amd import capi
from nimble_amd import encoder as E
from nimble_amd import wire as W
from nimble_amd import workloads as WL
from nimble_amd.decoder import Plan


def _tables():
    yield "wide", WL.wide_table_stripe(stripe=0, columns=700, rows=20_000)
    yield "mixed", [c for s in range(4) for c in WL.mixed_table_stripe(stripe=s, rows=30_000)]
    yield "fbw_rle", WL.fbw_rle_table(rows=40_000) * 5
    yield "dict_string", WL.dict_string_table(rows=300_000, rows_per_chunk=4_000)


@pytest.mark.parametrize("name,chunks", list(_tables()), ids=lambda v: v if isinstance(v, str) else "")
def test_plans_do_not_depend_on_the_number_of_threads(name, chunks):
    buf, offs = WL.pack_chunks(chunks)
    assert len(offs) >= 140
    for options in ({}, {"per_kernel": True}, {"keep_dictionary": True}, {"wave_scatter_min_tiles": 1, "lds_dictionary_min_tiles": 1}):
        serial = Plan(buf, offs, plan_threads=1, **options)
        want = serial.fingerprint()
        for threads in (2, 3, 8, 64):
            p = Plan(buf, offs, plan_threads=threads, **options)
            assert p.fingerprint() == want, f"{name}: {threads} threads, {options}"
            assert [bytes(a) for a in p.layouts] == [bytes(a) for a in serial.layouts]
            assert p.launches() == serial.launches()
            assert bytes(p.info) == bytes(serial.info)
            p.close()
        serial.close()


def test_fingerprint_tells_plans_apart():
    a = [E.encode(E.FixedBitWidth(bit_width=9), np.arange(500, dtype=np.int32), W.INT32)]
    b = [E.encode(E.FixedBitWidth(bit_width=10), np.arange(500, dtype=np.int32), W.INT32)]
    pa, pb, pa2 = Plan.from_encodings(a), Plan.from_encodings(b), Plan.from_encodings(a)
    assert pa.fingerprint() == pa2.fingerprint() != pb.fingerprint()
    for p in (pa, pb, pa2):
        p.close()


def test_the_first_damaged_chunk_is_reported_whatever_the_threads():
    """Every thread stops at the first chunk of its range that does not parse;
    the error of the lowest range is the one a single walk would have met."""
    chunks = [enc for _, enc in WL.wide_table_stripe(stripe=0, columns=600, rows=5_000)]
    bad_type = bytearray(chunks[200]); bad_type[0] = 99          # EncodingType out of range
    truncated = bytes(chunks[450][: len(chunks[450]) // 2])      # child beyond the end
    chunks[200], chunks[450] = bytes(bad_type), truncated
    messages = set()
    for threads in (1, 2, 4, 7):
        with pytest.raises(capi.NimbleCorruptError) as ei:
            Plan.from_encodings(chunks, plan_threads=threads)
        messages.add(ei.value.message)
    assert len(messages) == 1 and "EncodingType:99" in messages.pop()
