"""CPU: the synthetic benchmark tables are well-formed Nimble (the oracle decodes
every chunk, the planner accepts every chunk) at reduced sizes."""
import numpy as np

from nimble_amd import wire as W
from nimble_amd import workloads as WL
from nimble_amd.decoder import Plan


def _plan(chunks):
    buf, offs = WL.pack_chunks(chunks)
    return Plan(buf, offs)


def test_fbw_rle_table_small(oracle):
    chunks = WL.fbw_rle_table(50_000, seed=1)
    assert len(chunks) == 28  # 8 + 16 widths + 2 x 2 RLE columns, one chunk each at this size
    p = _plan(chunks)
    assert int(p.info.total_rows) == 28 * 50_000
    for label, enc in chunks[::5]:
        r = oracle.decode_chunk(enc)
        assert r["rows"] == 50_000, label
    # the header of an FBW chunk carries exactly the requested width
    for label, enc in chunks:
        if label.startswith("fbw/"):
            w = int(label.split("/w")[1].split("/")[0])
            es = W.elem_bytes(enc[1])
            assert enc[0] == W.FIXED_BIT_WIDTH and enc[6 + 1 + es] == w, label


def test_cpu_reference_column_round_trip(oracle):
    chunks, vals = WL.cpu_reference_column(100_000, seed=42)
    for label, enc in chunks:
        r = oracle.decode_chunk(enc)
        assert np.array_equal(r["values"], vals), label


def test_dict_string_table_small(oracle):
    chunks = WL.dict_string_table(30_000, seed=7, alphabets=(64, 4096), rows_per_chunk=20_000)
    assert len(chunks) == 2 * 3 * 2
    p = _plan(chunks)
    assert int(p.info.total_rows) == 6 * 30_000
    for i, (label, enc) in enumerate(chunks):
        lay = p.layouts[i]
        assert lay.data_type == W.STRING
        assert bool(lay.nullable) == ("null0.0/" not in label), label
        r = oracle.decode_chunk(enc)
        assert r["rows"] == lay.rows and r["non_null"] == lay.non_null_rows, label


def test_mixed_table_stripe_small(oracle):
    chunks = WL.mixed_table_stripe(stripe=2, rows=4_000)
    assert len(chunks) == 100
    p = _plan(chunks)
    assert int(p.info.max_depth) >= 3
    kinds = {}
    for i, (label, enc) in enumerate(chunks):
        r = oracle.decode_chunk(enc)
        assert r["rows"] == 4_000, label
        kinds[p.layouts[i].data_type] = kinds.get(p.layouts[i].data_type, 0) + 1
    assert kinds[W.STRING] == 15 and kinds[W.BOOL] == 10 and kinds[W.INT64] >= 25
    # a different stripe index gives different bytes (seeds are per stripe and column)
    assert WL.mixed_table_stripe(stripe=3, rows=4_000)[0][1] != chunks[0][1]


def test_wide_table_stripe_small(oracle):
    chunks = WL.wide_table_stripe(stripe=0, columns=200, rows=3_000)
    p = _plan(chunks)
    assert int(p.info.n_chunks) == 200
    assert int(p.info.n_launches) <= 40  # launches do not grow with the column count
    p2 = _plan(chunks[:20])
    assert int(p2.info.n_launches) <= int(p.info.n_launches)
    for label, enc in chunks[::9]:
        r = oracle.decode_chunk(enc)
        assert r["rows"] == 3_000 and r["data_type"] == W.FLOAT, label
