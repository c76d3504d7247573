#!/usr/bin/env python3
"""Writes tests/golden/golden_v1.json: hand-assembled encoded byte vectors with
their expected decoded output.

Independence: this script uses neither nimble_amd.encoder nor the oracle.  Every
vector is assembled with struct.pack from the byte layouts the reference
documents and hand-assembles in its own tests, and the expected output is the
input the reference's tests feed (so the answer is known by construction):

  prefix ............ encodings/common/EncodingPrefix.h:26-54
  FixedBitWidth ..... encodings/tests/FixedBitWidthEncodingTest.cpp:76-103 (hand assembly),
                      :122-138, :140-188 ({0..9}, {0..52}, bit-width byte 8 / 6)
  FixedBitArray ..... common/tests/FixedBitArrayTests.cpp:27-39 ((i+i*i)%1024 at 10 bits),
                      :151-170 (100-i at 7 bits); packing rule common/FixedBitArray.cpp:68-82
  RLE ............... encodings/RLEEncoding.h:102-106,428-431,741-745,
                      encodings/tests/RLEEncodingTest.cpp:142-148,184-195
  Nullable .......... encodings/tests/EncodingTest.cpp:1092-1105 (hand assembly)
  MainlyConstant .... encodings/tests/EncodingTest.cpp:1185-1202,1264-1282 (hand assembly)
  Dictionary ........ encodings/DictionaryEncoding.h:41-45
  SparseBool ........ encodings/SparseBoolEncoding.h:44-47, SparseBoolEncoding.cpp:215-217
  Constant .......... encodings/ConstantEncoding.h:33-35
  Trivial ........... encodings/TrivialEncoding.h:40-44,105-110,191-194
  Varint ............ encodings/VarintEncoding.h:37-40, common/Varint.h:70-76
  chunk header ...... common/ChunkHeader.h:25-47

Three origins (field "origin" of every case):
  reference-held ..... every byte is what one of the reference's tests assembles field by field
  reference-frame .... the frame (prefix, size fields, trailing value) and the expected output are
                       what a test of the reference assembles / asserts; the child encodings inside
                       it come from the reference's encoder there (forced to Trivial children) and
                       are assembled here from the documented layout
  self-assembled ..... put together here from the documented layout, with the reference tests' inputs

Run:  python tests/golden/make_golden.py
"""
import json
import os
import struct

HERE = os.path.dirname(os.path.abspath(__file__))

TRIVIAL, RLE, DICT, FBW, NULLABLE, SPARSE, VARINT, CONST, MC = 0, 1, 2, 3, 5, 6, 7, 9, 10
I8, U8, I16, U16, I32, U32, I64, U64, F32, F64, BOOL, STR = range(1, 13)
ES = {I8: 1, U8: 1, I16: 2, U16: 2, I32: 4, U32: 4, I64: 8, U64: 8, F32: 4, F64: 8}
FMT = {1: "B", 2: "H", 4: "I", 8: "Q"}


def varint(v):
    out = bytearray()
    while v >= 128:
        out.append(0x80 | (v & 0x7F))
        v >>= 7
    out.append(v)
    return bytes(out)


def prefix(enc, dt, rows, vr=False):
    return bytes([enc, dt]) + (varint(rows) if vr else struct.pack("<I", rows))


def raw(vals, dt):
    return b"".join(struct.pack("<" + FMT[ES[dt]], v & ((1 << (8 * ES[dt])) - 1)) for v in vals)


def pack_bits(vals, w):
    """FixedBitArray::set: value i OR-ed into bits [i*w, (i+1)*w), LSB first; +7 bytes of slop."""
    acc = 0
    for i, v in enumerate(vals):
        assert 0 <= v < (1 << w)
        acc |= v << (i * w)
    nbytes = (len(vals) * w + 7) // 8
    return acc.to_bytes(nbytes, "little") + bytes(7)


def fbw(vals, dt, w, base=0, vr=False):
    return (prefix(FBW, dt, len(vals), vr) + b"\x00" + raw([base], dt) + bytes([w]) +
            pack_bits([(v - base) & ((1 << (8 * ES[dt])) - 1) for v in vals], w))


def trivial(vals, dt, vr=False):
    if dt == BOOL:
        return prefix(TRIVIAL, BOOL, len(vals), vr) + b"\x00" + pack_bits([int(b) for b in vals], 1)
    if dt == STR:
        lens = trivial([len(s) for s in vals], U32, vr)
        return prefix(TRIVIAL, STR, len(vals), vr) + b"\x00" + struct.pack("<I", len(lens)) + lens + b"".join(vals)
    return prefix(TRIVIAL, dt, len(vals), vr) + b"\x00" + raw(vals, dt)


def runs_of(vals):
    lens, rv = [], []
    for v in vals:
        if rv and rv[-1] == v:
            lens[-1] += 1
        else:
            rv.append(v)
            lens.append(1)
    return lens, rv


def rle(vals, dt, child=trivial, vr=False):
    lens, rv = runs_of(vals)
    le = child(lens, U32, vr)
    head = prefix(RLE, dt, len(vals), vr) + struct.pack("<I", len(le)) + le
    if dt == BOOL:
        return head + bytes([int(vals[0])])
    return head + child(rv, dt, vr)


def dictionary(vals, dt, vr=False):
    alphabet = []
    for v in vals:
        if v not in alphabet:
            alphabet.append(v)
    idx = [alphabet.index(v) for v in vals]
    a = trivial(alphabet, dt, vr)
    return prefix(DICT, dt, len(vals), vr) + struct.pack("<I", len(a)) + a + trivial(idx, U32, vr)


def sparse_bool(bools, vr=False):
    n = len(bools)
    ones = sum(bools)
    sparse_value = not (ones > (n >> 1))
    idx = [i for i, b in enumerate(bools) if bool(b) == sparse_value] + [n]
    return prefix(SPARSE, BOOL, n, vr) + bytes([int(sparse_value)]) + trivial(idx, U32, vr)


def constant(value, rows, dt, vr=False):
    if dt == STR:
        return prefix(CONST, STR, rows, vr) + struct.pack("<I", len(value)) + value
    if dt == BOOL:
        return prefix(CONST, BOOL, rows, vr) + bytes([int(value)])
    return prefix(CONST, dt, rows, vr) + raw([value], dt)


def nullable(data_enc, mask, dt, nulls_enc=None, vr=False):
    ne = nulls_enc if nulls_enc is not None else trivial(mask, BOOL, vr)
    return prefix(NULLABLE, dt, len(mask), vr) + struct.pack("<I", len(data_enc)) + data_enc + ne


def mainly_constant(vals, dt, common, other_enc, is_common_enc=None, vr=False):
    ic = [v == common for v in vals]
    ice = is_common_enc if is_common_enc is not None else trivial(ic, BOOL, vr)
    cv = (struct.pack("<I", len(common)) + common) if dt == STR else raw([common], dt)
    return (prefix(MC, dt, len(vals), vr) + struct.pack("<I", len(ice)) + ice +
            struct.pack("<I", len(other_enc)) + other_enc + cv)


def varint_enc(vals, dt, base=None, vr=False):
    base = min(vals) if base is None else base
    return prefix(VARINT, dt, len(vals), vr) + raw([base], dt) + b"".join(varint(v - base) for v in vals)


CASES = []


def add(name, enc, dt, expected, nulls=None, vr=False, source="", origin="self-assembled", cursor=None, dictionary=None):
    """origin: see the module docstring.  cursor: [(rows to skip, rows to materialize, expected)]
    read after a reset().  dictionary: (size, entries as a set) the reference asserts for
    Encoding::dictionarySize() / dictionaryEntries()."""
    assert origin in ("reference-held", "reference-frame", "self-assembled")
    c = {"name": name, "data_type": dt, "varint_rows": vr, "hex": enc.hex(), "source": source, "origin": origin}
    if cursor:
        c["cursor"] = [{"skip": s, "materialize": m, "expected": list(e)} for s, m, e in cursor]
    if dictionary:
        size, entries = dictionary
        c["dictionary_size"] = size
        c["dictionary_entries"] = sorted(e.hex() if isinstance(e, bytes) else int(e) for e in entries)
    if dt == STR:
        c["expected_strings_hex"] = [s.hex() for s in expected]
    elif dt == BOOL:
        c["expected"] = [int(b) for b in expected]
    else:
        m = (1 << (8 * ES[dt])) - 1
        c["expected"] = [int(v) & m for v in expected]
    if nulls is not None:
        c["expected_non_null"] = [int(b) for b in nulls]
    CASES.append(c)


def main():
    r53 = list(range(53))
    add("fbw_u32_0to52_byte_rounded", fbw(r53, U32, 8), U32, r53,
        source="encodings/tests/FixedBitWidthEncodingTest.cpp:76-103 (encodeByteRounded: prefix, compression byte, "
               "u32 baseline 0, bit-width byte 8, FixedBitArray payload of bufferSize bytes), input {0..52} of :140-188",
        origin="reference-held")
    add("fbw_u32_0to52_exact_6bits", fbw(r53, U32, 6), U32, r53,
        source="FixedBitWidthEncodingTest.cpp:140-188 (bit-width byte == 6 with exact bits)")
    add("fbw_u32_0to52_varint_prefix", fbw(r53, U32, 8, vr=True), U32, r53, vr=True,
        source="encodings/tests/FixedBitWidthEncodingTest.cpp:76-103 with useVarintRowCount (FBWTestConfig<true>, :118)",
        origin="reference-held")
    r10 = list(range(10))
    add("fbw_u32_0to9", fbw(r10, U32, 8), U32, r10, source="FixedBitWidthEncodingTest.cpp:122-138")
    add("fbw_u32_single_42", fbw([42], U32, 8, base=42), U32, [42], source="FixedBitWidthEncodingTest.cpp:332-388")
    add("fbw_u32_all_100", fbw([100] * 100, U32, 8, base=100), U32, [100] * 100,
        source="FixedBitWidthEncodingTest.cpp:332-388")
    quad = [(i + i * i) % 1024 for i in range(10000)]
    add("fba_10bit_quadratic", fbw(quad, U32, 10), U32, quad, source="common/tests/FixedBitArrayTests.cpp:27-39")
    desc = [100 - i for i in range(27)]
    add("fba_7bit_descending", fbw(desc, U32, 7), U32, desc, source="common/tests/FixedBitArrayTests.cpp:151-170")
    add("fba_7bit_descending_into64", fbw(desc, U64, 7), U64, desc,
        source="common/tests/FixedBitArrayTests.cpp:165-169 (bulkGet32Into64)")
    wide = [(0x9E3779B97F4A7C15 * (i + 1)) & ((1 << 59) - 1) for i in range(300)]
    add("fbw_u64_59bit_two_word_stitch", fbw(wide, U64, 59), U64, wide,
        source="common/FixedBitArray.cpp:490-507 (widths 59..63 cross a word)")
    wide58 = [v & ((1 << 58) - 1) for v in wide]
    add("fbw_u64_58bit_boundary", fbw(wide58, U64, 58), U64, wide58,
        source="common/tests/FixedBitArrayTests.cpp:459-479 (58-bit boundary)")
    add("fbw_i64_baseline_12345", fbw([12345 + (i * 37) % 1000 for i in range(200)], I64, 16, base=12345), I64,
        [12345 + (i * 37) % 1000 for i in range(200)], source="encodings/benchmarks/FixedBitWidthBenchmark.cpp:40")
    add("trivial_i32", trivial([1, -2, 3, -4, 2**31 - 1, -2**31], I32), I32, [1, -2, 3, -4, 2**31 - 1, -2**31],
        source="encodings/TrivialEncoding.h:40-44")
    bools = [bool((i * 7) % 3 == 0) for i in range(75)]
    add("trivial_bool_75", trivial(bools, BOOL), BOOL, bools, source="encodings/TrivialEncoding.h:191-194")
    strs = [b"", b"a", b"nimble", b"\x00\xff", b"cascading stream encodings", b""]
    add("trivial_string", trivial(strs, STR), STR, strs, source="encodings/TrivialEncoding.h:105-110")
    add("rle_i32_233", rle([2, 3, 3], I32), I32, [2, 3, 3], source="encodings/tests/RLEEncodingTest.cpp:142-148")
    tri = [1, 2, 2, 3, 3, 3, 4, 4, 4, 4]
    add("rle_i32_triangle", rle(tri, I32), I32, tri, source="encodings/tests/RLEEncodingTest.cpp:142-148")
    add("rle_i64_triangle_fbw_children", rle(tri, I64, child=lambda v, dt, vr=False: fbw(v, dt, 8, vr=vr)), I64, tri,
        source="encodings/RLEEncoding.h:102-106,428-431")
    f32 = [0x00000000, 0x80000000, 0x7FC00000, 0x7FA00000, 0x7FC00003, 0x7FC00005]
    rl_f32 = [v for v in f32 for _ in range(3)]
    add("rle_f32_zero_signs_and_nan_payloads", rle(rl_f32, F32), F32, rl_f32,
        source="encodings/tests/RLEEncodingTest.cpp:172-195 (+-0.0 and NaN payloads survive bit-exactly)")
    rb = [True] * 3 + [False] * 5 + [True] + [False] * 2
    add("rle_bool", rle(rb, BOOL), BOOL, rb, source="encodings/RLEEncoding.h:741-745, RLEEncoding.cpp:39-46")
    dv = [7, 7, 3, 9, 3, 7, 9, 9, 1]
    add("dictionary_i32", dictionary(dv, I32), I32, dv, source="encodings/DictionaryEncoding.h:41-45,432-456")
    ds = [b"x", b"yy", b"x", b"", b"yy", b"zzz"]
    add("dictionary_string", dictionary(ds, STR), STR, ds, source="encodings/DictionaryEncoding.h:41-45")
    sb = [False] * 40
    for i in (3, 17, 18, 39):
        sb[i] = True
    add("sparse_bool_few_set", sparse_bool(sb), BOOL, sb, source="encodings/SparseBoolEncoding.cpp:190-217")
    sb2 = [not b for b in sb]
    add("sparse_bool_few_unset", sparse_bool(sb2), BOOL, sb2, source="encodings/SparseBoolEncoding.cpp:190-217")
    add("constant_i64", constant(-5, 33, I64), I64, [-5] * 33, source="encodings/ConstantEncoding.h:33-35")
    add("constant_string", constant(b"same", 4, STR), STR, [b"same"] * 4, source="encodings/ConstantEncoding.cpp:21-33")
    add("constant_bool", constant(True, 9, BOOL), BOOL, [True] * 9, source="encodings/ConstantEncoding.h:305-326")
    vv = [0, 1, 127, 128, 300, 16383, 16384, 2**21, 2**28 - 1, 2**28, 2**32 - 1]
    add("varint_u32", varint_enc(vv, U32, base=0), U32, vv, source="common/Varint.h:70-102")
    vv64 = vv + [2**35, 2**42 + 5, 2**49, 2**56 - 1, 2**63, 2**64 - 1]
    add("varint_u64", varint_enc(vv64, U64, base=0), U64, vv64, source="common/Varint.h:104-143")
    add("varint_i64_baseline", varint_enc([1000 + v for v in vv], I64, base=1000), I64, [1000 + v for v in vv],
        source="encodings/VarintEncoding.h:37-40,132-138")
    # Nullable{Dictionary, Trivial<bool>}: EncodingTest.cpp:1057-1115
    mask = [True, False, True, True, False, False, True, True, True, False]
    nn = [5, 6, 5, 7, 6, 5]
    full = []
    it = iter(nn)
    for m in mask:
        full.append(next(it) if m else 0)
    add("nullable_dictionary_i32", nullable(dictionary(nn, I32), mask, I32), I32, full, nulls=mask,
        source="encodings/tests/EncodingTest.cpp:1057-1115")
    add("nullable_all_null", nullable(trivial([], I64), [False] * 5, I64), I64, [0] * 5, nulls=[False] * 5,
        source="encodings/NullableEncoding.h:207-236")
    add("nullable_no_nulls_sparse", nullable(fbw(r10, I16, 8), [True] * 10, I16, nulls_enc=sparse_bool([True] * 10)), I16, r10,
        nulls=[True] * 10, source="encodings/NullableEncoding.h:38-42")
    # MainlyConstant{Trivial<bool>, Constant}: EncodingTest.cpp:1128-1205
    mc = [4, 4, 9, 4, 4, 4, 9, 4]
    add("mainly_constant_constant_others", mainly_constant(mc, I32, 4, constant(9, 2, I32)), I32, mc,
        source="encodings/tests/EncodingTest.cpp:1128-1205")
    mc2 = [1, 1, 2, 1, 3, 1, 1, 9, 1, 1, 1]
    add("mainly_constant_dictionary_others", mainly_constant(mc2, I64, 1, dictionary([2, 3, 9], I64)), I64, mc2,
        source="encodings/tests/EncodingTest.cpp:1207-1285")
    others = [v for v in mc2 if v != 1]
    add("mainly_constant_sparse_is_common",
        mainly_constant(mc2, I64, 1, trivial(others, I64), is_common_enc=sparse_bool([v == 1 for v in mc2])), I64, mc2,
        source="encodings/MainlyConstantEncoding.h:758-787 (SparseBool fast path)")
    ms = [b"c", b"c", b"other", b"c"]
    add("mainly_constant_string", mainly_constant(ms, STR, b"c", trivial([b"other"], STR)), STR, ms,
        source="encodings/MainlyConstantEncoding.cpp:21-43")
    # three levels: Nullable{MainlyConstant{Trivial<bool>, Dictionary}, Trivial<bool>}: EncodingTest.cpp:1295-1340
    nn2 = [1, 1, 2, 1, 3, 1]
    full2 = []
    it = iter(nn2)
    for m in mask[:9]:
        full2.append(next(it) if m else 0)
    add("nullable_mainly_constant_dictionary",
        nullable(mainly_constant(nn2, I32, 1, dictionary([2, 3], I32)), mask[:9], I32), I32, full2, nulls=mask[:9],
        source="encodings/tests/EncodingTest.cpp:1295-1340")
    # ---- frames and answers the reference's own tests hold ---------------------------
    # MainlyConstantEncodingTest.cpp:388-464: prefix | writeString(SparseBool isCommon) | writeString(Trivial<int32>
    # others) | write<uint32_t>(7); the test's Encoder<> writes children with Trivial leaves
    # (encodings/tests/TestUtils.h:207-274).  Asserted: {7,7,20,7,40,7,7,70,7,7}; reset, skip(3), materialize(4) = {7,40,7,7}
    ic = [True, True, False, True, False, True, True, False, True, True]
    for vr in (False, True):
        frame = (prefix(MC, I32, 10, vr) + struct.pack("<I", len(sparse_bool(ic, vr))) + sparse_bool(ic, vr) +
                 struct.pack("<I", len(trivial([20, 40, 70], I32, vr))) + trivial([20, 40, 70], I32, vr) + struct.pack("<I", 7))
        add("ref_mainly_constant_sparse_is_common" + ("_varint_rows" if vr else ""), frame, I32,
            [7, 7, 20, 7, 40, 7, 7, 70, 7, 7], vr=vr, origin="reference-frame", cursor=[(3, 4, [7, 40, 7, 7])],
            source="encodings/tests/MainlyConstantEncodingTest.cpp:388-464 (materializeSparseBoolIsCommon, useVarint = "
                   + ("true" if vr else "false") + ")")
    # ReadWithVisitorTest.cpp:3375-3483: Nullable<string>{Dictionary{Trivial, Trivial}, Trivial<bool>}, 50 rows, every
    # fifth null, value kAlphabet[i % 3]; frame writeChar x2 | writeUint32 | writeString(values) | writeBytes(nulls)
    # (:3434-3445).  Asserted: isNullable, dictionaryEnabled, alphabet[indices[i]] == kAlphabet[i % 3] on non-null rows
    alpha3 = [b"alpha", b"bravo", b"charlie"]
    mask50 = [i % 5 != 0 for i in range(50)]
    rows50 = [alpha3[i % 3] if mask50[i] else b"" for i in range(50)]
    add("ref_nullable_string_dictionary_50", nullable(dictionary([r for r, m in zip(rows50, mask50) if m], STR), mask50, STR), STR,
        rows50, nulls=mask50, origin="reference-frame", dictionary=(3, alpha3),
        source="encodings/tests/ReadWithVisitorTest.cpp:3375-3483 (readIndicesWithVisitorNullable)")
    # ReadWithVisitorTest.cpp:3491-3600: 20 rows, rows 0,4,8,.. null; "delta" sits only in null rows and must not be
    # in the dictionary: dictionarySize() == 3 (:3573)
    vals4 = [b"delta", b"alpha", b"bravo", b"charlie"]
    mask20 = [i % 4 != 0 for i in range(20)]
    rows20 = [vals4[i % 4] if mask20[i] else b"" for i in range(20)]
    add("ref_nullable_string_dictionary_masked_value", nullable(dictionary([r for r, m in zip(rows20, mask20) if m], STR), mask20, STR),
        STR, rows20, nulls=mask20, origin="reference-frame", dictionary=(3, alpha3),
        source="encodings/tests/ReadWithVisitorTest.cpp:3491-3600 (readIndicesWithVisitorNullableAlphabetValue)")
    # EncodingTest.cpp:1057-1115 encodeNullableDictionary (frame :1092-1105) with the inputs of nullableBasics
    # (:1513-1548: dictionarySize 3) and nullableAlphabetValue (:1550-1587: dictionarySize 2)
    m5 = [True, True, False, True, True]
    for tag, vals, size, where in (("basics", [1, 2, 1, 3, 2], 3, ":1513-1548"), ("alphabet_value", [1, 2, 3, 1, 2], 2, ":1550-1587")):
        for dt, dn in ((I32, "i32"), (I64, "i64"), (U8, "u8")):
            nn_ = [v for v, m in zip(vals, m5) if m]
            add(f"ref_nullable_dictionary_{tag}_{dn}", nullable(dictionary(nn_, dt), m5, dt), dt,
                [v if m else 0 for v, m in zip(vals, m5)], nulls=m5, origin="reference-frame", dictionary=(size, set(nn_)),
                source="encodings/tests/EncodingTest.cpp:1057-1115 (frame :1092-1105), inputs of " + where)
        names = {1: b"alpha", 2: b"bravo", 3: b"charlie"}
        svals = [names[v] for v in vals]
        nn_ = [v for v, m in zip(svals, m5) if m]
        add(f"ref_nullable_dictionary_{tag}_string", nullable(dictionary(nn_, STR), m5, STR), STR,
            [v if m else b"" for v, m in zip(svals, m5)], nulls=m5, origin="reference-frame", dictionary=(size, set(nn_)),
            source="encodings/tests/EncodingTest.cpp:1057-1115 (frame :1092-1105), inputs of " + where)
    # EncodingTest.cpp:1131-1205 serializeMainlyConstantWithConstantOthers (frame :1185-1202) with the input of
    # mainlyConstantWithSingleOtherValue (:3022-3070): 20 rows of 100, rows 3, 7, 11, 15 = 42
    d20 = [100] * 20
    for i in (3, 7, 11, 15):
        d20[i] = 42
    for dt, dn in ((I32, "i32"), (I64, "i64"), (U16, "u16")):
        add(f"ref_mainly_constant_constant_others_{dn}", mainly_constant(d20, dt, 100, constant(42, 4, dt)), dt, d20,
            origin="reference-frame",
            source="encodings/tests/EncodingTest.cpp:1131-1205 (frame :1185-1202), input of :3022-3070")
    sd20 = [b"common_value"] * 20
    for i in (3, 7, 11, 15):
        sd20[i] = b"other_value"
    add("ref_mainly_constant_constant_others_string",
        mainly_constant(sd20, STR, b"common_value", constant(b"other_value", 4, STR)), STR, sd20, origin="reference-frame",
        source="encodings/tests/EncodingTest.cpp:1131-1205 (frame :1185-1202), input of :3022-3070")
    # EncodingTest.cpp:1207-1285 serializeMainlyConstantDictionary (frame :1264-1282) with the input of :1825-1855:
    # 20 rows of 99, rows 3 and 15 = 1, row 7 = 2
    e20 = [99] * 20
    e20[3], e20[7], e20[15] = 1, 2, 1
    for dt, dn in ((I32, "i32"), (I64, "i64")):
        add(f"ref_mainly_constant_dictionary_others_{dn}", mainly_constant(e20, dt, 99, dictionary([1, 2, 1], dt)), dt, e20,
            origin="reference-frame",
            source="encodings/tests/EncodingTest.cpp:1207-1285 (frame :1264-1282), input of :1825-1855")
    # EncodingTest.cpp:1296-1340 encodeNullableMainlyConstantDictionary with the input of :2015-2037
    d7, m7 = [99, 1, 99, 2, 99, 99, 99], [True, True, False, True, True, True, True]
    nn7 = [v for v, m in zip(d7, m7) if m]
    add("ref_nullable_mainly_constant_dictionary", nullable(mainly_constant(nn7, I32, 99, dictionary([1, 2], I32)), m7, I32), I32,
        [v if m else 0 for v, m in zip(d7, m7)], nulls=m7, origin="reference-frame",
        source="encodings/tests/EncodingTest.cpp:1296-1340, input of :2015-2037")
    # ---- streams the reference's own tests assemble byte by byte --------------------
    # tools/tests/EncodingUtilitiesTest.cpp:42-70 buildTrivialUint32Stream: prefix | Uncompressed | raw u32
    for vals, where in (([10, 20, 30], ":194-201"), ([1, 2, 3], ":213-243"), ([10, 20], ":291-313 (encoded size 15)"),
                        ([0, 64], ":425-431 (BlockBitPacking baselines)"), ([0, 5], ":413 (PFOR exception positions)")):
        add("ref_trivial_u32_" + "_".join(map(str, vals)), trivial(vals, U32), U32, vals,
            source="tools/tests/EncodingUtilitiesTest.cpp:42-70 buildTrivialUint32Stream, used at " + where,
            origin="reference-held")
    assert len(trivial([10, 20], U32)) == 15  # EncodingUtilitiesTest.cpp:311-312
    add("ref_trivial_u32_varint_rows", trivial([10, 20, 30], U32, vr=True), U32, [10, 20, 30], vr=True,
        source="tools/tests/EncodingUtilitiesTest.cpp:49-70 buildTrivialUint32Stream(values, useVarintRowCount = true)",
        origin="reference-held")
    # tools/tests/EncodingUtilitiesTest.cpp:72-90 buildConstantUint32Stream: prefix | u32 value
    for value, rows, where in ((42, 100, ":203-208"), (99, 50, ":245-274")):
        add(f"ref_constant_u32_{value}x{rows}", constant(value, rows, U32), U32, [value] * rows,
            source="tools/tests/EncodingUtilitiesTest.cpp:72-90 buildConstantUint32Stream, used at " + where,
            origin="reference-held")
    # BlockBitPacking header with its three varint-sized metadata children and NO packed data
    # (tools/tests/EncodingUtilitiesTest.cpp:130-178 buildBlockBitPackingUint32Stream, children of :425-431):
    # the reference's traverseEncodings visits Baselines, BitWidths, DataOffsets in this order (:432-455)
    kids = [trivial([0, 64], U32), trivial([4, 5], U32), trivial([0, 32], U32)]
    bbp = prefix(15, U32, 100) + b"\x00" + varint(64) + varint(2)
    for k in kids:
        bbp += varint(len(k)) + k
    bbp += varint(64)
    headers = [{
        "name": "ref_block_bit_packing_header", "origin": "reference-held", "hex": bbp.hex(),
        "source": "tools/tests/EncodingUtilitiesTest.cpp:130-178,425-455",
        "encoding_type": 15, "data_type": U32, "rows": 100, "block_size": 64, "blocks": 2, "first_block_rows": 64,
        "children_hex": [k.hex() for k in kids], "children_names": ["Baselines", "BitWidths", "DataOffsets"],
    }]
    # encoding prefixes (encodings/tests/EncodingPrefixTest.cpp:42-83): EncodingPrefix::serialize output
    prefixes = [
        {"origin": "reference-held", "source": "encodings/tests/EncodingPrefixTest.cpp:42-54", "encoding_type": 8,
         "data_type": I64, "rows": 123, "varint_rows": False, "hex": prefix(8, I64, 123).hex()},
        {"origin": "reference-held", "source": "encodings/tests/EncodingPrefixTest.cpp:56-68", "encoding_type": TRIVIAL,
         "data_type": U32, "rows": 123, "varint_rows": False, "hex": prefix(TRIVIAL, U32, 123).hex()},
        {"origin": "reference-held", "source": "encodings/tests/EncodingPrefixTest.cpp:70-83", "encoding_type": RLE,
         "data_type": STR, "rows": 1000000, "varint_rows": True, "hex": prefix(RLE, STR, 1000000, vr=True).hex()},
    ]
    # chunk headers (common/tests/ChunkHeaderTest.cpp:23-52): u32 length | u8 CompressionType (Zstd = 1)
    chunk_headers = [
        {"origin": "reference-held", "source": "common/tests/ChunkHeaderTest.cpp:28-39", "length": 1234, "compression": 1,
         "hex": (struct.pack("<I", 1234) + b"\x01").hex()},
        {"origin": "reference-held", "source": "common/tests/ChunkHeaderTest.cpp:41-50", "length": 42, "compression": 0,
         "hex": (struct.pack("<I", 42) + b"\x00").hex()},
        {"origin": "reference-held", "source": "common/tests/ChunkHeaderTest.cpp:52-62", "length": 100000, "compression": 1,
         "hex": (struct.pack("<I", 100000) + b"\x01").hex()},
    ]
    # varint sizes at the boundaries (common/tests/VarintTests.cpp:64-76,95-103)
    varint_sizes = [{"value": v, "size": n} for v, n in
                    ((0, 1), (1, 1), (127, 1), (128, 2), (16383, 2), (16384, 3), (2097151, 3), (2097152, 4),
                     (268435455, 4), (268435456, 5), (2**32 - 1, 5), (2**64 - 1, 10))]
    for e in varint_sizes:
        assert len(varint(e["value"])) == e["size"]
    # run computation known answers (encodings/tests/RLETests.cpp:31-169): data -> (run lengths, run values);
    # used both ways: the encoder must compute these runs, and an RLE of them must decode to the data
    rle_runs = [{"source": "encodings/tests/RLETests.cpp:" + where, "data_type": dt, "data": d, "lengths": l, "values": v}
                for where, dt, d, l, v in (
                    ("42-56", I32, [42], [1], [42]),
                    ("58-72", I32, [5, 5, 5, 5, 5], [5], [5]),
                    ("74-88", I32, [1, 2, 3, 4, 5], [1, 1, 1, 1, 1], [1, 2, 3, 4, 5]),
                    ("90-104", I32, [1, 1, 1, 2, 2, 3, 3, 3, 3], [3, 2, 4], [1, 2, 3]),
                    ("106-120", I32, [-1, -1, 0, 0, 0, 1, 1], [2, 3, 2], [-1, 0, 1]),
                    ("122-136", I64, [100, 100, 200, 200, 200], [2, 3], [100, 200]),
                    ("138-153", U32, [0, 0, 1, 2, 2, 2], [2, 1, 3], [0, 1, 2]),
                    ("155-169", I32, [1, 2, 1, 2, 1, 2], [1, 1, 1, 1, 1, 1], [1, 2, 1, 2, 1, 2]))]
    for e in rle_runs:
        assert runs_of(e["data"]) == (e["lengths"], e["values"])
    # polarity and listed positions of SparseBool (encodings/tests/SparseBoolEncodingTest.cpp:145-216)
    T, F = True, False
    sparse_facts = [{"source": "encodings/tests/SparseBoolEncodingTest.cpp:" + where, "values": [int(b) for b in v],
                     "sparse_value": int(sv), "positions": pos}
                    for where, v, sv, pos in (
                        ("145-172", [T, T, F, T, F, T, T, F, T, T], False, [2, 4, 7]),
                        ("174-196", [F, T, F, F, F, F, T, F, F, F], True, [1, 6]),
                        ("198-207", [T, T, F, T, T, T, T, T, T, T], False, [2]),
                        ("209-216", [F, F, T, F, F, F, F, F, F, F], True, [2]))]
    out = {"format": 3,
           "note": "hand-assembled; regenerate with tests/golden/make_golden.py.  origin = reference-held: the bytes are "
                   "what one of the reference's own tests assembles field by field; reference-frame: frame and expected "
                   "output are a reference test's, the child encodings inside are assembled here; self-assembled: put "
                   "together here from the layout the reference documents",
           "chunk_header_size": 5,
           "cases": CASES,
           "rle_runs": {"origin": "reference-held", "entries": rle_runs},
           "sparse_bool_facts": {"origin": "reference-held", "entries": sparse_facts},
           "headers": headers,
           "prefixes": prefixes,
           "chunk_headers": chunk_headers,
           "varint_sizes": {"origin": "reference-held", "source": "common/tests/VarintTests.cpp:64-76,95-103",
                            "entries": varint_sizes}}
    with open(os.path.join(HERE, "golden_v1.json"), "w") as f:
        json.dump(out, f, indent=0)
    print(f"wrote {len(CASES)} cases")


if __name__ == "__main__":
    main()
