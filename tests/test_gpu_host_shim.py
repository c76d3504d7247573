"""GPU: the C++ host layer (GpuEncodingFactory -> Encoding, GpuStripeDecoder ->
Decoder) honours the reference's Encoding / Decoder contracts, with the GPU
doing the decode.  The same checks run against the CPU oracle in
test_oracle_semantics.py (encodings/tests/EncodingTest.cpp:682-1021,
velox/tests/ChunkedStreamDecoderTest.cpp:175-259)."""
import numpy as np
import pytest

from nimble_amd import encoder as E
from nimble_amd import wire as W

from encoding_contract import check_materialize_patterns, check_nullable_scattered, check_scattered_materialize
from util import patterns, phys, random_data, random_strings, rle_data

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def factory(gpu):
    import host_shim

    return host_shim.Factory()


SPECS = {
    "trivial": E.Trivial(), "fbw": E.FixedBitWidth(), "rle": E.RLE(), "dictionary": E.Dictionary(),
    "mainly_constant": E.MainlyConstant(), "varint": E.Varint(),
}


@pytest.mark.parametrize("dtype", [W.INT8, W.UINT16, W.INT32, W.INT64, W.FLOAT, W.DOUBLE])
def test_numeric_contract(factory, dtype):
    rng = np.random.default_rng(500 + dtype)
    for n in (1, 33, 700):
        for pname, data in patterns(rng, dtype, n):
            for sname, spec in SPECS.items():
                if sname == "varint" and W.elem_bytes(dtype) < 4:
                    continue
                e = factory.create(E.encode(spec, data, dtype))
                assert e.row_count == n and e.data_type == dtype
                check_materialize_patterns(e, phys(data, dtype), dtype, rng)
                e.reset()
                check_scattered_materialize(e, phys(data, dtype), dtype, rng)


def test_bool_contract(factory):
    rng = np.random.default_rng(5)
    for n in (1, 64, 65, 500):
        for p in (0.05, 0.5):
            b = rng.random(n) < p
            for spec in (E.Trivial(), E.SparseBool(), E.RLE()):
                e = factory.create(E.encode(spec, b, W.BOOL))
                check_materialize_patterns(e, b.astype(np.uint8), W.BOOL, rng)
                e.reset()
                check_scattered_materialize(e, b.astype(np.uint8), W.BOOL, rng)
                e.reset()
                words = np.zeros((n + 70) // 64 + 2, dtype=np.uint64)
                e.materialize_bools_as_bits(n // 2, words, 5)
                e.materialize_bools_as_bits(n - n // 2, words, 5 + n // 2)
                assert np.array_equal(np.unpackbits(words.view(np.uint8), bitorder="little")[5:5 + n].astype(bool), b)


def test_string_contract(factory):
    rng = np.random.default_rng(6)
    for n in (1, 50, 400):
        col = random_strings(rng, n)
        alpha = random_strings(rng, 17)
        picked = alpha.take(rng.integers(0, 17, n))
        for spec, c in ((E.Trivial(), col), (E.Dictionary(), picked), (E.MainlyConstant(other=E.Trivial()), picked),
                        (E.Constant(), alpha.take(np.zeros(n, dtype=np.int64)))):
            e = factory.create(E.encode(spec, c, W.STRING))
            check_materialize_patterns(e, (c.lengths, c.blob), W.STRING, rng)


@pytest.mark.parametrize("dtype", [W.INT32, W.INT64, W.DOUBLE])
def test_nullable_contract(factory, dtype):
    rng = np.random.default_rng(77 + dtype)
    for n in (1, 10, 333):
        for p_null in (0.0, 0.2, 0.8, 1.0):
            mask = rng.random(n) >= p_null
            vals = rle_data(rng, dtype, int(mask.sum())) if mask.any() else random_data(rng, dtype, 0)
            for spec in (E.Nullable(data=E.Trivial(), nulls=E.Trivial()), E.Nullable(data=E.FixedBitWidth(), nulls=E.SparseBool())):
                e = factory.create(E.encode_nullable(spec, vals, mask, dtype))
                assert e.is_nullable and e.null_count == n - int(mask.sum())
                check_nullable_scattered(e, phys(vals, dtype), mask, rng)
                e.reset()
                want = np.zeros(n, dtype=W.physical_dtype(dtype))
                want[mask] = phys(vals, dtype)
                assert np.array_equal(e.materialize(n), want)


def test_error_convention(factory):
    import host_shim

    ok = E.encode(E.Trivial(), np.arange(4, dtype=np.int32), W.INT32)
    with pytest.raises(host_shim.HostError, match="Unexpected trivial encoding end"):
        factory.create(ok + b"\x00")
    with pytest.raises(host_shim.HostError, match="invalid EncodingType"):
        factory.create(bytes([77]) + ok[1:])
    with pytest.raises(host_shim.HostError, match="not decoded on the GPU"):
        factory.create(bytes([8]) + ok[1:])  # Delta: the integration's fallback handles it
    e = factory.create(ok)
    with pytest.raises(host_shim.HostError, match="past the end"):
        e.materialize(5)
    # Encoding::debugString (encodings/common/Encoding.cpp:45-52)
    assert e.debug_string() == "Trivial<Int32> rowCount=4"
    assert factory.create(E.encode(E.FOR(), np.arange(9, dtype=np.uint64), W.UINT64)).debug_string(2) == "  FOR<Uint64> rowCount=9"


def _stream(rng, dtype, with_nulls, max_chunks=10, max_rows=10):
    chunks, values, masks = [], [], []
    for _ in range(int(rng.integers(1, max_chunks + 1))):
        n = int(rng.integers(1, max_rows + 1))
        nullable = with_nulls and bool(rng.integers(0, 2))
        mask = (rng.random(n) < 0.6) if nullable else np.ones(n, dtype=bool)
        v = random_data(rng, dtype, int(mask.sum()))
        if nullable:
            chunks.append(E.encode_nullable(E.Nullable(data=E.Trivial(), nulls=E.Trivial()), v, mask, dtype))
        else:
            chunks.append(E.encode(E.FixedBitWidth(), v, dtype))
        full = np.zeros(n, dtype=W.physical_dtype(dtype))
        full[mask] = phys(v, dtype)
        values.append(full)
        masks.append(mask)
    return E.stream(chunks), np.concatenate(values), np.concatenate(masks)


def test_stripe_decoder_matches_oracle_stream_decoder(factory, oracle):
    """Many streams of one stripe decoded in ONE GPU batch; every stream then
    read through Decoder::next in every batch size, dense and scattered."""
    import host_shim

    rng = np.random.default_rng(12)
    made = [_stream(rng, dt, nulls) for dt in (W.INT32, W.INT64, W.DOUBLE, W.INT16) for nulls in (False, True) for _ in range(3)]
    stripe = host_shim.StripeDecoder(factory, [m[0] for m in made])
    for si, (stream, values, mask) in enumerate(made):
        total = values.size
        gd = stripe.stream(si)
        od = oracle.StreamDecoder(stream)
        for batch in sorted({1, 2, 3, max(1, total // 2), total}):
            gd.reset()
            od.reset()
            pos = 0
            while pos < total:
                n = min(batch, total - pos)
                gout = np.zeros(n, dtype=values.dtype)
                oout = np.zeros(n, dtype=values.dtype)
                gn = np.zeros((n + 7) // 8 + 8, dtype=np.uint8)
                on = np.zeros((n + 7) // 8 + 8, dtype=np.uint8)
                gnn, greq = gd.next(n, gout, gn)
                onn, oreq = od.next(n, oout, on)
                assert (gnn, greq) == (onn, oreq)
                m = mask[pos:pos + n]
                assert gnn == int(m.sum())
                assert np.array_equal(gout[m], values[pos:pos + n][m])
                if greq:
                    assert np.array_equal(gn[:(n + 7) // 8], on[:(n + 7) // 8])
                pos += n
        # skip + scattered read
        skip = int(rng.integers(0, total))
        gd.reset()
        gd.skip(skip)
        n = total - skip
        scatter = []
        while sum(scatter) < n:
            scatter.append(bool(rng.integers(0, 2)))
        scatter = np.array(scatter, dtype=bool)
        sbits = np.zeros((scatter.size + 7) // 8 + 8, dtype=np.uint8)
        sbits[:(scatter.size + 7) // 8] = np.packbits(scatter.astype(np.uint8), bitorder="little")
        out = np.zeros(scatter.size, dtype=values.dtype)
        nulls = np.zeros((scatter.size + 7) // 8 + 8, dtype=np.uint8)
        nn, req = gd.next(n, out, nulls, sbits, scatter.size)
        m = mask[skip:]
        slots = np.flatnonzero(scatter)[m]
        assert nn == int(m.sum())
        assert np.array_equal(out[slots], values[skip:][m])
        if nn != scatter.size:
            want = np.zeros(scatter.size, dtype=bool)
            want[slots] = True
            assert np.array_equal(np.unpackbits(nulls, bitorder="little")[:scatter.size].astype(bool), want)


def test_stripe_decoder_strings(factory):
    import host_shim

    rng = np.random.default_rng(13)
    cols = [random_strings(rng, int(rng.integers(1, 40))) for _ in range(4)]
    stream = E.stream([E.encode(E.Trivial(), c, W.STRING) for c in cols])
    stripe = host_shim.StripeDecoder(factory, [stream])
    d = stripe.stream(0)
    total = sum(len(c) for c in cols)
    sv = np.zeros(2 * total, dtype=np.uint64)
    nn, _ = d.next(total, sv)
    assert nn == total and d.last_string_buffers >= 1
    lengths = np.zeros(total, dtype=np.uint32)
    n_bytes = host_shim.lib().nblh_sv_total(sv.ctypes.data, total)
    blob = np.zeros(max(int(n_bytes), 1), dtype=np.uint8)
    host_shim.lib().nblh_sv_flatten(sv.ctypes.data, total, lengths.ctypes.data, blob.ctypes.data)
    assert np.array_equal(lengths, np.concatenate([c.lengths for c in cols]))
    assert np.array_equal(blob[:int(n_bytes)], np.concatenate([c.blob for c in cols]))


def test_compressed_leaves_and_chunks_decode_like_their_uncompressed_twins(factory):
    """The reference writer's default: Zstd (or LZ4) leaf payloads, optionally
    whole-chunk compression.  The host layer inflates them into the staging
    buffer; results equal those of the same cascade written uncompressed."""
    pytest.importorskip("pyarrow")
    import host_shim

    rng = np.random.default_rng(21)
    n = 40_000
    ints = np.repeat(rng.integers(0, 500, n // 8 + 1), 8)[:n].astype(np.int64)
    strs = random_strings(rng, 30).take(rng.integers(0, 30, n))
    for codec in ("zstd", "lz4"):
        e = factory.create(E.encode(E.RLE(lengths=E.FixedBitWidth(compression=codec), values=E.Trivial(compression=codec)), ints, W.INT64))
        assert np.array_equal(e.materialize(n), ints.view(np.uint64))
        e = factory.create(E.encode(E.Dictionary(alphabet=E.Trivial(lengths=E.FixedBitWidth(compression=codec), compression=codec),
                                                 indices=E.FixedBitWidth(compression=codec)), strs, W.STRING))
        lengths, blob = e.materialize(n)
        assert np.array_equal(lengths, strs.lengths) and np.array_equal(blob, strs.blob)
        # stream with a compressed chunk, an uncompressed one and a chunk with compressed leaves
        a, b, c = ints[:15_000], ints[15_000:30_000], ints[30_000:]
        stream = (E.chunk(E.encode(E.FixedBitWidth(), a, W.INT64), codec) + E.chunk(E.encode(E.Trivial(), b, W.INT64)) +
                  E.chunk(E.encode(E.FixedBitWidth(compression=codec), c, W.INT64)))
        d = host_shim.StripeDecoder(factory, [stream]).stream(0)
        out = np.zeros(n, dtype=np.uint64)
        nn, _ = d.next(n, out)
        assert nn == n and np.array_equal(out, ints.view(np.uint64))


def test_decoding_statistics(gpu):
    """Options::decodingStats (encodings/common/Encoding.h:160-170): decompression is timed per inflated chunk and is
    a part of the decode time, as velox/selective/tests/SelectiveNimbleReaderTest.cpp:2805-2822 expects; a sample of
    the decode time is a batch.  Bytes moved, planning and device time are this build's additions."""
    pytest.importorskip("pyarrow")
    import host_shim

    f = host_shim.Factory(collect_stats=True)
    rng = np.random.default_rng(22)
    n = 200_000
    ints = np.repeat(rng.integers(0, 500, n // 8 + 1), 8)[:n].astype(np.int64)
    plain = E.encode(E.FixedBitWidth(), ints, W.INT64)
    e = f.create(plain)
    assert np.array_equal(e.materialize(n), ints.view(np.uint64))
    s = f.stats()
    assert s["decodeCPUTimeNanos"]["count"] == 1 and s["decompressCPUTimeNanos"]["count"] == 0
    assert s["encodedBytes"]["sum"] >= len(plain) and s["decodedBytes"]["sum"] >= 8 * n
    assert 0 < s["planNanos"]["sum"] <= s["decodeCPUTimeNanos"]["sum"] and s["deviceNanos"]["sum"] > 0
    # a stripe: one stream whose chunk is compressed as a whole, one whose leaves are, one plain
    zstd_leaves = E.encode(E.RLE(lengths=E.FixedBitWidth(compression="zstd"), values=E.Trivial(compression="zstd")), ints, W.INT64)
    streams = [E.chunk(plain, "zstd"), E.chunk(zstd_leaves), E.chunk(plain)]
    dec = host_shim.StripeDecoder(f, streams)
    for i in range(3):
        out = np.zeros(n, dtype=np.uint64)
        nn, _ = dec.stream(i).next(n, out)
        assert nn == n and np.array_equal(out, ints.view(np.uint64))
    s2 = f.stats()
    assert s2["decodeCPUTimeNanos"]["count"] == 2  # the stripe was one batch
    assert s2["decompressCPUTimeNanos"]["count"] >= 1  # the chunk with compressed leaves (whole-chunk compression is undone before the batch)
    assert 0 < s2["decompressCPUTimeNanos"]["sum"] <= s2["decodeCPUTimeNanos"]["sum"]
    assert s2["decodedBytes"]["sum"] - s["decodedBytes"]["sum"] >= 3 * 8 * n
    assert s2["decodeCPUTimeNanos"]["min"] <= s2["decodeCPUTimeNanos"]["max"]


def test_device_resident_batch(factory, oracle):
    """decodeToDevice leaves the decoded chunks in device memory; copied back,
    they are what the oracle decodes."""
    import host_shim
    from nimble_amd.decoder import slice_chunk
    from util import assert_chunk_equal, oracle_chunk

    rng = np.random.default_rng(31)
    n = 30_000
    mask = rng.random(n) < 0.7
    encs = [
        E.encode(E.FixedBitWidth(exact_bits=True), rng.integers(0, 1 << 11, n).astype(np.int32), W.INT32),
        E.encode(E.RLE(), np.repeat(rng.integers(0, 9, n // 5), 5).astype(np.int64), W.INT64),
        E.encode_nullable(E.Nullable(data=E.Trivial(), nulls=E.SparseBool()), rng.integers(0, 99, int(mask.sum())).astype(np.int64), mask, W.INT64),
        E.encode(E.Dictionary(), random_strings(rng, 12).take(rng.integers(0, 12, n)), W.STRING),
    ]
    batch = host_shim.DeviceBatch(factory, encs)
    assert batch.device_ptr != 0 and batch.bytes > 0
    host = batch.to_host()
    for enc, lay in zip(encs, batch.layouts):
        assert_chunk_equal(slice_chunk(host, lay), oracle_chunk(oracle, enc), "device batch")


def test_encoding_view_random_access(factory, oracle):
    """createEncodingView: readAt / range read are cursor free and agree with the
    oracle's decode for every encoding the reference has a view for
    (encodings/views/EncodingViewFactory.cpp:47-135, tests/EncodingViewTest.cpp)."""
    import host_shim

    rng = np.random.default_rng(41)
    n = 3000
    runs = np.repeat(rng.integers(0, 1 << 30, n // 9 + 1), 9)[:n].astype(np.int64)
    cases = [(spec, runs, W.INT64) for spec in (E.Trivial(), E.FixedBitWidth(exact_bits=True), E.RLE(), E.Dictionary(),
                                                E.MainlyConstant(), E.BlockBitPacking(block_size=100), E.FOR())]
    cases.append((E.Constant(), np.full(n, 7, dtype=np.int32), W.INT32))
    cases.append((E.SparseBool(), rng.random(n) < 0.03, W.BOOL))
    cases.append((E.Trivial(), rng.random(n) < 0.5, W.BOOL))
    cases.append((E.BlockBitPacking(), rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32).view(np.float32), W.FLOAT))
    for spec, data, dtype in cases:
        enc = E.encode(spec, data, dtype)
        view = host_shim.HostEncodingView(factory, enc)
        want = oracle.Encoding(enc).materialize(n)
        assert view.row_count == n and view.data_type == dtype and view.encoding_type == enc[0]
        assert host_shim.supports_encoding_view(enc[0])
        for _ in range(20):
            off = int(rng.integers(0, n + 1))
            length = int(rng.integers(0, n - off + 1))
            assert np.array_equal(view.read(off, length), want[off:off + length]), repr(spec)
            i = int(rng.integers(0, n))
            assert view.read_at(i) == want[i]
        assert np.array_equal(view.read(0, n), oracle.Encoding(enc).read(0, n))
        with pytest.raises(host_shim.HostError, match="beyond the row count"):
            view.read(n - 1, 2)
        with pytest.raises(host_shim.HostError, match="beyond the row count"):
            view.read_at(n)
    # strings: the view owns the bytes its string_views point into
    alpha = random_strings(rng, 30)
    col = alpha.take(rng.integers(0, 30, 500))
    for spec in (E.Trivial(), E.Dictionary(), E.RLE(values=E.Trivial()), E.MainlyConstant(other=E.Trivial())):
        view = host_shim.HostEncodingView(factory, E.encode(spec, col, W.STRING))
        lens, blob = view.read(100, 50)
        sub = col.take(np.arange(100, 150))
        assert np.array_equal(lens, sub.lengths) and np.array_equal(blob, sub.blob)
        lens, blob = view.read_at(499)
        assert blob.tobytes() == col.to_list()[499]
    # no view for Nullable / Varint, as in the reference
    for enc in (E.encode(E.Varint(), runs, W.INT64),
                E.encode_nullable(E.Nullable(), runs[:10], np.ones(10, dtype=bool), W.INT64)):
        assert not host_shim.supports_encoding_view(enc[0])
        with pytest.raises(host_shim.HostError, match="does not support EncodingView"):
            host_shim.HostEncodingView(factory, enc)


def test_block_encodings_through_the_host_layer(factory):
    """Encoding contract (cursor, skip, scattered materialize) over BlockBitPacking / FOR chunks,
    incl. compressed packed blocks inflated by the host pre-pass."""
    pytest.importorskip("pyarrow")
    rng = np.random.default_rng(43)
    for dtype in (W.INT16, W.INT32, W.UINT64):
        data = (np.repeat(rng.integers(0, 1000, 20), 100) + rng.integers(0, 9, 2000)).astype(W.physical_dtype(dtype))
        for spec in (E.BlockBitPacking(block_size=128), E.FOR(), E.BlockBitPacking(compression="zstd"), E.FOR(compression="lz4")):
            e = factory.create(E.encode(spec, data, dtype))
            assert e.row_count == 2000 and e.data_type == dtype
            check_materialize_patterns(e, phys(data, dtype), dtype, rng)
            e.reset()
            check_scattered_materialize(e, phys(data, dtype), dtype, rng)


def test_output_copy_in_slices(factory):
    """A stripe whose decoded output spans several slices of the copy back to the
    host: streams read last-first (each waits for its own slices only), a stripe
    dropped while its copy is still in flight (its pinned block must not be reused
    before the copy has landed), and a following stripe that reuses the buffers."""
    import host_shim

    rng = np.random.default_rng(321)
    rows = 1_500_000  # 12 MB per int64 stream: 9 streams = 108 MB of output, four 32 MiB slices

    def make(seed_shift):
        cols = [(rng.integers(0, 1 << 40, rows) + seed_shift + i).astype(np.int64) for i in range(9)]
        streams = [E.stream([E.encode(E.FixedBitWidth(), c[: rows // 2], W.INT64), E.encode(E.Trivial(), c[rows // 2:], W.INT64)])
                   for c in cols]
        return cols, streams

    cols_a, streams_a = make(0)
    cols_b, streams_b = make(7)
    for _ in range(2):
        dropped = host_shim.StripeDecoder(factory, streams_a)
        del dropped  # nothing read: the copy back is still running when the decoders go away
        stripe = host_shim.StripeDecoder(factory, streams_b)
        for i in reversed(range(9)):
            out = np.zeros(rows, dtype=np.int64)
            nn, _ = stripe.stream(i).next(rows, out)
            assert nn == rows and np.array_equal(out, cols_b[i]), i
        stripe2 = host_shim.StripeDecoder(factory, streams_a)  # queued behind the first one on the same stream
        out = np.zeros(rows, dtype=np.int64)
        stripe2.stream(4).next(rows, out)
        assert np.array_equal(out, cols_a[4])
        out2 = np.zeros(1000, dtype=np.int64)
        d = stripe.stream(0)
        d.reset()
        d.skip(rows - 1000)
        d.next(1000, out2)
        assert np.array_equal(out2, cols_b[0][-1000:])


def test_stripes_in_flight_do_not_overwrite_each_other(factory):
    """Five stripes made back to back, read only afterwards.  Their encodings
    are tiny (long runs) and their outputs large (256 MB each), so a stripe is
    made in a fraction of the time its copy back takes: the third stripe gets
    the device buffers of the first while that one's output is still on its way
    to the host, and its decode has to wait for that copy."""
    import host_shim

    rows, n_streams = 4_000_000, 8
    run = 50_000

    def make(k):
        cols, streams = [], []
        for i in range(n_streams):
            values = (np.arange(rows // run, dtype=np.int64) * 7919 + 1000003 * k + 17 * i) & ((1 << 47) - 1)
            col = np.repeat(values, run)
            cols.append(col)
            streams.append(E.stream([E.encode(E.RLE(), col, W.INT64)]))
        return cols, streams

    made = [make(k) for k in range(5)]
    assert sum(len(s) for s in made[0][1]) < 1 << 20  # the encodings really are small
    stripes = [host_shim.StripeDecoder(factory, streams) for _, streams in made]
    out = np.zeros(rows, dtype=np.int64)
    for k, (cols, _) in enumerate(made):
        for i in range(n_streams):
            nn, _ = stripes[k].stream(i).next(rows, out)
            assert nn == rows and np.array_equal(out, cols[i]), (k, i)


def test_copy_to_host_by_kernel(gpu):
    """nbl_copy_to_host_async -- the copy back of a decoded batch without a copy
    engine -- moves every byte, the ragged tail behind the last 16-byte vector
    included, touches nothing behind it, and refuses pointers it cannot vector."""
    import ctypes as C

    import torch

    from nimble_amd import capi

    lib = capi.lib()
    rng = np.random.default_rng(5)
    cap = (64 << 20) + 64
    host = torch.empty(cap, dtype=torch.uint8).pin_memory()
    src = torch.from_numpy(rng.integers(0, 256, cap, dtype=np.uint8)).to("cuda:0")
    stream = torch.cuda.Stream()
    for n in (0, 1, 15, 16, 17, 4096 + 5, (1 << 20) + 3, 64 << 20):
        host.fill_(0xA5)
        torch.cuda.synchronize()
        capi.check(lib.nbl_copy_to_host_async(C.c_void_p(host.data_ptr()), C.c_void_p(src.data_ptr()), n, C.c_void_p(stream.cuda_stream)))
        stream.synchronize()
        got = host.numpy()
        assert np.array_equal(got[:n], src[:n].cpu().numpy()), n
        assert (got[n:n + 64] == 0xA5).all(), n
    assert lib.nbl_copy_to_host_async(C.c_void_p(host.data_ptr() + 4), C.c_void_p(src.data_ptr()), 64, None) == capi.NBL_ERR_ARGUMENT
    assert b"16-byte" in lib.nbl_last_error()
