"""GPU suite (-m gpu): the raster-serial kernels are RESUMABLE (csrc/serial_engine.h) -- an image is worked
through a bounded number of rows per launch, its state record carries it from one launch to the next -- and the
drop-in decoders fetch a stream of unknown length on demand.  Everything against reference-held goldens and the
oracle, bit for bit; plus the damaged-input cases the advisor asked for (a stream may fail, never hang or fault)."""
import ctypes as C
import hashlib
import mmap
import time

import numpy as np
import pytest

import inputs

pytestmark = pytest.mark.gpu
u8p = C.POINTER(C.c_uint8)


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def test_model_and_decoders_resume_across_launches(gpu_ctx, pkg, golden, oracle):
    """The 64 x 16384 -n0 -e3 strip (config 5's width: LDS row cache boundary, 29 MB of statistics) in launches of
    16 rows -- four launches, three resumptions -- must give the reference's stream (7f9fa242...); its decode,
    resumed the same way, the input.  Then every mode at one and three rows per launch on small frames, encoders
    and decoders, NBLIC and QNBLIC, against the oracle."""
    manifest, _ = golden
    m = manifest["serial"]["syn1s1_64x16384_n0_e3"]
    img = inputs.syn1(64, 16384, 1)
    ctx = pkg.Context(device=0, n_slots=12, n_coders=2, n_groups=2, n_host_buffers=24)
    try:
        ctx.set_serial_rows(16)
        before = ctx.serial_launches()
        streams, recs = ctx.encode_modes([img], [0], [3])
        assert ctx.serial_launches() - before >= 4
        assert (len(streams[0]), sha(streams[0])) == (m["len"], m["sha256"])
        assert m["sha256"].startswith("7f9fa2424448")
        before = ctx.serial_launches()
        d = ctx.decode_batch(streams)[0]
        assert ctx.serial_launches() - before >= 4
        assert d is not None and np.array_equal(d[0], img) and d[1:] == (0, 3)
        modes = [(0, 2), (0, 3), (1, 1), (2, 2), (9, 1), (3, 3), (2, 1), (5, 2)]
        cases = []
        for k, (near, effort) in enumerate(modes):
            h, w = [(17, 13), (40, 37), (9, 130), (64, 64)][k % 4]
            cases.append((inputs.syn1(h, w, k + 2) if k % 2 else inputs.make("noise", h, w), near, effort))
        for rows in (1, 3):
            ctx.set_serial_rows(rows)
            ss, rr = ctx.encode_modes([c[0] for c in cases], [c[1] for c in cases], [c[2] for c in cases])
            q = ctx.qencode_batch([c[0] for c in cases[:4]])
            dec = ctx.decode_batch(ss + q)
            for (im, near, effort), s, r, dd in zip(cases, ss, rr, dec):
                ws, wrec, *_ = oracle.encode(im, near, effort)
                assert s == ws and np.array_equal(r, wrec), (rows, im.shape, near, effort)
                assert dd is not None and np.array_equal(dd[0], wrec) and dd[1:] == (near, effort), (rows, im.shape, near, effort)
            for (im, _, _), dd in zip(cases[:4], dec[len(cases):]):
                assert dd is not None and np.array_equal(dd[0], im), rows
    finally:
        ctx.close()


def test_dropin_decoders_fetch_the_stream_on_demand(gpu_ctx, pkg, oracle):
    """NBLICdecompress / QNBLICdecompress get no length (NBLIC.h:72, QNBLIC.h:16): the stream is fetched in steps;
    no more than the stream plus one step may have been read, and the image must be the oracle's."""
    img = inputs.syn1(256, 320, 4)
    try:
        for step in (4096, 16384):
            pkg.set_default_feed_chunk(step)
            for near, effort in [(0, 1), (2, 2), (0, 3)]:
                s = oracle.encode(img, near, effort)[0]
                padded = s + bytes(1 << 20)                                # a caller's buffer is usually larger than the stream
                launches = pkg.default_serial_launches()
                d = pkg.decompress(padded)
                want = oracle.decode(s)
                assert d is not None and np.array_equal(d[0], want[0]) and d[1:] == (near, effort)
                # the margin (4 w + 1024) makes the decoder ask for the next step a little early: one more step at most
                assert pkg.last_fed_bytes() <= len(s) + 2 * step + 4 * 320 + 1024, (step, near, effort, pkg.last_fed_bytes(), len(s))
                if step == 4096 and effort == 1:
                    assert pkg.default_serial_launches() - launches >= len(s) // step - 2      # it really was resumed step by step
            q = oracle.qencode(img)
            dq = pkg.qdecompress(q + bytes(1 << 20))
            assert dq is not None and np.array_equal(dq, img)
            assert pkg.last_fed_bytes() <= max(len(q), 65536) + 2 * step + 2 * 320 + 8
    finally:
        pkg.set_default_feed_chunk(0)
    d = pkg.decompress(oracle.encode(img, 0, 1)[0])                            # default step: the whole stream in one go
    assert d is not None and np.array_equal(d[0], img)


def test_dropin_decoder_at_the_very_end_of_a_mapping(gpu_ctx, pkg, oracle):
    """The stream's last byte is the last byte of a readable page and the next page is PROT_NONE: the reference reads
    exactly the stream and so must we -- the steps are copied by the kernel (a pipe write), which stops at the
    boundary instead of faulting."""
    libc = C.CDLL(None, use_errno=True)
    libc.mprotect.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    page = mmap.PAGESIZE
    lib = pkg.load_library()
    for img, near, effort, q in [(inputs.syn1(96, 128, 2), 0, 1, False), (inputs.syn1(40, 64, 3), 2, 2, False), (inputs.syn1(96, 128, 5), 0, 0, True)]:
        s = oracle.qencode(img) if q else oracle.encode(img, near, effort)[0]
        pages = (len(s) + page - 1) // page
        mm = mmap.mmap(-1, (pages + 1) * page)
        base = C.addressof(C.c_char.from_buffer(mm))
        start = pages * page - len(s)
        start &= ~1                                                        # QNBLIC takes uint16_t *
        mm[start:start + len(s)] = s
        assert libc.mprotect(base + pages * page, page, 0) == 0            # PROT_NONE behind the stream
        out = np.zeros(img.shape, np.uint8)
        hh, ww, n, e = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        if q:
            rc = lib.QNBLICdecompress(C.cast(base + start, C.POINTER(C.c_uint16)), out.ctypes.data_as(u8p), C.byref(hh), C.byref(ww))
        else:
            rc = lib.NBLICdecompress(0, C.cast(base + start, u8p), out.ctypes.data_as(u8p), C.byref(hh), C.byref(ww), C.byref(n), C.byref(e))
        assert rc == 0 and (hh.value, ww.value) == img.shape
        want = img if q else oracle.decode(s)[0]
        assert np.array_equal(out, want)
        assert pkg.last_fed_bytes() <= len(s) + 1                           # not a byte beyond the mapping (the odd start may add one in front)
        assert libc.mprotect(base + pages * page, page, 3) == 0
        del base
        mm.close()


def test_damaged_streams_fail_fast_and_never_fault(gpu_ctx, pkg, oracle):
    """(advisor, round 2) A QNBLIC stream whose histogram tables do not parse (it ends inside them) must never reach
    the GPU; four megabytes of nonsense behind a QNBLIC header parse as SOME tables and may decode to anything, but
    must not fault; an NBLIC stream that claims 10^8 pixels at -e3 and holds twenty bytes must fail at once instead
    of walking the whole image; a truncated stream stops at the row it runs dry in."""
    ctx = pkg.Context(device=0, n_slots=4, n_coders=2)
    try:
        short_q = b"Q0.2" + bytes([1, 0, 1, 0]) + bytes(200)                                    # ends inside the first histogram
        long_q = b"Q0.2" + bytes([1, 0, 1, 0]) + bytes([0xFF, 0xFE] * (2 << 20))                # 1 x 1, four megabytes of nonsense
        big = b"NBLIC0.3" + bytes([1, 0x27, 0x10, 0x27, 0x10, 0, 3, 3]) + bytes(4)              # 10000 x 10000, -e3, 20 bytes
        good = oracle.encode(inputs.syn1(200, 300, 9), 0, 2)[0]
        t0 = time.time()
        res = ctx.decode_batch([short_q, long_q, big, good[: len(good) // 2], good])
        assert time.time() - t0 < 60
        assert res[0] is None and res[2] is None and res[3] is None
        assert res[1] is None or res[1][0].shape == (1, 1)
        assert res[4] is not None and np.array_equal(res[4][0], inputs.syn1(200, 300, 9))
    finally:
        ctx.close()
    # (the drop-in decoders get no length: they cannot tell a truncated stream from one followed by readable memory, and
    # neither can the reference; their own refusals -- magic, size -- are in test_gpu_parity.py)


def test_band_stream_equals_one_piece_and_survives_checkpoints(gpu_ctx, pkg, oracle):
    """An image worked through in row bands (nblic_amd_stream): the bytes are the reference's; suspended after EVERY
    band, written down as a checkpoint, the encoder thrown away and a new one -- in a new context -- resumed from the
    checkpoint, the pieces still concatenate to the reference's stream, the running SHA-256 that travels with the
    checkpoint is the stream's, and the reconstruction is the reference's."""
    cases = [(inputs.syn1(200, 300, 3), 2, 2, 37), (inputs.syn1(96, 700, 4), 0, 3, 16), (inputs.make("noise", 64, 64), 1, 1, 7),
             (inputs.syn1(5, 52000, 6), 3, 2, 2), (inputs.syn1(33, 40, 8), 9, 1, 33)]
    for img, near, effort, band in cases:
        want, wrec, *_ = oracle.encode(img, near, effort)
        ctx = pkg.Context(device=0, n_slots=2, n_coders=1)
        try:
            st = ctx.stream(img, near, effort, band_rows=band)
            done, whole = st.run()
            assert done and whole == want, (img.shape, near, effort)
            assert st.progress()["sha256"] == hashlib.sha256(want).hexdigest()
            assert np.array_equal(st.recon()[0], wrec)
            st.close()
        finally:
            ctx.close()
        pieces, ck, steps, rec = [], None, 0, np.zeros_like(img)
        while True:
            ctx = pkg.Context(device=0, n_slots=2, n_coders=1)
            try:
                st = ctx.stream(img, near, effort, band_rows=band, checkpoint=ck)
                done, piece = st.run(1e-9)                               # the budget is spent after one band
                pieces.append(piece)
                steps += 1
                _, r0, r1 = st.recon(rec)                                # every object hands over the rows IT coded
                assert (r0, r1) == ((steps - 1) * band, min(steps * band, img.shape[0]))
                if done:
                    prog = st.progress()
                    st.close()
                    break
                ck = st.checkpoint()
                st.close()
            finally:
                ctx.close()
        assert steps == (img.shape[0] + band - 1) // band
        assert b"".join(pieces) == want, (img.shape, near, effort)
        assert prog["sha256"] == hashlib.sha256(want).hexdigest() and prog["bytes_total"] == len(want)
        assert np.array_equal(rec, wrec)


def test_two_wave_effort3_launches_are_exact_and_repeatable(gpu_ctx, pkg, oracle):
    """Effort-3 launches of at most 64 images give every image a second wave for the pixel's second system
    (serial_engine.hip model_body<..., WAVES = 2>: barriers order the hand-overs through LDS).  64 images side by side,
    three times over: every stream the oracle's, every time; then 65 images (one wave each) give the same streams."""
    imgs = [inputs.syn1(40 + (k % 5) * 7, 48 + (k % 7) * 9, k + 1) if k % 3 else inputs.make("noise", 33 + k % 11, 57) for k in range(65)]
    nears = [(0, 1, 3)[k % 3] for k in range(65)]
    want = [oracle.encode(im, n, 3)[0] for im, n in zip(imgs, nears)]
    ctx = pkg.Context(device=0, n_slots=65, n_coders=4, n_groups=1, n_host_buffers=80)
    try:
        for rep in range(3):
            got, _ = ctx.encode_modes(imgs[:64], nears[:64], [3] * 64)
            assert got == want[:64], rep
        got, _ = ctx.encode_modes(imgs, nears, [3] * 65)
        assert got == want
    finally:
        ctx.close()


def test_lean_decoder_many_images_side_by_side(gpu_ctx, pkg, oracle):
    """Decode launches of more than 256 images use the lean LDS image (serial_engine.hip DecodeLdsLean: the re-mappers'
    hit counts stay in the image's state record in memory, four waves per CU instead of one).  300 streams per effort,
    lossless and near-lossless, in one piece and in launches of 5 rows (the counts survive in the record between
    launches), against the oracle's reconstructions; then 256 of them (the full LDS image) give the same planes."""
    n = 300
    cases = []
    for k in range(n):
        h, w = 12 + (k % 7) * 3, 16 + (k % 5) * 9
        im = inputs.syn1(h, w, k + 1) if k % 4 else inputs.make("noise", h, w)
        cases.append((im, (0, 2, 0, 5)[k % 4]))
    ctx = pkg.Context(device=0, n_slots=8, n_coders=2, n_groups=1, n_host_buffers=16)
    try:
        for effort in (1, 2, 3):
            enc = [oracle.encode(im, near, effort) for im, near in cases]
            streams = [e[0] for e in enc]
            for rows in (0, 5):
                ctx.set_serial_rows(rows)
                dec = ctx.decode_batch(streams)
                for (im, near), e, d in zip(cases, enc, dec):
                    assert d is not None and np.array_equal(d[0], e[1]) and d[1:] == (near, effort), (effort, rows, im.shape, near)
            ctx.set_serial_rows(0)
            dec = ctx.decode_batch(streams[:256])
            assert all(d is not None and np.array_equal(d[0], e[1]) for e, d in zip(enc, dec)), effort
    finally:
        ctx.close()
