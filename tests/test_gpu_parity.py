"""GPU suite (-m gpu): the HIP path, called through the C ABI, against the oracle and the
committed golden fixtures.  Bit-exact is the bar for every stage and every stream."""
import hashlib

import numpy as np
import pytest

import inputs

pytestmark = pytest.mark.gpu


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def unpack_rec1(r):
    r = r.astype(np.uint32)
    px0 = r & 0xFF
    adr = (r >> 8) & 0x7FF
    qw = (r >> 19) & 31
    qu = (((r >> 16) & 7) << 1) | ((r >> 24) & 1)
    rel = (r >> 25) & 3
    qv = np.where(rel == 0, qu, np.where(rel == 1, qu + 1, qu - 1))
    return px0, adr, qu, qv, qw


def test_wave_primitives_selftest(gpu_ctx):
    assert gpu_ctx.selftest() == 0


STAGE_SHAPES = [(1, 1), (1, 9), (9, 1), (2, 2), (3, 5), (17, 13), (40, 37), (64, 64), (96, 128), (5, 300)]


@pytest.mark.parametrize("shape", STAGE_SHAPES)
def test_stage_parity(gpu_ctx, oracle, shape):
    h, w = shape
    for content in ("syn1", "noise", "checker", "const", "ramp"):
        img = inputs.make(content, h, w)
        st = oracle.stages(img)
        px0, adr, qu, qv, qw = unpack_rec1(gpu_ctx.debug_stage(img, "rec1"))
        assert np.array_equal(px0, st["px0"]), ("px0", content)
        assert np.array_equal(adr, st["adr"]), ("adr", content)
        assert np.array_equal(qu, st["qu"]) and np.array_equal(qv, st["qv"]) and np.array_equal(qw, st["qw"]), ("level", content)
        pxs = gpu_ctx.debug_stage(img, "pxs")
        assert np.array_equal(pxs & 0xFF, st["px"]) and np.array_equal(pxs >> 8, st["sign"]), ("S2", content)
        assert np.array_equal(gpu_ctx.debug_stage(img, "z"), st["z"]), ("S3", content)
        assert np.array_equal(gpu_ctx.debug_stage(img, "cnt"), st["ev_count"]), ("S4 count", content)
        ev = gpu_ctx.debug_stage(img, "events")
        assert len(ev) == len(st["cu"])
        e_qu, e_qv, node = ev & 15, (ev >> 4) & 15, (ev >> 8) & 255
        assert np.array_equal(e_qu * 256 + node, st["cu"]) and np.array_equal(e_qv * 256 + node, st["cv"]), ("S4 path", content)
        assert np.array_equal((ev >> 16) & 31, st["ev_qw"]) and np.array_equal((ev >> 21) & 1, st["ev_bin"]), ("S4 bins", content)
        coded = gpu_ctx.debug_stage(img, "coded")
        assert np.array_equal(coded & 0xFFF, st["prob"]), ("S5 prob", content)
        assert np.array_equal(coded >> 15, st["ev_bin"]), ("S5 bin", content)


def test_batch_streams_equal_golden(gpu_ctx, golden):
    _, streams = golden
    imgs, want = [], []
    for (h, w) in inputs.SMALL_SHAPES:
        for content in inputs.CONTENTS:
            imgs.append(inputs.make(content, h, w))
            want.append(streams[inputs.case_id(content, h, w, 0, 1)].tobytes())
    got = gpu_ctx.encode_batch(imgs)
    for k, (g, wnt) in enumerate(zip(got, want)):
        assert g == wnt, k


def test_mixed_size_batch_equals_oracle(gpu_ctx, oracle):
    from oracle.oracle import syn1
    imgs = [syn1(512, 512, 1), inputs.make("noise", 300, 200), syn1(768, 512, 5), inputs.make("const", 1, 1),
            syn1(100, 1000, 3), inputs.make("checker", 257, 255), syn1(512, 768, 2)]
    got = gpu_ctx.encode_batch(imgs)
    for img, g in zip(imgs, got):
        assert g == oracle.encode(img, 0, 1)[0], img.shape


def test_flat_and_structured_images_multi_block(gpu_ctx, oracle):
    """Long context chains whose coupled warm-up cannot meet (constant error) must fall back to
    the in-order replay and still be exact; mixtures exercise both paths inside one chain."""
    half = inputs.make("const", 200, 300).copy()
    half[:, 150:] = inputs.make("noise", 200, 150)
    bands = inputs.make("ramp", 256, 256).copy()
    bands[64:192] = 200
    imgs = [inputs.make("const", 200, 300), inputs.make("ramp", 256, 256), inputs.make("checker", 300, 300), half, bands,
            np.zeros((130, 1000), np.uint8), np.full((500, 90), 255, np.uint8)]
    got = gpu_ctx.encode_batch(imgs)
    for k, (img, g) in enumerate(zip(imgs, got)):
        assert g == oracle.encode(img, 0, 1)[0], k


def test_output_capacity_is_enforced(gpu_ctx):
    img = inputs.make("noise", 64, 64)
    outs = [np.empty(600, np.uint8)]                      # noise needs > 4 KB
    with pytest.raises(RuntimeError):
        gpu_ctx.encode_ptrs([img.ctypes.data], [img.shape], False, outs)


def test_config2_4096_golden_and_round_trip(gpu_ctx, golden, oracle):
    """BASELINE config 2: 4096x4096 SYN-1, -n0 -e1.  Golden hash from the compiled reference;
    the stream must also decode (oracle decoder) back to the input."""
    from oracle.oracle import syn1
    manifest, _ = golden
    m = manifest["large"]["syn1s1_4096x4096_n0_e1"]
    img = syn1(4096, 4096, 1)
    assert sha(img.tobytes()) == m["input_sha256"]
    s = gpu_ctx.encode_batch([img])[0]
    assert len(s) == m["len"] == 8900446
    assert sha(s) == m["sha256"]
    dec = oracle.decode(s)
    assert dec is not None and np.array_equal(dec[0], img)


def test_device_resident_inputs(gpu_ctx, oracle):
    torch = pytest.importorskip("torch")
    from oracle.oracle import syn1
    imgs = [syn1(256, 384, s) for s in (1, 2, 3, 4)]
    dev = [torch.from_numpy(i).cuda() for i in imgs]
    torch.cuda.synchronize()
    outs, lens = gpu_ctx.encode_ptrs([d.data_ptr() for d in dev], [i.shape for i in imgs], True)
    for img, o, n in zip(imgs, outs, lens):
        assert o[:int(n)].tobytes() == oracle.encode(img, 0, 1)[0]


# ---- drop-in entry points, every mode (serial engine for everything but -n0 -e1) ------------
@pytest.mark.parametrize("near,effort", inputs.PARAM_CLASSES)
def test_dropin_compress_all_modes_small(pkg, golden, near, effort):
    manifest, streams = golden
    for (h, w) in [(1, 1), (1, 7), (7, 1), (3, 5), (17, 13)]:
        for content in ("syn1", "noise", "checker"):
            img = inputs.make(content, h, w)
            cid = inputs.case_id(content, h, w, near, effort)
            s, rec, n_out, e_out = pkg.compress(img, near, effort)
            assert s == streams[cid].tobytes(), cid
            m = manifest["small"][cid]
            assert (n_out, e_out) == (m["near_out"], m["effort_out"])
            assert sha(rec.tobytes()) == m["recon_sha256"], cid


@pytest.mark.parametrize("near,effort", [(0, 1), (2, 1), (0, 2), (3, 3)])
def test_dropin_decompress_reference_streams(pkg, golden, oracle, near, effort):
    _, streams = golden
    for (h, w) in [(1, 1), (2, 2), (5, 3), (17, 13), (64, 64) if effort == 1 else (17, 13)]:
        img = inputs.make("syn1", h, w)
        s = streams[inputs.case_id("syn1", h, w, near, effort)].tobytes()
        d = pkg.decompress(s)
        assert d is not None
        want = oracle.decode(s)
        assert np.array_equal(d[0], want[0]) and d[1:] == want[1:]
        assert int(np.abs(d[0].astype(int) - img.astype(int)).max()) <= near


def test_dropin_rejects_bad_input(pkg):
    assert pkg.decompress(b"Q0.2" + bytes(60)) is None                  # not an NBLIC stream
    assert pkg.compress(np.zeros((1, 1), np.uint8), 0, 1)[0] is not None


# ---- QNBLIC (effort 0): BASELINE config 1 on the GPU path -------------------------------------
def test_q_batch_streams_equal_golden(gpu_ctx, golden):
    _, streams = golden
    imgs, want = [], []
    for (h, w) in inputs.SMALL_SHAPES:
        for content in inputs.CONTENTS:
            imgs.append(inputs.make(content, h, w))
            want.append(streams[f"q_{content}_{h}x{w}"].tobytes())
    got = gpu_ctx.qencode_batch(imgs)
    for k, (g, wnt) in enumerate(zip(got, want)):
        assert g == wnt, k


def test_q_config1_512_and_4096_hashes(gpu_ctx, golden, oracle):
    """BASELINE config 1 (512x512 SYN-1, effort 0): golden 50fdb1a0...; and the 4096^2 frame."""
    from oracle.oracle import syn1
    manifest, _ = golden
    for key, hw in (("syn1s1_512x512_q0", 512), ("syn1s1_4096x4096_q0", 4096)):
        img = syn1(hw, hw, 1)
        s = gpu_ctx.qencode_batch([img])[0]
        assert len(s) == manifest["large"][key]["len"] and sha(s) == manifest["large"][key]["sha256"], key
    assert manifest["large"]["syn1s1_512x512_q0"]["sha256"].startswith("50fdb1a0a3cac171")
    small = syn1(300, 200, 3)
    s = gpu_ctx.qencode_batch([small, inputs.make("noise", 100, 333), inputs.make("const", 150, 150)])
    assert s[0] == oracle.qencode(small) and s[1] == oracle.qencode(inputs.make("noise", 100, 333))
    assert s[2] == oracle.qencode(inputs.make("const", 150, 150))


def test_q_dropin_round_trip(pkg, golden, oracle):
    _, streams = golden
    for (h, w) in [(1, 1), (1, 7), (7, 1), (3, 5), (17, 13), (64, 64), (2, 256)]:
        for content in ("syn1", "noise", "checker"):
            img = inputs.make(content, h, w)
            want = streams[f"q_{content}_{h}x{w}"].tobytes()
            assert pkg.qcompress(img) == want, (content, h, w)
            dec = pkg.qdecompress(want)                               # a stream the REFERENCE produced
            assert dec is not None and np.array_equal(dec, img), (content, h, w)
    assert pkg.qdecompress(b"NBLIC0.3" + bytes(40)) is None


def test_config3_kodak_shaped_batch(gpu_ctx, oracle):
    """BASELINE config 3: 24 images of Kodak's shapes (18 of 512x768 rows x cols... 768x512 and 6
    portrait), batched.  The Kodak files themselves never leave the build container, so the GPU
    box runs SYN-1 frames of the same shapes (seeds 1..24) against the oracle; parity of the
    oracle on the real Kodak pixels is pinned on the CPU side (test_oracle.py)."""
    from oracle.oracle import syn1
    shapes = [(512, 768)] * 18 + [(768, 512)] * 6
    imgs = [syn1(h, w, seed) for seed, (h, w) in enumerate(shapes, start=1)]
    got = gpu_ctx.encode_batch(imgs)
    gotq = gpu_ctx.qencode_batch(imgs[:6] + imgs[-3:])
    for k in (0, 5, 11, 17, 18, 23):
        assert got[k] == oracle.encode(imgs[k], 0, 1)[0], k
    assert len(set(got)) == 24
    for g, im in zip(gotq, imgs[:6] + imgs[-3:]):
        assert g == oracle.qencode(im)


@pytest.mark.parametrize("key", ["syn1s1_256x256_n0_e2", "syn1s1_256x256_n0_e3", "syn1s1_256x256_n2_e2", "syn1s1_512x512_n2_e1"])
def test_serial_modes_moderate_sizes(pkg, golden, oracle, key):
    """Stand-ins for BASELINE configs 4/5 (raster-serial modes) at sizes the serial engine finishes
    in seconds: stream and reconstruction hashes from the compiled reference, then a GPU decode."""
    from oracle.oracle import syn1
    manifest, _ = golden
    m = manifest["large"][key]
    name, dims, n, e = key.split("_")
    h, w = map(int, dims.split("x"))
    img = syn1(h, w, int(name[5:]))
    s, rec, _, _ = pkg.compress(img, int(n[1:]), int(e[1:]))
    assert (len(s), sha(s)) == (m["len"], m["sha256"])
    assert sha(rec.tobytes()) == m["recon_sha256"]
    d = pkg.decompress(s)
    assert d is not None and np.array_equal(d[0], rec) and d[1:] == (int(n[1:]), int(e[1:]))


def test_extreme_shapes_and_buffer_growth(gpu_ctx, oracle):
    """Maximum width / height (65535, NBLIC.h:29-30), and a noise frame whose 9.5 bins/px exceed the
    6 bins/px the event buffers are provisioned for (they are re-sized from the measured total)."""
    from oracle.oracle import syn1
    imgs = [syn1(1, 65535, 3), syn1(65535, 1, 4), syn1(3, 40000, 5), inputs.noise(1024, 1024, 11)]
    got = gpu_ctx.encode_batch(imgs)
    for k, (img, g) in enumerate(zip(imgs, got)):
        assert g == oracle.encode(img, 0, 1)[0], k
    gq = gpu_ctx.qencode_batch(imgs[:3])
    for img, g in zip(imgs[:3], gq):
        assert g == oracle.qencode(img)


def test_size_limits(pkg, gpu_ctx):
    # the reference refuses > 100,000,000 pixels (NBLIC.c:726): so do the drop-in symbols and the batch API
    big = np.zeros((1, 1), np.uint8)
    import ctypes as C
    lib = pkg.load_library()
    out = np.empty(64, np.uint8)
    n, e = C.c_int(0), C.c_int(1)
    u8p = C.POINTER(C.c_uint8)
    assert lib.NBLICcompress(0, out.ctypes.data_as(u8p), big.ctypes.data_as(u8p), 10001, 10000, C.byref(n), C.byref(e)) == -1
    assert lib.NBLICcompress(0, out.ctypes.data_as(u8p), big.ctypes.data_as(u8p), 0, 5, C.byref(n), C.byref(e)) == -1
    assert lib.QNBLICcompress(out.ctypes.data_as(C.POINTER(C.c_uint16)), big.ctypes.data_as(u8p), 70000, 1) == -1
    with pytest.raises(RuntimeError):
        gpu_ctx.encode_ptrs([big.ctypes.data], [(10001, 10000)], False, [np.empty(64, np.uint8)])


@pytest.mark.gpu
def test_sixteen_image_packs_and_chunked_streaming(pkg, oracle):
    """Many images, few coder threads: the coder threads take full 16-image packs (two AVX-512
    registers in lock-step), stream each image's bins from HBM in chunks (256 Kbin here, so the images
    span up to a dozen chunks and all end at different bins) and wait for packs to fill mid-batch.
    Every stream must still be the oracle's, byte for byte."""
    rng = np.random.default_rng(3)
    imgs = []
    for k in range(44):
        h, w = int(rng.integers(200, 760)), int(rng.integers(200, 900))
        imgs.append(inputs.make(inputs.CONTENTS[k % len(inputs.CONTENTS)], h, w) if k % 4 else inputs.syn1(h, w, seed=k + 1))
    import os
    os.environ["NBLIC_AMD_CHUNK_BINS"] = "262144"               # the pipeline's 4 Mbin chunks would swallow these images whole
    try:
        ctx = pkg.Context(device=0, n_slots=16, n_coders=2, n_groups=4, n_host_buffers=64)
    finally:
        del os.environ["NBLIC_AMD_CHUNK_BINS"]
    try:
        got = ctx.encode_batch(imgs)
        again = ctx.encode_batch(imgs[:5])                      # a short batch right after: singles
    finally:
        ctx.close()
    for k, (img, g) in enumerate(zip(imgs, got)):
        assert g == oracle.encode(img, 0, 1)[0], (k, img.shape)
    assert again == got[:5]


@pytest.mark.gpu
def test_production_shaped_context_many_small_images(pkg, oracle):
    """The bench's context shape (6 groups of 8, 16 coder threads, device-side backlog) fed 200
    images of every content and size class, twice over: all groups, driver threads, copy streams and
    coder threads are busy at once, packs form from images that end at wildly different bins."""
    rng = np.random.default_rng(17)
    imgs = []
    for k in range(200):
        h, w = int(rng.integers(1, 420)), int(rng.integers(1, 640))
        imgs.append(inputs.make(inputs.CONTENTS[k % len(inputs.CONTENTS)], h, w) if k % 3 else inputs.syn1(h, w, seed=k + 1))
    want = [oracle.encode(img, 0, 1)[0] for img in imgs]
    ctx = pkg.Context(device=0, n_slots=48, n_coders=16, n_groups=6, n_host_buffers=336)
    try:
        for _ in range(2):
            got = ctx.encode_batch(imgs)
            assert got == want
    finally:
        ctx.close()


@pytest.mark.gpu
def test_overlapping_batches_begin_end(pkg, oracle):
    """Three batches in flight at once (nblic_amd_encode_batch_begin / _end), ended out of order:
    every batch gets exactly its own streams."""
    rng = np.random.default_rng(23)
    batches = []
    for b in range(3):
        imgs = []
        for k in range(30):
            h, w = int(rng.integers(40, 500)), int(rng.integers(40, 600))
            imgs.append(inputs.make(inputs.CONTENTS[(k + b) % len(inputs.CONTENTS)], h, w) if k % 2 else inputs.syn1(h, w, seed=100 * b + k + 1))
        batches.append(imgs)
    ctx = pkg.Context(device=0, n_slots=12, n_coders=4, n_groups=3, n_host_buffers=64)
    try:
        tickets = [ctx.encode_begin([i.ctypes.data for i in imgs], [i.shape for i in imgs], False) for imgs in batches]
        results = {}
        for b in (1, 0, 2):
            outs, lens = ctx.encode_end(tickets[b])
            results[b] = [o[:int(n)].tobytes() for o, n in zip(outs, lens)]
        again = ctx.encode_batch(batches[0][:3])                 # the synchronous call still works afterwards
    finally:
        ctx.close()
    for b, imgs in enumerate(batches):
        assert results[b] == [oracle.encode(img, 0, 1)[0] for img in imgs], b
    assert again == results[0][:3]
