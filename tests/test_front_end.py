"""The front end either side of the hot path: the reference tool's switch grammar
and its PGM / BMP files.  CPU tests cover the grammar, the file formats (against an independent
construction of the bytes, against the reference's own FileIO.c compiled in place, and on the Kodak
BMPs -- the last two only where /root/reference exists) and that the reference's own main() links
against libnblic_amd.so; the -m gpu tests run the command line end to end."""
import hashlib
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

import inputs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SRC = "/root/reference/src"


def bmp_bytes(img):
    """8-bit gray BMP as FileIO.c:229-287 lays it out, built independently here."""
    h, w = img.shape
    stride = (w + 3) & ~3
    head = struct.pack("<HIII", 0x4D42, 1078 + stride * h, 0, 1078)
    dib = struct.pack("<IiiHHIIiiII", 40, w, h, 1, 8, 0, 0, 0xEC4, 0xEC4, 256, 0)
    pal = b"".join(bytes([k, k, k, 0xFF]) for k in range(256))
    rows = b"".join(img[i].tobytes() + bytes(stride - w) for i in range(h - 1, -1, -1))
    return head + dib + pal + rows


# ---- switch grammar (NBLIC_main.c:52-112) ------------------------------------------------------
def test_switch_grammar(pkg):
    p = pkg.cli_parse
    assert p([]) == {"decompress": 0, "near": 0, "effort": 1, "verbose": 0, "multithread": 0, "large": 0, "device": -1, "src": None, "dst": None}
    a = p(["-cn2e2V", "in.bmp", "out.nblic"])
    assert (a["decompress"], a["near"], a["effort"], a["verbose"], a["src"], a["dst"]) == (0, 2, 2, 2, "in.bmp", "out.nblic")
    a = p(["-d", "-v", "x.nblic", "y.pgm", "z.bmp"])                       # the LAST plain argument is the output
    assert (a["decompress"], a["verbose"], a["src"], a["dst"]) == (1, 1, "x.nblic", "z.bmp")
    assert p(["-N12E0t", "a", "b"])["near"] == 12 and p(["-N12E0t", "a", "b"])["effort"] == 0 and p(["-N12E0t", "a", "b"])["multithread"] == 1
    assert p(["-cVtn0e3", "a", "b"])["effort"] == 3
    assert p(["-exn4", "a", "b"])["effort"] == 1 and p(["-exn4", "a", "b"])["near"] == 4    # 'e' eats the next character whatever it is
    assert p(["-n", "a", "b"])["near"] == 0 and p(["-e", "a", "b"])["effort"] == 1
    assert p(["-c", "-d", "-C", "a", "b"])["decompress"] == 0
    assert p(["-zQw", "a", "b"])["src"] == "a"                              # unknown letters are ignored
    assert p(["-Lg3", "a", "b"])["large"] == 1 and p(["-Lg3", "a", "b"])["device"] == 3
    assert p(["only-one"])["dst"] is None


def test_usage_when_files_are_missing(pkg, capfd):
    assert pkg.cli(["-c", "lonely.pgm"]) == -1
    assert "nblic_codec_amd" in capfd.readouterr().out


# ---- files --------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(1, 1), (1, 2), (2, 3), (3, 5), (7, 8), (13, 17), (64, 61)])
def test_pgm_and_bmp_round_trip_and_bytes(pkg, tmp_path, shape):
    img = inputs.make("noise", *shape)
    pgm, bmp = str(tmp_path / "a.pgm"), str(tmp_path / "a.BmP")
    assert pkg.write_gray(pgm, img, False) and pkg.write_gray(bmp, img, True)
    assert open(pgm, "rb").read() == b"P5\n%d %d\n255\n" % (shape[1], shape[0]) + img.tobytes()
    assert open(bmp, "rb").read() == bmp_bytes(img)
    for path, kind in ((pgm, "PGM"), (bmp, "BMP")):
        got = pkg.read_gray(path)
        assert got is not None and got[1] == kind and np.array_equal(got[0], img)


def test_readers_accept_and_reject_like_the_reference(pkg, tmp_path):
    img = inputs.make("ramp", 4, 6)
    def put(name, data):
        p = str(tmp_path / name)
        open(p, "wb").write(data)
        return p
    ok = pkg.read_gray(put("ws.pgm", b"P5 \t6\r\n4\n\n200\n" + img.tobytes() + b"trailing"))
    assert ok is not None and np.array_equal(ok[0], img)                      # any white space, maxval < 255 taken as is, extra bytes ignored
    assert pkg.read_gray(put("c.pgm", b"P5\n# comment\n6 4\n255\n" + img.tobytes())) is None      # fscanf("%d") stops at '#'
    assert pkg.read_gray(put("m.pgm", b"P5\n6 4\n256\n" + img.tobytes())) is None
    assert pkg.read_gray(put("m0.pgm", b"P5\n6 4\n0\n" + img.tobytes())) is None
    assert pkg.read_gray(put("s.pgm", b"P5\n6 4\n255\n" + img.tobytes()[:-1])) is None                # short
    assert pkg.read_gray(put("p2.pgm", b"P2\n6 4\n255\n" + img.tobytes())) is None
    good = bmp_bytes(img)
    assert pkg.read_gray(put("x.anything", good))[1] == "BMP"                                         # probing ignores the name
    assert np.array_equal(pkg.read_gray(put("nopad.bmp", good[:-2]))[0], img)                         # padding of the last stored row may be missing
    assert pkg.read_gray(put("short.bmp", good[:-3])) is None
    bad = bytearray(good); bad[28] = 24
    assert pkg.read_gray(put("rgb.bmp", bytes(bad))) is None
    bad = bytearray(good); bad[30] = 1
    assert pkg.read_gray(put("rle.bmp", bytes(bad))) is None
    bad = bytearray(good); bad[22:26] = struct.pack("<i", -4)
    assert pkg.read_gray(put("topdown.bmp", bytes(bad))) is None
    far = bytearray(good[:1078]) + bytes(10) + good[1078:]; far[10:14] = struct.pack("<I", 1088)
    assert np.array_equal(pkg.read_gray(put("gap.bmp", bytes(far)))[0], img)                          # pixels start where the header says
    assert pkg.read_gray(str(tmp_path / "missing.pgm")) is None


@pytest.mark.skipif(not os.path.isdir(inputs.KODAK_DIR), reason="Kodak BMPs live only in the build container")
def test_kodak_bmps_read_like_the_reference_reader(pkg, golden):
    manifest, _ = golden
    for name in sorted(os.listdir(inputs.KODAK_DIR)):
        got = pkg.read_gray(os.path.join(inputs.KODAK_DIR, name))
        assert got is not None and got[1] == "BMP"
        assert hashlib.sha256(got[0].tobytes()).hexdigest() == manifest["kodak_e1"][name]["input_sha256"], name


@pytest.fixture(scope="module")
def ref_fileio():
    """The reference's own FileIO.c compiled in place into oracle/_ref (container only)."""
    import ctypes as C
    if not os.path.isdir(REF_SRC):
        pytest.skip("reference sources absent")
    so = os.path.join(ROOT, "oracle", "_ref", "libfileio_ref.so")
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-o", so, os.path.join(REF_SRC, "FileIO.c")], check=True)
    return C.CDLL(so)


def test_files_against_the_reference_fileio(pkg, ref_fileio, tmp_path):
    import ctypes as C
    u8p = C.POINTER(C.c_uint8)
    for shape in [(1, 1), (3, 5), (17, 13), (64, 61), (30, 128)]:
        img = inputs.make("noise", *shape)
        for as_bmp, writer, reader in ((False, ref_fileio.writePGMImageFile, ref_fileio.loadPGMImageFile),
                                       (True, ref_fileio.writeBMPGrayImageFile, ref_fileio.loadBMPGrayImageFile)):
            mine, theirs = str(tmp_path / "mine"), str(tmp_path / "theirs")
            assert pkg.write_gray(mine, img, as_bmp)
            assert writer(theirs.encode(), img.ctypes.data_as(u8p), shape[0], shape[1]) == 0
            assert open(mine, "rb").read() == open(theirs, "rb").read(), (shape, as_bmp)
            back = np.zeros_like(img)
            h, w = C.c_int(), C.c_int()
            assert reader(mine.encode(), back.ctypes.data_as(u8p), C.byref(h), C.byref(w)) == 0      # the reference reads what we wrote
            assert (h.value, w.value) == shape and np.array_equal(back, img)
            assert np.array_equal(pkg.read_gray(theirs)[0], img)                                      # and we read what it wrote


@pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="reference sources absent")
def test_reference_main_links_against_the_library(pkg):
    """The reference's own caller (NBLIC_main.c + FileIO.c) built against libnblic_amd.so instead of
    NBLIC.c / QNBLIC.c: it links without an unresolved symbol and starts (usage text, exit status -1)."""
    pkg.build()
    out = os.path.join(ROOT, "tests", "_build", "ref_main_on_amd")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    lib_dir = os.path.dirname(pkg.LIB_PATH)
    r = subprocess.run(["gcc", "-O2", "-w", "-o", out, os.path.join(REF_SRC, "NBLIC_main.c"), os.path.join(REF_SRC, "FileIO.c"),
                        "-L" + lib_dir, "-lnblic_amd", "-Wl,-rpath," + lib_dir, "-Wl,--no-undefined"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([out], capture_output=True, text=True, timeout=120)
    assert r.returncode == 255 and "nblic_codec" in r.stdout


# ---- end to end on the GPU ------------------------------------------------------------------------
@pytest.mark.gpu
def test_cli_round_trips_like_verify_py(pkg, oracle, tmp_path):
    """verify.py's check (verify.py:58-72: max |original - decoded| <= near) through the command line, for
    PGM and BMP inputs and every effort, plus byte equality of each .nblic file with the oracle's stream."""
    img = inputs.syn1(61, 45, seed=9)
    src = {"pgm": str(tmp_path / "in.pgm"), "bmp": str(tmp_path / "in.bmp")}
    assert pkg.write_gray(src["pgm"], img, False) and pkg.write_gray(src["bmp"], img, True)
    for k, (near, effort, fmt) in enumerate([(0, 0, "bmp"), (0, 1, "pgm"), (0, 2, "bmp"), (0, 3, "pgm"), (2, 0, "bmp"), (3, 2, "pgm"), (12, 1, "bmp")]):
        coded, back = str(tmp_path / f"{k}.nblic"), str(tmp_path / (f"{k}.BMP" if k % 2 else f"{k}.pgm"))
        assert pkg.cli([f"-cn{near}e{effort}", src[fmt], coded]) == 0
        data = open(coded, "rb").read()
        if near == 0 and effort == 0:
            assert data == oracle.qencode(img)
        else:
            assert data == oracle.encode(img, near, effort)[0]
        assert pkg.cli(["-d", coded, back]) == 0
        got = pkg.read_gray(back)
        assert got is not None and got[1] == ("BMP" if k % 2 else "PGM")
        assert int(np.abs(got[0].astype(int) - img.astype(int)).max()) <= min(near, 9)
    assert pkg.cli(["-c", str(tmp_path / "nope.pgm"), str(tmp_path / "x.nblic")]) == -1
    assert pkg.cli(["-d", src["pgm"], str(tmp_path / "x.pgm")]) == -1                                # not a stream


@pytest.mark.gpu
def test_cli_executable(pkg, oracle, tmp_path):
    img = inputs.syn1(40, 33, seed=4)
    src, coded, back = str(tmp_path / "in.bmp"), str(tmp_path / "o.nblic"), str(tmp_path / "o.bmp")
    assert pkg.write_gray(src, img, True)
    r = subprocess.run([pkg.CLI_PATH, "-cVn0e1", src, coded], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "input image format = BMP" in r.stdout and "compression bpp" in r.stdout, r.stdout + r.stderr
    assert open(coded, "rb").read() == oracle.encode(img, 0, 1)[0]
    r = subprocess.run([pkg.CLI_PATH, "-d", "-v", coded, back], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "output image format= BMP" in r.stdout
    assert open(back, "rb").read() == bmp_bytes(img)
