"""nblic-image-compression_amd -- MI355X-native NBLIC v0.3 hot path (host-side Python layer).

The product is ``libnblic_amd.so`` (hand-written HIP kernels for gfx950 + a thin C++ host
pipeline), whose C ABI is declared in ``include/nblic_amd.h``.  This module is only the
binding that tests, ``bench.py`` and Python callers use; it mirrors the reference's
operator interface for the path (``NBLICcompress`` / ``NBLICdecompress``: same argument
meaning, same clamping and error conventions, reference: src/NBLIC.h:54,72) and adds the
batch interface.  There is no CPU fallback: if the library is missing or no HIP device is
usable, calls raise.

The directory name contains a hyphen, so import it with
``importlib.import_module("nblic-image-compression_amd")``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Sequence, Tuple

import numpy as np

# The batch pipeline keeps 6 group streams + 8 copy streams busy; the HIP runtime multiplexes
# streams onto 4 hardware queues by default and reads this when it initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NBLIC_AMD_LIB") or os.path.join(_HERE, "libnblic_amd.so")      # (NBLIC_AMD_LIB: another build of the same ABI, for A/B runs on one box)
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(_HERE), "include", "nblic_amd.h")

_u8p = C.POINTER(C.c_uint8)
_lib = None

# every symbol include/nblic_amd.h declares
EXPORTS = (
    "NBLICcompress", "NBLICdecompress", "QNBLICcompress", "QNBLICdecompress", "QNBLICcompressMultiThread",
    "nblic_amd_create", "nblic_amd_create_ex", "nblic_amd_destroy", "nblic_amd_encode_batch", "nblic_amd_encode_batch_begin", "nblic_amd_encode_batch_end", "nblic_amd_qencode_batch", "nblic_amd_set_max_pixels",
    "nblic_amd_enable_timing", "nblic_amd_stage_times", "nblic_amd_last_launches", "nblic_amd_last_stats", "nblic_amd_debug_stage",
    "nblic_amd_encode_batch_modes", "nblic_amd_decode_batch", "nblic_amd_serial_selftest",
    "nblic_amd_set_serial_rows", "nblic_amd_serial_launches", "nblic_amd_set_feed_chunk", "nblic_amd_last_fed_bytes",
    "nblic_amd_stream_begin", "nblic_amd_stream_resume", "nblic_amd_stream_run", "nblic_amd_stream_checkpoint", "nblic_amd_stream_progress",
    "nblic_amd_stream_recon", "nblic_amd_stream_end",
    "nblic_amd_cli_main", "nblic_amd_cli_parse", "nblic_amd_read_gray", "nblic_amd_write_gray",
    "nblic_amd_set_device_coder", "nblic_amd_device_coder_stats",
    "nblic_amd_range_code", "nblic_amd_range_code_multi", "nblic_amd_range_code_chunked", "nblic_amd_selftest", "nblic_amd_syn1", "nblic_amd_version",
)


def build(force: bool = False) -> str:
    """Compile libnblic_amd.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [INCLUDE]
    stale = force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if stale:
        r = subprocess.run(["make", "-C", CSRC], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("building libnblic_amd.so failed:\n" + r.stdout + r.stderr)
    return LIB_PATH


def load_library() -> C.CDLL:
    """dlopen the HIP library.  Raises (loudly) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run __graft_entry__.build() (hipcc, gfx950). "
                           "There is no CPU fallback for the NBLIC hot path.")
    lib = C.CDLL(LIB_PATH)
    ip = C.POINTER(C.c_int)
    lib.NBLICcompress.restype = C.c_int
    lib.NBLICcompress.argtypes = [C.c_int, _u8p, _u8p, C.c_int, C.c_int, ip, ip]
    lib.NBLICdecompress.restype = C.c_int
    lib.NBLICdecompress.argtypes = [C.c_int, _u8p, _u8p, ip, ip, ip, ip]
    u16p = C.POINTER(C.c_uint16)
    lib.QNBLICcompress.restype = C.c_int
    lib.QNBLICcompress.argtypes = [u16p, _u8p, C.c_int, C.c_int]
    lib.QNBLICcompressMultiThread.restype = C.c_int
    lib.QNBLICcompressMultiThread.argtypes = [u16p, _u8p, C.c_int, C.c_int]
    lib.QNBLICdecompress.restype = C.c_int
    lib.QNBLICdecompress.argtypes = [u16p, _u8p, ip, ip]
    lib.nblic_amd_create.restype = C.c_void_p
    lib.nblic_amd_create.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.nblic_amd_create_ex.restype = C.c_void_p
    lib.nblic_amd_create_ex.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.nblic_amd_destroy.restype = None
    lib.nblic_amd_destroy.argtypes = [C.c_void_p]
    lib.nblic_amd_encode_batch.restype = C.c_int
    lib.nblic_amd_encode_batch.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_int, ip, ip,
                                           C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_long)]
    lib.nblic_amd_encode_batch_begin.restype = C.c_void_p
    lib.nblic_amd_encode_batch_begin.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_int, ip, ip,
                                                 C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_long)]
    lib.nblic_amd_encode_batch_end.restype = C.c_int
    lib.nblic_amd_encode_batch_end.argtypes = [C.c_void_p, C.c_void_p]
    lib.nblic_amd_qencode_batch.restype = C.c_int
    lib.nblic_amd_qencode_batch.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_int, ip, ip,
                                            C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_long)]
    lib.nblic_amd_encode_batch_modes.restype = C.c_int
    lib.nblic_amd_encode_batch_modes.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_int, ip, ip, ip, ip,
                                                 C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_long), C.POINTER(C.c_void_p)]
    lib.nblic_amd_decode_batch.restype = C.c_int
    lib.nblic_amd_decode_batch.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p),
                                           C.POINTER(C.c_size_t), ip, ip, ip, ip, ip]
    lib.nblic_amd_serial_selftest.restype = C.c_int
    lib.nblic_amd_serial_selftest.argtypes = [C.c_void_p]
    lib.nblic_amd_cli_main.restype = C.c_int
    lib.nblic_amd_cli_main.argtypes = [C.c_int, C.POINTER(C.c_char_p)]
    lib.nblic_amd_cli_parse.restype = None
    lib.nblic_amd_cli_parse.argtypes = [C.c_int, C.POINTER(C.c_char_p), ip, C.c_char_p, C.c_char_p, C.c_size_t]
    lib.nblic_amd_read_gray.restype = C.c_int
    lib.nblic_amd_read_gray.argtypes = [C.c_char_p, _u8p, C.c_size_t, ip, ip]
    lib.nblic_amd_write_gray.restype = C.c_int
    lib.nblic_amd_write_gray.argtypes = [C.c_char_p, _u8p, C.c_int, C.c_int, C.c_int]
    lib.nblic_amd_set_device_coder.restype = C.c_int
    lib.nblic_amd_set_device_coder.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.nblic_amd_device_coder_stats.restype = None
    lib.nblic_amd_device_coder_stats.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_long), C.POINTER(C.c_long)]
    lib.nblic_amd_set_max_pixels.restype = None
    lib.nblic_amd_set_max_pixels.argtypes = [C.c_void_p, C.c_long]
    lib.nblic_amd_set_serial_rows.restype = None
    lib.nblic_amd_set_serial_rows.argtypes = [C.c_void_p, C.c_int]
    lib.nblic_amd_serial_launches.restype = C.c_long
    lib.nblic_amd_serial_launches.argtypes = [C.c_void_p]
    lib.nblic_amd_set_feed_chunk.restype = None
    lib.nblic_amd_set_feed_chunk.argtypes = [C.c_void_p, C.c_size_t]
    lib.nblic_amd_last_fed_bytes.restype = C.c_long
    lib.nblic_amd_last_fed_bytes.argtypes = [C.c_void_p]
    lib.nblic_amd_stream_begin.restype = C.c_void_p
    lib.nblic_amd_stream_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.nblic_amd_stream_resume.restype = C.c_void_p
    lib.nblic_amd_stream_resume.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    lib.nblic_amd_stream_run.restype = C.c_int
    lib.nblic_amd_stream_run.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.nblic_amd_stream_checkpoint.restype = C.c_size_t
    lib.nblic_amd_stream_checkpoint.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.nblic_amd_stream_progress.restype = C.c_int
    lib.nblic_amd_stream_progress.argtypes = [C.c_void_p, ip, C.POINTER(C.c_ulonglong), C.c_void_p, C.POINTER(C.c_double)]
    lib.nblic_amd_stream_recon.restype = C.c_int
    lib.nblic_amd_stream_recon.argtypes = [C.c_void_p, C.c_void_p, ip, ip]
    lib.nblic_amd_stream_end.restype = None
    lib.nblic_amd_stream_end.argtypes = [C.c_void_p]
    lib.nblic_amd_enable_timing.restype = None
    lib.nblic_amd_enable_timing.argtypes = [C.c_void_p, C.c_int]
    lib.nblic_amd_stage_times.restype = C.c_int
    lib.nblic_amd_stage_times.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_char_p), C.c_int]
    lib.nblic_amd_last_launches.restype = C.c_long
    lib.nblic_amd_last_launches.argtypes = [C.c_void_p]
    lib.nblic_amd_last_stats.restype = None
    lib.nblic_amd_last_stats.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.nblic_amd_debug_stage.restype = C.c_long
    lib.nblic_amd_debug_stage.argtypes = [C.c_void_p, _u8p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    lib.nblic_amd_range_code.restype = C.c_size_t
    lib.nblic_amd_range_code.argtypes = [C.POINTER(C.c_uint16), C.c_size_t, _u8p, C.c_size_t]
    lib.nblic_amd_range_code_multi.restype = C.c_int
    lib.nblic_amd_range_code_multi.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int, C.POINTER(C.c_void_p),
                                               C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    lib.nblic_amd_range_code_chunked.restype = C.c_int
    lib.nblic_amd_range_code_chunked.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int, C.POINTER(C.c_void_p),
                                                 C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_size_t]
    lib.nblic_amd_selftest.restype = C.c_int
    lib.nblic_amd_selftest.argtypes = [C.c_void_p]
    lib.nblic_amd_syn1.restype = None
    lib.nblic_amd_syn1.argtypes = [_u8p, C.c_int, C.c_int, C.c_uint32]
    lib.nblic_amd_version.restype = C.c_char_p
    _lib = lib
    return lib


def out_capacity(h: int, w: int) -> int:
    """Output provision per image: the reference CLI's 2 B/px (NBLIC_main.c:141) plus slack."""
    return 2 * h * w + 4096


# ---------------------------------------------------------------------------------------------
# drop-in operators (same meaning as the reference's NBLICcompress / NBLICdecompress)
# ---------------------------------------------------------------------------------------------
def compress(img: np.ndarray, near: int = 0, effort: int = 1) -> Tuple[Optional[bytes], np.ndarray, int, int]:
    """``NBLICcompress``: returns (stream or None on -1, reconstruction, clamped near, clamped effort).

    Like the reference the encoder overwrites its input plane with the reconstruction
    (NBLIC.c:876); here the caller's array is left alone and the overwritten copy is returned.
    """
    lib = load_library()
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    rec = img.copy()
    out = np.empty(out_capacity(h, w), np.uint8)
    n, e = C.c_int(near), C.c_int(effort)
    ln = lib.NBLICcompress(0, out.ctypes.data_as(_u8p), rec.ctypes.data_as(_u8p), h, w, C.byref(n), C.byref(e))
    if ln < 0:
        return None, rec, n.value, e.value
    return out[:ln].tobytes(), rec, n.value, e.value


def decompress(stream: bytes) -> Optional[Tuple[np.ndarray, int, int]]:
    """``NBLICdecompress``: returns (image, near, effort) or None on -1."""
    lib = load_library()
    if len(stream) < 16:
        return None
    buf = np.frombuffer(bytes(stream), np.uint8).copy()
    h = (int(buf[9]) << 8) | int(buf[10])
    w = (int(buf[11]) << 8) | int(buf[12])
    img = np.zeros((max(h, 1), max(w, 1)), np.uint8)
    hh, ww, n, e = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    rc = lib.NBLICdecompress(0, buf.ctypes.data_as(_u8p), img.ctypes.data_as(_u8p), C.byref(hh), C.byref(ww), C.byref(n), C.byref(e))
    if rc != 0:
        return None
    return img[:hh.value, :ww.value], n.value, e.value


def set_default_feed_chunk(n: int) -> None:
    """Step in which the drop-in decoders fetch a stream of unknown length (``nblic_amd_set_feed_chunk(NULL, n)``)."""
    load_library().nblic_amd_set_feed_chunk(None, n)


def set_default_serial_rows(rows: int) -> None:
    """Rows per launch of the resumable serial kernels behind the drop-in operators (0 = automatic)."""
    load_library().nblic_amd_set_serial_rows(None, rows)


def default_serial_launches() -> int:
    return int(load_library().nblic_amd_serial_launches(None))


def last_fed_bytes() -> int:
    """Bytes the last drop-in decode read from the caller's stream (``nblic_amd_last_fed_bytes(NULL)``)."""
    return int(load_library().nblic_amd_last_fed_bytes(None))


def set_default_max_pixels(n: int) -> None:
    """Opt-in pixel limit of the context behind the drop-in operators (``nblic_amd_set_max_pixels(NULL, n)``)."""
    load_library().nblic_amd_set_max_pixels(None, n)


CLI_PATH = os.path.join(_HERE, "nblic_codec_amd")


def cli(args: Sequence[str]) -> int:
    """Run the reference tool's command line in-process (``nblic_amd_cli_main``); args exclude argv[0]."""
    lib = load_library()
    argv = [b"nblic_codec_amd"] + [a.encode() for a in args]
    arr = (C.c_char_p * len(argv))(*argv)
    return int(lib.nblic_amd_cli_main(len(argv), arr))


def cli_parse(args: Sequence[str]) -> dict:
    """The switch grammar alone (``nblic_amd_cli_parse``)."""
    lib = load_library()
    argv = [b"nblic_codec_amd"] + [a.encode() for a in args]
    arr = (C.c_char_p * len(argv))(*argv)
    f = (C.c_int * 8)()
    src, dst = C.create_string_buffer(4096), C.create_string_buffer(4096)
    lib.nblic_amd_cli_parse(len(argv), arr, f, src, dst, 4096)
    return {"decompress": f[0], "near": f[1], "effort": f[2], "verbose": f[3], "multithread": f[4], "large": f[5], "device": f[6],
            "src": src.value.decode() if f[7] & 1 else None, "dst": dst.value.decode() if f[7] & 2 else None}


def read_gray(path: str) -> Optional[Tuple[np.ndarray, str]]:
    """An 8-bit gray PGM or BMP as the reference's front end reads it.  Returns (image, "PGM" | "BMP") or None."""
    lib = load_library()
    h, w = C.c_int(), C.c_int()
    probe = np.empty(1, np.uint8)
    kind = lib.nblic_amd_read_gray(path.encode(), probe.ctypes.data_as(_u8p), 0, C.byref(h), C.byref(w))
    if kind == 0:
        return None
    img = np.empty((h.value, w.value), np.uint8)
    kind = lib.nblic_amd_read_gray(path.encode(), img.ctypes.data_as(_u8p), img.size, C.byref(h), C.byref(w))
    return (img, "PGM" if kind == 1 else "BMP") if kind > 0 else None


def write_gray(path: str, img: np.ndarray, as_bmp: bool) -> bool:
    img = np.ascontiguousarray(img, np.uint8)
    return load_library().nblic_amd_write_gray(path.encode(), img.ctypes.data_as(_u8p), img.shape[0], img.shape[1], int(as_bmp)) == 0


def syn1(h: int, w: int, seed: int = 1) -> np.ndarray:
    """SYN-1 synthetic frame (``nblic_amd_syn1``)."""
    img = np.empty((h, w), np.uint8)
    load_library().nblic_amd_syn1(img.ctypes.data_as(_u8p), h, w, seed)
    return img


def qcompress(img: np.ndarray) -> Optional[bytes]:
    """``QNBLICcompress`` (effort 0, lossless): the stream as bytes, or None on -1."""
    lib = load_library()
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.empty(out_capacity(h, w) // 2, np.uint16)
    words = lib.QNBLICcompress(out.ctypes.data_as(C.POINTER(C.c_uint16)), img.ctypes.data_as(_u8p), h, w)
    return None if words < 0 else out[:words].tobytes()


def qdecompress(stream: bytes) -> Optional[np.ndarray]:
    """``QNBLICdecompress``: the image, or None on -1."""
    lib = load_library()
    if len(stream) < 8:
        return None
    buf = np.frombuffer(bytes(stream) + b"\0" * 8, np.uint8).copy().view(np.uint16)
    h, w = int(buf[2]), int(buf[3])
    img = np.zeros((max(h, 1), max(w, 1)), np.uint8)
    hh, ww = C.c_int(), C.c_int()
    rc = lib.QNBLICdecompress(buf.ctypes.data_as(C.POINTER(C.c_uint16)), img.ctypes.data_as(_u8p), C.byref(hh), C.byref(ww))
    return None if rc != 0 else img[: hh.value, : ww.value]


def range_code(coded: np.ndarray, cap: Optional[int] = None) -> Optional[bytes]:
    """Host range-coder stage alone (``nblic_amd_range_code``); needs no GPU."""
    lib = load_library()
    coded = np.ascontiguousarray(coded, np.uint16)
    cap = coded.size * 4 + 16 if cap is None else cap
    out = np.empty(max(cap, 1), np.uint8)
    n = lib.nblic_amd_range_code(coded.ctypes.data_as(C.POINTER(C.c_uint16)), coded.size, out.ctypes.data_as(_u8p), cap)
    if n == C.c_size_t(-1).value:
        return None
    return out[:n].tobytes()


def range_code_multi(streams: Sequence[np.ndarray], caps: Optional[Sequence[int]] = None):
    """Several bin streams through ``nblic_amd_range_code_multi``.  Returns (list of bytes or None, used_simd)."""
    lib = load_library()
    arrs = [np.ascontiguousarray(a, np.uint16) for a in streams]
    k = len(arrs)
    caps = [a.size * 4 + 16 for a in arrs] if caps is None else list(caps)
    outs = [np.empty(max(c, 1), np.uint8) for c in caps]
    cp = (C.c_void_p * k)(*[C.c_void_p(a.ctypes.data) for a in arrs])
    nn = (C.c_size_t * k)(*[a.size for a in arrs])
    op = (C.c_void_p * k)(*[C.c_void_p(o.ctypes.data) for o in outs])
    cc = (C.c_size_t * k)(*caps)
    ln = (C.c_size_t * k)()
    simd = lib.nblic_amd_range_code_multi(cp, nn, k, op, cc, ln)
    bad = C.c_size_t(-1).value
    return [None if ln[i] == bad else outs[i][: ln[i]].tobytes() for i in range(k)], simd


def range_code_chunked(streams: Sequence[np.ndarray], chunk: int, caps: Optional[Sequence[int]] = None):
    """Up to 16 bin streams fed ``chunk`` bins at a time through the resumable coders
    (``nblic_amd_range_code_chunked``) -- the coder threads' path, minus the GPU."""
    lib = load_library()
    arrs = [np.ascontiguousarray(a, np.uint16) for a in streams]
    k = len(arrs)
    caps = [a.size * 4 + 16 for a in arrs] if caps is None else list(caps)
    outs = [np.empty(max(c, 1), np.uint8) for c in caps]
    cp = (C.c_void_p * k)(*[C.c_void_p(a.ctypes.data) for a in arrs])
    nn = (C.c_size_t * k)(*[a.size for a in arrs])
    op = (C.c_void_p * k)(*[C.c_void_p(o.ctypes.data) for o in outs])
    cc = (C.c_size_t * k)(*caps)
    ln = (C.c_size_t * k)()
    if lib.nblic_amd_range_code_chunked(cp, nn, k, op, cc, ln, C.c_size_t(chunk)) != 0:
        raise ValueError("nblic_amd_range_code_chunked: 1..16 streams, chunk > 0")
    bad = C.c_size_t(-1).value
    return [None if ln[i] == bad else outs[i][: ln[i]].tobytes() for i in range(k)]


# ---------------------------------------------------------------------------------------------
# batch context
# ---------------------------------------------------------------------------------------------
class Context:
    """Several images in flight on one GPU (``nblic_amd_create``)."""

    def __init__(self, device: int = 0, n_slots: int = 4, n_coders: int = 4, n_groups: int = 0, n_host_buffers: int = 0):
        self.lib = load_library()
        if n_groups > 0:
            self.handle = self.lib.nblic_amd_create_ex(device, n_groups, (n_slots + n_groups - 1) // n_groups, n_coders,
                                                       n_host_buffers or 2 * n_slots)
        else:
            self.handle = self.lib.nblic_amd_create(device, n_slots, n_coders)
        if not self.handle:
            raise RuntimeError("nblic_amd_create failed: no usable HIP device (the hot path has no CPU fallback)")
        self.device, self.n_slots, self.n_coders = device, n_slots, n_coders

    def close(self):
        if self.handle:
            self.lib.nblic_amd_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def selftest(self) -> int:
        return self.lib.nblic_amd_selftest(self.handle)

    def set_device_coder(self, n_packs: int, min_outstanding: int = 0) -> int:
        """Range-coder stage on the GPU for part of the backlog (``nblic_amd_set_device_coder``)."""
        return int(self.lib.nblic_amd_set_device_coder(self.handle, n_packs, min_outstanding))

    def device_coder_stats(self) -> dict:
        b, p, i = C.c_double(), C.c_long(), C.c_long()
        self.lib.nblic_amd_device_coder_stats(self.handle, C.byref(b), C.byref(p), C.byref(i))
        return {"bins": b.value, "packs": p.value, "images": i.value}

    def serial_selftest(self) -> int:
        return self.lib.nblic_amd_serial_selftest(self.handle)

    def encode_modes(self, imgs: Sequence[np.ndarray], nears: Sequence[int], efforts: Sequence[int], want_recon: bool = True):
        """Any-mode batch encode (``nblic_amd_encode_batch_modes``): per image (near, effort).  Returns
        (streams, reconstructions or None)."""
        planes = [np.ascontiguousarray(i, np.uint8) for i in imgs]
        k = len(planes)
        outs = [np.empty(out_capacity(*p.shape), np.uint8) for p in planes]
        recs = [np.empty_like(p) for p in planes] if want_recon else None
        ip_ = (C.c_void_p * k)(*[C.c_void_p(p.ctypes.data) for p in planes])
        hs = (C.c_int * k)(*[p.shape[0] for p in planes])
        ws = (C.c_int * k)(*[p.shape[1] for p in planes])
        nn = (C.c_int * k)(*[int(v) for v in nears])
        ee = (C.c_int * k)(*[int(v) for v in efforts])
        op = (C.c_void_p * k)(*[C.c_void_p(o.ctypes.data) for o in outs])
        caps = (C.c_size_t * k)(*[o.size for o in outs])
        lens = (C.c_long * k)()
        rp = (C.c_void_p * k)(*[C.c_void_p(r.ctypes.data) for r in recs]) if want_recon else None
        if self.lib.nblic_amd_encode_batch_modes(self.handle, k, ip_, 0, hs, ws, nn, ee, op, caps, lens, rp) != 0:
            raise RuntimeError(f"nblic_amd_encode_batch_modes failed (lengths {list(lens)[:8]}...)")
        return [o[: lens[i]].tobytes() for i, o in enumerate(outs)], recs

    def decode_batch(self, streams: Sequence[bytes]):
        """Batch decode of NBLIC / QNBLIC streams (``nblic_amd_decode_batch``).  Returns a list of
        (image, near, effort) or None per stream."""
        k = len(streams)
        bufs = [np.frombuffer(bytes(s), np.uint8).copy() for s in streams]
        dims = []
        for b in bufs:
            if b.size >= 16 and bytes(b[:8]) == b"NBLIC0.3":
                dims.append(((int(b[9]) << 8) | int(b[10]), (int(b[11]) << 8) | int(b[12])))
            elif b.size >= 8 and bytes(b[:4]) == b"Q0.2":
                dims.append((int(b[4]) | (int(b[5]) << 8), int(b[6]) | (int(b[7]) << 8)))
            else:
                dims.append((1, 1))
        imgs = [np.zeros((max(h, 1), max(w, 1)), np.uint8) for (h, w) in dims]
        sp = (C.c_void_p * k)(*[C.c_void_p(b.ctypes.data) for b in bufs])
        sl = (C.c_size_t * k)(*[b.size for b in bufs])
        op = (C.c_void_p * k)(*[C.c_void_p(i.ctypes.data) for i in imgs])
        caps = (C.c_size_t * k)(*[i.size for i in imgs])
        hs, ws, nn, ee, st = ((C.c_int * k)() for _ in range(5))
        self.lib.nblic_amd_decode_batch(self.handle, k, sp, sl, op, caps, hs, ws, nn, ee, st)
        return [None if st[i] != 0 else (imgs[i][: hs[i], : ws[i]], nn[i], ee[i]) for i in range(k)]

    def enable_timing(self, on=True):
        """0 / False off, 1 / True every stage, 2 only the stages the bench line reports against a roof."""
        self.lib.nblic_amd_enable_timing(self.handle, int(on))

    def set_max_pixels(self, n: int):
        self.lib.nblic_amd_set_max_pixels(self.handle, n)

    def stream(self, img: np.ndarray, near: int, effort: int, band_rows: int = 0, checkpoint: Optional[bytes] = None) -> "BandStream":
        """One image in row bands (``nblic_amd_stream_*``): bounded workspace, suspend / resume through checkpoints."""
        return BandStream(self, img, near, effort, band_rows, checkpoint)

    def set_serial_rows(self, rows: int):
        """Rows per launch of the resumable serial kernels (``nblic_amd_set_serial_rows``); 0 = automatic."""
        self.lib.nblic_amd_set_serial_rows(self.handle, rows)

    def serial_launches(self) -> int:
        return int(self.lib.nblic_amd_serial_launches(self.handle))

    def encode_ptrs(self, ptrs: Sequence[int], shapes: Sequence[Tuple[int, int]], on_device: bool,
                    outs: Optional[List[np.ndarray]] = None) -> Tuple[List[np.ndarray], np.ndarray]:
        """Encode planes given as raw addresses (host or device).  Returns (out buffers, lengths)."""
        k = len(ptrs)
        if outs is None:
            outs = [np.empty(out_capacity(h, w), np.uint8) for (h, w) in shapes]
        imgs = (C.c_void_p * k)(*[C.c_void_p(int(p)) for p in ptrs])
        hs = (C.c_int * k)(*[int(s[0]) for s in shapes])
        ws = (C.c_int * k)(*[int(s[1]) for s in shapes])
        op = (C.c_void_p * k)(*[C.c_void_p(o.ctypes.data) for o in outs])
        caps = (C.c_size_t * k)(*[o.size for o in outs])
        lens = (C.c_long * k)()
        rc = self.lib.nblic_amd_encode_batch(self.handle, k, imgs, int(on_device), hs, ws, op, caps, lens)
        arr = np.array(list(lens), dtype=np.int64)
        if rc != 0:
            raise RuntimeError(f"nblic_amd_encode_batch failed (lengths {arr.tolist()})")
        return outs, arr

    def encode_begin(self, ptrs: Sequence[int], shapes: Sequence[Tuple[int, int]], on_device: bool,
                     outs: Optional[List[np.ndarray]] = None):
        """Submit a batch (``nblic_amd_encode_batch_begin``); returns a ticket for :meth:`encode_end`.
        Several batches may be in flight; the ticket keeps every argument array alive."""
        k = len(ptrs)
        if outs is None:
            outs = [np.empty(out_capacity(h, w), np.uint8) for (h, w) in shapes]
        imgs = (C.c_void_p * k)(*[C.c_void_p(int(p)) for p in ptrs])
        hs = (C.c_int * k)(*[int(s[0]) for s in shapes])
        ws = (C.c_int * k)(*[int(s[1]) for s in shapes])
        op = (C.c_void_p * k)(*[C.c_void_p(o.ctypes.data) for o in outs])
        caps = (C.c_size_t * k)(*[o.size for o in outs])
        lens = (C.c_long * k)()
        handle = self.lib.nblic_amd_encode_batch_begin(self.handle, k, imgs, int(on_device), hs, ws, op, caps, lens)
        if not handle:
            raise RuntimeError("nblic_amd_encode_batch_begin failed")
        return {"handle": handle, "keep": (imgs, hs, ws, op, caps), "lens": lens, "outs": outs, "n": k}

    def encode_end(self, ticket) -> Tuple[List[np.ndarray], np.ndarray]:
        """Wait for a batch submitted with :meth:`encode_begin`.  Returns (out buffers, lengths)."""
        rc = self.lib.nblic_amd_encode_batch_end(self.handle, ticket["handle"])
        lens = np.array(ticket["lens"][:], np.int64)
        if rc != 0:
            raise RuntimeError(f"nblic_amd_encode_batch_end failed (lengths {list(lens[:8])}...)")
        return ticket["outs"], lens

    def encode_batch(self, imgs: Sequence[np.ndarray]) -> List[bytes]:
        """-n0 -e1 encode of host planes; returns the .nblic streams."""
        planes = [np.ascontiguousarray(i, np.uint8) for i in imgs]
        outs, lens = self.encode_ptrs([p.ctypes.data for p in planes], [p.shape for p in planes], False)
        return [o[:int(n)].tobytes() for o, n in zip(outs, lens)]

    def qencode_ptrs(self, ptrs: Sequence[int], shapes: Sequence[Tuple[int, int]], on_device: bool,
                     outs: Optional[List[np.ndarray]] = None) -> Tuple[List[np.ndarray], np.ndarray]:
        """Effort-0 (QNBLIC) encode of planes given as raw addresses (host or device).  Returns
        (uint16 out buffers, lengths in 16-bit words)."""
        k = len(ptrs)
        if outs is None:
            outs = [np.empty(out_capacity(h, w) // 2, np.uint16) for (h, w) in shapes]
        ip_ = (C.c_void_p * k)(*[C.c_void_p(int(p)) for p in ptrs])
        hs = (C.c_int * k)(*[int(s[0]) for s in shapes])
        ws = (C.c_int * k)(*[int(s[1]) for s in shapes])
        op = (C.c_void_p * k)(*[C.c_void_p(o.ctypes.data) for o in outs])
        caps = (C.c_size_t * k)(*[o.size for o in outs])
        lens = (C.c_long * k)()
        if self.lib.nblic_amd_qencode_batch(self.handle, k, ip_, int(on_device), hs, ws, op, caps, lens) != 0:
            raise RuntimeError(f"nblic_amd_qencode_batch failed (lengths {list(lens)})")
        return outs, np.array(lens[:], np.int64)

    def qencode_batch(self, imgs: Sequence[np.ndarray]) -> List[bytes]:
        """Effort-0 (QNBLIC) encode of host planes; returns the streams as bytes (little-endian words)."""
        planes = [np.ascontiguousarray(i, np.uint8) for i in imgs]
        k = len(planes)
        outs = [np.empty(out_capacity(*p.shape) // 2, np.uint16) for p in planes]
        ip_ = (C.c_void_p * k)(*[C.c_void_p(p.ctypes.data) for p in planes])
        hs = (C.c_int * k)(*[p.shape[0] for p in planes])
        ws = (C.c_int * k)(*[p.shape[1] for p in planes])
        op = (C.c_void_p * k)(*[C.c_void_p(o.ctypes.data) for o in outs])
        caps = (C.c_size_t * k)(*[o.size for o in outs])
        lens = (C.c_long * k)()
        if self.lib.nblic_amd_qencode_batch(self.handle, k, ip_, 0, hs, ws, op, caps, lens) != 0:
            raise RuntimeError(f"nblic_amd_qencode_batch failed (lengths {list(lens)})")
        return [o[: lens[i]].tobytes() for i, o in enumerate(outs)]

    def stage_times(self) -> dict:
        ms = (C.c_double * 64)()
        names = (C.c_char_p * 64)()
        n = self.lib.nblic_amd_stage_times(self.handle, ms, names, 64)
        return {names[i].decode(): ms[i] for i in range(n)}

    def last_launches(self) -> int:
        return int(self.lib.nblic_amd_last_launches(self.handle))

    def last_stats(self) -> Tuple[float, float]:
        b, s = C.c_double(), C.c_double()
        self.lib.nblic_amd_last_stats(self.handle, C.byref(b), C.byref(s))
        return b.value, s.value

    _STAGE = {"rec1": (0, np.uint32), "pxs": (1, np.uint16), "z": (2, np.uint8), "cnt": (3, np.uint8),
              "events": (4, np.uint32), "coded": (5, np.uint16), "dbg": (6, np.uint64), "totals": (7, np.uint32)}

    def debug_stage(self, img: np.ndarray, name: str) -> np.ndarray:
        """Intermediate array of the staged -e1 pipeline for one image (kernel parity tests)."""
        which, dt = self._STAGE[name]
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        out = np.empty(40 * h * w + 64, dt)
        cnt = self.lib.nblic_amd_debug_stage(self.handle, img.ctypes.data_as(_u8p), h, w, which,
                                             C.c_void_p(out.ctypes.data), out.nbytes)
        if cnt < 0:
            raise RuntimeError("nblic_amd_debug_stage failed")
        return out[:cnt].copy()


class BandStream:
    """An encode in progress (``nblic_amd_stream``).  ``run(budget_seconds)`` returns (finished, bytes of this call);
    ``checkpoint()`` the state to hand to ``Context.stream(..., checkpoint=...)`` in another call or process."""

    def __init__(self, ctx: Context, img: np.ndarray, near: int, effort: int, band_rows: int = 0, checkpoint: Optional[bytes] = None):
        self.ctx, self.lib = ctx, ctx.lib
        self.img = np.ascontiguousarray(img, np.uint8)
        h, w = self.img.shape
        if checkpoint is None:
            self.handle = self.lib.nblic_amd_stream_begin(ctx.handle, C.c_void_p(self.img.ctypes.data), 0, h, w, near, effort, band_rows)
        else:
            self._ck = np.frombuffer(bytes(checkpoint), np.uint8).copy()
            self.handle = self.lib.nblic_amd_stream_resume(ctx.handle, C.c_void_p(self.img.ctypes.data), 0, C.c_void_p(self._ck.ctypes.data), self._ck.size)
        if not self.handle:
            raise RuntimeError("nblic_amd_stream_begin / _resume failed")
        self.out = np.empty(h * w + h * w // 8 + 65536, np.uint8)

    def run(self, budget_seconds: float = 0.0) -> Tuple[bool, bytes]:
        n = C.c_size_t(0)
        rc = self.lib.nblic_amd_stream_run(self.handle, float(budget_seconds), C.c_void_p(self.out.ctypes.data), self.out.size, C.byref(n))
        if rc < 0:
            raise RuntimeError("nblic_amd_stream_run failed")
        return rc == 1, self.out[: n.value].tobytes()

    def progress(self) -> dict:
        rows, total, ms = C.c_int(), C.c_ulonglong(), C.c_double()
        digest = (C.c_ubyte * 32)()
        state = self.lib.nblic_amd_stream_progress(self.handle, C.byref(rows), C.byref(total), digest, C.byref(ms))
        return {"state": state, "rows_done": rows.value, "bytes_total": total.value, "sha256": bytes(digest).hex(), "model_kernel_ms": ms.value}

    def checkpoint(self) -> bytes:
        need = self.lib.nblic_amd_stream_checkpoint(self.handle, None, 0)
        buf = np.empty(need, np.uint8)
        if self.lib.nblic_amd_stream_checkpoint(self.handle, C.c_void_p(buf.ctypes.data), need) != need:
            raise RuntimeError("nblic_amd_stream_checkpoint failed")
        return buf.tobytes()

    def recon(self, plane: Optional[np.ndarray] = None) -> Tuple[np.ndarray, int, int]:
        """Writes the rows this object has coded so far into `plane` (a whole h x w array, allocated if None);
        returns (plane, first_row, end_row).  After a resume the earlier rows came out of the earlier objects."""
        rec = np.zeros_like(self.img) if plane is None else plane
        a, b = C.c_int(), C.c_int()
        if self.lib.nblic_amd_stream_recon(self.handle, C.c_void_p(rec.ctypes.data), C.byref(a), C.byref(b)) != 0:
            raise RuntimeError("nblic_amd_stream_recon failed")
        return rec, a.value, b.value

    def close(self):
        if self.handle:
            self.lib.nblic_amd_stream_end(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
