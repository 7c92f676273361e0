#!/usr/bin/env python3
"""Side measurements quoted in DESIGN.md (not the headline bench): effort 0 throughput, single
image latency, PCIe-inclusive rate, and the raster-serial engine against the CPU oracle."""
import importlib, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # repo root
import torch
torch.cuda.init()
pkg = importlib.import_module("nblic-image-compression_amd")
from oracle.oracle import Oracle, Reference
o = Oracle()
ref = Reference() if Reference.available() else None
out = {}

def timed(f, reps=1):
    best = 1e9
    for _ in range(reps):
        t = time.perf_counter(); r = f(); best = min(best, time.perf_counter() - t)
    return r, best

H = W = 4096
frames = [pkg.syn1(H, W, k + 1) for k in range(32)]
ctx = pkg.Context(device=0, n_slots=32, n_coders=16, n_groups=4, n_host_buffers=64)
ctx.encode_batch(frames[:8]); ctx.qencode_batch(frames[:8])                  # warm-up / allocation
_, t = timed(lambda: ctx.encode_batch(frames), 2)
out["e1_host_inputs_32x4096_Mpx_s"] = round(32 * H * W / t / 1e6, 1)
_, t = timed(lambda: ctx.qencode_batch(frames), 2)
out["e0_host_inputs_32x4096_Mpx_s"] = round(32 * H * W / t / 1e6, 1)
_, t = timed(lambda: ctx.encode_batch(frames[:1]), 3)
out["e1_single_4096_latency_ms"] = round(t * 1e3, 1)
_, t = timed(lambda: ctx.qencode_batch(frames[:1]), 3)
out["e0_single_4096_latency_ms"] = round(t * 1e3, 1)
if ref:
    _, t = timed(lambda: ref.qencode(frames[0]))
    out["e0_reference_cpu_Mpx_s"] = round(H * W / t / 1e6, 2)
    _, t = timed(lambda: ref.encode(frames[0], 0, 1))
    out["e1_reference_cpu_Mpx_s"] = round(H * W / t / 1e6, 2)
ctx.close()
# effort 0 with the frames resident in HBM, a batch large enough to fill the pipeline
ctx = pkg.Context(device=0, n_slots=48, n_coders=16, n_groups=6, n_host_buffers=128)
dev = [torch.from_numpy(f).to("cuda:0") for f in frames] * 8                     # 256 planes (32 distinct)
torch.cuda.synchronize()
outs = [np.empty((H * W + 8192) // 2, np.uint16) for _ in dev]
shapes = [(H, W)] * len(dev)
ctx.qencode_ptrs([d.data_ptr() for d in dev], shapes, True, outs)
(_, lens_q), t = timed(lambda: ctx.qencode_ptrs([d.data_ptr() for d in dev], shapes, True, outs), 2)
out["e0_device_inputs_256x4096_Mpx_s"] = round(len(dev) * H * W / t / 1e6, 1)
assert outs[0][: int(lens_q[0])].tobytes() == pkg.qcompress(frames[0])
ctx.close()
# raster-serial engine (one lane) vs the CPU oracle on small frames
for name, (h, w, near, effort) in {"e1_n2_256": (256, 256, 2, 1), "e2_n0_96": (96, 96, 0, 2), "e3_n0_64": (64, 64, 0, 3)}.items():
    img = pkg.syn1(h, w, 1)
    (s, rec, _, _), tg = timed(lambda: pkg.compress(img, near, effort))
    (so, *_), tc = timed(lambda: o.encode(img, near, effort))
    assert s == so
    d, td = timed(lambda: pkg.decompress(s))
    out["serial_" + name] = {"gpu_enc_ms": round(tg * 1e3, 1), "gpu_dec_ms": round(td * 1e3, 1), "cpu_oracle_enc_ms": round(tc * 1e3, 2)}
img = pkg.syn1(512, 512, 1)
q = pkg.qcompress(img)
d, td = timed(lambda: pkg.qdecompress(q))
out["serial_e0_decode_512_ms"] = round(td * 1e3, 1)
print(json.dumps(out))
