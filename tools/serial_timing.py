"""Serial-mode timing on the GPU box: per-image microseconds per pixel of the model stage + entropy
stages (encode) and of the fused decoder, one image alone and many images side by side, next to the
compiled reference's single thread when oracle/_ref travelled.  Prints JSON lines.

    python tools/serial_timing.py [--size 512] [--batch 64]
"""
import argparse, importlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--modes", default="0:1,2:1,0:2,2:2,0:3")
ap.add_argument("--groups", type=int, default=2, help="groups the images in flight are split into (each launches its own kernels on its own stream)")
args = ap.parse_args()
pkg = importlib.import_module("nblic-image-compression_amd")
from oracle.oracle import Reference, syn1
ref = Reference() if Reference.available() else None
H = W = args.size
ctx = pkg.Context(device=0, n_slots=max(2, args.batch), n_coders=8, n_groups=args.groups, n_host_buffers=2 * args.batch + 16)
for mode in args.modes.split(","):
    near, effort = map(int, mode.split(":"))
    img = syn1(H, W, 1)
    ctx.encode_modes([img[:64, :64]], [near], [effort])                       # warm-up (allocations, code load)
    t0 = time.perf_counter(); s1, _ = ctx.encode_modes([img], [near], [effort]); t1 = time.perf_counter() - t0
    imgs = [syn1(H, W, k + 1) for k in range(args.batch)]
    ctx.encode_modes(imgs, [near] * args.batch, [effort] * args.batch)        # the batch once untimed: every slot's statistics buffers are (re)allocated for this effort -- hipFree synchronises the device, which halved the timed rate
    t0 = time.perf_counter(); sb, _ = ctx.encode_modes(imgs, [near] * args.batch, [effort] * args.batch); tb = time.perf_counter() - t0
    t0 = time.perf_counter(); d1 = ctx.decode_batch(s1); td1 = time.perf_counter() - t0
    t0 = time.perf_counter(); db = ctx.decode_batch(sb); tdb = time.perf_counter() - t0
    line = {"mode": f"-n{near} -e{effort}", "size": f"{H}x{W}", "encode_one_us_per_px": round(t1 / (H * W) * 1e6, 3),
            "encode_batch": args.batch, "encode_batch_Mpx_s": round(args.batch * H * W / tb / 1e6, 2),
            "decode_one_us_per_px": round(td1 / (H * W) * 1e6, 3), "decode_batch_Mpx_s": round(args.batch * H * W / tdb / 1e6, 2),
            "decode_ok": all(d is not None for d in db)}
    if ref is not None:
        t0 = time.perf_counter(); rs = ref.encode(img, near, effort)[0]; tr = time.perf_counter() - t0
        line["reference_thread_us_per_px"] = round(tr / (H * W) * 1e6, 3)
        line["bit_exact"] = (rs == s1[0])
    print(json.dumps(line), flush=True)
# effort 0 (QNBLIC): batch decode
imgs = [syn1(H, W, k + 1) for k in range(args.batch)]
q = ctx.qencode_batch(imgs)
t0 = time.perf_counter(); d1 = ctx.decode_batch(q[:1]); td1 = time.perf_counter() - t0
t0 = time.perf_counter(); db = ctx.decode_batch(q); tdb = time.perf_counter() - t0
print(json.dumps({"mode": "-e0 decode (QNBLIC)", "size": f"{H}x{W}", "decode_one_us_per_px": round(td1 / (H * W) * 1e6, 3),
                  "decode_batch": args.batch, "decode_batch_Mpx_s": round(args.batch * H * W / tdb / 1e6, 2),
                  "decode_ok": all(d is not None and (d[0] == im).all() for d, im in zip(db, imgs))}), flush=True)
ctx.close()
