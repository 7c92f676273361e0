"""Batch plateau of the raster-serial modes: N images of SIZE x SIZE side by side (one wave each), N swept, per
mode; model-stage occupancy figures beside it.  Prints JSON lines.

    python tools/serial_batch_sweep.py [--size 512] [--batches 256,512,1024,2048] [--modes 0:2,0:3,2:1]
"""
import argparse, importlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--batches", default="256,512,1024,2048")
ap.add_argument("--modes", default="2:1,0:2,0:3")
ap.add_argument("--groups", type=int, default=2)
ap.add_argument("--decode", action="store_true")
args = ap.parse_args()
pkg = importlib.import_module("nblic-image-compression_amd")
from oracle.oracle import syn1
H = W = args.size
batches = [int(b) for b in args.batches.split(",")]
top = max(batches)
# registers per lane of the model kernels (hipcc -S metadata of csrc/serial_engine.hip at this commit) -> waves per SIMD (512 / VGPRs)
VGPRS = {1: 123, 2: 160, 3: 190}
LDS_STATIC = 11 * 1024 + 3 * ((W + 15) & ~15)
ctx = pkg.Context(device=0, n_slots=top, n_coders=16, n_groups=args.groups, n_host_buffers=top + 64)
base = [syn1(H, W, k + 1) for k in range(min(top, 64))]
for mode in args.modes.split(","):
    near, effort = map(int, mode.split(":"))
    for n in batches:
        imgs = [base[k % len(base)] for k in range(n)]
        ctx.encode_modes(imgs, [near] * n, [effort] * n, want_recon=False)      # untimed: buffers of this effort are allocated
        t0 = time.perf_counter(); s, _ = ctx.encode_modes(imgs, [near] * n, [effort] * n, want_recon=False); dt = time.perf_counter() - t0
        per_group = (n + args.groups - 1) // args.groups
        waves_per_simd = 512 // VGPRS[effort]
        line = {"mode": f"-n{near} -e{effort}", "size": f"{H}x{W}", "images": n, "encode_Mpx_s": round(n * H * W / dt / 1e6, 1), "seconds": round(dt, 3),
                "waves_per_launch": per_group, "waves_per_CU_offered": round(per_group / 256, 2),
                "occupancy_limits": {"vgprs": VGPRS[effort], "waves_per_SIMD_by_vgprs": waves_per_simd, "waves_per_CU_by_vgprs": 4 * waves_per_simd,
                                     "lds_bytes_per_wave": LDS_STATIC, "waves_per_CU_by_lds": (160 * 1024) // LDS_STATIC}}
        if args.decode:
            t0 = time.perf_counter(); d = ctx.decode_batch(s); dd = time.perf_counter() - t0
            line["decode_Mpx_s"] = round(n * H * W / dd / 1e6, 1); line["decode_ok"] = all(x is not None for x in d)
        print(json.dumps(line), flush=True)
if args.decode:                                          # effort 0 (QNBLIC): batch decode, the same sweep
    for n in batches:
        imgs = [base[k % len(base)] for k in range(n)]
        q = ctx.qencode_batch(imgs)
        ctx.decode_batch(q[: min(n, 64)])
        t0 = time.perf_counter(); d = ctx.decode_batch(q); dd = time.perf_counter() - t0
        print(json.dumps({"mode": "-e0 decode (QNBLIC)", "size": f"{H}x{W}", "images": n, "decode_Mpx_s": round(n * H * W / dd / 1e6, 1),
                          "decode_ok": all(x is not None and (x[0] == im).all() for x, im in zip(d, imgs))}), flush=True)
ctx.close()
