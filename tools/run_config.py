"""Runs one of BASELINE.json's raster-serial configurations on the HIP path and checks the stream against
the compiled reference's golden (tests/golden/manifest.json["serial"], produced in the build container by
tests/golden/make_golden_large.py).  Prints one JSON line.

The image is worked through in ROW BANDS (include/nblic_amd.h, nblic_amd_stream_*): bounded device workspace, no
kernel longer than a band, and the run can be SUSPENDED: with --budget SECONDS the encoder stops between two bands
once the budget is spent, writes a checkpoint to --checkpoint FILE and exits with code 3; started again with the
same arguments it resumes from that file.  A running SHA-256 of the stream travels in the checkpoint, so the hash of
the whole stream is known at the end although no run ever held all of it.

    python tools/run_config.py --config 4            # 8192x8192 SYN-1, -n2 -e2
    python tools/run_config.py --config 5 --budget 1000 --checkpoint gpurun_out/config5.ckpt    # 16384x16384 SYN-1, -n0 -e3
    python tools/run_config.py --shape 64x16384 --near 0 --effort 3 [--one-piece]
"""
import argparse, hashlib, importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=0)
ap.add_argument("--shape", default="")
ap.add_argument("--near", type=int, default=0)
ap.add_argument("--effort", type=int, default=1)
ap.add_argument("--band-rows", type=int, default=0)
ap.add_argument("--budget", type=float, default=0.0, help="seconds of encoding after which the run suspends itself (0: run to the end)")
ap.add_argument("--checkpoint", default="", help="file the suspended state is written to / resumed from")
ap.add_argument("--one-piece", action="store_true", help="the whole image in one nblic_amd_encode_batch_modes call instead of bands")
ap.add_argument("--decode", action="store_true", help="also decode the stream on the GPU and compare with the reconstruction (needs the whole stream: not after a resume)")
args = ap.parse_args()
if args.config == 4:
    h, w, near, effort = 8192, 8192, 2, 2
elif args.config == 5:
    h, w, near, effort = 16384, 16384, 0, 3
else:
    h, w = map(int, args.shape.split("x")); near, effort = args.near, args.effort
import threading

def heartbeat():                                        # a long single-image run prints nothing for minutes: keep a sign of life on stderr
    t0 = time.time()
    while True:
        time.sleep(45)
        print(f"[run_config] still running, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)

threading.Thread(target=heartbeat, daemon=True).start()
pkg = importlib.import_module("nblic-image-compression_amd")
img = pkg.syn1(h, w, 1)
key = f"syn1s1_{h}x{w}_n{near}_e{effort}"
gold = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest.json"))).get("serial", {}).get(key)
ctx = pkg.Context(device=0, n_slots=2, n_coders=2, n_groups=2, n_host_buffers=4)
if h * w > 100000000:
    ctx.set_max_pixels(1 << 33)                          # opt-in: the reference refuses > 1e8 pixels (NBLIC.h:31)
line = {"config": args.config or None, "workload": f"{h}x{w} SYN-1, -n{near} -e{effort}"}
stream_bytes = None
t0 = time.perf_counter()
if args.one_piece:
    streams, recs = ctx.encode_modes([img], [near], [effort])
    dt = time.perf_counter() - t0
    stream_bytes, rec = streams[0], recs[0]
    line.update({"mode": "one piece", "bytes": len(stream_bytes), "sha256": hashlib.sha256(stream_bytes).hexdigest()})
else:
    ck, history = None, []
    if args.checkpoint and os.path.exists(args.checkpoint):
        ck = open(args.checkpoint, "rb").read()
        if os.path.exists(args.checkpoint + ".json"):
            history = json.load(open(args.checkpoint + ".json"))
    st = ctx.stream(img, near, effort, band_rows=args.band_rows, checkpoint=ck)
    start = st.progress()
    done, piece = st.run(args.budget)
    dt = time.perf_counter() - t0
    prog = st.progress()
    history.append({"rows": [start["rows_done"], prog["rows_done"]], "seconds": round(dt, 2), "bytes": len(piece),
                    "piece_sha256": hashlib.sha256(piece).hexdigest(), "model_kernel_s": round(prog["model_kernel_ms"] / 1e3, 2)})
    line.update({"mode": "row bands", "band_rows": args.band_rows or "auto", "runs": history, "rows_done": prog["rows_done"],
                 "bytes": prog["bytes_total"], "sha256": prog["sha256"]})
    if not done:
        assert args.checkpoint, "--budget without --checkpoint"
        os.makedirs(os.path.dirname(os.path.abspath(args.checkpoint)), exist_ok=True)
        open(args.checkpoint, "wb").write(st.checkpoint())
        json.dump(history, open(args.checkpoint + ".json", "w"))
        line.update({"suspended": True, "checkpoint_bytes": os.path.getsize(args.checkpoint)})
        print(json.dumps(line), flush=True)
        st.close(); ctx.close()
        sys.exit(3)
    rec, r0, r1 = st.recon()
    if ck is None:
        stream_bytes = piece
    elif near > 0:
        line["recon_rows_checked"] = [r0, r1]                # a resumed run holds only the rows it coded itself
    else:
        rec = img                                            # lossless: the reconstruction is the input (NBLIC.c:876)
    st.close()
total_s = sum(r["seconds"] for r in line.get("runs", [])) or dt
line.update({"encode_seconds": round(total_s, 2), "encode_us_per_px": round(total_s / (h * w) * 1e6, 3), "encode_Mpixel_per_s": round(h * w / total_s / 1e6, 3),
             "recon_sha256": hashlib.sha256(rec.tobytes()).hexdigest()})
rows = slice(*line["recon_rows_checked"]) if "recon_rows_checked" in line else slice(0, h)
line["max_abs_error"] = int(abs(rec[rows].astype(np.int16) - img[rows].astype(np.int16)).max())
if gold:
    line["golden"] = {"bytes": gold["len"], "sha256": gold["sha256"], "reference_thread_seconds": gold.get("ref_seconds"), "limit_raised": gold.get("limit_raised")}
    line["bit_exact"] = (line["bytes"] == gold["len"] and line["sha256"] == gold["sha256"] and line["recon_sha256"] == gold["recon_sha256"])
if args.decode and stream_bytes is not None:
    t0 = time.perf_counter()
    d = ctx.decode_batch([stream_bytes])[0]
    line["decode_seconds"] = round(time.perf_counter() - t0, 2)
    line["decode_ok"] = d is not None and bool((d[0] == rec).all())
print(json.dumps(line), flush=True)
ctx.close()
