import importlib, sys, os, numpy as np
sys.path.insert(0, '/root/repo')
pkg = importlib.import_module("nblic-image-compression_amd")
img = pkg.syn1(4096, 4096, 1)
ctx = pkg.Context(device=0, n_slots=1, n_coders=1, n_groups=1)
d = ctx.debug_stage(img, "dbg")
d = ctx.debug_stage(img, "dbg")
print("mapper blocks (stage, walk, flush cycles; rounds):")
for b in range(8):
    print(b, d[64 + b*4: 64 + b*4 + 4])
print("counter chains > 100k touches: cycles, touches, windows, key")
rows = d[256:256+1024].reshape(256, 4)
for r in rows[rows[:,1] > 0][:10]:
    print("cycles %d touches %d windows %d | cycles/window %.0f | halvings %d (%.2f/window)" % (
        r[0], r[1], r[2], r[0] / max(1, r[2]), r[3], r[3] / max(1, r[2])))

print("touch scatter waves (parity, segment/64): cycles total, setup, match, place, close; events")
for k in range(8):
    t = d[2048 + k * 8: 2048 + k * 8 + 6].astype(np.int64)
    if t[0]:
        rows_ = max(1, int(t[5]) // 64)
        print("  p%d s%d: total %d setup %d | per 64-event row: total %.0f match %.0f place %.0f close %.0f" % (
            k // 4, (k % 4) * 64, t[0], t[1], (t[0] - t[1]) / rows_, t[2] / rows_, t[3] / rows_, t[4] / rows_))
