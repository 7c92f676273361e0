#!/usr/bin/env python3
"""Randomised soak of the HIP -e1 / -e0 batch paths against the oracle (run by hand on a GPU box):
random context shapes, image sizes and contents, synchronous and overlapping batches, odd chunk
lengths.  Prints one line per round; exits non-zero on the first mismatch."""
import importlib, os, sys, time
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import torch
torch.cuda.init()
import inputs
from oracle.oracle import Oracle
pkg = importlib.import_module("nblic-image-compression_amd")
o = Oracle()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
t_end = time.time() + budget
rnd = 0
while time.time() < t_end:
    rnd += 1
    groups = int(rng.integers(1, 7)); gsize = int(rng.integers(1, 9)); coders = int(rng.integers(1, 17))
    chunk = int(rng.choice([4096, 10000, 65536, 262144, 1 << 22]))
    os.environ["NBLIC_AMD_CHUNK_BINS"] = str(chunk)
    ctx = pkg.Context(device=0, n_slots=groups * gsize, n_coders=coders, n_groups=groups, n_host_buffers=int(rng.integers(8, 200)))
    del os.environ["NBLIC_AMD_CHUNK_BINS"]
    try:
        batches = []
        for b in range(int(rng.integers(1, 4))):
            imgs = []
            for k in range(int(rng.integers(1, 60))):
                big = rng.random() < 0.1
                h, w = (int(rng.integers(300, 900)), int(rng.integers(300, 1100))) if big else (int(rng.integers(1, 200)), int(rng.integers(1, 260)))
                c = inputs.CONTENTS[int(rng.integers(0, len(inputs.CONTENTS)))]
                imgs.append(inputs.make(c, h, w) if c != "syn1" else inputs.syn1(h, w, seed=int(rng.integers(1, 1000))))
            batches.append(imgs)
        want = [[o.encode(i, 0, 1)[0] for i in imgs] for imgs in batches]
        if rng.random() < 0.5:
            tickets = [ctx.encode_begin([i.ctypes.data for i in imgs], [i.shape for i in imgs], False) for imgs in batches]
            order = rng.permutation(len(batches))
            got = {}
            for b in order:
                outs, lens = ctx.encode_end(tickets[b]); got[b] = [x[:int(n)].tobytes() for x, n in zip(outs, lens)]
            ok = all(got[b] == want[b] for b in range(len(batches)))
            mode = "overlapped"
        else:
            ok = all(ctx.encode_batch(imgs) == w_ for imgs, w_ in zip(batches, want))
            mode = "synchronous"
        if rng.random() < 0.3:                                   # an effort-0 batch through the same context
            q = batches[0][:8]
            ok = ok and ctx.qencode_batch(q) == [o.qencode(i) for i in q]
            mode += "+e0"
    finally:
        ctx.close()
    print("round %d: %d groups x %d, %d coders, chunk %d, %s, %d images: %s" % (rnd, groups, gsize, coders, chunk, mode, sum(len(b) for b in batches), "ok" if ok else "MISMATCH"), flush=True)
    if not ok:
        sys.exit(1)
print("soak ok: %d rounds" % rnd)
