import importlib, sys, time, os
import numpy as np
sys.path.insert(0, '/root/repo')
import torch
torch.cuda.init()
pkg = importlib.import_module("nblic-image-compression_amd")
rng = np.random.default_rng(1)
n = 6_000_000
p1 = np.clip((rng.normal(0.5, 0.35, n) * 4096).astype(np.int64), 1, 4095)
bins = (rng.random(n) < p1 / 4096.0)
base = (p1.astype(np.uint16) | (bins.astype(np.uint16) << 15))
plain = [np.roll(base, 977 * k).copy() for k in range(16)]
pinned_t = [torch.empty(n, dtype=torch.int16, pin_memory=True) for _ in range(16)]
pinned = []
for t, a in zip(pinned_t, plain):
    v = t.numpy().view(np.uint16); v[:] = a; pinned.append(v)
# pinned memory just written by the GPU (DMA), never touched by the CPU since
dev = [torch.from_numpy(a.view(np.int16)).cuda() for a in plain]
def timed(f):
    t = time.perf_counter(); r = f(); return r, time.perf_counter() - t
for rep in range(3):
    (_, _), tp = timed(lambda: pkg.range_code_multi(plain))
    (_, _), tq = timed(lambda: pkg.range_code_multi(pinned))
    for t, d in zip(pinned_t, dev): t.copy_(d, non_blocking=True)
    torch.cuda.synchronize()
    (_, _), tr = timed(lambda: pkg.range_code_multi(pinned))
    print("x16 Mbins/s: pageable %.0f  pinned (CPU-written) %.0f  pinned (just DMA-written) %.0f" % (16*n/tp/1e6, 16*n/tq/1e6, 16*n/tr/1e6))
# pageable memory first touched by this thread, then page-locked in place (hipHostRegister)
reg = [a.copy() for a in plain]
rt = torch.cuda.cudart()
for a in reg:
    rc = rt.cudaHostRegister(a.ctypes.data, a.nbytes, 0)
    assert int(rc) == 0, rc
for rep in range(3):
    (_, _), tg = timed(lambda: pkg.range_code_multi(reg))
    for a, d in zip(reg, dev):
        torch.from_numpy(a.view(np.int16)).copy_(d, non_blocking=True)
    torch.cuda.synchronize()
    (_, _), th = timed(lambda: pkg.range_code_multi(reg))
    print("x16 Mbins/s: registered (CPU-written) %.0f  registered (just DMA-written) %.0f" % (16*n/tg/1e6, 16*n/th/1e6))
