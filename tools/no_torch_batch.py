"""The batch pipeline driven WITHOUT torch in the process, so that libnblic_amd.so binds to the system HIP
runtime (/opt/rocm) rather than the one bundled in the torch wheel -- to tell apart what the pipeline does from
what a runtime version does (e.g. which engine performs the device->host chunk copies).  The planes are put
into HBM through the same runtime with ctypes (hipMalloc / hipMemcpy), so the timed region is the bench's.

    rocprofv3 --kernel-trace --memory-copy-trace --stats ... -- python3 tools/no_torch_batch.py [frames] [rounds] [torch-first|-] [distinct planes]
"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
if len(sys.argv) > 3 and sys.argv[3] == "torch-first":       # the comparison: the wheel's bundled runtime
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    import torch
    torch.cuda.init()
pkg = importlib.import_module("nblic-image-compression_amd")
H = W = 4096
distinct = int(sys.argv[4]) if len(sys.argv) > 4 else 8
planes = [pkg.syn1(H, W, 1 + k) for k in range(distinct)]
ctx = pkg.Context(0, n_slots=48, n_coders=16, n_groups=6, n_host_buffers=336)
import ctypes as C
hip = C.CDLL("libamdhip64.so")                      # already in the process: the one libnblic_amd.so bound to
with open("/proc/self/maps") as f:
    print("runtime:", sorted({l.split()[-1] for l in f if "libamdhip64" in l}), flush=True)
dev = []
for p in planes:
    d = C.c_void_p()
    assert hip.hipMalloc(C.byref(d), C.c_size_t(p.size)) == 0
    assert hip.hipMemcpy(d, C.c_void_p(p.ctypes.data), C.c_size_t(p.size), 1) == 0
    dev.append(d.value)
ptrs = [dev[k % distinct] for k in range(frames)]
shapes = [(H, W)] * frames
out_sets = [[np.empty(pkg.out_capacity(H, W), np.uint8) for _ in range(frames)] for _ in range(2)]
ctx.encode_ptrs(ptrs, shapes, True, out_sets[0])                 # warm-up: rings, buffers
t0 = time.perf_counter()
pending = []
for r in range(rounds):                                          # two batches in flight, like bench.py
    if len(pending) == 2:
        lens = ctx.encode_end(pending.pop(0))[1]
    pending.append(ctx.encode_begin(ptrs, shapes, True, out_sets[r % 2]))
while pending:
    lens = ctx.encode_end(pending.pop(0))[1]
dt = time.perf_counter() - t0
print("%d rounds of %d frames, two in flight: %.3f s per round, %.0f Mpixel/s  first length %d" % (rounds, frames, dt / rounds, rounds * frames * H * W / dt / 1e6, int(lens[0])), flush=True)
ctx.close()
