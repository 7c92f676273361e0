"""Writes the coded-bin records (prob | bin << 15, what the host coder is fed) of a few SYN-1 frames to
<prefix>K.u16, for tools/feed_bench.cpp.   python tools/dump_coded.py /tmp/coded_ 8 [size]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
pkg = importlib.import_module("nblic-image-compression_amd")
prefix, count = sys.argv[1], int(sys.argv[2])
size = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
ctx = pkg.Context(0, n_slots=1, n_coders=1, n_groups=1)
for k in range(count):
    coded = ctx.debug_stage(pkg.syn1(size, size, k + 1), "coded")
    coded.tofile("%s%d.u16" % (prefix, k))
    print(k, coded.size, flush=True)
