#!/usr/bin/env python3
"""Discrete-event model behind the coder threads' take policy (pipeline.hip, coder_take).

The GPU delivers groups of 8 images every ~29 ms after a ~100 ms ramp; a coder thread codes k
images together in T(k) seconds (measured on the GPU box's EPYC 9575F: scalar 450 Mbins/s,
8 AVX-512 lanes ~137 Mbins/s per lane, 16 lanes ~91 Mbins/s per lane; 74.5 Mbins per 4096x4096
SYN-1 frame).  Prints the whole-batch Mpixel/s each policy would reach for several batch sizes.
No GPU needed.

What the model does NOT capture, and the in-situ runs showed: the host share is a hard 16-CPU
quota, so once every thread is busy the pipeline is paced by bins per CPU-second, not by per-thread
speed.  Measured at 512 frames per step: greedy packs 3257 Mpixel/s; waiting mid-batch for full
16-image packs 3382; an even-share drain from half-way through the batch 2976 (too many small,
inefficient packs).  pipeline.hip ships the second.  Also tried in situ: a two-slot scheduler
per thread (packs of eight that start and finish independently at chunk boundaries and are stepped
in lock-step whenever both are running) -- bit-exact, busy 88 % of the time, but 3250-3340
Mpixel/s: the periods with one pack running alone cost what the continuous refill gains."""
import heapq

BINS = 74.5e6

def T(k):
    """seconds a thread needs for k images together (in-situ rates, Mbins/s per thread)"""
    if k == 1: return BINS / 400e6
    rate = 650e6 * min(k, 8) / 4 if k <= 4 else (1300e6 + (k - 8) * 72e6 if k >= 8 else 650e6 + (k - 4) * 162e6)   # 2x2..2x4 lanes, then up to 2x8 = 1875
    return BINS * k / rate

def run(B, policy, threads=16, t0=0.10, gap=0.029, group=8):
    events = [(t0 + i * gap, 'a', group) for i in range(B // group)]
    heapq.heapify(events)
    q, to_come, idle, tend = 0, B, threads, 0.0
    while events:
        t, kind, n = heapq.heappop(events)
        if kind == 'a': q += n; to_come -= n
        else: idle += 1; tend = t
        while idle > 0 and q > 0:
            k = policy(q, to_come, threads, idle)
            if k <= 0: break
            k = min(k, q); q -= k; idle -= 1
            heapq.heappush(events, (t + T(k), 'd', k))
    return tend

def idle_driven(mult, use16):
    def pol(q, to_come, threads, idle):
        left = q + to_come
        if q >= 8 and idle == 1 and left >= mult * threads:
            return 16 if (use16 and q >= 16 and left >= 8 * threads) else 8
        return 1
    return pol

def even_share(q, to_come, threads, idle):
    left = q + to_come; k = -(-left // threads)
    if k < 3: return 1
    want = 16 if k >= 14 else min(k, 8)
    if q >= want: return want
    return q if to_come == 0 else 0

if __name__ == "__main__":
    def greedy(q, to_come, threads, idle):               # as many as are queued, up to 16
        return min(q, 16)
    def greedy_tail(mult):
        def pol(q, to_come, threads, idle):
            left = q + to_come
            if left < mult * threads: return 1 if left < 2 * threads else min(q, max(1, -(-left // threads)))
            return min(q, 16)
        return pol
    pols = {"singles": lambda *a: 1, "always-8": lambda q, *a: 8 if q >= 8 else 1, "even-share": even_share,
            "idle-driven 8": idle_driven(3, 0), "idle-driven 8/16": idle_driven(3, 1), "greedy<=16": greedy,
            "greedy+tail4": greedy_tail(4), "greedy+tail8": greedy_tail(8)}
    for B in (16, 64, 256, 512, 1024):
        print("B=%4d  " % B + "  ".join("%s %.0f" % (n, B * 16.777216 / run(B, p)) for n, p in pols.items()))
