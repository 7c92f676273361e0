#!/usr/bin/env python3
"""bench.py -- Mpixel/s of bit-exact -n0 -e1 NBLIC encode on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of B (default 512) synthetic 4096x4096 8-bit gray
frames (SYN-1, BASELINE config 2) per GPU: the frames are already resident in HBM when the
timed region starts; a step is complete when every byte-exact .nblic stream is in host memory
(and, for N > 1, gathered on rank 0 over RCCL).  Steps are submitted back to back, as a continuous
feed would be -- step k+1 is handed to the pipeline (nblic_amd_encode_batch_begin) before step k is
collected -- and every one of the K timed steps is complete before the closing barrier; use
--no-overlap-steps to collect each step before submitting the next.  Prints ONE JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
"""
import argparse
import hashlib
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The pipeline keeps 6 group streams + 8 copy streams + 1 serial-engine stream busy.  The HIP runtime
# multiplexes streams onto 4 hardware queues by default, which puts the coder threads' copies
# behind other groups' kernels; it reads this when it initialises, i.e. before torch is imported.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 measured copy


def cpu_baseline(frames, effort=1, reps_budget_s=12.0):
    """Single-thread CPU encode of the same frames: the compiled reference when oracle/_ref
    travelled with the snapshot, else our CPU port.  Bounded sample (a few frames).  The thread is PINNED to
    one core of the rank's slice for the measurement (taskset) -- the last one, which no
    coder thread is bound to -- and the line says which."""
    from oracle.oracle import Oracle, Reference
    if Reference.available():
        r = Reference()
        enc, kind = (r.encode if effort else (lambda im, n, e: (r.qencode(im),))), "reference"
    else:
        o = Oracle()
        enc, kind = ((lambda im, n, e: o.encode(im, n, e)) if effort else (lambda im, n, e: (o.qencode(im),))), "port"
    pinned, before = None, None
    try:
        before = os.sched_getaffinity(0)
        pinned = max(before)
        os.sched_setaffinity(0, {pinned})               # pid 0 = the calling thread only
    except (AttributeError, OSError, ValueError):
        pinned = None
    px, t_total, used, streams = 0, 0.0, 0, []
    try:
        for f in frames:
            t0 = time.perf_counter()
            streams.append(enc(f, 0, 1)[0])
            t_total += time.perf_counter() - t0
            px += f.size
            used += 1
            if t_total > reps_budget_s:
                break
    finally:
        if before is not None:
            try:
                os.sched_setaffinity(0, before)
            except OSError:
                pass
    return {"value": round(px / t_total / 1e6, 3), "unit": "Mpixel/s", "cores": 1, "kind": kind, "pinned_core": pinned,
            "sample": f"{used} of the batch's {frames[0].shape[0]}x{frames[0].shape[1]} SYN-1 frames, -n0 -e{effort}, one thread pinned to one core"}, streams, enc


def verify_frames(enc, make_frame, indices, got, workers):
    """Byte-for-byte check of the GPU streams of the frames `indices` against the CPU encoder (outside the
    timed region; the C encoders release the GIL, so `workers` threads run side by side)."""
    from concurrent.futures import ThreadPoolExecutor

    def one(k):
        return enc(make_frame(k), 0, 1)[0] == got(k)
    with ThreadPoolExecutor(max_workers=max(1, workers)) as ex:
        return list(ex.map(one, indices))


def profiled_traffic(kernel, images_per_launch):
    """HBM BYTES per launch of `kernel` from the committed rocprofv3 PMC summary (separate
    --pmc FETCH_SIZE / --pmc WRITE_SIZE passes, FETCH doubled as the gfx950 guide prescribes;
    profiles/*_hbm_traffic.csv, produced by tools/summarize_profile.py), rescaled to this run's
    images per launch.  None when no summary has been committed for that kernel."""
    import csv, glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.csv")))
    if not files:
        return None
    meta = {}
    try:
        with open(files[-1].replace("_hbm_traffic.csv", "_pmc_config.json")) as f:
            meta = json.load(f)
    except OSError:
        pass
    prof_ipl = float(meta.get("images_per_launch", 8))
    for row in csv.DictReader(open(files[-1])):
        if row["kernel"].split("(")[0].split("::")[-1].split("<")[0] == kernel:     # "void nblic::k_touch_scatter<false>(...)" -> k_touch_scatter
            col = next(c for c in row if c.startswith("hbm_MB_per_launch_corrected"))
            return int(float(row[col]) * 1048576 * images_per_launch / prof_ipl)
    return None


def launch_ranks(n_gpus, backend):
    """`python bench.py --gpus N` started directly (no WORLD_SIZE in the environment): this process becomes
    the launcher.  It touches neither torch nor the GPU; it starts N children of this same script, one
    rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as torch.distributed.run would set them),
    lets rank 0 write the JSON line to our stdout, and exits non-zero if any rank does."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n_gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_gpus), LOCAL_WORLD_SIZE=str(n_gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    for r, pr in enumerate(procs):
        code = pr.wait()
        if code != 0:
            print(f"[bench] rank {r} exited with {code}", file=sys.stderr)
            rc = rc or code or 1
    sys.exit(rc)


def parse_cpulist(text):
    """'0-15,128-143' -> [0, ..., 15, 128, ..., 143]"""
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out += list(range(int(lo), int(hi or lo) + 1))
    return out


def gpu_local_cpulists(sysfs="/sys"):
    """local_cpulist (the CPUs of the GPU's own NUMA node) of every AMD GPU, in render-node order -- the order the HIP
    runtime enumerates them in; HIP_VISIBLE_DEVICES (a list of ordinals) is applied.  [] when sysfs does not say."""
    import glob
    nodes = []
    for d in glob.glob(os.path.join(sysfs, "class", "drm", "renderD*")):
        try:
            with open(os.path.join(d, "device", "vendor")) as f:
                if int(f.read().strip(), 16) != 0x1002:
                    continue
            with open(os.path.join(d, "device", "local_cpulist")) as f:
                cpus = parse_cpulist(f.read())
            numa = -1
            try:
                with open(os.path.join(d, "device", "numa_node")) as f:
                    numa = int(f.read().strip())
            except (OSError, ValueError):
                pass
            nodes.append((int(os.path.basename(d)[7:]), numa, cpus))
        except (OSError, ValueError):
            continue
    nodes.sort()
    vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")
    if vis:
        try:
            nodes = [nodes[int(v)] for v in vis.split(",") if v.strip() != ""]
        except (ValueError, IndexError):
            pass
    return [(numa, cpus) for _, numa, cpus in nodes]


def cpu_share(local_rank, gpus_per_node, sysfs="/sys", avail=None):
    """The logical CPUs this rank confines itself to: whole physical cores (both SMT siblings) of ITS GPU's NUMA node
    (sysfs local_cpulist of the GPU's PCI device), that node's cores split evenly, in core order, between the GPUs that
    hang off it.  At N = 1 the rank still takes only the share one of `gpus_per_node` GPUs gets, so the single-GPU figure
    is what a rank of the 8-GPU job has.  Where sysfs says nothing about the GPUs (containers, this build box) the host's
    cores are split evenly by rank as before.  Returns (cpus, how) or (None, why)."""
    try:
        avail = sorted(os.sched_getaffinity(0)) if avail is None else sorted(avail)
        cores = {}
        for c in avail:
            with open(os.path.join(sysfs, "devices", "system", "cpu", f"cpu{c}", "topology", "thread_siblings_list")) as f:
                first = parse_cpulist(f.read())[0]
            cores.setdefault(first, []).append(c)
    except (OSError, ValueError, AttributeError, IndexError):
        return None, "cpu topology unreadable"
    gpus = gpu_local_cpulists(sysfs)
    slot = local_rank % max(1, gpus_per_node)
    if len(gpus) > slot and gpus[slot][1]:
        numa, local = gpus[slot]
        local = set(local)
        mine = [k for k in sorted(cores) if k in local]                       # physical cores of the GPU's node that we may use
        peers = [i for i, (nn, cc) in enumerate(gpus[:max(gpus_per_node, len(gpus))]) if set(cc) == local]
        share_of = max(len(peers), gpus_per_node * len(mine) // max(1, len(cores)))   # a lone visible GPU still leaves room for the node's other GPUs
        idx = peers.index(slot) if slot in peers else 0
        per = len(mine) // max(1, share_of)
        if per >= 1:
            part = mine[idx * per:(idx + 1) * per]
            return sorted(c for k in part for c in cores[k]), f"numa node {numa} of GPU {slot}: cores {part[0]}..{part[-1]} ({idx + 1} of {share_of} on that node)"
    ordered = [cores[k] for k in sorted(cores)]
    per = len(ordered) // max(1, gpus_per_node)
    if per < 1:
        return None, "fewer cores than GPUs"
    mine = ordered[slot * per:(slot + 1) * per]
    return sorted(c for core in mine for c in core), f"even split by rank (sysfs names no NUMA node for GPU {slot})"


def host_description():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        cpus = len(os.sched_getaffinity(0))
    except AttributeError:
        cpus = os.cpu_count() or 1
    toggles = {k: os.environ[k] for k in ("GPU_MAX_HW_QUEUES", "NBLIC_AMD_HOSTMALLOC", "NBLIC_AMD_CHUNK_BINS", "NBLIC_AMD_DBG",
                                          "NBLIC_AMD_NO_SIMD", "NBLIC_AMD_DEVICE", "NBLIC_BENCH_DEVICE", "NBLIC_AMD_COPY_STREAMS",
                                          "NBLIC_BENCH_NO_STAGE_TIMING", "NBLIC_BENCH_SYSTEM_HIP", "DEBUG_CLR_LIMIT_BLIT_WG", "HSA_CU_MASK") if k in os.environ}
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:           # "max 100000" or "<quota us> <period us>": CPUs' worth of time the cgroup may use
            q, per = f.read().split()[:2]
            quota = "unlimited" if q == "max" else round(int(q) / int(per), 2)
    except (OSError, ValueError):
        pass
    return {"cpu_model": model, "cpus_used_by_this_rank": cpus, "cpus_total": os.cpu_count(), "cgroup_cpu_quota_cpus": quota, "env": toggles}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)       # steps are submitted back to back: the pipeline's fill and drain (~1 s) are inside the region ONCE, so a longer run of steps weighs them less
    ap.add_argument("--warmup", type=int, default=3)       # the coder threads' rings grow (and are page-locked) during the first steps
    ap.add_argument("--batch", type=int, default=512, help="frames per GPU per step (the pipeline's fill and drain, ~0.1 s + ~0.8 s, are inside every step)")
    ap.add_argument("--height", type=int, default=4096)
    ap.add_argument("--width", type=int, default=4096)
    ap.add_argument("--coders", type=int, default=0, help="host range-coder threads per GPU (0 = CPU share)")
    ap.add_argument("--slots", type=int, default=0, help="images in flight per GPU (0 = min(batch, 48))")
    ap.add_argument("--groups", type=int, default=6, help="launch groups the images in flight are split into")
    ap.add_argument("--host-buffers", type=int, default=0, help="coded-bin buffers in HBM between the GPU and the coder threads (0 = slots + 16*coders + 32)")
    ap.add_argument("--host-inputs", action="store_true", help="hand host buffers over (PCIe-inclusive rate)")
    ap.add_argument("--gather-chunk", type=int, default=256, help="frames per exchange of the N>1 gather (bounds rank 0's receive buffers: world x 2.3 GB at 256)")
    ap.add_argument("--no-overlap-steps", dest="overlap_steps", action="store_false",
                    help="collect every step before submitting the next (default: step k+1 is submitted with nblic_amd_encode_batch_begin "
                         "before step k is collected, as a continuous feed would; the pipeline's fill and drain are then paid once per "
                         "run of steps, inside the timed region, instead of once per step)")
    ap.add_argument("--device-packs", type=int, default=0, help="pack threads of the device range coder (64 images per wave, one lane per image), a supplement to the host coder threads; 0 = host threads only "
                                                                 "(default: measured on MI355X a lane codes ~7 Mbins/s, a pack of 64 4096^2 frames takes ~11 s and its kernel stalls the streams that share its hardware queue: 5.4 -> 2.9 Gpx/s with one pack, DESIGN.md section 4)")
    ap.add_argument("--steps-in-flight", type=int, default=0, help="steps submitted ahead of the one being collected (0 = 4 with the device coder, else 2)")
    ap.add_argument("--device-min-outstanding", type=int, default=-1, help="images that must be unfinished for the device coder to take a pack (-1 = 2.8 x batch: its seconds of latency never become the tail of the run)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stage-timing", choices=("roofline", "all", "off"), default="roofline",
                    help="HIP events around k_touch_scatter and k_predict only (default), around every stage, or none")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL) for real runs; gloo lets the N>1 path be rehearsed with every rank on one GPU")
    ap.add_argument("--node-gpus", type=int, default=8, help="GPUs the host is shared between: a rank confines itself (threads and all) to 1/NODE_GPUS of the host's cores, both SMT siblings of each; 0 = no confinement")
    ap.add_argument("--launch-check", action="store_true", help="no GPU work: every rank joins a gloo group, rank 0 prints how many ranks it saw (tests the --gpus N launcher on a CPU-only box)")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the extra measurements (single frame, 8-frame batch, the H2D-inclusive leg, the all-stage timing passes)")
    ap.add_argument("--effort", type=int, choices=(0, 1), default=1, help="1: the graded -n0 -e1 path (BASELINE config 2); 0: QNBLIC (config 1's mode) on the same frames, one synchronous batch call per step")
    ap.add_argument("--h2d-steps", type=int, default=10, help="timed steps of the H2D-inclusive leg (SURVEY 8(d): >= 10 after >= 3 warm-ups); 0 skips it")
    ap.add_argument("--h2d-warmup", type=int, default=3)
    ap.add_argument("--pinned-budget-gb", type=float, default=24.0, help="page-locked host memory one rank may hold (output slabs + coder rings); the coder threads' chunk is halved until the total fits, so eight ranks stay under 8 x this")
    args = ap.parse_args()

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        launch_ranks(args.gpus, args.dist_backend)           # does not return
    if env_world is not None and int(env_world) != args.gpus:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={env_world}: refusing to report a mislabelled run", file=sys.stderr)
        sys.exit(2)

    share, share_how = None, "no confinement"
    if args.node_gpus > 0:
        fake_cpus = os.environ.get("NBLIC_BENCH_ASSUME_CPUS")             # tests: a topology that is not this machine's (with NBLIC_BENCH_SYSFS)
        share, share_how = cpu_share(int(os.environ.get("LOCAL_RANK", "0")), max(args.node_gpus, int(os.environ.get("LOCAL_WORLD_SIZE", "1"))),
                                     os.environ.get("NBLIC_BENCH_SYSFS", "/sys"), range(int(fake_cpus)) if fake_cpus else None)
        if share and not args.launch_check and not fake_cpus:
            os.sched_setaffinity(0, share)              # before torch / HIP start their threads: they inherit it

    if os.environ.get("NBLIC_BENCH_SYSTEM_HIP"):            # experiment: /opt/rocm's HIP + HSA runtimes instead of the ones bundled in the torch wheel
        import ctypes
        for name in ("libhsa-runtime64.so", "libamdhip64.so"):
            ctypes.CDLL(name, mode=ctypes.RTLD_GLOBAL)
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.launch_check:
        seen, slices = 1, [{"rank": rank, "cpus": share, "how": share_how}]
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo")
            t = torch.tensor([1], dtype=torch.int64)
            dist.all_reduce(t)
            seen = int(t.item())
            every = [None] * world
            dist.all_gather_object(every, slices[0])
            slices = every
            dist.destroy_process_group()
        if rank == 0:
            line = {"launch_check": True, "n_gpus": world, "ranks_seen": seen, "local_rank": local_rank}
            if os.environ.get("NBLIC_BENCH_SYSFS") or os.environ.get("NBLIC_BENCH_SHOW_SLICES"):
                line["cpu_slices"] = slices
            print(json.dumps(line), flush=True)
        return
    gpu = int(os.environ.get("NBLIC_BENCH_DEVICE", local_rank))       # rehearsal: all ranks on one GPU
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", gpu))
        else:
            dist.init_process_group(args.dist_backend)
    torch.cuda.set_device(gpu)
    dev = torch.device("cuda", gpu)
    comm_dev = dev if args.dist_backend == "nccl" else torch.device("cpu")

    pkg = importlib.import_module("nblic-image-compression_amd")
    H, W, B = args.height, args.width, args.batch
    try:
        cpus = len(os.sched_getaffinity(0))
    except AttributeError:
        cpus = os.cpu_count() or 1
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if share:
        coders = args.coders or max(1, min(B, 16, (len(share) + 1) // 2))    # one coder thread per physical CORE of the slice (measured: 32 threads on 16 cores + SMT siblings 3918 vs 5210 Mpx/s)
    else:
        coders = args.coders or max(1, min(B, 16, cpus // max(1, local_world)))   # 16 = one GPU's CPU share
    slots = args.slots or min(B, 48)

    # frames are generated and uploaded 128 at a time: only the CPU baseline's sample (and, with
    # --host-inputs, everything) stays on the host
    from concurrent.futures import ThreadPoolExecutor
    frames, dev_frames = [], []
    with ThreadPoolExecutor(max_workers=max(1, min(16, cpus if share else cpus // max(1, local_world)))) as ex:   # the C generator releases the GIL
        for k0 in range(0, B, 128):
            part = list(ex.map(lambda k: pkg.syn1(H, W, seed=rank * B + k + 1), range(k0, min(B, k0 + 128))))
            dev_frames += [torch.from_numpy(f).to(dev) for f in part]
            torch.cuda.synchronize()
            frames += part if args.host_inputs else part[: max(0, 8 - len(frames))]
    dev_packs = args.device_packs if (args.overlap_steps and B >= 256) else 0            # the device coder needs a deep backlog to be worth its latency
    in_flight = args.steps_in_flight or (4 if dev_packs else 2)
    if not args.overlap_steps:
        in_flight = 1
    host_buffers = args.host_buffers or (min(B + 16, slots + 24 * coders + 32) + 64 * dev_packs)   # groups in flight + every thread's sixteen + a queue (+ the device coder's packs)
    # page-locked host memory of this rank: the output slabs (a set per step in flight) and every coder thread's ring
    # (3 slots x 24 lanes x chunk bins x 13/8 bytes).  Eight ranks share one host: the chunk is halved until the rank is
    # under its budget (the library reads NBLIC_AMD_CHUNK_BINS when the context is created)
    cap = H * W + H * W // 4 + 4096 if args.host_inputs else H * W * 3 // 4 + 4096   # SYN-1 codes to 0.53 B/px
    chunk_bins = int(os.environ.get("NBLIC_AMD_CHUNK_BINS", 1 << 22))
    slab_bytes = in_flight * B * cap
    ring_bytes = lambda ch: coders * 3 * 24 * ch * 13 // 8
    while slab_bytes + ring_bytes(chunk_bins) > args.pinned_budget_gb * 1e9 and chunk_bins > (1 << 18):
        chunk_bins //= 2
    if chunk_bins != (1 << 22):
        os.environ["NBLIC_AMD_CHUNK_BINS"] = str(chunk_bins)
    pinned = {"output_slabs_GB": round(slab_bytes / 1e9, 2), "coder_rings_GB": round(ring_bytes(chunk_bins) / 1e9, 2), "budget_GB": args.pinned_budget_gb,
              "coder_chunk_bins": chunk_bins}
    ctx = pkg.Context(device=gpu, n_slots=slots, n_coders=coders, n_groups=max(1, min(args.groups, slots)),
                      n_host_buffers=host_buffers)
    # HIP events around the two kernels the line reports against a roof (k_touch_scatter, the dominant one by time per
    # launch in every full-timing run and in profiles/*_kernel_stats.csv, and k_predict); --stage-timing all times all 31
    # stages (2 % slower: thirty-two events per group launch), off none
    ctx.enable_timing(0 if os.environ.get("NBLIC_BENCH_NO_STAGE_TIMING") else {"roofline": 2, "all": 1, "off": 0}[args.stage_timing])
    if dev_packs:
        ctx.set_device_coder(dev_packs, args.device_min_outstanding if args.device_min_outstanding >= 0 else int(2.8 * B))
    # streams land in one pinned slab (a slot per frame: worst case seen is 1.0025 B/px) so that the
    # multi-GPU gather can stage them to HBM with plain async copies
    slab = torch.empty((B, cap), dtype=torch.uint8, pin_memory=True)
    outs = [slab[k].numpy() for k in range(B)]
    slabs, out_sets = [slab], [outs]
    for _ in range(1, in_flight):                         # several steps in flight: a set of output buffers each
        slabs.append(torch.empty((B, cap), dtype=torch.uint8, pin_memory=True))
        out_sets.append([slabs[-1][k].numpy() for k in range(B)])
    shapes = [(H, W)] * B
    GATHER_CHUNK = max(1, args.gather_chunk)              # frames per exchange: bounds rank 0's receive buffers (world x 2.3 GB at 256)
    dev_pack = torch.empty(min(B, GATHER_CHUNK) * cap, dtype=torch.uint8, device=comm_dev) if world > 1 else None
    ptrs = [f.ctypes.data for f in frames] if args.host_inputs else [d.data_ptr() for d in dev_frames]
    gather = None
    if world > 1:
        gather = importlib.import_module("nblic-image-compression_amd.gather")

    last = {}

    all_lens = []
    step_done = []                                        # when each step of the current run() was complete (streams in host memory, exchanged)
    outs16 = [[o[: o.size & ~1].view(np.uint16) for o in oset] for oset in out_sets]     # effort 0 writes 16-bit words (QNBLIC.h:14)

    def exchange(lens, which):
        last["lens"], last["outs"] = lens, out_sets[which]
        all_lens.append(np.array(lens, copy=True))
        if world > 1:                                   # the one exchange of the path: streams -> rank 0 (stay in HBM)
            for k0 in range(0, B, GATHER_CHUNK):
                k1 = min(B, k0 + GATHER_CHUNK)
                off = 0
                for k in range(k0, k1):
                    n = int(lens[k])
                    dev_pack[off:off + n].copy_(slabs[which][k, :n], non_blocking=True)
                    off += n
                last["gathered"] = None                 # release the previous receive buffers BEFORE the next ones are allocated
                last["gathered"] = gather.gather_packed(dev_pack[:off], torch.from_numpy(lens[k0:k1]).to(comm_dev))
                last["gathered_first"] = k0
            torch.cuda.current_stream().synchronize()   # the slab rows are free for a later step's coders only now

    def run(n_steps, src=None, on_device=None):
        src = ptrs if src is None else src
        on_device = (not args.host_inputs) if on_device is None else on_device
        del step_done[:]
        if args.effort == 0:                             # QNBLIC: one synchronous batch call per step (lengths come back in words)
            for _ in range(n_steps):
                _, words = ctx.qencode_ptrs(src, shapes, on_device, outs16[0])
                exchange(words * 2, 0)
                step_done.append(time.perf_counter())
            return
        if not args.overlap_steps:
            for _ in range(n_steps):
                _, lens = ctx.encode_ptrs(src, shapes, on_device, outs)
                exchange(lens, 0)
                step_done.append(time.perf_counter())
            return
        pending = []                                    # up to in_flight steps are submitted before the oldest is collected (and exchanged)
        for i in range(n_steps):
            if len(pending) == in_flight:
                t, which = pending.pop(0)
                exchange(ctx.encode_end(t)[1], which)
                step_done.append(time.perf_counter())
            pending.append((ctx.encode_begin(src, shapes, on_device, out_sets[i % in_flight]), i % in_flight))
        for t, which in pending:
            exchange(ctx.encode_end(t)[1], which)
            step_done.append(time.perf_counter())

    def step_stats(t_start):
        """median / min of the intervals between the completions of consecutive steps (the first one is measured from
        the start of the region and, with steps in flight, contains the pipeline's fill)"""
        marks = [t_start] + list(step_done)
        iv = sorted(b - a for a, b in zip(marks[:-1], marks[1:]))
        if not iv:
            return None, None
        return round(iv[len(iv) // 2] * 1e3 if len(iv) % 2 else (iv[len(iv) // 2 - 1] + iv[len(iv) // 2]) * 0.5e3, 3), round(iv[0] * 1e3, 3)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run(args.warmup)
    del all_lens[:]
    fence()
    t0 = time.perf_counter()
    run(args.steps)
    fence()
    dt = time.perf_counter() - t0
    step_median_ms, step_min_ms = step_stats(t0)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    timed_stage = ctx.stage_times()                   # the two stages timed INSIDE the region (summed over its group launches)
    timed_launches = max(1, ctx.last_launches())
    timed_bins, timed_coder_s = ctx.last_stats()

    # ---- reporting (outside the timed region) ------------------------------------------------
    if os.environ.get("NBLIC_BENCH_THREAD_CPU"):          # who used the rank's CPU share: CPU seconds per thread name since process start
        by_name = {}
        tick = os.sysconf("SC_CLK_TCK")
        for tid in os.listdir("/proc/self/task"):
            try:
                with open(f"/proc/self/task/{tid}/comm") as f:
                    name = f.read().strip()
                with open(f"/proc/self/task/{tid}/stat") as f:
                    fields = f.read().rsplit(")", 1)[1].split()
                u, sy = int(fields[11]) / tick, int(fields[12]) / tick
            except (OSError, IndexError, ValueError):
                continue
            e = by_name.setdefault(name, [0, 0.0, 0.0])
            e[0] += 1; e[1] += u; e[2] += sy
        for name, (cnt, u, sy) in sorted(by_name.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
            print(f"[bench] threads {name!r} x{cnt}: user {u:.1f} s, system {sy:.1f} s", file=sys.stderr)
        try:
            with open("/sys/fs/cgroup/cpu.stat") as f:
                print("[bench] cgroup cpu.stat: " + " ".join(f.read().split()), file=sys.stderr)
        except OSError:
            pass
    lens = last["lens"]
    timed_lens = np.array(lens, copy=True)
    timed_outs, timed_all_lens = last["outs"], list(all_lens)
    bins, coder_s = timed_bins, timed_coder_s
    dev_stats = ctx.device_coder_stats() if dev_packs else None
    steps_counted = args.steps if (args.overlap_steps and args.effort == 1) else 1      # the library's counters run over all overlapped steps
    imgs_per_launch = B * steps_counted / timed_launches
    timed_per_launch = {k: v / timed_launches for k, v in timed_stage.items() if k != "host_gap" and v > 0}

    gathered_ok = None
    if world > 1 and rank == 0:
        payloads, lens_all = last["gathered"]             # the step's last exchange (frames gathered_first ..)
        k0 = last["gathered_first"]
        gathered_ok = (len(payloads) == world and all(int(l.sum()) == p.numel() for p, l in zip(payloads, lens_all)) and
                       hashlib.sha256(payloads[0][: int(lens_all[0][0])].cpu().numpy().tobytes()).hexdigest() ==
                       hashlib.sha256(last["outs"][k0][: int(lens[k0])].tobytes()).hexdigest())

    # ---- correctness of what was timed (outside the timed region; before the extra legs reuse the buffers) --------
    bit_exact, checks = None, {}
    if rank == 0:
        try:
            with open(os.path.join(ROOT, "tests", "golden", "manifest.json")) as f:
                m = json.load(f)["large"].get(f"syn1s1_{H}x{W}_n0_e1" if args.effort else f"syn1s1_{H}x{W}_q0")
            if m:
                s0 = last["outs"][0][: int(lens[0])].tobytes()
                checks["frame0_golden_sha"] = (len(s0) == m["len"] and hashlib.sha256(s0).hexdigest() == m["sha256"])
        except OSError:
            pass
        checks["every_step_same_lengths"] = all(np.array_equal(l, all_lens[0]) for l in all_lens) and len(all_lens) == args.steps
        if args.effort == 1 and args.overlap_steps and args.steps >= in_flight and in_flight >= 2 and args.warmup + args.steps >= in_flight:   # the last steps' slabs, byte for byte
            la = torch.from_numpy(np.asarray(lens)).clamp(max=cap)
            same = True
            for k in range(B):
                n = int(la[k])
                same = same and all(bool(torch.equal(slabs[0][k, :n], sl[k, :n])) for sl in slabs[1:])
            checks[f"last_{in_flight}_steps_identical"] = same
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu, ref_streams, enc = cpu_baseline(frames, args.effort)
        ok = [ref_streams[k] == last["outs"][k][: int(lens[k])].tobytes() for k in range(len(ref_streams))]
        spread = [k for k in sorted({(B * j) // 16 + (j % 16) for j in range(1, 16)}) if len(ref_streams) <= k < B]   # other packs, other lanes
        ok += verify_frames(enc, lambda k: pkg.syn1(H, W, seed=rank * B + k + 1), spread,
                            lambda k: last["outs"][k][: int(lens[k])].tobytes(), min(16, cpus))
        checks["frames_vs_cpu_encoder"] = f"{sum(ok)}/{len(ok)}"
        checks["frames_checked"] = list(range(len(ref_streams))) + spread
        checks["all_frames_vs_cpu_ok"] = all(ok)
    if rank == 0:
        bit_exact = all(v for k, v in checks.items() if isinstance(v, bool))

    # ---- extra legs (outside the headline's region): the H2D-inclusive rate under the same protocol, every stage timed
    # under load and alone, one frame alone, the 8-frame batch north_star names ----------------------------------------
    extra = {}
    all_per_launch, alone_per_launch = {}, {}

    def timed_call(fn):
        fence()
        t0 = time.perf_counter()
        fn()
        fence()
        d = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([d], dtype=torch.float64, device=comm_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            d = float(tt.item())
        return d
    if not args.no_extra_legs and args.effort == 1:
        # every one of the 31 stages timed in THIS run: two overlapped steps under the same six-group concurrency as the
        # headline (thirty-two events per group launch cost ~2 %, which is why the timed region carries only two stages) ...
        ctx.enable_timing(1)
        run(min(2, max(1, args.steps)))
        fence()
        st, ln = ctx.stage_times(), max(1, ctx.last_launches())
        all_per_launch = {k: v / ln for k, v in st.items() if k != "host_gap"}
        # ... and one group of frames ALONE on the GPU (nothing else in flight): what each kernel needs by itself
        n_alone = max(1, min(B, slots // max(1, min(args.groups, slots))))
        ctx.encode_ptrs(ptrs[:n_alone], shapes[:n_alone], not args.host_inputs, outs[:n_alone])
        fence()
        st, ln = ctx.stage_times(), max(1, ctx.last_launches())
        alone_per_launch = {k: v / ln for k, v in st.items() if k != "host_gap"}
        extra["frames_per_launch_alone"] = n_alone / ln
        ctx.enable_timing(0)
    if not args.no_extra_legs:
        one = max(1, 8 // world)                                     # 8 frames over the whole job: 8 / N per GPU
        enc_some = (lambda k: ctx.qencode_ptrs(ptrs[:k], shapes[:k], not args.host_inputs, outs16[0][:k])) if args.effort == 0 else \
                   (lambda k: ctx.encode_ptrs(ptrs[:k], shapes[:k], not args.host_inputs, outs[:k]))
        d8 = sorted(timed_call(lambda: enc_some(one)) for _ in range(5))
        extra["batch8_Mpixel_per_s"] = round(one * world * H * W / d8[len(d8) // 2] / 1e6, 2)
        extra["batch8_ms"] = {"median": round(d8[len(d8) // 2] * 1e3, 2), "min": round(d8[0] * 1e3, 2), "repetitions": len(d8)}
        extra["batch8_frames_per_gpu"] = one
        if world == 1:
            d1 = sorted(timed_call(lambda: enc_some(1)) for _ in range(13))[:-3]     # 13 repetitions, the three slowest (warm-up) dropped
            extra["single_frame_ms"] = round(d1[len(d1) // 2] * 1e3, 2)
            extra["single_frame"] = {"median_ms": round(d1[len(d1) // 2] * 1e3, 2), "min_ms": round(d1[0] * 1e3, 2), "repetitions": len(d1)}
    if not args.no_extra_legs and world == 1 and not args.host_inputs and args.h2d_steps > 0 and args.effort == 1:
        # SURVEY 8(d)'s quantity: input in (pinned) host memory, every frame uploaded over PCIe INSIDE the region, same
        # protocol as the headline: warm-up steps, then K timed steps submitted back to back, barrier to barrier
        hslab = torch.empty((B, H * W), dtype=torch.uint8, pin_memory=True)
        hrows = [hslab[k].numpy() for k in range(B)]
        with ThreadPoolExecutor(max_workers=max(1, min(16, cpus))) as ex:
            list(ex.map(lambda k: np.copyto(hrows[k], pkg.syn1(H, W, seed=rank * B + k + 1).reshape(-1)), range(B)))
        hptrs = [r.ctypes.data for r in hrows]
        run(args.h2d_warmup, hptrs, False)
        fence()
        th0 = time.perf_counter()
        run(args.h2d_steps, hptrs, False)
        fence()
        dh = time.perf_counter() - th0
        h_med, h_min = step_stats(th0)
        same = all(np.array_equal(l, timed_lens) for l in all_lens[-args.h2d_steps:])
        extra["value_h2d_inclusive"] = {"value": round(args.h2d_steps * B * H * W / dh / 1e6, 2), "unit": "Mpixel/s", "steps": args.h2d_steps, "warmup": args.h2d_warmup,
                                        "ms_per_step": round(dh / args.h2d_steps * 1e3, 3), "ms_per_step_median": h_med, "ms_per_step_min": h_min,
                                        "same_streams_as_headline": bool(same), "pinned_input_GB": round(B * H * W / 1e9, 2),
                                        "note": "frames in pinned host memory, uploaded inside the timed region (SURVEY 8(d)'s metric); same submission pattern as `value`"}
        extra["pcie_inclusive_Mpixel_per_s"] = extra["value_h2d_inclusive"]["value"]
        if not same:
            bit_exact = False
        del hslab, hrows
    lens = timed_lens

    if rank == 0:
        total_px = float(H) * W * B * world * args.steps
        value = total_px / dt / 1e6
        # the dominant kernel by time per launch, chosen from THIS run's own timers: the all-stage pass when it ran (31
        # stages under the same concurrency as the headline), else the stages timed inside the region
        pool = {k: v for k, v in (all_per_launch or timed_per_launch).items() if v > 0} or {"k_touch_scatter": 0.0}
        dom = max(pool, key=pool.get)
        in_region = dom in timed_per_launch
        launch_ms = timed_per_launch[dom] if in_region else pool[dom]
        alg_bytes = (H * W + float(np.mean(lens))) * imgs_per_launch   # 1 B/px read + L/N B/px written
        achieved = alg_bytes / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
        k_pred_ms = timed_per_launch.get("k_predict") or all_per_launch.get("k_predict") or 0.0
        mode = "-n0 -e1" if args.effort else "-n0 -e0 (QNBLIC)"
        line = {
            "metric": "Mpixel/s encode (bit-exact) 4096x4096 gray -e1 lossless" if args.effort else "Mpixel/s encode (bit-exact) gray -e0 lossless (QNBLIC)",
            "value": round(value, 2), "unit": "Mpixel/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "ms_per_step_median": step_median_ms, "ms_per_step_min": step_min_ms,
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{B} x {H}x{W} SYN-1 gray frames per GPU per step, {mode}, inputs resident in HBM" + (f", steps submitted back to back ({in_flight} in flight)" if (args.overlap_steps and args.effort) else "")
                       if not args.host_inputs else f"{B} x {H}x{W} SYN-1 frames per GPU per step, {mode}, host inputs (PCIe-inclusive)",
                       "frames_per_gpu": B, "images_in_flight": slots, "host_coder_threads_per_gpu": coders, "coded_bin_buffers_in_hbm": host_buffers, "steps_overlapped": bool(args.overlap_steps and args.effort), "steps_in_flight": in_flight if args.effort else 1,
                       "device_coder_pack_threads": dev_packs,
                       "parallelism": f"image-per-GPU x{world}, RCCL gather of streams" if world > 1 else "single GPU",
                       "host": dict(host_description(), cpu_slice=share_how, pinned_host_memory=pinned)},
            "bit_exact": bit_exact, "bit_exact_checks": checks, "gathered_ok": gathered_ok,
            "bits_per_pixel": round(8.0 * float(np.mean(lens)) / (H * W), 4),
        }
        if args.effort:
            line.update({
                "bins_per_pixel": round((bins + (dev_stats["bins"] if dev_stats else 0.0)) / (H * W * B * steps_counted), 3),
                "host_coder_Mbins_per_s_per_thread": round(bins / coder_s / 1e6, 1) if coder_s > 0 else None,
                "device_coder": dev_stats,
                # coded bins cross PCIe as 13-bit groups: 64 bins in thirteen 64-bit words (range_coder.h kGroupWords)
                "d2h_bytes_per_bin": 13.0 / 8.0, "d2h_GB_per_s": round(bins / steps_counted * args.steps * (13.0 / 8.0) / dt / 1e9, 2),
                "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": profiled_traffic(dom, imgs_per_launch),
                             "traffic_source": "the committed rocprofv3 PMC summary (profiles/*_hbm_traffic.csv), rescaled to this run's frames per launch; not measured in this run",
                             "launch_ms": round(launch_ms, 4),
                             "launch_ms_measured": "HIP events inside the timed region" if in_region else "HIP events in the all-stage pass right after the timed region",
                             "launch_ms_note": f"average over {timed_launches} launches while up to {min(args.groups, slots)} groups share the GPU: it includes the time a launch waits for CUs",
                             "launch_ms_alone": round(alone_per_launch[dom], 4) if alone_per_launch.get(dom) else None,
                             "achieved_alone": round((H * W + float(np.mean(lens))) * extra.get("frames_per_launch_alone", 0) / (alone_per_launch[dom] * 1e-3) / 1e9, 3) if alone_per_launch.get(dom) else None,
                             "algorithmic_bytes_per_launch": int(alg_bytes), "images_per_launch": imgs_per_launch,
                             "chosen_from": "all 31 stage timers of this run" if all_per_launch else "the stages timed inside the region"},
                "s1_roofline": {"kernel": "k_predict", "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                                "launch_ms": round(k_pred_ms, 4), "launch_ms_alone": round(alone_per_launch["k_predict"], 4) if alone_per_launch.get("k_predict") else None,
                                "algorithmic_bytes_per_launch": int(5 * H * W * imgs_per_launch),      # 1 B/px read + its 4 B/px record written
                                "achieved": round(5 * H * W * imgs_per_launch / (k_pred_ms * 1e-3) / 1e9, 2) if k_pred_ms else None,
                                "frac": round(5 * H * W * imgs_per_launch / (k_pred_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if k_pred_ms else None,
                                "achieved_alone": round(5 * H * W * extra.get("frames_per_launch_alone", 0) / (alone_per_launch["k_predict"] * 1e-3) / 1e9, 2) if alone_per_launch.get("k_predict") else None},
                "kernel_ms_per_launch": {k: round(v, 4) for k, v in (all_per_launch or timed_per_launch).items()},
                "kernel_ms_per_launch_alone": {k: round(v, 4) for k, v in alone_per_launch.items()} or None,
            })
        line.update(extra)
        if cpu is not None:
            line["cpu_baseline"] = cpu
            line["speedup_vs_cpu_baseline"] = round(value / cpu["value"], 2)
        print(json.dumps(line), flush=True)
        if bit_exact is False:
            print("[bench] a stream differs from the CPU encoder / golden hash / previous step", file=sys.stderr)
            ctx.close()
            sys.exit(3)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
