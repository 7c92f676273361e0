"""CPU suite: the N > 1 path (one process per GPU, a single rooted gather of the finished
streams) exercised with world_size 2 over gloo.  The same function runs over RCCL in bench.py."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _streams_for(rank, n):
    rng = np.random.default_rng(100 + rank)
    return [rng.integers(0, 256, int(rng.integers(0, 5000)), dtype=np.uint8).tobytes() for _ in range(n)]


def _worker(rank, world, port, counts, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        gather = importlib.import_module("nblic-image-compression_amd.gather")
        mine = _streams_for(rank, counts[rank])
        got = gather.gather_streams(mine, torch.device("cpu"))
        if rank == 0:
            ok = got is not None and len(got) == world
            for r in range(world):
                ok = ok and got[r] == _streams_for(r, counts[r])
            q.put(bool(ok))
        else:
            assert got is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("counts", [(3, 3), (4, 1), (2, 0)])
def test_gather_streams_world2_gloo(counts):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, counts, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=10) is True


def test_pack_roundtrip():
    gather = importlib.import_module("nblic-image-compression_amd.gather")
    payload, lens = gather.pack([b"abc", b"", b"defgh"])
    assert payload.tobytes() == b"abcdefgh" and lens.tolist() == [3, 0, 5]


def test_bench_gpus_flag_launches_ranks():
    """`python bench.py --gpus 2` started directly must itself become two ranks (one per GPU): here with
    --launch-check, which joins a gloo group and touches no GPU; rank 0's line reports what it saw."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line == {"launch_check": True, "n_gpus": 2, "ranks_seen": 2, "local_rank": 0}


def test_bench_refuses_mislabelled_world():
    import subprocess
    env = dict(os.environ, WORLD_SIZE="3", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def _fake_sysfs(root, nps=2):
    """A two-socket, 128-core (256-thread), 8-GPU host: GPUs 0-3 on NUMA node 0, 4-7 on node 1; plus one render node
    that is not an AMD GPU."""
    import os
    for c in range(256):
        d = os.path.join(root, "devices", "system", "cpu", f"cpu{c}", "topology")
        os.makedirs(d)
        with open(os.path.join(d, "thread_siblings_list"), "w") as f:
            f.write(f"{c % 128},{c % 128 + 128}\n")
    for g in range(8):
        d = os.path.join(root, "class", "drm", f"renderD{128 + g}", "device")
        os.makedirs(d)
        node = g // 4
        with open(os.path.join(d, "vendor"), "w") as f: f.write("0x1002\n")
        with open(os.path.join(d, "numa_node"), "w") as f: f.write(f"{node}\n")
        with open(os.path.join(d, "local_cpulist"), "w") as f: f.write(f"{64 * node}-{64 * node + 63},{128 + 64 * node}-{128 + 64 * node + 63}\n")
    d = os.path.join(root, "class", "drm", "renderD136", "device")
    os.makedirs(d)
    with open(os.path.join(d, "vendor"), "w") as f: f.write("0x1a03\n")
    with open(os.path.join(d, "local_cpulist"), "w") as f: f.write("0-255\n")


def test_cpu_slices_follow_the_gpus_numa_nodes(tmp_path):
    """Each rank's CPU slice comes from ITS GPU's NUMA node (sysfs local_cpulist), the node's cores split between the
    GPUs on it; a lone visible GPU still takes only the share it would have in the full node; without sysfs the old
    even split by rank."""
    sys.path.insert(0, ROOT)
    import bench
    _fake_sysfs(str(tmp_path))
    seen = set()
    for r in range(8):
        cpus, how = bench.cpu_share(r, 8, str(tmp_path), range(256))
        node, idx = r // 4, r % 4
        first = 64 * node + 16 * idx
        assert cpus == list(range(first, first + 16)) + list(range(first + 128, first + 144)), (r, how)
        assert f"numa node {node}" in how
        assert not (seen & set(cpus))
        seen |= set(cpus)
    assert len(seen) == 256
    os.environ["HIP_VISIBLE_DEVICES"] = "5"                    # a one-GPU container on that host: GPU 5 is ordinal 0
    try:
        cpus, how = bench.cpu_share(0, 8, str(tmp_path), range(256))
    finally:
        del os.environ["HIP_VISIBLE_DEVICES"]
    assert cpus == list(range(64, 80)) + list(range(192, 208)) and "numa node 1" in how
    empty = tmp_path / "nothing"
    (empty / "class" / "drm").mkdir(parents=True)
    for c in range(16):
        d = empty / "devices" / "system" / "cpu" / f"cpu{c}" / "topology"
        d.mkdir(parents=True)
        (d / "thread_siblings_list").write_text(f"{c % 8},{c % 8 + 8}\n")
    cpus, how = bench.cpu_share(1, 4, str(empty), range(16))
    assert cpus == [2, 3, 10, 11] and "even split" in how
    assert bench.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]


def test_bench_eight_ranks_report_distinct_numa_correct_slices(tmp_path):
    """`bench.py --gpus 8 --launch-check` on the fake topology: eight ranks, eight distinct slices, each inside its
    GPU's node (no GPU is touched; the ranks only join a gloo group)."""
    import json
    import subprocess
    _fake_sysfs(str(tmp_path))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(NBLIC_BENCH_SYSFS=str(tmp_path), NBLIC_BENCH_ASSUME_CPUS="256")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--launch-check"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["ranks_seen"] == 8 and len(line["cpu_slices"]) == 8
    used = set()
    for s in sorted(line["cpu_slices"], key=lambda e: e["rank"]):
        node = s["rank"] // 4
        assert len(s["cpus"]) == 32 and all((c % 128) // 64 == node for c in s["cpus"]), s
        assert not (used & set(s["cpus"]))
        used |= set(s["cpus"])
