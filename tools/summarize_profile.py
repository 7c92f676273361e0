#!/usr/bin/env python3
"""Condenses a gpurun_out/prof_* directory (rocprofv3 --kernel-trace --stats run plus separate
--pmc FETCH_SIZE / --pmc WRITE_SIZE runs of bench.py) into the small files kept under profiles/.

    python tools/summarize_profile.py gpurun_out/prof_r1 profiles/r01
"""
import collections
import csv
import glob
import os
import shutil
import sqlite3
import sys


def pmc(dirname, counter):
    """kernel -> (launches, mean counter value) from a --pmc pass: CSV output of older rocprofv3
    builds, or the rocpd sqlite database (counters_collection view) of newer ones."""
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(dirname, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            a = agg[r["Kernel_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    for f in glob.glob(os.path.join(dirname, "**", "*_results.db"), recursive=True):
        db = sqlite3.connect(f)
        for name, n, total in db.execute("select kernel_name, count(*), sum(value) from counters_collection where counter_name = ? group by kernel_name", (counter,)):
            a = agg[name]
            a[0] += n
            a[1] += total
    return {k: (n, v / n) for k, (n, v) in agg.items()}


def kernel_stats_from_db(src, dst):
    """The --stats table (per-kernel calls / total / average / min / max duration) from the rocpd database."""
    files = glob.glob(os.path.join(src, "trace", "**", "*_results.db"), recursive=True)
    if not files:
        return False
    db = sqlite3.connect(files[0])
    rows = list(db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc"))
    whole = sum(r[2] for r in rows) or 1
    with open(dst + "_kernel_stats.csv", "w") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for name, calls, total, avg, lo, hi in rows:
            w.writerow([name, calls, int(total), round(avg, 3), round(100.0 * total / whole, 2), int(lo), int(hi)])
    return True


def main(src, dst):
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    stats = glob.glob(os.path.join(src, "trace", "**", "*_kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], dst + "_kernel_stats.csv")
    else:
        kernel_stats_from_db(src, dst)
    for name in ("bench_plain.json", "bench_trace.json"):
        p = os.path.join(src, name)
        if os.path.exists(p):
            shutil.copy(p, dst + "_" + name)
    fetch, write = pmc(os.path.join(src, "pmc_fetch"), "FETCH_SIZE"), pmc(os.path.join(src, "pmc_write"), "WRITE_SIZE")
    with open(dst + "_hbm_traffic.csv", "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches", "FETCH_SIZE_KB_per_launch_raw", "WRITE_SIZE_KB_per_launch", "hbm_MB_per_launch_corrected(2*fetch+write)"])
        for k in sorted(set(fetch) | set(write), key=lambda k: -(2 * fetch.get(k, (0, 0))[1] + write.get(k, (0, 0))[1])):
            fn, fv = fetch.get(k, (0, 0.0))
            wn, wv = write.get(k, (0, 0.0))
            w.writerow([k, max(fn, wn), round(fv, 1), round(wv, 1), round((2 * fv + wv) / 1024, 1)])
    print("wrote", dst + "_*")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
