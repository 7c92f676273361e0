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
    # gfx950's FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 bytes (MI355X_MICROARCH.md,
    # HBM section): doubled for kernels whose reads are coalesced wave instructions of >= 128 bytes; kernels whose
    # fetches are dominated by byte loads or by random 2-byte gathers (one 64-byte request per lane) read true at x1.
    narrow = ("k_mix", "k_map_count", "k_count_bins", "k_predict_border", "k_q_symbols", "k_q_predict", "k_serial_")
    total_raw = total_corr = 0.0
    with open(dst + "_hbm_traffic.csv", "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches", "FETCH_SIZE_KB_per_launch_raw", "WRITE_SIZE_KB_per_launch", "fetch_factor", "hbm_MB_per_launch_corrected(factor*fetch+write)"])
        def corrected(k):
            fac = 1 if any(t in k for t in narrow) else 2
            return fac, fac * fetch.get(k, (0, 0.0))[1] + write.get(k, (0, 0.0))[1]
        for k in sorted(set(fetch) | set(write), key=lambda k: -corrected(k)[1]):
            fn, fv = fetch.get(k, (0, 0.0))
            wn, wv = write.get(k, (0, 0.0))
            fac, corr = corrected(k)
            w.writerow([k, max(fn, wn), round(fv, 1), round(wv, 1), fac, round(corr / 1024, 1)])
            n = max(fn, wn)
            total_raw += (fv + wv) * n
            total_corr += corr * n
        w.writerow(["TOTAL_KB_all_launches", "", round(total_raw, 1), "", "", round(total_corr / 1024, 1)])
    print("wrote", dst + "_*", "total raw KB", round(total_raw), "corrected MB", round(total_corr / 1024))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
