#!/bin/bash
# How many copy streams should the coder threads share?  For each count: the bench's headline, and (traced, short) the share of GPU
# kernel time spent in the runtime's blit-kernel copies.  usage: tools/copy_stream_sweep.sh gpurun_out/sweep "1 2 4 8"
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/$1; mkdir -p $OUT
for n in $2; do
  NBLIC_AMD_COPY_STREAMS=$n timeout -k 10 200 python3 $R/bench.py --steps 6 --warmup 1 --no-extra-legs --no-cpu-baseline > $OUT/bench_cs$n.json 2> $OUT/bench_cs$n.err || exit 1
  python3 -c "import json,sys; d=json.loads(open('$OUT/bench_cs$n.json').read().strip().splitlines()[-1]); print('copy streams $n: value', d['value'], 'coder Mbins/s/thread', d['host_coder_Mbins_per_s_per_thread'])"
  (cd /tmp && export TMPDIR=/tmp && NBLIC_AMD_COPY_STREAMS=$n timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_cs$n -o t -- python3 $R/bench.py --steps 1 --warmup 0 --batch 128 --no-extra-legs --no-cpu-baseline > $OUT/trace_cs$n.json 2> $OUT/trace_cs$n.err) || exit 1
  python3 - <<PY
import csv
rows=list(csv.DictReader(open('$OUT/trace_cs$n/t_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows:
    if 'copyBuffer' in r['Name'] or 'interleave' in r['Name']:
        print('   ', r['Name'][:40], r['Calls'], 'calls', round(100*float(r['TotalDurationNs'])/tot,1), '% of kernel time, avg ms', round(float(r['AverageNs'])/1e6,2))
PY
done
