#!/bin/bash
# Collects what profiles/ keeps for a round, on the GPU box:
#   trace pass   rocprofv3 --kernel-trace --stats of the default bench shape (2 steps)  -> per-kernel durations
#   pmc passes   rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate runs, kernel trace only) of a small batch
# usage: tools/profile_round.sh gpurun_out/prof_r2a      (then: python tools/summarize_profile.py gpurun_out/prof_r2a profiles/r02a)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $R/bench.py --steps 2 --warmup 1 --no-extra-legs > $OUT/bench_trace.json 2> $OUT/bench_trace.err || exit 1
echo "trace pass done"
SMALL="--batch 32 --groups 4 --slots 32 --steps 1 --warmup 0 --no-cpu-baseline --no-extra-legs --no-overlap-steps"
timeout -k 10 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python3 $R/bench.py $SMALL > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err || exit 1
echo "fetch pass done"
timeout -k 10 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python3 $R/bench.py $SMALL > $OUT/pmc_write.json 2> $OUT/pmc_write.err || exit 1
echo "write pass done"
