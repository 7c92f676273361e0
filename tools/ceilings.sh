#!/bin/bash
# The slices of the headline, measured at the current code: the device side alone (bins produced,
# neither copied nor coded), everything but the coding (bins reach the host), and the plain line.  usage: tools/ceilings.sh OUTDIR
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/$1
mkdir -p $OUT
COMMON="--steps 8 --warmup 2 --no-extra-legs --no-cpu-baseline"
NBLIC_AMD_DBG=16 python3 $R/bench.py $COMMON > $OUT/device_side_only.json 2> $OUT/device_side_only.err
NBLIC_AMD_DBG=128 python3 $R/bench.py $COMMON > $OUT/no_coding.json 2> $OUT/no_coding.err
python3 $R/bench.py $COMMON > $OUT/plain8.json 2> $OUT/plain8.err
python3 - <<PY
import json
for n in ("device_side_only", "no_coding", "plain8"):
    try:
        d = json.load(open("$OUT/" + n + ".json"))
        print(n, d["value"], "Mpixel/s", "bit_exact", d.get("bit_exact"))
    except Exception as e:
        print(n, "failed:", e)
PY
