#!/bin/bash
# A/B of an environment switch: tools/gpu/ab.sh VAR "0 1" [bench args]
set -e
mkdir -p gpurun_out
VAR=$1; VALS=$2; shift 2
for v in $VALS; do
  env $VAR=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 10 "$@" > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_$v.json")); print("$VAR=$v", round(d["ms_per_step"],4), d["phase_ms"])
PY
done
