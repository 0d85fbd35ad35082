#!/bin/bash
# block pipeline through bench.py: single-device line, the sharded per-rank program at one rank (rehearsal), and the same
# under torch.distributed.run with a world of one (RCCL initialised)
set -e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --mode blocks --steps 10 --no-cpu-baseline --no-extras > gpurun_out/bb_single.json 2> gpurun_out/bb_single.err || (tail -20 gpurun_out/bb_single.err; exit 1)
timeout -k 10 300 python bench.py --mode blocks --steps 10 --no-cpu-baseline --no-extras --force-sharded > gpurun_out/bb_sharded1.json 2> gpurun_out/bb_sharded1.err || (tail -20 gpurun_out/bb_sharded1.err; exit 1)
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29771 bench.py --gpus 1 --mode blocks --steps 10 --no-cpu-baseline --no-extras --force-sharded > gpurun_out/bb_rccl1.json 2> gpurun_out/bb_rccl1.err || (tail -20 gpurun_out/bb_rccl1.err; exit 1)
python - <<'PY'
import json
for f in ("bb_single","bb_sharded1","bb_rccl1"):
    d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
    print(f, "ms/step %.3f"%d["ms_per_step"], "single_same %.3f"%d.get("single_gpu_same_workload_ms",0), "speedup %.2f"%d.get("speedup_vs_single_gpu_same_workload",0), d.get("labels_identical_to_single_gpu_run"), d["config"]["clusters"])
PY
