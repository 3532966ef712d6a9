#!/bin/bash
# round 3, GPU call P: bench.py's N>1 code path on one GPU (2 ranks, gloo; group engine over devices 0,0) and the --engine group line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
BENCH_SINGLE_GPU_TEST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 4 --warmup 2 --bins 6000000 --segments 40 --cpu-sample 0 --exact-extra 0 --n128-extra 0 > gpurun_out/p_bench_2rank.json 2> gpurun_out/p_bench_2rank.err
echo "2-rank bench rc=$?"; tail -3 gpurun_out/p_bench_2rank.err | cut -c1-300
python - <<'PY'
import json
r = json.loads(open("gpurun_out/p_bench_2rank.json").read().strip().splitlines()[-1])
print({k: r[k] for k in ("value", "n_gpus", "ms_per_step", "scaling")}); print("strong", r.get("strong_scaling")); print("group", json.dumps(r.get("group_engine"))[:1500])
PY
timeout 600 python bench.py --engine group --gpus 2 --group-devices 0,0 --scaling weak --steps 6 --warmup 3 --bins 6000000 --segments 40 --cpu-sample 0 > gpurun_out/p_bench_group00.json 2> gpurun_out/p_bench_group00.err; echo "group 0,0 weak rc=$?"; cut -c1-1200 gpurun_out/p_bench_group00.json
