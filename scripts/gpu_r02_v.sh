#!/bin/bash
# round 2, GPU call V: fused back half for 128 states (fuse128) -- parity under varying poison, A/B of config 5; bench.py's N>1 code path on one GPU
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
PSMC_HIP_POISON=vary timeout 600 python -m pytest tests/test_gpu_estep.py -m gpu -q --maxfail=30 -k "n128" > gpurun_out/v_pytest.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/v_pytest.log | cut -c1-250; grep -n "^E  " gpurun_out/v_pytest.log | head -8 | cut -c1-250
for cfg in "fuse128=1" "fuse128=0"; do
  tag=$(echo $cfg | tr ' =' '__')
  opts=""; for kv in $cfg; do opts="$opts --opt $kv"; done
  timeout 300 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --exact-extra 0 --n128-extra 1 $opts > gpurun_out/v_bench_$tag.json 2> gpurun_out/v_bench_$tag.err
  echo "bench [$cfg] rc=$?"
  python - <<PY
import json
r=json.load(open("gpurun_out/v_bench_$tag.json"))["n128"]
print("   n128 full %.2f ms  kernels %s" % (r["ms_per_step"], {k: round(v, 2) for k, v in r["kernels_ms"].items()}))
PY
done
BENCH_SINGLE_GPU_TEST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --bins 6000000 --cpu-sample 0 --exact-extra 0 --n128-extra 0 > gpurun_out/v_bench_2rank.json 2> gpurun_out/v_bench_2rank.err
echo "2-rank bench rc=$?"; cut -c1-700 gpurun_out/v_bench_2rank.json; tail -3 gpurun_out/v_bench_2rank.err | cut -c1-300
