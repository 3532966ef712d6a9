#!/bin/bash
# round 3, GPU call C: one-round plan + merged phase 1: parity suite, then the sweep over shard sizes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --no-header -p no:cacheprovider > gpurun_out/c_pytest.log 2>&1
echo "pytest rc=$?"; tail -6 gpurun_out/c_pytest.log | cut -c1-250; grep -n "^E  " gpurun_out/c_pytest.log | head -8 | cut -c1-250
OLD="struct_tiles=8192 two_phase=2 merge1=0 warm_shift=1 kc_sub=4"
timeout 900 python scripts/shard_sweep.py --cfg "" --cfg "$OLD" --cfg "merge1=0" --cfg "warm_shift=1" --cfg "walk_impl=0" --cfg "kc_sub=4" --cfg "two_phase=2" --cfg "two_phase=2 merge1=0" \
   --shares 8,4,2 --out gpurun_out/c_sweep.json > gpurun_out/c_sweep.log 2> gpurun_out/c_sweep.err
echo "sweep rc=$?"; tail -3 gpurun_out/c_sweep.err | cut -c1-300
timeout 900 python scripts/shard_sweep.py --cfg "" --cfg "merge1=1" --cfg "chunk=7424 two_phase=0" --cfg "chunk=7424 two_phase=0 merge1=1" --cfg "chunk=5568 two_phase=0 merge1=1" --cfg "struct_tiles=4096 two_phase=0 merge1=1 warm_shift=0" \
   --shares 1 --chr 0 --out gpurun_out/c_sweep_full.json > gpurun_out/c_sweep_full.log 2> gpurun_out/c_sweep_full.err
echo "sweep full rc=$?"; tail -3 gpurun_out/c_sweep_full.err | cut -c1-300
python - <<'PY'
import json
for f in ("gpurun_out/c_sweep.json", "gpurun_out/c_sweep_full.json"):
  for r in json.load(open(f)):
    if "error" in r: print(r["workload"], r["cfg"], "ERROR", r["error"][:100]); continue
    k = r["kernels_ms"]
    print("%-16s %-62s %7.3f ms (min %6.3f) tiles %5d x %5d rep %s  tot %.2f fwd %.2f bwd %.2f cnt %.2f" % (r["workload"], r["cfg"], r["ms_median"], r["ms_min"], r["tiles"], r["tile_len"], r["repairs"], k["total"], k["fwd_sweep"], k["bwd_sweep"], k["expect"]))
PY
