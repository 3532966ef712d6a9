#!/bin/bash
# round 3, GPU call G: split-C fused back half (count_waves 1/2/4): parity, then the sweep over shard sizes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider > gpurun_out/g_pytest.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/g_pytest.log | cut -c1-250; grep -n "^E  \|^FAILED" gpurun_out/g_pytest.log | head -12 | cut -c1-250
show() { python - "$1" <<'PY'
import json, sys
for r in json.load(open(sys.argv[1])):
    if "error" in r: print(r["workload"], r["cfg"], "ERROR", r["error"][:100]); continue
    k = r["kernels_ms"]
    print("%-16s %-40s %7.3f ms (min %6.3f) tiles %5d x %5d rep %s  tot %.2f fwd %.2f bwd %.2f cnt %.2f" % (r["workload"], r["cfg"], r["ms_median"], r["ms_min"], r["tiles"], r["tile_len"], r["repairs"], k["total"], k["fwd_sweep"], k["bwd_sweep"], k["expect"]))
PY
}
timeout 900 python scripts/shard_sweep.py --cfg "" --cfg "count_waves=1" --cfg "count_waves=2" --cfg "count_waves=4" --shares 8,4,2 --out gpurun_out/g_sweep.json > gpurun_out/g_sweep.log 2> gpurun_out/g_sweep.err
echo "sweep rc=$?"; tail -3 gpurun_out/g_sweep.err | cut -c1-300; show gpurun_out/g_sweep.json
timeout 900 python scripts/shard_sweep.py --cfg "" --cfg "struct_tiles=2048 count_waves=2 two_phase=0" --cfg "struct_tiles=4096 count_waves=2 two_phase=0" --cfg "struct_tiles=4096 count_waves=2 two_phase=2" --cfg "struct_tiles=2048 count_waves=4 two_phase=2" \
   --shares 1 --chr 0 --out gpurun_out/g_sweep_full.json > gpurun_out/g_sweep_full.log 2> gpurun_out/g_sweep_full.err
echo "sweep full rc=$?"; tail -3 gpurun_out/g_sweep_full.err | cut -c1-300; show gpurun_out/g_sweep_full.json
