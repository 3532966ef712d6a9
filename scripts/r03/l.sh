#!/bin/bash
# round 3, GPU call L: forward sweep split into the two lists' launches (split_fwd) A/B, interleaved repeats; parity of the fast tests; timeline
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
R=$PWD
show() { python - "$1" <<'PY'
import json, sys, collections
agg = collections.defaultdict(list)
for r in json.load(open(sys.argv[1])):
    if "error" in r: print(r["workload"], r["cfg"], "ERROR", r["error"][:100]); continue
    k = r["kernels_ms"]
    agg[(r["workload"], r["cfg"])].append(r["ms_median"])
    print("%-16s %-26s %7.3f ms (min %6.3f) tiles %5d x %5d rep %s  tot %.2f fwd %.2f cnt %.2f" % (r["workload"], r["cfg"], r["ms_median"], r["ms_min"], r["tiles"], r["tile_len"], r["repairs"], k["total"], k["fwd_sweep"], k["expect"]))
for k, v in agg.items(): print("==", k, "median of repeats %.3f  all %s" % (sorted(v)[len(v) // 2], [round(x, 2) for x in v]))
PY
}
timeout 900 python scripts/shard_sweep.py --cfg "" --cfg "split_fwd=0" --repeat 3 --shares 1 --chr 0 --warmup 10 --steps 14 --out gpurun_out/l_sweep.json > gpurun_out/l_sweep.log 2> gpurun_out/l_sweep.err
echo "sweep rc=$?"; tail -3 gpurun_out/l_sweep.err | cut -c1-300; show gpurun_out/l_sweep.json
timeout 1200 python -m pytest tests/test_gpu_estep.py tests/test_gpu_scale.py -m gpu -q --no-header -p no:cacheprovider -k "fast or config3 or shard" > gpurun_out/l_pytest.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/l_pytest.log | cut -c1-250; grep -n "^E  \|^FAILED" gpurun_out/l_pytest.log | head -12 | cut -c1-250
cd /tmp; rm -rf $R/gpurun_out/prof/l_tl*
timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof -o l_tl -- python $R/scripts/shard_sweep.py --shares 1 --chr 0 --steps 4 --warmup 8 > $R/gpurun_out/l_tl.log 2>&1; echo "rocprof rc=$?"
cd $R; python scripts/prof_timeline.py $(ls gpurun_out/prof/l_tl*.db | tail -1) > gpurun_out/l_timeline.txt 2>&1; cat gpurun_out/l_timeline.txt | cut -c1-120
