#!/bin/bash
# round 3, GPU call A: where the current plan stands on shard-sized inputs (baseline for the new planner)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
R=$PWD
CFGS=()
for ch in 0 256 512 1024 2048; do for tp in 2 0; do for ws in 1 0; do
  c="two_phase=$tp warm_shift=$ws"; [ $ch != 0 ] && c="chunk=$ch $c"
  CFGS+=(--cfg "$c")
done; done; done
timeout 900 python scripts/shard_sweep.py "${CFGS[@]}" --shares 8,4,2 --out gpurun_out/a_sweep.json > gpurun_out/a_sweep.log 2> gpurun_out/a_sweep.err
echo "sweep rc=$?"; tail -3 gpurun_out/a_sweep.err | cut -c1-300
python - <<'PY'
import json
for r in json.load(open("gpurun_out/a_sweep.json")):
    if "error" in r: print(r["workload"], r["cfg"], "ERROR", r["error"][:100]); continue
    k = r["kernels_ms"]
    print("%-16s %-40s %7.3f ms (min %6.3f) tiles %5d x %5d rep %s  tot %.2f fwd %.2f bwd %.2f cnt %.2f" % (r["workload"], r["cfg"], r["ms_median"], r["ms_min"], r["tiles"], r["tile_len"], r["repairs"], k["total"], k["fwd_sweep"], k["bwd_sweep"], k["expect"]))
PY
cd /tmp
for w in "--shares 8 --chr 0" "--shares '' --chr 500000"; do
  tag=$(echo $w | tr -dc 0-9 | cut -c1-3)
  rm -rf $R/gpurun_out/prof/a_tl$tag*
  eval timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof -o a_tl$tag -- python $R/scripts/shard_sweep.py $w --steps 4 --warmup 6 > $R/gpurun_out/a_tl$tag.log 2>&1
  echo "rocprof [$w] rc=$?"
  (cd $R && python scripts/prof_timeline.py $(ls gpurun_out/prof/a_tl$tag*.db | tail -1) k_reduce2 all > gpurun_out/a_timeline_$tag.txt 2>&1; cat gpurun_out/a_timeline_$tag.txt | cut -c1-120)
done
