#!/bin/bash
# round 3, GPU call D: kernel timelines of the one-round plan (500 k, 3.75 M share, 7.5 M share)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
R=$PWD
cd /tmp
for w in "--shares 8 --chr 0" "--shares '' --chr 500000" "--shares 4 --chr 0"; do
  tag=$(echo $w | tr -dc 0-9 | cut -c1-3)
  rm -rf $R/gpurun_out/prof/d_tl$tag*
  eval timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof -o d_tl$tag -- python $R/scripts/shard_sweep.py $w --steps 4 --warmup 6 ${SWEEP_ARGS:-} > $R/gpurun_out/d_tl$tag.log 2>&1
  echo "rocprof [$w] rc=$?"
  (cd $R && python scripts/prof_timeline.py $(ls gpurun_out/prof/d_tl$tag*.db | tail -1) k_reduce2 all > gpurun_out/d_timeline_$tag.txt 2>&1; cat gpurun_out/d_timeline_$tag.txt | cut -c1-120)
done
