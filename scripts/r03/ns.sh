#!/bin/bash
# round 3, GPU call NS: the north-star target on one GPU (psmc -N25 + 100 bootstraps at -N25, exact and fast) + the -p 64*2 trajectory
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python scripts/northstar.py gpurun_out/r03_northstar.json gpurun_out/traj_n128.json > gpurun_out/ns.log 2> gpurun_out/ns.err
echo "northstar rc=$?"; tail -5 gpurun_out/ns.err | cut -c1-300; tail -40 gpurun_out/ns.log | cut -c1-300
