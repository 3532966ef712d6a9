#!/bin/bash
# wall-clock per EM iteration of the psmc binary on a genome-sized synthetic input (exact / fast, thread counts)
set -u
R=$GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
python - <<'PY'
import numpy as np, os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from psmc_amd import sim
g = np.load("tests/golden/hmm_params.npz")
a, e, a0 = g["n64_curve.a"], g["n64_curve.e"], g["n64_curve.a0"]
lens = sim.human_like_lengths(30_000_000, n_seg=90)
segs = sim.simulate_genome(a, e, a0, lens, seed=43)
conv = np.frombuffer(b"TKN", dtype=np.uint8)
with open("/tmp/genome.psmcfa", "wb") as fh:
    for i, s in enumerate(segs):
        fh.write(b">chr%d\n" % i)
        t = conv[s]
        pad = (-len(t)) % 60
        body = np.concatenate([t, np.full(pad, 10, np.uint8)]).reshape(-1, 60)
        rows = np.concatenate([body, np.full((body.shape[0], 1), 10, np.uint8)], axis=1).ravel()
        fh.write(rows.tobytes().rstrip(b"\n") + b"\n")
print("wrote", os.path.getsize("/tmp/genome.psmcfa") / 1e6, "MB")
PY
for cfg in "exact 1" "exact 4" "exact 8" "fast 1"; do
  set -- $cfg
  echo "== PSMC_HIP_MODE=$1 PSMC_THREADS=$2"
  ( time PSMC_HIP_MODE=$1 PSMC_THREADS=$2 PSMC_TIMING=1 psmc_amd/host/psmc -N3 -t15 -r5 -p "4+25*2+4+6" -o /tmp/out_$1_$2.psmc /tmp/genome.psmcfa ) 2>&1 | grep -v "^$" | tail -8
done
cmp /tmp/out_exact_1.psmc /tmp/out_exact_4.psmc && cmp /tmp/out_exact_1.psmc /tmp/out_exact_8.psmc && echo "exact outputs identical across thread counts"
grep "^LK" /tmp/out_exact_1.psmc | head -4; grep "^LK" /tmp/out_fast_1.psmc | head -4
