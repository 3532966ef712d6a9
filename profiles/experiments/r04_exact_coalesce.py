#!/usr/bin/env python3
"""Driver of profiles/experiments/r04_exact_coalesce.c: 25 EM rounds x 120 tile starts, CPU only.

    python profiles/experiments/r04_exact_coalesce.py [out.txt]
"""
import json, os, struct, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from psmc_amd import hostlib, sim

def main():
    tj = json.load(open(os.path.join(ROOT, "tests", "golden", "traj_n64.json")))
    rounds = [hostlib.hmm_params(tj["pattern"], r["params"]) for r in tj["rounds"] if r["round"] >= 1][:25]
    a, e, a0 = rounds[0]
    n, L, W, H, NS = a.shape[0], 600_000, 3072, 8192, 120
    seq = sim.simulate_segment(a, e, a0, L, np.random.default_rng(11)).astype(np.uint8)
    starts = np.linspace(W + 1000, L - H - 1000, NS).astype(np.int32)
    path = "/tmp/exact_coalesce.bin"
    with open(path, "wb") as f:
        f.write(struct.pack("6i", n, L, len(rounds), NS, W, H))
        f.write(seq.tobytes())
        for a, e, a0 in rounds:
            f.write(np.ascontiguousarray(a, np.float64).tobytes()); f.write(np.ascontiguousarray(e, np.float64).tobytes()); f.write(np.ascontiguousarray(a0, np.float64).tobytes())
        f.write(starts.tobytes())
    exe = "/tmp/exact_coalesce"
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "scripts", "r04", "exact_coalesce.c")], check=True)
    out = subprocess.run([exe, path], check=True, capture_output=True, text=True).stdout
    print(out)
    if len(sys.argv) > 1: open(sys.argv[1], "w").write(out)

if __name__ == "__main__":
    main()
