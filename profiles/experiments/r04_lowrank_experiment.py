#!/usr/bin/env python3
"""How many directions survive a speculative warm-up?  (CPU, numpy; round 4.)

A tile's speculation fails where the forward operator over the W bins before it, K_W = prod_p diag(e[o_p]) a^T, is not rank 1
to the tolerance: the start vector still matters.  The run machinery then pays a full 64-column transfer matrix per tile
(~21x a plain sweep of the tile).  If K_W is numerically rank q with q = 2..4 there, q warm-started candidate vectors span
every possible start vector and a run tile needs q sweeps instead of 64 columns.  This prints, over all tile boundaries of a
simulated chromosome, the distribution of the numerical rank of K_W at 1e-13 / 1e-12.

    python profiles/experiments/r04_lowrank_experiment.py [bins=500000] [tile=1856] [W=3072] [round=10]
"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from psmc_amd import hostlib, sim

def main():
    bins = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 1856
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 3072
    rnd = int(sys.argv[4]) if len(sys.argv) > 4 else 10
    tj = json.load(open(os.path.join(ROOT, "tests", "golden", "traj_n64.json")))
    rounds = [hostlib.hmm_params(tj["pattern"], r["params"]) for r in tj["rounds"] if r["round"] >= 1]
    a0_, e0_, p0_ = rounds[0]
    seq = sim.simulate_segment(a0_, e0_, p0_, bins, np.random.default_rng(7))
    a, e, a0 = rounds[rnd]
    n = a.shape[0]
    M = [np.diag(e[s]) @ a.T for s in range(3)]          # x_p = M[o_p] x_{p-1}
    for dirn in ("forward", "backward"):
        Mb = [a @ np.diag(e[s]) for s in range(3)]        # backward: z_p = a diag(e[o_{p+1}]) z_{p+1} (same spectrum question)
        ranks12, ranks13, s2 = [], [], []
        for lo in range(T + 1, bins - W - T, T):
            K = np.eye(n)
            rng_ = range(lo - W, lo) if dirn == "forward" else range(lo + W, lo, -1)
            for i, p in enumerate(rng_):
                K = (M if dirn == "forward" else Mb)[seq[p]] @ K
                if i % 64 == 63: K /= K.sum()
            sv = np.linalg.svd(K, compute_uv=False)
            sv /= sv[0]
            ranks12.append(int((sv > 1e-12).sum())); ranks13.append(int((sv > 1e-13).sum())); s2.append(sv[1])
        r12, r13 = np.array(ranks12), np.array(ranks13)
        print("%s: %d boundaries, tile %d, warm-up %d, round %d" % (dirn, len(r12), T, W, rnd))
        for name, r in (("1e-12", r12), ("1e-13", r13)):
            print("   numerical rank of K_W at %s: " % name + "  ".join("%d: %d" % (q, (r == q).sum()) for q in range(1, 9)) + "  >8: %d" % (r > 8).sum())
        print("   boundaries that fail plain speculation (rank > 1 at 1e-12): %d (%.1f %%); of those rank 2: %d, <= 3: %d, <= 4: %d" % (
            (r12 > 1).sum(), 100.0 * (r12 > 1).mean(), (r12 == 2).sum(), ((r12 > 1) & (r12 <= 3)).sum(), ((r12 > 1) & (r12 <= 4)).sum()))

if __name__ == "__main__":
    main()
