#!/usr/bin/env python3
"""CPU experiment (round 3, VERDICT r2 lever 4a): how many warm-up bins does a tile need when its warm-up starts from the PREVIOUS
EM round's forward vector at that position instead of the stationary vector?  Dense numpy recursion, 25-round trajectory
(tests/golden/traj_n64.json), 12 positions of a 120 k-bin simulated segment.  Result: DESIGN.md section 3, round 3 table."""
import sys, os, json, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench
from psmc_amd import sim
traj,_ = bench.load_trajectory(os.path.join(ROOT, 'tests', 'golden', 'traj_n64.json'))
a,e,a0 = bench.load_params()
rng = np.random.default_rng(5)
L = 120000
seq = sim.simulate_segment(a,e,a0,L,rng)
def fwd_all(par, x0, lo, hi):
    """normalised forward vectors at positions lo..hi (0-based idx), starting with x0 as X_{lo-1}"""
    A,E,_ = par
    E3 = np.vstack([E[:2], np.ones(64)])
    x = x0.copy(); out = np.empty((hi-lo+1,64))
    for p in range(lo,hi+1):
        x = E3[seq[p]]*(x@A); x /= x.sum(); out[p-lo]=x
    return out
def stationary(par):
    A=par[0]; x=np.full(64,1/64.)
    for _ in range(4000): x=x@A; x/=x.sum()
    return x
pts = list(range(20000, L, 9000))
Wmax = 16000
res=[]
exact=[None]*len(traj)
for r in range(len(traj)):
    # "exact" vectors over the whole sequence from the true start
    A,E,A0 = traj[r]
    x = A0*np.vstack([E[:2],np.ones(64)])[seq[0]]; x/=x.sum()
    exact[r] = np.vstack([x[None], fwd_all(traj[r], x, 1, L-1)])
print("exact done", file=sys.stderr)
def need(par, x0, start, target_pos, ex):
    """steps needed from start so that at all later positions up to target the rel err <= 1e-12: return first step count after which err stays below"""
    out = fwd_all(par, x0, start+1, target_pos)
    err = np.abs(out-ex[start+1:target_pos+1]).max(1)/ex[start+1:target_pos+1].max(1)
    bad = np.nonzero(err>1e-12)[0]
    return 0 if len(bad)==0 else int(bad[-1])+1
for r in range(1,len(traj)):
    st = stationary(traj[r])
    ns=[]; no=[]; dl=[]
    for p in pts:
        s = p-Wmax
        ns.append(need(traj[r], st, s, p, exact[r]))
        no.append(need(traj[r], exact[r-1][s], s, p, exact[r]))
        dl.append(np.abs(exact[r][p]-exact[r-1][p]).max()/exact[r][p].max())
    ns=np.array(ns); no=np.array(no)
    print("round %2d: delta med %.1e max %.1e | stationary start: need med %5d p90 %5d max %5d | prev-round start: med %5d p90 %5d max %5d | ratio med %.2f" % (r, np.median(dl), max(dl), np.median(ns), np.percentile(ns,90), ns.max(), np.median(no), np.percentile(no,90), no.max(), np.median(no/np.maximum(ns,1))))
