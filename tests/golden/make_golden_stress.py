#!/usr/bin/env python3
"""Real-data-shaped stress fixture: every other large input of the suite is drawn from the model with
missing runs of 10-90 bins; a real .psmcfa carries centromere / assembly-gap runs of 1e4 .. 3e5 `N` bins
(utils/fq2psmcfa.c:114-127 turns every window without enough called bases into `N`), long runs of homozygosity and
het-dense stretches -- exactly where the forgetting length that sizes the fast mode's speculative warm-ups changes.
This generator plants them in a 2.2 M-bin, 6-segment input drawn from the n64 `curve` model:

  seg 0  700 k bins   an N run of 200,000 bins; a run of homozygosity of 50,000 bins (no het, no missing)
  seg 1  500 k        an N run of 20,000; a het-dense stretch (20 % het) of 30,000
  seg 2  400 k        three N runs of 20,000 separated by 500 called bins (a gap-riddled region)
  seg 3  250 k        N runs at both ends (3,000 each: unplaced telomeres)
  seg 4  150 k        as drawn
  seg 5  200,200      100 called bins, 200,000 N, 100 called bins

and dumps, from the REAL reference (oracle/_ref): the .psmc of `psmc -N3 -t15 -r5 -p "4+25*2+4+6"` (exact mode must
reproduce it byte for byte) and the E-step statistics A, E, LL at the parameters of rounds 0, 1, 2 (exact: bit for
bit; fast: within its stated tolerance, at the genome plan's options and at the default plan).

    python tests/golden/make_golden_stress.py        (~2 minutes: the reference runs 4 E-steps over 2.2 M bins)
"""
import gzip
import os
import subprocess
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE); sys.path.insert(0, ROOT)
import make_golden as mg  # noqa: E402
from make_golden import orc  # noqa: E402
from psmc_amd import sim  # noqa: E402

OUT = os.path.join(HERE, "stress")
ARGS = ["-N3", "-t15", "-r5", "-p", "4+25*2+4+6"]


def build_segments():
    g = np.load(os.path.join(HERE, "hmm_params.npz"))
    a, e, a0 = g["n64_curve.a"], g["n64_curve.e"], g["n64_curve.a0"]
    rng = np.random.default_rng(20260927)
    lens = [700_000, 500_000, 400_000, 250_000, 150_000]
    segs = [sim.simulate_segment(a, e, a0, L, rng) for L in lens]
    s = segs[0]; s[150_000:350_000] = 2; s[500_000:550_000] = 0
    s = segs[1]; s[100_000:120_000] = 2; s[300_000:330_000] = (rng.random(30_000) < 0.2).astype(np.uint8)
    s = segs[2]
    for k in range(3): s[200_000 + k * 20_500:200_000 + k * 20_500 + 20_000] = 2
    s = segs[3]; s[:3000] = 2; s[-3000:] = 2
    last = np.concatenate([sim.simulate_segment(a, e, a0, 100, rng, miss_rate=0.0), np.full(200_000, 2, np.uint8),
                           sim.simulate_segment(a, e, a0, 100, rng, miss_rate=0.0)])
    segs.append(last)
    return segs


def main():
    os.makedirs(OUT, exist_ok=True)
    orc.build_oracle(with_ref=True)
    R = orc.Reference()
    segs = build_segments()
    conv = np.frombuffer(b"TKN", dtype=np.uint8)
    fa = os.path.join(OUT, "stress.psmcfa")
    with open(fa, "wb") as fh:
        for i, s in enumerate(segs):
            fh.write((">stress%d\n" % i).encode())
            t = conv[s]
            for j in range(0, len(t), 60):
                fh.write(t[j:j + 60].tobytes() + b"\n")
    txt, err = mg.run_ref(ARGS + ["stress.psmcfa"], OUT)
    subprocess.run(["gzip", "-9", "-n", "-f", fa], check=True)
    with gzip.open(os.path.join(OUT, "stress_N3.psmc.gz"), "wt") as fh:
        fh.write(txt)
    open(os.path.join(OUT, "stress_N3.args"), "w").write(" ".join(ARGS + ["stress.psmcfa.gz"]) + "\n")
    # the parameters of rounds 0..2 from the PA lines (9 decimals, what a restart would read) -> E-step statistics of the reference
    out = {}
    pas = [l.split("\t")[1].split() for l in txt.splitlines() if l.startswith("PA")]
    for rd in range(3):
        par = np.array([float(x) for x in pas[rd][1:]])
        hp = R.hmm_params("4+25*2+4+6", par)
        r = R.estep(hp["a"], hp["e"], hp["a0"], segs, per_seg=True)
        k = "rd%d" % rd
        out[k + ".params"] = par
        for f in ("a", "e", "a0"):
            out["%s.%s" % (k, f)] = np.asarray(hp[f])
        out[k + ".A"] = r["A"]; out[k + ".E"] = r["E"]; out[k + ".LL"] = np.array(r["LL"]); out[k + ".seg_LL"] = r["seg_LL"]
        out[k + ".seg_chk"] = r["seg_chk"]
    np.savez_compressed(os.path.join(OUT, "stress_estep.npz"), **out)
    print("stress fixtures:", {f: os.path.getsize(os.path.join(OUT, f)) for f in sorted(os.listdir(OUT))})


if __name__ == "__main__":
    main()
