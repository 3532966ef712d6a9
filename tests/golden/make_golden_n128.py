#!/usr/bin/env python3
"""Golden vectors for the 128-state configuration (`-p "64*2"`, README of the
reference; BASELINE.json configs[5]) from the REAL reference (oracle/_ref, built
from the unmodified sources).  Separate from make_golden.py so that the other
fixtures stay byte-stable.  Outputs: estep_n128.npz and cli/small_n128_N2.*.

    python tests/golden/make_golden_n128.py
"""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (helpers only: orc, bottleneck_lambdas, run_ref)
from make_golden import orc  # noqa: E402


def main():
    orc.build_oracle(with_ref=True)
    R = orc.Reference()
    segs_npz = np.load(os.path.join(HERE, "segments_small.npz"))
    segs = [segs_npz[k] for k in sorted(segs_npz.files)]
    n, nf, pm = R.parse_pattern(mg.PAT128)
    assert n + 1 == 128 and nf == 64
    out = {}
    for nm, par in (("curve", np.concatenate([[0.0625, 0.0131, 15.0], mg.bottleneck_lambdas(nf)])),):
        hp = R.hmm_params(mg.PAT128, par)
        k = "n128_" + nm
        for f in ("a", "e", "a0"):
            out["%s.%s" % (k, f)] = np.asarray(hp[f])
        out["%s.params" % k] = par
        r = R.estep(hp["a"], hp["e"], hp["a0"], segs, per_seg=True)
        out["%s.E" % k] = r["E"]; out["%s.LL" % k] = np.array(r["LL"]); out["%s.A0" % k] = r["A0"]
        out["%s.seg_LL" % k] = r["seg_LL"]; out["%s.seg_chk" % k] = r["seg_chk"]
        out["%s.seg_A_rowsum" % k] = r["seg_A"].sum(2)
        if nm == "curve":
            out["%s.A" % k] = r["A"]
            out["%s.seg_A_colsum" % k] = r["seg_A"].sum(1)
            out["%s.seg_E" % k] = r["seg_E"]
            f, b, s, lk = R.fwd_bwd(hp["a"], hp["e"], hp["a0"], segs[5])
            out["%s.f65" % k] = f; out["%s.b65" % k] = b; out["%s.s65" % k] = s; out["%s.lk65" % k] = np.array(lk)
    np.savez_compressed(os.path.join(HERE, "estep_n128.npz"), **out)
    cli = os.path.join(HERE, "cli")
    args = ["-N2", "-p", mg.PAT128, "small.psmcfa"]
    txt, err = mg.run_ref(args, cli)
    open(os.path.join(cli, "small_n128_N2.psmc"), "w").write(txt)
    open(os.path.join(cli, "small_n128_N2.args"), "w").write(" ".join(args) + "\n")
    args = ["-N1", "-d", "-p", mg.PAT128, "small.psmcfa"]
    txt, err = mg.run_ref(args, cli)
    open(os.path.join(cli, "small_n128_d.psmc"), "w").write(txt)
    open(os.path.join(cli, "small_n128_d.args"), "w").write(" ".join(args) + "\n")
    print("n128 fixtures: %.1f KB" % (os.path.getsize(os.path.join(HERE, "estep_n128.npz")) / 1024))


if __name__ == "__main__":
    main()
