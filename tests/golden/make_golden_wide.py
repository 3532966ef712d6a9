#!/usr/bin/env python3
"""Golden vectors beyond 128 hidden states -- `-p "100*2"` (200 states) and the odd `-p "4+47*3+4"` (149) -- from the
REAL reference (oracle/_ref, built from the unmodified sources): the reference allocates for any n (khmm.c:10-23) and
cli.c:66-99 accepts any pattern; round 5's wide exact kernels (psmc_amd/csrc/estep_wide.hip) must reproduce it bit for
bit.  Separate from the other generators so that their fixtures stay byte-stable.
Outputs: estep_wide.npz and cli/small_n200_N2.*, cli/small_n149_d.*, cli/small_n149_D.* (gz).

    python tests/golden/make_golden_wide.py
"""
import gzip
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (helpers only: orc, bottleneck_lambdas, run_ref)
from make_golden import orc  # noqa: E402

PATS = {"n200": "100*2", "n149": "4+47*3+4"}


def main():
    orc.build_oracle(with_ref=True)
    R = orc.Reference()
    segs_npz = np.load(os.path.join(HERE, "segments_small.npz"))
    segs = [segs_npz[k] for k in sorted(segs_npz.files)]
    out = {}
    for k, pat in PATS.items():
        n, nf, pm = R.parse_pattern(pat)
        par = np.concatenate([[0.0625, 0.0131, 15.0], mg.bottleneck_lambdas(nf)])
        hp = R.hmm_params(pat, par)
        assert hp["a"].shape[0] == n + 1 == int(k[1:])
        for f in ("a", "e", "a0"):
            out["%s.%s" % (k, f)] = np.asarray(hp[f])
        out["%s.params" % k] = par
        use = segs[:8]   # 8 of the 13 small segments: the reference needs ~n^2 per bin
        r = R.estep(hp["a"], hp["e"], hp["a0"], use, per_seg=True)
        out["%s.A" % k] = r["A"]; out["%s.E" % k] = r["E"]; out["%s.LL" % k] = np.array(r["LL"]); out["%s.A0" % k] = r["A0"]
        out["%s.seg_LL" % k] = r["seg_LL"]; out["%s.seg_chk" % k] = r["seg_chk"]; out["%s.seg_E" % k] = r["seg_E"]
        out["%s.seg_A_rowsum" % k] = r["seg_A"].sum(2); out["%s.seg_A_colsum" % k] = r["seg_A"].sum(1)
        f, b, s, lk = R.fwd_bwd(hp["a"], hp["e"], hp["a0"], segs[5])
        # the tables of one segment, thinned to every 7th position (1 .. L): 200 states x 2 tables stay small
        out["%s.f65" % k] = f[1::7]; out["%s.b65" % k] = b[1::7]; out["%s.s65" % k] = s[1:]; out["%s.lk65" % k] = np.array(lk)
    np.savez_compressed(os.path.join(HERE, "estep_wide.npz"), **out)
    cli = os.path.join(HERE, "cli")
    for name, args in (("small_n200_N2", ["-N2", "-p", PATS["n200"], "small.psmcfa"]),
                       ("small_n149_d", ["-N1", "-d", "-p", PATS["n149"], "small.psmcfa"])):
        txt, err = mg.run_ref(args, cli)
        open(os.path.join(cli, name + ".psmc"), "w").write(txt)
        open(os.path.join(cli, name + ".args"), "w").write(" ".join(args) + "\n")
    print("wide fixtures: %.1f KB" % (os.path.getsize(os.path.join(HERE, "estep_wide.npz")) / 1024))


if __name__ == "__main__":
    main()
