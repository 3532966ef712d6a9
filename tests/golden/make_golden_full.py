#!/usr/bin/env python3
"""Golden output of the REAL reference at config 2's full size (BASELINE.json configs[1]: one ~500 k-bin segment,
-p "4+25*2+4+6", -N25): tests/golden/full/chr22like.psmcfa.gz + chr22like_N25.psmc.gz.

Runs only in the build container (needs oracle/_ref/psmc_ref, built by `make -C oracle ref` from the unmodified
sources under /root/reference); takes ~2.5 minutes of one CPU core.  The outputs are data: a synthetic observation
stream (seeded draw from the 64-state model of hmm_params.npz) and the reference binary's output on it.  Kept apart
from tests/golden/cli/ because the CPU test-suite replays every case there through the oracle backend, and this one
would take minutes; the GPU tests replay it through the HIP E-step (tests/test_host_cli.py).

    python tests/golden/make_golden_full.py
"""
import gzip
import os
import subprocess
import sys
import time
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import orc  # noqa: E402
from psmc_amd import sim  # noqa: E402

ARGS = ["-N25", "-t15", "-r5", "-p", "4+25*2+4+6", "chr22like.psmcfa.gz"]   # README:12 of the reference


def main():
    g = np.load(os.path.join(HERE, "hmm_params.npz"))
    a, e, a0 = g["n64_curve.a"], g["n64_curve.e"], g["n64_curve.a0"]
    seg = sim.simulate_segment(a, e, a0, 500_000, np.random.default_rng(42))
    out = os.path.join(HERE, "full")
    os.makedirs(out, exist_ok=True)
    conv = np.array(list("TKN"))
    t = ''.join(conv[seg])
    with gzip.GzipFile(os.path.join(out, "chr22like.psmcfa.gz"), "wb", mtime=0) as fh:
        fh.write((">chr22like\n" + '\n'.join(t[j:j + 60] for j in range(0, len(t), 60)) + '\n').encode())
    t0 = time.time()
    r = subprocess.run([orc.REF_BIN] + ARGS, cwd=out, capture_output=True, text=True, check=True)
    with gzip.GzipFile(os.path.join(out, "chr22like_N25.psmc.gz"), "wb", mtime=0) as fh:
        fh.write(r.stdout.encode())
    open(os.path.join(out, "chr22like_N25.args"), "w").write(" ".join(ARGS) + "\n")
    print("reference ran %.0f s; %d bins, het %.4f, missing %.4f" % (time.time() - t0, len(seg), (seg == 1).mean(), (seg == 2).mean()))


if __name__ == "__main__":
    main()
