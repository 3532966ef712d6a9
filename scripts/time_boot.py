#!/usr/bin/env python3
"""Config 4 at its stated size on the MI355X box: 100 bootstrap replicates over splitfa-like trunks of a 30 M-bin genome
(utils/splitfa.c:20-35 of the reference: 500,000-bin trunks, a tail shorter than 1.5 trunks stays whole), n = 64,
`psmc_boot -R 100 ... -- -N<iters> -t15 -r5 -p "4+25*2+4+6"`, exact and fast mode, with PSMC_TIMING per-iteration
times; next to it two single `psmc -b` runs (what the reference's xargs farm would start 100 times).
Writes JSON to argv[1] (default gpurun_out/r02_boot_timing.json)."""
import json
import os
import re
import subprocess
import sys
import time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
HOST = os.path.join(ROOT, "psmc_amd", "host")


def main():
    out_json = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r02_boot_timing.json")
    n_rep = int(os.environ.get("BOOT_REPLICATES", "100"))
    iters = int(os.environ.get("BOOT_ITERS", "3"))
    import northstar_data as nd
    tmp = os.environ.get("TMPDIR", "/tmp")
    fd = nd.files(tmp, want=("split",))
    path = fd["split"]
    res = dict(workload="%d trunks (%d bins, longest %d), %d replicates, -N%d -t15 -r5 -p 4+25*2+4+6" % (fd["n_trunks"], fd["trunk_bins"], fd["longest_trunk"], n_rep, iters), runs={})
    args = ["-N%d" % iters, "-t15", "-r5", "-p", "4+25*2+4+6", path]
    for mode in (("fast",) if os.environ.get("BOOT_FAST_ONLY") else ("exact",) if os.environ.get("BOOT_EXACT_ONLY") else ("exact", "fast")):
        env = dict(os.environ, PSMC_HIP_MODE=mode, PSMC_TIMING="1")
        t0 = time.time()
        r = subprocess.run([os.path.join(HOST, "psmc_boot"), "-R", str(n_rep), "-S", "1000", "-O", os.path.join(tmp, "boot_%s-%%d.psmc" % mode), "--"] + args,
                           capture_output=True, text=True, env=env)
        wall = time.time() - t0
        its = [(float(m.group(1)), float(m.group(2))) for m in re.finditer(r"E-steps ([0-9.]+) ms on \d+ device\(s\), M-steps ([0-9.]+) ms", r.stderr)]
        res["runs"]["psmc_boot_" + mode] = dict(rc=r.returncode, wall_s=round(wall, 2), per_iteration_ms=[dict(esteps=x, msteps=y) for x, y in its],
                                               stderr_tail=r.stderr[-400:] if r.returncode else "")
        sys.stderr.write("[time_boot] %s: %.1f s, iterations %s\n" % (mode, wall, its))
        for ln in r.stderr.splitlines():   # PSMC_HIP_DEBUG_TIMES=1: the library's per-group breakdown of the exact batch
            if "batch launch" in ln or "fast batch" in ln or "batch:" in ln: sys.stderr.write(ln[:260] + "\n")
        t0 = time.time()
        one = subprocess.run([os.path.join(HOST, "psmc"), "-b"] + args, capture_output=True, text=True, env=dict(env, PSMC_SEED="1000"))
        w1 = time.time() - t0
        ts = [(float(m.group(1)), float(m.group(2))) for m in re.finditer(r"E-step ([0-9.]+) ms, M-step ([0-9.]+) ms", one.stderr)]
        same = None
        try:
            same = open(os.path.join(tmp, "boot_%s-0.psmc" % mode)).read() == one.stdout
        except Exception:
            pass
        res["runs"]["single_psmc_b_" + mode] = dict(rc=one.returncode, wall_s=round(w1, 2), per_iteration_ms=[dict(estep=x, mstep=y) for x, y in ts],
                                                  replicate0_identical_to_psmc_boot=same)
    json.dump(res, open(out_json, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
