#!/usr/bin/env python3
"""Randomised end-to-end run of the `psmc` binary (exact mode on the GPU) against the reference's own binary built from its sources
(oracle/_ref/psmc_ref; test infrastructure), byte for byte (round 6; `python scripts/fuzz_cli.py SECONDS [SEED0]`).
Every case: a random .psmcfa (1-6 sequences of 60..40 k bins, heterozygosity drifting along the sequence, runs of N), a random pattern
(3..150 hidden states: the 64-state, 128-state and wide kernels all come up), -N1..4, random -t / -r, sometimes -d or -d -D; FUZZ_MORE=1 adds -s, -C, -T and -l;
FUZZ_ONE=<seed> replays one case and keeps both outputs."""
import json
import os
import subprocess
import sys
import time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import northstar_data as nd

OURS = os.path.join(ROOT, "psmc_amd", "host", "psmc")
REF = os.path.join(ROOT, "oracle", "_ref", "psmc_ref")
PATTERNS = ["4+5*3+4", "6*1", "3+2*2+3", "2*8", "1+1+1", "4+25*2+4+6", "20*3+4", "10*1+5*2", "1*3+2*10", "64*2", "30*5", "3*1+1*60", "33*2"]


def make_input(rng, path):
    segs = []
    for _ in range(int(rng.integers(1, 7))):
        L = int(np.exp(rng.uniform(np.log(60), np.log(40_000))))
        het = np.exp(rng.normal(np.log(0.01), 1.0))
        rate = np.clip(het * np.exp(np.cumsum(rng.normal(0, 0.02, size=L))), 1e-4, 0.3)
        s = (rng.random(L) < rate).astype(np.uint8)
        for _ in range(int(rng.integers(0, 4))):
            w = int(np.exp(rng.uniform(0, np.log(max(2, L // 3))))); at = int(rng.integers(0, L - w + 1)); s[at:at + w] = 2
        segs.append(s)
    nd.write_psmcfa(path, segs, "s")
    return [len(s) for s in segs]


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    tmp = os.environ.get("TMPDIR", "/tmp")
    if not os.path.exists(REF):
        print("oracle/_ref/psmc_ref is not here (make -C oracle ref needs the reference checkout)"); return 2
    t_end = time.time() + budget
    stats = dict(cases=0, failures=[], by_pattern={})
    seed = seed0
    one = os.environ.get("FUZZ_ONE")   # replay one seed; both outputs stay in $TMPDIR as fuzz_cli_ref.psmc / fuzz_cli_ours.psmc
    if one: seed = int(one); t_end = time.time() + 1e9
    while time.time() < t_end:
        rng = np.random.default_rng(seed)
        fa = os.path.join(tmp, "fuzz_cli.psmcfa")
        lens = make_input(rng, fa)
        pat = PATTERNS[int(rng.integers(len(PATTERNS)))]
        args = ["-N%d" % rng.integers(1, 5), "-t%g" % round(rng.uniform(3, 20), 1), "-r%g" % round(rng.uniform(1, 8), 1), "-p", pat]
        u = rng.random()
        if u < 0.25: args.append("-d")
        elif u < 0.4: args += ["-d", "-D"]
        if os.environ.get("FUZZ_MORE"):   # (second generation of cases: the draws above stay what they were for a given seed)
            v = rng.random()
            if v < 0.12: args += ["-d", "-s"] if "-d" not in args else ["-s"]
            elif v < 0.24: args += (["-d"] if "-d" not in args else []) + ["-C", str(int(rng.integers(1, 12)))]
            if rng.random() < 0.15: args += ["-T", "%g" % round(rng.uniform(0.05, 2.0), 2)]
            if rng.random() < 0.1: args += ["-l", "%g" % round(rng.uniform(0.05, 0.3), 2)]
        case = dict(seed=seed, lens=lens, args=args)
        outs = []
        for exe, env in ((REF, os.environ), (OURS, dict(os.environ, PSMC_HIP_MODE="exact"))):
            o = os.path.join(tmp, "fuzz_cli_%s.psmc" % ("ref" if exe == REF else "ours"))
            r = subprocess.run([exe] + args + ["-o", o, fa], capture_output=True, text=True, env=env, timeout=600)
            outs.append((r.returncode, open(o, "rb").read() if os.path.exists(o) else b"", r.stderr[-300:]))
            if os.path.exists(o) and not one: os.remove(o)
        if outs[0][0] < 0 and outs[1][0] == 0 and outs[1][1].startswith(outs[0][1][:outs[0][1].rfind(b"\n") + 1]):
            stats["reference_aborted"] = stats.get("reference_aborted", 0) + 1   # (the reference's own crash, e.g. -d with three states: what it wrote before is a prefix of ours)
        elif outs[0][0] != outs[1][0] or outs[0][1] != outs[1][1]:
            first = next((i for i, (x, y) in enumerate(zip(outs[0][1].splitlines(), outs[1][1].splitlines())) if x != y), -1)
            stats["failures"].append(dict(case, rc=[outs[0][0], outs[1][0]], sizes=[len(outs[0][1]), len(outs[1][1])], first_diff_line=first, stderr=outs[1][2]))
            print("FAIL", json.dumps(stats["failures"][-1]), flush=True)
        stats["cases"] += 1; stats["by_pattern"][pat] = stats["by_pattern"].get(pat, 0) + 1
        seed += 1
        if one: print("args", args, "lens", lens); break
    stats["seeds"] = [seed0, seed - 1]
    print(json.dumps(stats, indent=1))
    return 1 if stats["failures"] else 0


if __name__ == "__main__":
    sys.exit(main())
