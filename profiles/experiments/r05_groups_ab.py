#!/usr/bin/env python3
"""Round 5: psmc_boot's E / M pipeline (PSMC_BOOT_GROUPS) and the tail fill of the exact batch ("batch_tailfill"), 100 replicates of the
north-star trunks, no main run: per-iteration wall clock and the launches of the last iteration.  The replicates' files of every
variant are compared byte for byte with the first one's.  -> gpurun_out/r05_groups_ab.json"""
import json, os, re, subprocess, sys, time, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import northstar_data as nd
HOST = os.path.join(ROOT, "psmc_amd", "host")
out = {}
f = nd.files(want=("split",))
tmp = os.environ.get("TMPDIR", "/tmp")
args = ["-t15", "-r5", "-p", "4+25*2+4+6"]
KEEP = re.compile(r"iteration|batch launch|batch:|error|cannot|failed", re.I)
first = {}
VARIANTS = [("exact", "1", "batch_tailfill=0", 4), ("exact", "1", "", 4), ("exact", "2", "", 4), ("fast", "1", "", 8), ("fast", "2", "", 8)]
for mode, groups, opts, iters in VARIANTS:
    tag = "%s groups=%s %s" % (mode, groups, opts)
    e = dict(os.environ, PSMC_HIP_MODE=mode, PSMC_TIMING="1", PSMC_HIP_DEBUG_TIMES="1", PSMC_BOOT_GROUPS=groups, PSMC_HIP_OPTIONS=opts)
    cmd = [os.path.join(HOST, "psmc_boot"), "-R", "100", "-S", "1000", "-O", os.path.join(tmp, "gab-%d.psmc"), "--", "-N%d" % iters] + args + [f["split"]]
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, env=e)
    wall = time.time() - t0
    lines = [l[:260] for l in r.stderr.splitlines() if KEEP.search(l)]
    its = [(float(m.group(1)), float(m.group(2)), float(m.group(3))) for m in re.finditer(r"E-steps ([0-9.]+) ms on \d+ device\(s\), M-steps ([0-9.]+) ms, \d+ group\(s\), wall ([0-9.]+) ms", r.stderr)]
    h = hashlib.sha256()
    for k in range(100):
        try: h.update(open(os.path.join(tmp, "gab-%d.psmc" % k), "rb").read())
        except OSError: h.update(b"missing")
    same = first.setdefault(mode, h.hexdigest()) == h.hexdigest()
    it_idx = [i for i, l in enumerate(lines) if "iteration" in l]
    last = lines[(it_idx[-2] + 1 if len(it_idx) > 1 else 0):] if it_idx else lines[-12:]
    out[tag] = dict(rc=r.returncode, wall_s=round(wall, 2), iterations_E_M_wall_ms=its, files_equal_first_variant_of_mode=same, last_iteration=last, stderr_tail=r.stderr[-400:] if r.returncode else "")
    print(tag, r.returncode, round(wall, 1), "same files" if same else "FILES DIFFER", [tuple(round(x) for x in t) for t in its], flush=True)
    for l in last: print("    ", l, flush=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r05_groups_ab.json"), "w"), indent=1)
