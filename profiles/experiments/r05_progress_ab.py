#!/usr/bin/env python3
"""Round 5: M-steps under the batch (psmc_hip_estep_batch_cb + "batch_major").  psmc_boot --main, exact mode, 100 replicates of the
north-star trunks at -N6 (the main run alive for the first three batch iterations): batch_major=1 (replicates complete launch by
launch) against 0 (everything by length: they complete in the last launch); then the fast job at -N25.  Files compared byte for
byte between the exact variants.  -> gpurun_out/r05_progress_ab.json"""
import json, os, re, subprocess, sys, time, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import northstar_data as nd
HOST = os.path.join(ROOT, "psmc_amd", "host")
f = nd.files()
tmp = os.environ.get("TMPDIR", "/tmp")
out = {}; first = {}
KEEP = re.compile(r"iteration|batch:|usable|main run|error|cannot|failed", re.I)
for mode, opts, iters, main in [v.split("/") for v in (sys.argv[1:] or ["exact/batch_major=1/6/1", "exact/batch_major=0/6/1", "fast//25/1"])]:
    tag = "%s %s -N%s%s" % (mode, opts, iters, " --main" if main == "1" else "")
    e = dict(os.environ, PSMC_HIP_MODE=mode, PSMC_TIMING="1", PSMC_HIP_DEBUG_TIMES="1" if mode == "exact" else "", PSMC_HIP_OPTIONS=opts)
    if not e["PSMC_HIP_DEBUG_TIMES"]: del e["PSMC_HIP_DEBUG_TIMES"]
    cmd = [os.path.join(HOST, "psmc_boot"), "-R", "100", "-S", "1000", "-O", os.path.join(tmp, "pab-%d.psmc")]
    if main == "1": cmd += ["--main", os.path.join(tmp, "pab-main.psmc"), "--main-input", f["genome"]]
    cmd += ["--", "-N" + iters, "-t15", "-r5", "-p", "4+25*2+4+6", f["split"]]
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, env=e)
    wall = time.time() - t0
    its = [(float(m.group(1)), float(m.group(2)), float(m.group(3)), float(m.group(4))) for m in
           re.finditer(r"E-steps ([0-9.]+) ms on \d+ device\(s\), M-steps ([0-9.]+) ms after the last batch \(([0-9.]+) ms of work.*?wall ([0-9.]+) ms", r.stderr)]
    h = hashlib.sha256()
    for k in list(range(100)) + ["main"]:
        try: h.update(open(os.path.join(tmp, "pab-%s.psmc" % k), "rb").read())
        except OSError: h.update(b"missing")
    same = first.setdefault(mode, h.hexdigest()) == h.hexdigest()
    lines = [l[:300] for l in r.stderr.splitlines() if KEEP.search(l)]
    out[tag] = dict(rc=r.returncode, wall_s=round(wall, 2), iterations_E_Mtail_Mwork_wall_ms=its, files_equal_first_variant_of_mode=same, lines=lines[-14:], stderr_tail=r.stderr[-400:] if r.returncode else "")
    print(tag, r.returncode, round(wall, 1), "same files" if same else "FILES DIFFER", [tuple(round(x) for x in t) for t in its], flush=True)
    for l in lines[-8:]: print("    ", l[:260], flush=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r05_progress_ab.json"), "w"), indent=1)
