#!/usr/bin/env python3
"""Round 5: fast bootstrap, 100 replicates of the north-star trunks at -N8 (FC_ITERS): contexts per device (PSMC_HIP_DEVICES=0 / 0,0 / 0,0,0 / 0,0,0,0)
and the E / M pipeline (PSMC_BOOT_GROUPS 1 / 2).  -> gpurun_out/r05_fast_contexts.json"""
import json, os, re, subprocess, sys, time, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import northstar_data as nd
HOST = os.path.join(ROOT, "psmc_amd", "host")
f = nd.files(want=("split",))
tmp = os.environ.get("TMPDIR", "/tmp")
out = {}
ref = None
for devs, groups in [tuple(v.split("/")) for v in (sys.argv[1:] or ["0,0/1", "0,0/2", "0/2", "0,0,0/2", "0,0,0,0/2"])]:
    tag = "devices=%s groups=%s" % (devs, groups)
    e = dict(os.environ, PSMC_HIP_MODE="fast", PSMC_TIMING="1", PSMC_BOOT_GROUPS=groups, PSMC_HIP_DEVICES=devs)
    cmd = [os.path.join(HOST, "psmc_boot"), "-R", "100", "-S", "1000", "-O", os.path.join(tmp, "fc-%d.psmc"), "--", "-N%s" % os.environ.get("FC_ITERS", "8"), "-t15", "-r5", "-p", "4+25*2+4+6", f["split"]]
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, env=e)
    wall = time.time() - t0
    its = [(float(m.group(1)), float(m.group(2)), float(m.group(3))) for m in re.finditer(r"E-steps ([0-9.]+) ms on \d+ device\(s\), M-steps ([0-9.]+) ms, \d+ group\(s\), wall ([0-9.]+) ms", r.stderr)]
    h = hashlib.sha256()
    for k in range(100):
        try: h.update(open(os.path.join(tmp, "fc-%d.psmc" % k), "rb").read())
        except OSError: h.update(b"missing")
    if devs == "0,0":
        same = (ref is None) or ref == h.hexdigest(); ref = ref or h.hexdigest()
    else:
        same = None   # another dealing of the replicates over contexts: another learning history
    out[tag] = dict(rc=r.returncode, wall_s=round(wall, 2), iterations_E_M_wall_ms=its, files_equal_groups_1=same, stderr_tail=r.stderr[-300:] if r.returncode else "")
    print(tag, r.returncode, round(wall, 1), same, [round(t[2]) for t in its], "M", [round(t[1]) for t in its], flush=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r05_fast_contexts.json"), "w"), indent=1)
