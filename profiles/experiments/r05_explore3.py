#!/usr/bin/env python3
"""Round 5, third GPU call: (1) which compute units do the bits of a CU mask name (SE balance), (2) psmc_boot --main unmasked with
entry slots kept free vs masked, (3) the fast bootstrap's first iteration: fresh device vs after a process that dirtied 270 GB.
-> gpurun_out/r05_explore3.json"""
import json, os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import northstar_data as nd
HOST = os.path.join(ROOT, "psmc_amd", "host")
out = {}
def save(): json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r05_explore3.json"), "w"), indent=1)
which = sys.argv[1:] or ["map", "fast_fresh", "main0", "main32", "fast_after"]
if "map" in which:
    from psmc_amd import hip
    seen, order = set(), []
    for k in range(8, 129, 8):
        r = hip.cumask_probe(k, 4 * k, 64, 400)
        new = [c for c in r["a"]["cus"] if c not in seen]
        seen.update(new); order.append(dict(bits="%d..%d" % (k - 8, k - 1), new=new))
        print("bits %3d..%3d ->" % (k - 8, k - 1), new, flush=True)
    out["cu_mask_bit_order"] = order; save()
f = nd.files()
tmp = os.environ.get("TMPDIR", "/tmp")
args = ["-t15", "-r5", "-p", "4+25*2+4+6"]
KEEP = re.compile(r"iteration|batch launch|batch:|fast batch|main run|\[psmc\] E-step|error|cannot|failed", re.I)
def boot(mode, n_rep, iters, tag, env=None, main=False, dbg=True):
    e = dict(os.environ, PSMC_HIP_MODE=mode, PSMC_TIMING="1", **(env or {}))
    if dbg: e["PSMC_HIP_DEBUG_TIMES"] = "1"
    cmd = [os.path.join(HOST, "psmc_boot"), "-R", str(n_rep), "-S", "1000", "-O", os.path.join(tmp, tag + "-%d.psmc")]
    if main: cmd += ["--main", os.path.join(tmp, tag + "-main.psmc"), "--main-input", f["genome"]]
    cmd += ["--", "-N%d" % iters] + args + [f["split"]]
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, env=e)
    res = dict(rc=r.returncode, wall_s=round(time.time() - t0, 2), lines=[l[:230] for l in r.stderr.splitlines() if KEEP.search(l)][-60:])
    print(tag, res["rc"], res["wall_s"], flush=True)
    for l in res["lines"]: print("   ", l, flush=True)
    return res
for w in which:
    if w == "fast_fresh": out[w] = boot("fast", 100, 3, "f0", dbg=False)
    elif w == "fast_after": out[w] = boot("fast", 100, 3, "f1")
    elif w.startswith("main"): out[w] = boot("exact", 100, 3, w, env=dict(PSMC_BOOT_MAIN_CUS=w[4:]), main=True)
    save()
