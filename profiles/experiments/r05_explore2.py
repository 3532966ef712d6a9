#!/usr/bin/env python3
"""Round 5, second GPU call: exact bootstrap batch with the entry schedule (batch_sort 1 vs 0), psmc_boot --main with the device
split by compute-unit masks, and the fast bootstrap's first iteration on a fresh device vs after a process that dirtied 270 GB.
-> gpurun_out/r05_explore2.json"""
import json, os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import northstar_data as nd
HOST = os.path.join(ROOT, "psmc_amd", "host")
out = {}
def save(): json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r05_explore2.json"), "w"), indent=1)
f = nd.files()
tmp = os.environ.get("TMPDIR", "/tmp")
args = ["-t15", "-r5", "-p", "4+25*2+4+6"]
KEEP = re.compile(r"iteration|batch launch|batch:|fast batch|main run|\[psmc\] E-step|error|cannot|failed", re.I)
def boot(mode, n_rep, iters, tag, env=None, main=False):
    e = dict(os.environ, PSMC_HIP_MODE=mode, PSMC_TIMING="1", PSMC_HIP_DEBUG_TIMES="1", **(env or {}))
    if e.pop("PSMC_HIP_DEBUG_TIMES_OFF", None): e.pop("PSMC_HIP_DEBUG_TIMES", None)
    cmd = [os.path.join(HOST, "psmc_boot"), "-R", str(n_rep), "-S", "1000", "-O", os.path.join(tmp, tag + "-%d.psmc")]
    if main: cmd += ["--main", os.path.join(tmp, tag + "-main.psmc"), "--main-input", f["genome"]]
    cmd += ["--", "-N%d" % iters] + args + [f["split"]]
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, env=e)
    res = dict(rc=r.returncode, wall_s=round(time.time() - t0, 2), lines=[l[:230] for l in r.stderr.splitlines() if KEEP.search(l)][-70:])
    print(tag, res["rc"], res["wall_s"], flush=True)
    for l in res["lines"]: print("   ", l, flush=True)
    return res
which = sys.argv[1:] or ["fast_fresh", "sort1", "sort0", "main24", "main16", "fast_after"]
for w in which:
    if w == "fast_fresh": out[w] = boot("fast", 100, 3, "f0", env=dict(PSMC_HIP_DEBUG_TIMES_OFF="1"))
    elif w == "sort1": out[w] = boot("exact", 100, 3, "s1")
    elif w == "sort0": out[w] = boot("exact", 100, 3, "s0", env=dict(PSMC_HIP_OPTIONS="batch_sort=0"))
    elif w.startswith("main"): out[w] = boot("exact", 100, 4, w, env=dict(PSMC_BOOT_MAIN_CUS=w[4:]), main=True)
    elif w == "fast_after": out[w] = boot("fast", 100, 3, "f1", env=dict(PSMC_HIP_DEBUG_TIMES_OFF="1"))
    elif w == "fast_main": out[w] = boot("fast", 100, 3, "fm", env=dict(PSMC_HIP_DEBUG_TIMES_OFF="1"), main=True)
    save()
# byte checks: the replicates do not depend on the schedule or on the main run beside them
def same(a, b, n=100):
    return all(open(os.path.join(tmp, "%s-%d.psmc" % (a, r))).read() == open(os.path.join(tmp, "%s-%d.psmc" % (b, r))).read() for r in range(n))
chk = {}
for a, b in (("s1", "s0"), ("s1", "main24"), ("s1", "main16")):
    try: chk["%s==%s" % (a, b)] = same(a, b)
    except Exception as ex: chk["%s==%s" % (a, b)] = str(ex)
out["replicates_identical"] = chk; save()
print("identical", chk, flush=True)
