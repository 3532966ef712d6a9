#!/usr/bin/env python3
"""Round 5, first GPU call: (1) host cores the box really gives us, (2) do per-stream compute-unit masks partition the device,
(3) where the first iterations of the fast bootstrap go (PSMC_HIP_DEBUG_TIMES), (4) what naive sharing costs: psmc (exact, main run)
beside psmc_boot (exact, 100 replicates) as two processes.  -> gpurun_out/r05_explore1.json"""
import json, os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import northstar_data as nd
HOST = os.path.join(ROOT, "psmc_amd", "host")
out = {}
def save(): json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r05_explore1.json"), "w"), indent=1)
# ---- 1
h = dict(cpu_count=os.cpu_count(), affinity=len(os.sched_getaffinity(0)))
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/memory.max"):
    try: h[p] = open(p).read().strip()
    except Exception: pass
try: h["loadavg"] = open("/proc/loadavg").read().strip()
except Exception: pass
out["host"] = h; save()
print("host", h, flush=True)
# ---- 2
from psmc_amd import hip
cm = {}
for (ca, wa, wb) in ((0, 96, 1024), (24, 96, 928), (24, 96, 1024), (16, 90, 4000), (32, 128, 896), (8, 32, 992)):
    try:
        r = hip.cumask_probe(ca, wa, wb, 3328)
        cm["%d cus: %d + %d waves" % (ca, wa, wb)] = r
        print("cumask", ca, wa, wb, json.dumps(r), flush=True)
    except Exception as ex:
        cm["%d cus: %d + %d waves" % (ca, wa, wb)] = dict(error=str(ex)); print("cumask", ca, "ERR", ex, flush=True)
out["cumask_probe"] = cm; save()
# ---- data
t0 = time.time(); f = nd.files(); print("data %.1f s" % (time.time() - t0), f, flush=True)
tmp = os.environ.get("TMPDIR", "/tmp")
args = ["-t15", "-r5", "-p", "4+25*2+4+6"]
def boot(mode, n_rep, iters, tag, extra_env=None, wait=True):
    env = dict(os.environ, PSMC_HIP_MODE=mode, PSMC_TIMING="1", PSMC_HIP_DEBUG_TIMES="1", **(extra_env or {}))
    cmd = [os.path.join(HOST, "psmc_boot"), "-R", str(n_rep), "-S", "1000", "-O", os.path.join(tmp, tag + "-%d.psmc"), "--", "-N%d" % iters] + args + [f["split"]]
    t0 = time.time()
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    if not wait: return p, t0
    so, se = p.communicate()
    return dict(rc=p.returncode, wall_s=round(time.time() - t0, 2), stderr=[l[:300] for l in se.splitlines() if "psmc" in l][-60:])
# ---- 3 fast bootstrap, 100 replicates x 4 iterations
out["boot_fast_100"] = boot("fast", 100, 4, "x1f"); save()
print("boot fast", json.dumps(out["boot_fast_100"], indent=0)[:6000], flush=True)
out["boot_fast_16"] = boot("fast", 16, 3, "x1g"); save()
print("boot fast 16", json.dumps(out["boot_fast_16"], indent=0)[:3000], flush=True)
# ---- 4 exact: boot alone, main alone, both at once (main first, so that its tables exist before the batch sizes its own)
out["boot_exact_alone"] = boot("exact", 100, 3, "x1e"); save()
print("boot exact alone", json.dumps(out["boot_exact_alone"], indent=0)[:5000], flush=True)
def main_run(iters, wait=True):
    env = dict(os.environ, PSMC_HIP_MODE="exact", PSMC_TIMING="1")
    t0 = time.time()
    p = subprocess.Popen([os.path.join(HOST, "psmc"), "-N%d" % iters] + args + ["-o", os.path.join(tmp, "x1main.psmc"), f["genome"]], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    if not wait: return p, t0
    so, se = p.communicate()
    return dict(rc=p.returncode, wall_s=round(time.time() - t0, 2), stderr=[l[:200] for l in se.splitlines()][-12:])
out["main_exact_alone"] = main_run(3); save()
print("main alone", out["main_exact_alone"], flush=True)
pm, tm = main_run(6, wait=False)
time.sleep(6.0)   # input parsing + upload + first table allocation of the main run
pb, tb = boot("exact", 100, 3, "x1h", wait=False)
sb, eb = pb.communicate(); wb_ = time.time() - tb
sm, em = pm.communicate(); wm_ = time.time() - tm
out["both_at_once"] = dict(boot=dict(rc=pb.returncode, wall_s=round(wb_, 2), stderr=[l[:300] for l in eb.splitlines() if "psmc" in l][-40:]),
                           main=dict(rc=pm.returncode, wall_s=round(wm_, 2), stderr=[l[:200] for l in em.splitlines()][-12:]))
save()
print("both", json.dumps(out["both_at_once"], indent=0)[:8000], flush=True)
