#!/usr/bin/env python3
"""Round 6: why does the fast north-star job take 24 s when it follows the exact one and 19.8 s on a fresh device?  Runs
psmc_boot --main (100 replicates, -N25, fast) on a fresh box, then an exact job of one iteration (250 GB of tables: leaves the
device memory dirty), then the fast job again -- every stderr line of the fast jobs stamped with its time since process start."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import northstar_data as nd
tmp = os.environ.get("TMPDIR", "/tmp")
f = nd.files(tmp)
HOST = os.path.join(ROOT, "psmc_amd", "host")
ARGS = ["-t15", "-r5", "-p", "4+25*2+4+6"]
def job(mode, iters, tag):
    env = dict(os.environ, PSMC_HIP_MODE=mode, PSMC_TIMING="1", PSMC_SEED="4242")
    cmd = [os.path.join(HOST, "psmc_boot"), "-R", "100", "-S", "1000", "-O", os.path.join(tmp, "x_%s-%%d.psmc" % tag),
           "--main", os.path.join(tmp, "x_%s-main.psmc" % tag), "--main-input", f["genome"], "--", "-N%d" % iters] + ARGS + [f["split"]]
    t0 = time.time()
    p = subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True, env=env)
    marks = []
    for line in p.stderr:
        marks.append((time.time() - t0, line.rstrip()[:110]))
    p.wait()
    wall = time.time() - t0
    print("== %s (%s, -N%d): %.2f s, rc %d" % (tag, mode, iters, wall, p.returncode))
    its = [m for m in marks if "iteration" in m[1] and "E-steps" in m[1]]
    print("   first stderr line at %.2f s; first iteration line at %.2f s; last iteration line at %.2f s; process ended %.2f s after it" % (
        marks[0][0] if marks else -1, its[0][0] if its else -1, its[-1][0] if its else -1, wall - (its[-1][0] if its else 0)))
    for t, l in marks[:4] + marks[-3:]: print("   %7.2f  %s" % (t, l))
job("fast", 25, "fresh")
job("exact", 1, "dirty")
job("fast", 25, "after_exact")
