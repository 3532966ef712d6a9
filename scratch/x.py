import os, subprocess, sys, time
ROOT="/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import northstar_data as nd
tmp = os.environ.get("TMPDIR", "/tmp")
f = nd.files(tmp)
HOST = os.path.join(ROOT, "psmc_amd", "host")
ARGS = ["-t15", "-r5", "-p", "4+25*2+4+6"]
env = dict(os.environ, **({"PSMC_BOOT_QUICK_EXIT": "1"} if len(sys.argv) > 1 else {}), PSMC_HIP_MODE="exact", PSMC_TIMING="1", PSMC_SEED="4242", PSMC_HIP_DEBUG_TIMES="1")
cmd = [os.path.join(HOST, "psmc_boot"), "-R", "100", "-S", "1000", "-O", os.path.join(tmp, "x-%d.psmc"),
       "--main", os.path.join(tmp, "x-main.psmc"), "--main-input", f["genome"], "--", "-N1"] + ARGS + [f["split"]]
t0 = time.time()
p = subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True, env=env)
for line in p.stderr:
    print("%7.2f  %s" % (time.time() - t0, line.rstrip()[:230]))
p.wait(); print("wall %.2f" % (time.time() - t0))
