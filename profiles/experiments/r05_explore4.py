#!/usr/bin/env python3
"""Round 5, psmc_boot --main with the main run ALIVE during the measured iterations (-N10: the main run lasts ~36 s, the batch's
iterations 2 and 3 lie inside it, 5-10 after it): compute-unit masks of 32 / 64 units against no masks with 128 / 224 entry slots
kept free.  (r05_explore2/3 used -N3/-N4: the main run had finished before the batch's second iteration -- their "free" main run
was never concurrent.)  -> gpurun_out/r05_explore4.json"""
import json, os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import northstar_data as nd
HOST = os.path.join(ROOT, "psmc_amd", "host")
out = {}
f = nd.files()
tmp = os.environ.get("TMPDIR", "/tmp")
args = ["-t15", "-r5", "-p", "4+25*2+4+6"]
KEEP = re.compile(r"iteration|batch launch|main run|\[psmc\] E-step|error|cannot|failed", re.I)
for tag, env in [(a, dict(x.split("=") for x in a.split(","))) for a in (sys.argv[1:] or ["PSMC_BOOT_MAIN_CUS=32", "PSMC_BOOT_MAIN_CUS=0,PSMC_BOOT_MAIN_SLOTS=128", "PSMC_BOOT_MAIN_CUS=0,PSMC_BOOT_MAIN_SLOTS=224"])]:
    e = dict(os.environ, PSMC_HIP_MODE="exact", PSMC_TIMING="1", PSMC_HIP_DEBUG_TIMES="1", **env)
    cmd = [os.path.join(HOST, "psmc_boot"), "-R", "100", "-S", "1000", "-O", os.path.join(tmp, "x4-%d.psmc"), "--main", os.path.join(tmp, "x4-main.psmc"), "--main-input", f["genome"],
           "--", "-N10"] + args + [f["split"]]
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, env=e)
    lines = [l[:230] for l in r.stderr.splitlines() if KEEP.search(l)]
    its = [float(m.group(1)) for m in re.finditer(r"E-steps ([0-9.]+) ms", r.stderr)]
    mes = [float(m.group(1)) for m in re.finditer(r"\[psmc\] E-step ([0-9.]+) ms", r.stderr)]
    out[tag] = dict(rc=r.returncode, wall_s=round(time.time() - t0, 2), batch_iterations_ms=its, main_esteps_ms=mes, lines=lines)
    print(tag, r.returncode, out[tag]["wall_s"], "batch", [round(x) for x in its], "main", [round(x) for x in mes], flush=True)
    # the launches of the third iteration (main alive) and of the last one (main gone)
    it_idx = [i for i, l in enumerate(lines) if "iteration" in l]
    for which in (2, len(it_idx) - 1):
        if which < len(it_idx):
            lo = it_idx[which - 1] + 1 if which > 0 else 0
            for l in lines[lo:it_idx[which] + 1]:
                if "batch launch" in l or "iteration" in l: print("    ", l, flush=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r05_explore4.json"), "w"), indent=1)
