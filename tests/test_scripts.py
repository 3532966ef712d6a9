"""The GPU-lease tooling cannot run here (no GPU), but it must at least parse: a typo in scripts/lease.sh costs a GPU call."""
import ast
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lease_script_parses():
    subprocess.run(["bash", "-n", os.path.join(ROOT, "scripts", "lease.sh")], check=True)
    text = open(os.path.join(ROOT, "scripts", "lease.sh")).read()
    for task in ("suite", "bench", "sweep", "timeline", "prof", "wtrace", "probes", "trace", "northstar", "final"):
        assert ("  %s)" % task) in text or ("  %s) " % task) in text, task   # every documented task has a case


def test_python_scripts_parse():
    for d in ("scripts", os.path.join("profiles", "experiments")):
        for f in sorted(os.listdir(os.path.join(ROOT, d))):
            if f.endswith(".py"):
                ast.parse(open(os.path.join(ROOT, d, f)).read(), filename=f)


def test_sq_summary_kernel_names():
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import sq_summary
    assert sq_summary.short("void psmc::k_fwd_struct<false, 4, true>(double const*, double const*)") == "k_fwd_struct<false,4,true>"
    assert sq_summary.short("psmc::k_bwd_count4f_struct(double const*, double const*, double const*)") == "k_bwd_count4f_struct"
    assert sq_summary.short("__amd_rocclr_copyBuffer") == "__amd_rocclr_copyBuffer"
