"""SURVEY.md section 5 (race / memory checking of the host driver): the C host code (command line, reader, model,
M-step, writer, decoding, bootstrap driver) built with -fsanitize=address,undefined and with -fsanitize=thread, the
CPU oracle as E-step backend (tests/host_oracle_main.c, tests/host_oracle_boot_main.c), run over the golden command
lines: byte-identical output AND no sanitizer report.  CPU only; a few seconds per case."""
import glob
import gzip
import os
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "psmc_amd", "host")
CLI = os.path.join(ROOT, "tests", "golden", "cli")
BUILD = "/tmp/psmc_test_build"
SRC = [os.path.join(HOST, f) for f in ("pattern.c", "input.c", "model.c", "mstep.c", "run.c", "sim.c", "boot.c")]


def build(name, main_c, san, extra=()):
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, name)
    fastq = os.path.join(BUILD, name + "_fastq.o")   # the libmvec restatement keeps its own flags (no sanitizer: -ffast-math SIMD code, checked by test_fast_mstep_logfactors_reject_alike)
    subprocess.run(["gcc", "-O3", "-g", "-fPIC", "-std=gnu99", "-I" + HOST, "-mavx2", "-mfma", "-ffast-math", "-fopenmp-simd", "-c", "-o", fastq,
                    os.path.join(HOST, "fastq.c")], check=True)
    cmd = ["gcc", "-O1", "-g", "-fno-omit-frame-pointer", "-ffp-contract=off", "-std=gnu99", "-fopenmp"] + list(san) + list(extra) + \
          ["-I" + HOST, "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "oracle"), "-o", exe, os.path.join(ROOT, "tests", main_c)] + SRC + \
          [fastq, os.path.join(ROOT, "oracle", "psmc_oracle.c"), "-lz", "-lmvec", "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def golden_text(name):
    p = os.path.join(CLI, name + ".psmc")
    return open(p).read() if os.path.exists(p) else gzip.open(p + ".gz", "rt").read()


@pytest.fixture(scope="module")
def asan_psmc():
    return build("psmc_oracle_asan", "host_oracle_main.c", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"])


CASES = [os.path.basename(f)[:-5] for f in sorted(glob.glob(os.path.join(CLI, "*.args")))]
# (the wide-state goldens run minutes under ASan; their host path is the same code as the small ones)
SMALL = [c for c in CASES if "n128" not in c and "mid_n64" not in c and "n200" not in c and "n149" not in c] or CASES[:4]


@pytest.mark.parametrize("name", SMALL)
def test_host_driver_asan_ubsan_clean(asan_psmc, name):
    args = open(os.path.join(CLI, name + ".args")).read().split()
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=97", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=98")
    r = subprocess.run([asan_psmc] + args, cwd=CLI, capture_output=True, text=True, env=env)
    assert r.returncode == 0, (r.returncode, r.stderr[-3000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr and "LeakSanitizer" not in r.stderr, r.stderr[-3000:]
    assert r.stdout == golden_text(name)


def tsan_reports_between_our_threads(stderr):
    """ThreadSanitizer reports with a frame of psmc_amd/host in them.  boot.c's threads -- one per (pretend) device, a pool for the
    M-steps, the main run's -- meet only under one mutex and its condition variable, which TSan follows: any report is a real race."""
    return ["WARNING: ThreadSanitizer" + rep[:1500] for rep in stderr.split("WARNING: ThreadSanitizer")[1:] if "/psmc_amd/host/" in rep]


@pytest.mark.parametrize("fast_mstep", ["0", "1"])
def test_boot_driver_asan_and_tsan_clean(tmp_path, fast_mstep):
    """boot.c: one driver thread per (pretend) device sending its replicates' E-steps as a batch, the batch's progress callback
    queueing each finished replicate's M-step for a pool of threads, devices moving on when their own replicates are through -- under
    AddressSanitizer and under ThreadSanitizer (two EM iterations: the hand-over between the stages happens in the second)."""
    args = ["-N2", "-I", "0.3", os.path.join(CLI, "mid.psmcfa.gz")]
    outs = {}
    for tag, san, env_extra in (("asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"], dict(ASAN_OPTIONS="detect_leaks=1:exitcode=97")),
                                ("tsan", ["-fsanitize=thread"], dict(TSAN_OPTIONS="exitcode=0:history_size=4"))):
        exe = build("psmc_oracle_boot_" + tag, "host_oracle_boot_main.c", san)
        env = dict(os.environ, PSMC_FAST_MSTEP=fast_mstep, PSMC_FACTORED=fast_mstep, OMP_NUM_THREADS="4", **env_extra)
        run_args = args
        r = subprocess.run([exe, "4", "17", str(tmp_path / (tag + "-%d.psmc"))] + run_args, capture_output=True, text=True, env=env)
        if tag == "asan":
            assert r.returncode == 0, (tag, r.returncode, r.stderr[-3000:])
            assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, (tag, r.stderr[-3000:])
        else:
            ours = tsan_reports_between_our_threads(r.stderr)
            assert not ours, "\n".join(ours)[:4000]
        outs[tag] = [open(tmp_path / (tag + "-%d.psmc" % k)).read() for k in range(4)]
    assert len(set(outs["asan"])) == 4 and len(set(outs["tsan"])) == 4   # the replicates really differ
    assert outs["asan"] == outs["tsan"]                                     # the same runs under both sanitizers


def test_boot_output_pattern_is_not_a_format_string(tmp_path):
    """-O is the user's string.  Exactly one %d, %% for a literal percent sign; anything else is refused
    before a file is touched (it used to be handed to snprintf as the format)."""
    exe = build("psmc_oracle_boot_asan", "host_oracle_boot_main.c", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"])
    args = ["-N1", os.path.join(CLI, "mid.psmcfa.gz")]
    for bad in ("out-%s.psmc", "out-%d-%d.psmc", "out-%n%d.psmc", "out.psmc", "out-%ld.psmc", "out-%"):
        r = subprocess.run([exe, "2", "1", str(tmp_path / bad)] + args, capture_output=True, text=True)
        assert r.returncode != 0 and "AddressSanitizer" not in r.stderr, (bad, r.returncode, r.stderr[-500:])
        assert not list(tmp_path.glob("out*")), bad
    r = subprocess.run([exe, "2", "1", str(tmp_path / "ok-100%%-%d.psmc")] + args, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1000:]
    assert sorted(p.name for p in tmp_path.glob("ok-*")) == ["ok-100%-0.psmc", "ok-100%-1.psmc"]
