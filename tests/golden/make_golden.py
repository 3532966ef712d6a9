#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference.

Runs only in the build container: needs oracle/_ref (built by `make -C oracle
ref` from the unmodified sources under /root/reference).  The outputs are pure
data -- inputs and the reference's outputs -- and are committed; the GPU box,
which has no reference checkout, tests against them.

    python tests/golden/make_golden.py
"""
import os
import subprocess
import sys
import gzip
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import orc  # noqa: E402
from psmc_amd import sim  # noqa: E402

PAT64 = "4+25*2+4+6"   # README:12 of the reference: 64 states, 28 free lambdas
PAT23 = "4+5*3+4"      # cli.c:12 default: 23 states
PAT128 = "64*2"


def bottleneck_lambdas(n_free):
    x = np.arange(n_free) / max(n_free - 1, 1)
    lam = 1.0 + 2.5 * np.exp(-((x - 0.15) / 0.08) ** 2) - 0.7 * np.exp(-((x - 0.45) / 0.1) ** 2) + 1.5 * x ** 2
    return np.maximum(lam, 0.1)


def param_sets(R):
    out = {}
    for tag, pat in (("n64", PAT64), ("n23", PAT23)):
        n, nf, pm = R.parse_pattern(pat)
        flat = np.concatenate([[0.06, 0.012, 15.0], np.ones(nf)])                 # RD 0-like start
        curve = np.concatenate([[0.0625, 0.0131, 15.0], bottleneck_lambdas(nf)])  # a fitted-looking history
        for nm, par in (("flat", flat), ("curve", curve)):
            hp = R.hmm_params(pat, par)
            out["%s_%s" % (tag, nm)] = dict(pattern=pat, params=par, par_map=pm, **hp)
    return out


def t10k_text():
    """The 10 000-bin plumbing fixture of SURVEY.md section 8c (Python's random, seed 1)."""
    import random
    random.seed(1)
    L = 10000
    s = []
    i = 0
    while i < L:
        r = random.random()
        if r < 0.02 / 50:
            k = min(L - i, random.randint(10, 90)); s.extend('N' * k); i += k
        else:
            s.append('K' if random.random() < 0.01 else 'T'); i += 1
    s = ''.join(s)
    return '>1\n' + '\n'.join(s[j:j + 60] for j in range(0, L, 60)) + '\n'


def to_psmcfa(segs, names=None):
    conv = np.array(list("TKN"))
    out = []
    for i, s in enumerate(segs):
        out.append(">%s" % (names[i] if names else "seg%d" % i))
        t = ''.join(conv[s])
        out.extend(t[j:j + 60] for j in range(0, len(t), 60))
    return '\n'.join(out) + '\n'


def cnt_file_bytes(lengths, n_cnt=5, seed=5, short_by=None):
    """A cntcpg-style count file (utils/cntcpg.c:55-92 of the reference: int32 n_cnt, then per sequence int32 length
    and length*n_cnt int32 counts) with seeded pseudo-random counts; short_by[i] bins are cut off record i so that the
    reader's length-mismatch path (aux.c:206-210) is exercised."""
    rng = np.random.default_rng(seed)
    out = [np.array([n_cnt], dtype=np.int32).tobytes()]
    for i, L in enumerate(lengths):
        l = int(L) - (short_by or {}).get(i, 0)
        out.append(np.array([l], dtype=np.int32).tobytes())
        out.append(rng.integers(0, 40, size=(l, n_cnt), dtype=np.int32).tobytes())
    return b"".join(out)


def cli_decode_c_cases(cli, segs):
    """-c goldens (psmc_decode's CT lines, aux.c:202-231): counts alone, and together with -d."""
    with open(os.path.join(cli, "small.cnt"), "wb") as fh:
        fh.write(cnt_file_bytes([len(segs[8]), len(segs[9]), len(segs[3])], short_by={1: 7}))
    return {"small_decode_c": ["-N1", "-c", "small.cnt", "small.psmcfa"],
            "small_decode_dc": ["-N1", "-d", "-c", "small.cnt", "small.psmcfa"]}


def run_ref(args, cwd):
    r = subprocess.run([orc.REF_BIN] + args, cwd=cwd, capture_output=True, text=True, check=True)
    return r.stdout, r.stderr


def main():
    orc.build_oracle(with_ref=True)
    R = orc.Reference()
    P = param_sets(R)
    # ---------------- parameters (psmc_update_hmm KATs, core.c:61-133)
    np.savez_compressed(os.path.join(HERE, "hmm_params.npz"),
                        **{"%s.%s" % (k, f): np.asarray(v) for k, d in P.items() for f, v in d.items()})
    # ---------------- observation segments drawn from the n64 'curve' model (our own simulator)
    g = P["n64_curve"]
    rng = np.random.default_rng(20260926)
    lens = [1, 2, 3, 63, 64, 65, 127, 129, 1000, 4097, 20000]
    segs = [sim.simulate_segment(g["a"], g["e"], g["a0"], L, rng, miss_rate=0.002) for L in lens]
    segs.append(np.full(300, 2, dtype=np.uint8))                 # all missing
    segs.append(np.tile(np.array([0, 1], dtype=np.uint8), 100))  # alternating hom/het
    np.savez_compressed(os.path.join(HERE, "segments_small.npz"), **{"s%02d" % i: s for i, s in enumerate(segs)})
    # ---------------- E-step goldens (em.c:33-55) for every parameter set
    est = {}
    for k, d in P.items():
        r = R.estep(d["a"], d["e"], d["a0"], segs, per_seg=True)
        for f, v in r.items():
            if f == "seg_A":      # keep the fixture small: three segments' full he->A + everyone's row sums
                est["%s.seg_A_pick" % k] = v[[0, 5, 9]]
                est["%s.seg_A_rowsum" % k] = v.sum(2)
            else:
                est["%s.%s" % (k, f)] = np.asarray(v)
    # full forward/backward tables of one short segment (khmm.c:145-241)
    for k in ("n64_curve", "n23_flat"):
        d = P[k]
        f, b, s, lk = R.fwd_bwd(d["a"], d["e"], d["a0"], segs[5])
        est["%s.f65" % k] = f; est["%s.b65" % k] = b; est["%s.s65" % k] = s; est["%s.lk65" % k] = np.array(lk)
    np.savez_compressed(os.path.join(HERE, "estep_small.npz"), **est)
    # ---------------- a mid-size genome-like batch: summed statistics only
    rng = np.random.default_rng(7)
    lens2 = [60000, 35000, 20000, 12000, 5000, 800]
    segs2 = [sim.simulate_segment(g["a"], g["e"], g["a0"], L, rng) for L in lens2]
    np.savez_compressed(os.path.join(HERE, "segments_mid.npz"), **{"s%02d" % i: s for i, s in enumerate(segs2)})
    mid = {}
    for k in ("n64_curve", "n64_flat"):
        d = P[k]
        r = R.estep(d["a"], d["e"], d["a0"], segs2, per_seg=True)
        mid["%s.A" % k] = r["A"]; mid["%s.E" % k] = r["E"]; mid["%s.LL" % k] = np.array(r["LL"])
        mid["%s.seg_LL" % k] = r["seg_LL"]; mid["%s.seg_chk" % k] = r["seg_chk"]
    # one whole EM round from the flat start (E + M step; em.c:27-78)
    d = P["n64_flat"]
    em = R.em_round(PAT64, d["params"], segs2)
    for f, v in em.items():
        mid["em_n64_flat.%s" % f] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, "estep_mid.npz"), **mid)
    # ---------------- pattern / resample / reader / minimiser KATs
    kat = {}
    for pat in (PAT23, PAT64, PAT128, "1+1+1", "2*3+1*4+7"):
        n, nf, pm = R.parse_pattern(pat)
        kat["pattern.%s" % pat] = np.concatenate([[n, nf], pm])
    L500 = np.array([500000] * 55 + [612345, 731000, 500001, 499999, 250000, 1], dtype=np.int32)
    for seed in (1, 42, 1000):
        kat["resample.%d" % seed] = R.resample(seed, L500)
    kat["resample.lens"] = L500
    x, fx = R.kmin_quad(np.array([3.0, -2.0, 0.5, 0.0, 10.0]), np.array([1.0, 2.0, -3.0, 0.25, 0.0]))
    kat["kmin.x"] = x; kat["kmin.fx"] = np.array(fx)
    np.savez_compressed(os.path.join(HERE, "host_kats.npz"), **kat)
    # ---------------- end-to-end .psmc outputs of the reference binary
    cli = os.path.join(HERE, "cli")
    os.makedirs(cli, exist_ok=True)
    with open(os.path.join(cli, "t10k.psmcfa"), "w") as fh:
        fh.write(t10k_text())
    with gzip.open(os.path.join(cli, "mid.psmcfa.gz"), "wt") as fh:
        fh.write(to_psmcfa(segs2))
    with open(os.path.join(cli, "small.psmcfa"), "w") as fh:
        fh.write(to_psmcfa([segs[8], segs[9], segs[3]], names=["a", "b", "c"]))
    runs = {
        "t10k_default_N5": ["-N5", "t10k.psmcfa"],
        "t10k_n64_N3": ["-N3", "-t15", "-r5", "-p", PAT64, "t10k.psmcfa"],
        "mid_n64_N4": ["-N4", "-t15", "-r5", "-p", PAT64, "mid.psmcfa.gz"],
        "small_decode_d": ["-N2", "-d", "small.psmcfa"],
        "small_decode_D": ["-N1", "-D", "small.psmcfa"],
        "small_decode_s": ["-N1", "-s", "small.psmcfa"],
        "small_T": ["-N2", "-T", "0.5", "small.psmcfa"],
    }
    runs.update(cli_decode_c_cases(cli, segs))
    for name, args in runs.items():
        out, err = run_ref(args, cli)
        if len(out) > 65536:   # the -D dump is large: keep it gzipped
            with gzip.open(os.path.join(cli, name + ".psmc.gz"), "wt") as fh:
                fh.write(out)
        else:
            with open(os.path.join(cli, name + ".psmc"), "w") as fh:
                fh.write(out)
        with open(os.path.join(cli, name + ".args"), "w") as fh:
            fh.write(" ".join(args) + "\n")
    # restart from a PA line (-i), aux.c:84-113
    pa = [l for l in open(os.path.join(cli, "t10k_default_N5.psmc")) if l.startswith("PA\t")][-1][3:]
    with open(os.path.join(cli, "t10k_restart.par"), "w") as fh:
        fh.write(pa)
    out, err = run_ref(["-N2", "-i", "t10k_restart.par", "t10k.psmcfa"], cli)
    open(os.path.join(cli, "t10k_restart_N2.psmc"), "w").write(out)
    open(os.path.join(cli, "t10k_restart_N2.args"), "w").write("-N2 -i t10k_restart.par t10k.psmcfa\n")
    rd = R.read_psmcfa(os.path.join(cli, "t10k.psmcfa"))
    np.savez_compressed(os.path.join(HERE, "reader_t10k.npz"), seq=rd["segs"][0], L=rd["L"], L_e=rd["L_e"],
                        n_e=rd["n_e"], sum_L=np.array(rd["sum_L"]), sum_n=np.array(rd["sum_n"]))
    tot = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(HERE) for f in fs)
    print("golden fixtures written, %.1f KB total" % (tot / 1024))


if __name__ == "__main__":
    main()
