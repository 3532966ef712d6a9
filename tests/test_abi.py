"""CPU-side checks of the C-ABI library: it builds, loads, exports every symbol
include/psmc_hip.h declares, and fails loudly (no CPU fallback) without a GPU."""
import ctypes as C
import os
import re
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "psmc_amd", "csrc")], check=True)
    from psmc_amd import hip
    return hip.load_library()


def declared_symbols(header="psmc_hip.h"):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(psmc_hip_[a-z_0-9]+)\s*\(", txt)))


def _exported(path):
    out = subprocess.run(["nm", "-D", "--defined-only", path], check=True, capture_output=True, text=True).stdout
    return set(re.findall(r"\bT (psmc_hip_[a-z_0-9]+)$", out, flags=re.M))


def test_exports_match_header(lib):
    """The product library exports the E-step ABI and NOTHING of the lab bench; the diagnostics live in
    libpsmc_hip_diag.so with a header of their own."""
    from psmc_amd import hip
    names = declared_symbols()
    assert names == sorted(hip.EXPORTS)
    for nm in names:
        assert hasattr(lib, nm), nm
    assert _exported(hip.lib_path()) == set(names)
    dnames = declared_symbols("psmc_hip_diag.h")
    assert dnames == sorted(hip.DIAG_EXPORTS)
    assert not set(dnames) & set(names)
    diag = hip.load_diag()
    for nm in dnames:
        assert hasattr(diag, nm), nm
    assert _exported(os.path.join(os.path.dirname(hip.lib_path()), "libpsmc_hip_diag.so")) == set(dnames)


def test_no_silent_cpu_fallback(lib):
    from psmc_amd import hip
    if lib.psmc_hip_device_count() > 0:
        pytest.skip("a GPU is visible")
    h = C.c_void_p()
    assert lib.psmc_hip_create(C.byref(h), 64, 0, 0) < 0 and not h.value
    with pytest.raises(hip.HipError):
        hip.HipEStep(64)
    assert lib.psmc_hip_strerror(-3).decode() == "HIP runtime error"


def test_argument_validation(lib):
    h = C.c_void_p()
    assert lib.psmc_hip_create(C.byref(h), 0, 0, 0) == -1       # EINVAL
    assert lib.psmc_hip_create(C.byref(h), 1025, 0, 0) == -4    # ENOTSUP: beyond PSMC_HIP_MAX_STATES (one thread per state in a work-group)
    assert lib.psmc_hip_create(C.byref(h), 64, 0, 7) == -1
    assert lib.psmc_hip_create(None, 64, 0, 0) == -1


_EXACT_ASM = []


def _exact_asm():
    """gfx950 assembly of the exact kernels (compiled once per test session: ~30 s)"""
    if not _EXACT_ASM:
        out = "/tmp/psmc_exact_audit_%d.s" % os.getpid()
        src = os.path.join(ROOT, "psmc_amd", "csrc", "estep_exact.hip")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                        "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", out, src],
                       check=True, stderr=subprocess.DEVNULL)
        _EXACT_ASM.append(open(out).read())
        os.unlink(out)
    return _EXACT_ASM[0]


_WIDE_ASM = []


def _wide_asm():
    if not _WIDE_ASM:
        out = "/tmp/psmc_wide_audit_%d.s" % os.getpid()
        src = os.path.join(ROOT, "psmc_amd", "csrc", "estep_wide.hip")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                        "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", out, src],
                       check=True, stderr=subprocess.DEVNULL)
        _WIDE_ASM.append(open(out).read())
        os.unlink(out)
    return _WIDE_ASM[0]


def test_exact_kernels_have_no_fma():
    """hipcc contracts a*b+c by default; the exact mode must never contain an FMA outside the IEEE division expansion
     -- with ONE deliberate exception that is checked here operand by operand: the ordered sums
    s = s + x_k are issued as v_fmac_f64_dpp s, x, ONE (wave_prims.h add_bcast16), and fma(x, 1.0, s) is the IEEE sum bit
    for bit.  Every such instruction must multiply by a register that the kernel loads with the constant 1.0 and never
    writes otherwise."""
    txt = _exact_asm()
    n_div = len(re.findall(r"v_div_fmas_f64", txt))
    n_fma = len(re.findall(r"\bv_fma_f64|\bv_fmac_f64(?!_dpp)", txt))
    assert n_div > 0
    # each IEEE f64 division expands to v_div_scale x2, v_rcp, 5-6 v_fma, v_div_fmas, v_div_fixup
    assert n_fma <= 7 * n_div, (n_fma, n_div)
    assert "v_mfma" not in txt and "v_pk_fma" not in txt
    # the wide kernels (129 .. 1024 states, estep_wide.hip) have no deliberate exception at all
    wide = _wide_asm()
    n_div_w = len(re.findall(r"v_div_fmas_f64", wide))
    assert n_div_w > 0
    assert len(re.findall(r"\bv_fma_f64|\bv_fmac_f64", wide)) <= 7 * n_div_w
    assert "v_mfma" not in wide and "v_pk_fma" not in wide
    n_sum = 0
    wr = re.compile(r"(?:v_|ds_|global_load|buffer_load|flat_load|scratch_load)\w*\s+(v\d+|v\[\d+:\d+\])")
    def regs(tok):
        m = re.match(r"v\[(\d+):(\d+)\]", tok)
        return set(range(int(m.group(1)), int(m.group(2)) + 1)) if m else {int(tok[1:])}
    for fn in re.split(r"^\.Lfunc_end\d+:", txt, flags=re.M):
        lines = [l.strip() for l in fn.split("\n")]
        uses = [(i, re.match(r"v_fmac_f64_dpp (v\[\d+:\d+\]), (v\[\d+:\d+\]), (v\[\d+:\d+\])", l)) for i, l in enumerate(lines)]
        uses = [(i, m.group(3)) for i, m in uses if m]
        n_sum += len(uses)
        for reg in set(r for _, r in uses):
            first, last = min(i for i, r in uses if r == reg), max(i for i, r in uses if r == reg)
            loads = [i for i, l in enumerate(lines) if l == "v_mov_b64_e32 %s, 1.0" % reg]
            assert loads and min(loads) < first, (reg, "not loaded with 1.0 before its first use")
            # from that load to the last sum that multiplies by it (text order covers the loops in between) nothing else writes the pair
            for l in lines[min(loads) + 1:last]:
                m = wr.match(l)
                if m and not l.startswith("v_fmac_f64_dpp") and (regs(m.group(1)) & regs(reg)):
                    assert l == "v_mov_b64_e32 %s, 1.0" % reg, (reg, l)
    assert n_sum > 0 and n_sum % 16 == 0


def test_recompute_consumers_keep_in_flight_registers_untouched():
    """k_expect_exact_rf2's consumers hand registers with LDS loads IN FLIGHT from one asm block to the next (estep_exact.hip,
    PSMC_CPOS).  That is sound only while nothing but asm stands between the blocks of a position: checked on the compiled code."""
    L = _exact_asm().split("\n")
    blocks, i = [], 0
    while i < len(L):
        if "ASMSTART" in L[i]:
            j = i
            while "ASMEND" not in L[j]: j += 1
            if any(".Lc4j_" in x for x in L[i:j]): blocks.append((i, j))
            i = j
        i += 1
    assert len(blocks) >= 2 * 256      # 2 entries x 16 positions x 8 blocks, for each of the two rep_impl instances
    n_handover = 0
    for (i0, j0), (i1, j1) in zip(blocks, blocks[1:]):
        body1 = [x.strip() for x in L[i1 + 1:j1] if x.strip()]
        if body1[0].startswith("s_waitcnt lgkmcnt(0)") and body1[1].startswith("ds_read"): continue   # first block of a position: nothing in flight
        if body1[0].startswith("s_waitcnt lgkmcnt(0)"):
            pass                                                                                       # last block: waits for everything first
        infl = [re.match(r"\s*ds_read_b64 (v\[\d+:\d+\])", x).group(1) for x in L[i0 + 1:j0] if x.strip().startswith("ds_read_b64")][-4:]
        between = [x.strip() for x in L[j0 + 1:i1] if x.strip() and not x.strip().startswith(";")]
        assert all(x.startswith("s_nop") for x in between), between
        assert len(infl) == 4
        n_handover += 1
    assert n_handover >= 2 * 2 * 16 * 7


def test_every_option_is_documented():
    """Each key psmc_hip_set_option accepts is described in include/psmc_hip.h (the ABI's only documentation)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    api = open(os.path.join(root, "psmc_amd", "csrc", "api.hip")).read()
    hdr = open(os.path.join(root, "include", "psmc_hip.h")).read()
    keys = re.findall(r'k == "([a-z_0-9]+)"', api)
    assert len(keys) >= 20
    missing = [k for k in keys if '"%s"' % k not in hdr]
    assert not missing, missing
