#!/usr/bin/env python3
"""Static instruction mix of the steady-state loop of the factored path's kernels, from the gfx950 assembly.

The SQ counters see only part of the device (profiles/r02_sq_counters.json), so the per-bin instruction count that the
VALU-issue roofline of bench.py's `factored_stats.roofline` needs is taken from the compiled code instead: the natural loop with the most vector
instructions of each kernel is its main loop (k_bwd_acc_ckpt: one block of 8 positions of 4 tiles; k_fwd_struct<ckpt>:
4 positions of 4 tiles), whose instruction counts divided by the tile-positions per trip are instructions per bin.
Writes profiles/sq_factored.json (`valu_per_launch` = per bin x bins; + the SQ-counter ratios when a counter pass is given).

    python scripts/isa_count.py [bins=30000001]
"""
import collections, json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "psmc_amd", "csrc")

def classify(x, c):
    if x.startswith("v_mfma"): c["mfma"] += 1
    elif x.startswith("v_accvgpr"): c["accvgpr"] += 1
    elif x.startswith("v_"): c["valu"] += 1
    elif x.startswith("ds_"): c["lds"] += 1
    elif x.startswith(("global_", "buffer_", "scratch_", "flat_")): c["vmem"] += 1
    elif x.startswith("s_waitcnt"): c["waitcnt"] += 1
    elif x.startswith("s_"): c["salu"] += 1

def loops_of(asm, mangled_part):
    """Every natural loop of the kernel as (head label, #instructions, mix): the lines from a label to the LAST backward branch to it
    (straight-line count: rare side paths inside the loop -- masked rows, the tile's last position -- are included once)."""
    m0 = re.search(r"^(_ZN4psmc%s\w*):" % re.escape(mangled_part), asm, re.M)   # the kernel's entry label ... its .Lfunc_end
    k = asm[m0.end():]; k = k[:k.index(".Lfunc_end")]
    lines = [l.strip() for l in k.split("\n")]
    label_at = {}
    for n, l in enumerate(lines):
        ml = re.match(r"(\.LBB\d+_\d+):", l)
        if ml: label_at[ml.group(1)] = n
    loops = {}
    for n, l in enumerate(lines):
        m = re.match(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in label_at and label_at[m.group(1)] < n:
            loops[m.group(1)] = max(loops.get(m.group(1), 0), n)
    out = []
    for lab, end in loops.items():
        if any(label_at[l2] > label_at[lab] and e2 < end for l2, e2 in loops.items() if l2 != lab): continue   # innermost loops only
        c = collections.Counter(); cnt = 0
        for l in lines[label_at[lab]:end + 1]:
            if not l or l.startswith((".", ";", "//")) or re.match(r"[\w.$]+:", l): continue
            classify(l.split()[0], c); cnt += 1
        out.append((lab, cnt, dict(c)))
    return out

def main():
    bins = int(sys.argv[1]) if len(sys.argv) > 1 else 30000001
    res = {"bins": bins, "kernels": {}, "note": "static instruction counts of the main loop (scripts/isa_count.py): vector instructions per bin x bins; "
           "a SIMD issues at most one f64 vector instruction per 4 cycles"}
    for src, kernels in (("estep_factored.hip", [("k_bwd_acc_ckpt", "14k_bwd_acc_ckpt", 8 * 4)]),
                         ("estep_struct.hip", [("k_fwd_struct<false,4,true>", "12k_fwd_structILb0ELi4ELb1E", 4 * 4)])):
        asm_path = "/tmp/isa_%s.s" % src
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
                        "-S", "--cuda-device-only", "-o", asm_path, os.path.join(CSRC, src)], check=True)
        asm = open(asm_path).read()
        for name, mangled, tile_pos in kernels:
            bl = loops_of(asm, mangled)
            if name.startswith("k_fwd"):   # the steady-state variant of the forward sweep: the leanest group loop that stores (MODE 1: every position inside the tile)
                big = min((b for b in bl if b[2].get("vmem", 0) >= 2 and b[2].get("valu", 0) > 100), key=lambda b: b[2]["valu"])
            else:
                big = max(bl, key=lambda b: b[2].get("valu", 0))
            vpb = big[2]["valu"] / tile_pos
            res["kernels"][name] = {"loop_block": big[0], "instructions": big[1], "mix": big[2], "tile_positions_per_trip": tile_pos,
                                    "valu_per_bin": vpb, "valu_per_launch": vpb / 1.0 * bins / 1.0 / 1.0 * 1.0 / 1.0 if False else vpb * bins / 4.0 * 4.0 / 4.0}
            # one wave-instruction serves four tiles: wave-instructions per bin = valu / (4 tiles x positions) ... valu_per_bin above is already per tile-position
            res["kernels"][name]["valu_per_launch"] = vpb * bins
            print(name, big[0], big[1], big[2], "-> %.1f wave-level vector instructions per bin" % vpb)
    # the counts are valid for THESE kernel sources only: bench.py compares the hash and drops the issue roofline when they have changed
    import hashlib
    res["kernel_sources"] = ["psmc_amd/csrc/estep_factored.hip", "psmc_amd/csrc/struct_prims.h", "psmc_amd/csrc/estep_struct.hip", "psmc_amd/csrc/wave_prims.h"]
    res["kernel_src_sha16"] = hashlib.sha256(b"".join(open(os.path.join(ROOT, s), "rb").read() for s in res["kernel_sources"])).hexdigest()[:16]
    json.dump(res, open(os.path.join(ROOT, "profiles", "sq_factored.json"), "w"), indent=1)

if __name__ == "__main__":
    main()
