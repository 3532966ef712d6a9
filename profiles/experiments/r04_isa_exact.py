#!/usr/bin/env python3
"""Static instruction mix of the innermost loops of the exact-mode kernels (estep_exact.hip), from the gfx950 assembly.

A wave alone on its SIMD -- the exact sweeps of a single E-step: one wave per segment -- issues one instruction of ANY kind
per four cycles, so the loop's instruction count x 4 cycles x positions is the floor of the sweep (DESIGN.md section 8,
"Exact mode, instruction by instruction").  Prints, per kernel, its largest natural loops: label, instructions, mix.

    python profiles/experiments/r04_isa_exact.py > profiles/r04b_exact_isa_counts.txt
"""
import os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_count as ic
ROOT = ic.ROOT
CSRC = ic.CSRC


def main():
    asm_path = "/tmp/isa_exact_%d.s" % os.getpid()
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
                    "-S", "--cuda-device-only", "-o", asm_path, os.path.join(CSRC, "estep_exact.hip")], check=True, stderr=subprocess.DEVNULL)
    asm = open(asm_path).read()
    os.unlink(asm_path)
    print("# python profiles/experiments/r04_isa_exact.py -- innermost natural loops of the exact kernels (label, instructions, mix); straight-line count:")
    print("# side paths inside a loop (the rare symbols) are included once.  k_fwd_exact<REP, STORE_F>: REP 1 = v_permlane swaps (single E-step),")
    print("# 0 = ds_bpermute (batch).  The position loops of k_expect_exact / k_expect_exact_rf2 are unrolled asm blocks: see DESIGN.md for their counts.")
    for m in re.finditer(r"^(_ZN4psmc\w+):", asm, re.M):
        name = m.group(1)
        try:
            bl = ic.loops_of(asm, name[len("_ZN4psmc"):])
        except Exception:
            continue
        for lab, cnt, mix in sorted(bl, key=lambda b: -b[1])[:2]:
            if cnt >= 100:
                print("%-56s %-10s %5d %s" % (name[8:64], lab, cnt, mix))


if __name__ == "__main__":
    main()
