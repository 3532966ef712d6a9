"""INTEGRATION.md section B is the reference-side binding of the drop-in boundary (what a maintainer of lh3/psmc would
paste into em.c).  Prose rots: this test extracts the C block as written, compiles it against the reference's OWN
headers (psmc.h, khmm.h) and our public header, and links it with libpsmc_hip.so -- a signature drift of
psmc_hip_create / _load_segments / _estep / _last_error, or a renamed field of the reference's structs, fails here.
Build container only: the reference checkout does not travel to the GPU box."""
import os
import re
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
BUILD = "/tmp/psmc_test_build"


def c_blocks(section_title):
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text.split(section_title, 1)[1]
    sec = re.split(r"\n## ", sec, 1)[0]
    return re.findall(r"```c\n(.*?)```", sec, flags=re.S)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout absent (GPU box)")
def test_section_b_snippet_compiles_against_reference_headers_and_links():
    blocks = c_blocks("## B. Keep the reference's host code")
    assert blocks and "hip_estep" in blocks[0]
    os.makedirs(BUILD, exist_ok=True)
    src = os.path.join(BUILD, "integration_b.c")
    with open(src, "w") as f:
        f.write("#include <stdio.h>\n#include <stdlib.h>\n#include <string.h>\n#include <stdint.h>\n"
                '#include "psmc.h"\n#include "khmm.h"\n')
        f.write(blocks[0])
        # the call site the comment at the end of the block describes (em.c:33 and :56 onward stay the reference's)
        f.write("\nint main(int argc, char **argv)\n{\n\tpsmc_par_t *pp = 0; psmc_data_t *pd = 0; hmm_exp_t *he_sum = 0; double LL = 0;\n"
                "\tif (argc > 100) hip_estep(pp, pd, he_sum, &LL);\n\treturn 0;\n}\n")
    lib = os.path.join(ROOT, "psmc_amd")
    assert os.path.exists(os.path.join(lib, "libpsmc_hip.so")), "build the library first (__graft_entry__.build)"
    exe = os.path.join(BUILD, "integration_b")
    r = subprocess.run(["gcc", "-std=gnu99", "-Wall", "-Werror=implicit-function-declaration", "-Werror=incompatible-pointer-types",
                        "-Werror=int-conversion", "-I" + REF, "-I" + os.path.join(ROOT, "include"), "-o", exe, src,
                        "-L" + lib, "-lpsmc_hip", "-Wl,-rpath," + lib], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # every psmc_hip_* symbol the snippet calls is declared in the header and exported by the library
    used = set(re.findall(r"\b(psmc_hip_[a-z_]+)\s*\(", blocks[0]))
    nm = subprocess.run(["nm", "-D", "--defined-only", os.path.join(lib, "libpsmc_hip.so")], capture_output=True, text=True).stdout
    header = open(os.path.join(ROOT, "include", "psmc_hip.h")).read()
    for s in used:
        assert re.search(r"\b%s\b" % s, nm), s
        assert re.search(r"\b%s\s*\(" % s, header), s


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout absent (GPU box)")
def test_section_b_and_d_fragments_name_real_entry_points():
    """The one-line fragments (decode calls, batch and group bindings) are not complete C; check that every entry point
    they name exists with the argument count shown."""
    header = open(os.path.join(ROOT, "include", "psmc_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    frags = "".join(re.findall(r"```c\n(.*?)```", text, flags=re.S)[1:])
    frags = re.sub(r"/\*.*?\*/", "", frags, flags=re.S)
    for m in re.finditer(r"\b(psmc_hip_[a-z_]+)\s*\(([^;]*?)\);", frags, flags=re.S):
        name, args = m.group(1), m.group(2)
        decl = re.search(r"\b%s\s*\(([^;]*?)\);" % name, header, flags=re.S)
        assert decl, name
        assert len(args.split(",")) == len(decl.group(1).split(",")), (name, args, decl.group(1))
