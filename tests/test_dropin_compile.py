"""CPU tier: the drop-in claim, checked with the reference's own files.

(1) The UNMODIFIED reference sources that use the kept header surface -
    serial_new/sweep-tt-multistart.c, examples/example_floatbox.c, tools/vconvert.c -
    compile against THIS repository's include/ with the reference's own flags
    (serial_new/Makefile:1-3,11: cc -O3 -Wfatal-errors -I../include); example_floatbox
    also runs and prints its "passed" line (examples/example_floatbox.c:8-47).
(2) The forward of sweepXYZ that INTEGRATION.md section 1 documents is cut out of that
    file and built - around the reference translation unit, included from where it lies -
    against libttsweep.so, so the documented binding cannot rot.  Linked, not run: there
    is no GPU in the CPU tier (the same forward runs in the GPU tier as the repository's
    own host program, tests/test_host_program.py).

Nothing of the reference is stored here; the tests are skipped where /root/reference is
absent (the GPU box)."""
import os
import re
import subprocess

import pytest

from conftest import ROOT

REF = "/root/reference"
INC = os.path.join(ROOT, "include")
CSRC = os.path.join(ROOT, "uoparallel-seismic-project_amd", "csrc")

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")


def cc(args, **kw):
    return subprocess.run(["gcc"] + args, capture_output=True, text=True, **kw)


@pytest.mark.parametrize("source", ["serial_new/sweep-tt-multistart.c", "examples/example_floatbox.c",
                                    "tools/vconvert.c"])
def test_unmodified_reference_source_compiles_against_our_headers(tmp_path, source):
    exe = tmp_path / "prog"
    r = cc(["-O3", "-Wfatal-errors", "-I", INC, os.path.join(REF, source), "-o", str(exe), "-lm"])
    assert r.returncode == 0, r.stderr
    if source.endswith("example_floatbox.c"):
        run = subprocess.run([str(exe)], capture_output=True, text=True, cwd=str(tmp_path))
        assert run.returncode == 0 and "passed" in run.stdout, run.stdout + run.stderr


def integration_forward():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"/\* BEGIN sweepXYZ forward.*?/\* END sweepXYZ forward \*/", text, re.S)
    assert m, "INTEGRATION.md lost its sweepXYZ forward block"
    return m.group(0)


def test_documented_sweepxyz_forward_builds_against_the_library(tmp_path, pkg):
    """The reference TU with its own sweepXYZ renamed out of the way, then the forward
    exactly as INTEGRATION.md prints it; linked against libttsweep.so and the HIP runtime."""
    pkg._lib.build()        # incremental make
    tu = tmp_path / "patched.c"
    tu.write_text(
        "#define main reference_main\n"
        "#define sweepXYZ reference_sweepXYZ\n"
        f'#include "{REF}/serial_new/sweep-tt-multistart.c"\n'
        "#undef main\n"
        "#undef sweepXYZ\n"
        '#include "ttsweep.h"\n'
        "_Static_assert(sizeof(struct FS) == sizeof(ttsweep_fs), \"struct FS layout\");\n"
        "_Static_assert(sizeof(struct START) == sizeof(ttsweep_start), \"struct START layout\");\n"
        + integration_forward() +
        "\nint main(void) {\n"
        "  int anychange = 1;\n"
        "  while (anychange) anychange = sweepXYZ(4, 4, 4, 0, 0, 1);\n"
        "  return 0;\n"
        "}\n")
    exe = tmp_path / "patched"
    r = cc(["-O3", "-Wfatal-errors", "-I", INC, str(tu), "-o", str(exe), "-L", CSRC, "-lttsweep",
            "-L/opt/rocm/lib", "-lamdhip64", "-lm", f"-Wl,-rpath,{CSRC}", "-Wl,-rpath,/opt/rocm/lib"])
    assert r.returncode == 0, r.stderr
    nm = subprocess.run(["nm", "-u", str(exe)], capture_output=True, text=True).stdout
    for sym in ("ttsweep_create", "ttsweep_set_velocity", "ttsweep_solve", "ttsweep_last_error"):
        assert sym in nm, sym
