"""ia_advance (csrc/t_advance.h): the closed form of k iterated float additions used by the traversal's expansion pass
must reproduce the plain recurrence bit for bit -- fuzzed on the CPU (same header, gcc, no FMA contraction), including
the round-half-even tie cases (steps with trailing-zero mantissas), t = 0 starts, binade crossings and stuck sums."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_advance_closed_form_is_bit_exact(tmp_path):
    exe = str(tmp_path / "fuzz_advance")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-I", os.path.join(ROOT, "intrinsicavatar_amd", "csrc"),
                           "-o", exe, os.path.join(ROOT, "tests", "native", "fuzz_advance.c"), "-lm"])
    out = subprocess.run([exe, "3000000"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:]
    assert "0 mismatches" in out.stdout
