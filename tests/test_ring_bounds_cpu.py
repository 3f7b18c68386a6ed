"""CPU: the lazy-reduction bookkeeping of the NTT / constraint / polynomial kernels, checked for the WORST case.
dot_ring_amd/csrc/ring_body.hip.h holds the arithmetic bodies the device kernels run, written over an abstract field type;
tests/native/ring_bounds_check.cpp instantiates them with an interval type (range of the limbs, range of the value in units of p)
whose operations assert the preconditions csrc/fr29.hip.h states (limb products <= 2^59.3, |a b| <= 35 p^2, int32 limbs, the
input ranges of reduce_small / canon29 / canon29_small).  A violated precondition makes the checker exit non-zero."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ring_kernel_bodies_keep_their_bounds(tmp_path):
    exe = tmp_path / "ring_bounds_check"
    subprocess.run(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "dot_ring_amd", "csrc"),
                    os.path.join(ROOT, "tests", "native", "ring_bounds_check.cpp"), "-o", str(exe)], check=True)
    proc = subprocess.run([str(exe)], capture_output=True, text=True)
    assert proc.returncode == 0, proc.stderr
    assert "all bounds hold" in proc.stdout
    for name in ("constraints (Bandersnatch)", "constraints (JubJub)", "constraints (hidden rows)", "quotient", "horner", "agg8", "lin3", "ntt (normal input)",
                 "ntt (constraint-kernel input)", "ntt (16 stages, raw output)"):
        assert name in proc.stdout
