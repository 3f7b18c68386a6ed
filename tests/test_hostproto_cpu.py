"""Host arithmetic of the batch verifier and the worker pool (dot_ring_amd/csrc/hostproto.hpp), compiled with plain g++: parallel_for visits
every index exactly once for any size and grain, also from two posting threads, and carries an item's exception to the caller; the two routines that were split
around their field inversion so that sixteen proofs share one (te_add_affine, ring_verifier_terms) must give what the unsplit
routines give, batch_inv must equal single inversions and keep zeros, and an evaluation point inside the domain must be refused by
the first half already (tests/native/hostproto_check.cpp)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_split_verifier_arithmetic_matches_the_unsplit_routines(tmp_path):
    exe = tmp_path / "hostproto_check"
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "dot_ring_amd", "csrc"),
                    os.path.join(ROOT, "tests", "native", "hostproto_check.cpp"), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("ok "), out.stdout + out.stderr
    assert int(out.stdout.split()[1]) > 1000
