"""CPU: the scalar recoders of the G1 Pippenger (dot_ring_amd/csrc/msm_recode.hip.h), compiled for the host as they are.
tests/native/recode_check.cpp runs for_each_wnaf_digit (width-w non-adjacent form over bit-row tables, w = 9..13) and for_each_digit
(signed windows, c = 7..16) on the edges of the field, runs of ones across every slot boundary, alternating patterns, 2^w - 1 and the
sign thresholds at every position, and 200 000 random scalars: digits odd and in range, one per slot, w positions apart, rows < 256,
and sum d 2^position = k."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_scalar_recoders_reconstruct_every_scalar(tmp_path):
    exe = tmp_path / "recode_check"
    subprocess.run(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "dot_ring_amd", "csrc"),
                    os.path.join(ROOT, "tests", "native", "recode_check.cpp"), "-o", str(exe)], check=True)
    proc = subprocess.run([str(exe)], capture_output=True, text=True)
    assert proc.returncode == 0, proc.stderr
    assert "recoding ok" in proc.stdout
