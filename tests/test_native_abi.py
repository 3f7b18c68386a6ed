"""CPU: the C-ABI library loads, exports every symbol include/dotring_hip.h declares, and its host-side
helpers (no GPU involved) agree with the oracle.  No GPU compute is attempted here."""
import os
import random
import re

import pytest

from dot_ring_amd import _native
from oracle import coracle
from oracle.pyref import kzg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "dotring_hip.h")).read()
    declared = set(re.findall(r"DR_API [^;(]*?\b(dr_\w+)\(", header))
    assert len(declared) >= 30
    lib = _native.lib()
    missing = [name for name in sorted(declared) if not hasattr(lib, name)]
    assert not missing, missing
    assert declared == set(_native.EXPORTED_SYMBOLS), declared ^ set(_native.EXPORTED_SYMBOLS)
    assert b"gfx950" in lib.dr_version()


def test_no_cpu_fallback_without_gpu():
    if _native.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_native.DotRingHipError):
        _native.Context(0)


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "dot_ring_amd")):
        for name in files:
            if name.endswith((".py", ".hip", ".hip.h", ".hpp")):
                text = open(os.path.join(dirpath, name), errors="replace").read()
                assert "oracle" not in text.replace("random-oracle", ""), f"{name} mentions the oracle"


def test_host_fr_sqrt():
    rng = random.Random(5)
    p = coracle.FR_P
    for _ in range(20):
        a = rng.randrange(1, p)
        r = _native.fr_sqrt(a * a % p)
        assert r in (a, p - a)
        assert r == coracle.fr_sqrt(a * a % p)       # same Tonelli-Shanks schedule as the reference -> same root
    assert _native.fr_sqrt(0) == 0
    with pytest.raises(ValueError):
        _native.fr_sqrt(5)
    with pytest.raises(ValueError):
        _native.fr_sqrt(p)


def test_host_g1_codecs_and_sum():
    rng = random.Random(6)
    pts = [coracle.g1_mul(kzg.G1_GEN, rng.randrange(1, coracle.FR_P)) for _ in range(6)]
    for pt in pts:
        ser = kzg.serialize(pt)
        comp = _native.g1_compress(ser)
        assert comp == kzg.compress(pt)
        assert _native.g1_decompress(comp) == ser
    assert _native.g1_compress(None) == b"\xc0" + bytes(47)
    assert _native.g1_decompress(b"\xc0" + bytes(47)) is None
    for bad in (bytes(48), b"\xe0" + bytes(47), b"\xc0" + bytes(46) + b"\x01", b"\x9f" + b"\xff" * 47):
        with pytest.raises(ValueError):
            _native.g1_decompress(bad)
    with pytest.raises(ValueError):
        _native.g1_decompress(bytes(47))
    # sum of points incl. infinity, a duplicate (doubling) and an inverse pair
    neg0 = (pts[0][0], coracle.FP_P - pts[0][1])
    terms = [pts[0], pts[1], None, pts[1], pts[2], neg0]
    want = None
    for t in terms:
        want = coracle.g1_add(want, t)
    got = _native.g1_sum([None if t is None else kzg.serialize(t) for t in terms])
    assert got == kzg.serialize(want)
    assert _native.g1_sum([kzg.serialize(pts[0]), kzg.serialize(neg0)]) is None
    assert _native.g1_sum([]) is None


def test_plain_c_consumer_of_the_abi(tmp_path):
    """The boundary is a C ABI: a C99 program includes include/dotring_hip.h, dlopens the library and calls host-side
    entry points (hashing, field square root, G1 compression) — no Python, no C++, no GPU."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "abi_smoke"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "c", "abi_smoke.c"),
                    "-ldl", "-o", str(exe)], check=True)
    out = subprocess.run([str(exe), os.path.join(root, "dot_ring_amd", "libdotring_hip.so")], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert "abi ok" in out.stdout
