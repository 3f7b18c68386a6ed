"""ORACLE (test infrastructure only) — ctypes access to oracle/liboracle.so and oracle/_ref.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (dot_ring_amd) must never import anything under oracle/.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_REF_PATH = os.path.join(_HERE, "_ref", "libbls_scalar_ref.so")

FR_P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
FP_P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
BSN_N = 0x1CFB69D4CA675F520CCE760202687600FF8F87007419047174FD06B52876E7E1


def build(force: bool = False) -> None:
    """Compile the C restatement (and oracle/_ref when /root/reference is present)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.run(["make", "-C", os.path.join(_HERE, "c")], check=True, capture_output=True)


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.orc_fr_sqrt.restype = ctypes.c_int
        _lib.orc_ntt.restype = ctypes.c_int
        _lib.orc_g1_on_curve.restype = ctypes.c_int
        _lib.orc_g1_recover_y.restype = ctypes.c_int
        _lib.orc_te_pippenger_window.restype = ctypes.c_int
        _lib.orc_te_pippenger_window.argtypes = [ctypes.c_size_t]
    return _lib


def ref_lib() -> ctypes.CDLL | None:
    """The reference's own bls12_381_scalar.c compiled where it lies (oracle/_ref), or None."""
    if not os.path.exists(_REF_PATH):
        return None
    return ctypes.CDLL(_REF_PATH)


def _b32(v: int) -> bytes:
    return int(v).to_bytes(32, "little")


def _b48(v: int) -> bytes:
    return int(v).to_bytes(48, "little")


def _out(n: int):
    return ctypes.create_string_buffer(n)


# ---------------------------------------------------------------- Fr
def fr_add(a: int, b: int) -> int:
    o = _out(32)
    lib().orc_fr_add(_b32(a), _b32(b), o)
    return int.from_bytes(o.raw, "little")


def fr_sub(a: int, b: int) -> int:
    o = _out(32)
    lib().orc_fr_sub(_b32(a), _b32(b), o)
    return int.from_bytes(o.raw, "little")


def fr_mul(a: int, b: int) -> int:
    o = _out(32)
    lib().orc_fr_mul(_b32(a), _b32(b), o)
    return int.from_bytes(o.raw, "little")


def fr_mul_mont_raw(a: int, b: int) -> int:
    o = _out(32)
    lib().orc_fr_mul_mont_raw(_b32(a), _b32(b), o)
    return int.from_bytes(o.raw, "little")


def fr_inv(a: int) -> int:
    o = _out(32)
    lib().orc_fr_inv(_b32(a), o)
    return int.from_bytes(o.raw, "little")


def fr_pow(a: int, e: int) -> int:
    o = _out(32)
    lib().orc_fr_pow(_b32(a), _b32(e), o)
    return int.from_bytes(o.raw, "little")


def fr_sqrt(a: int) -> int | None:
    o = _out(32)
    ok = lib().orc_fr_sqrt(_b32(a), o)
    return int.from_bytes(o.raw, "little") if ok else None


# ---------------------------------------------------------------- Bandersnatch (affine ints in/out)
def te_pack(points) -> bytes:
    return b"".join(_b32(x) + _b32(y) for x, y in points)


def te_unpack(raw: bytes):
    return [
        (int.from_bytes(raw[i : i + 32], "little"), int.from_bytes(raw[i + 32 : i + 64], "little"))
        for i in range(0, len(raw), 64)
    ]


def scalars_pack(scalars) -> bytes:
    return b"".join(_b32(s) for s in scalars)


def te_add(p, q):
    o = _out(64)
    lib().orc_te_add(te_pack([p]), te_pack([q]), o)
    return te_unpack(o.raw)[0]


def te_mul(p, k: int, glv: bool = False):
    o = _out(64)
    fn = lib().orc_te_mul_glv if glv else lib().orc_te_mul_naive
    fn(te_pack([p]), _b32(k % BSN_N), o)
    return te_unpack(o.raw)[0]


def te_mul2(p1, k1: int, p2, k2: int):
    o = _out(64)
    lib().orc_te_mul2_w2(te_pack([p1]), _b32(k1), te_pack([p2]), _b32(k2), o)
    return te_unpack(o.raw)[0]


def te_mul_batch_raw(pts: bytes, ks: bytes, n: int, glv: bool = True) -> bytes:
    o = _out(64 * n)
    lib().orc_te_mul_batch(pts, ks, ctypes.c_size_t(n), o, ctypes.c_int(1 if glv else 0))
    return o.raw


def te_msm(points, scalars, window_bits: int = 0):
    n = len(points)
    o = _out(64)
    lib().orc_te_msm(te_pack(points), scalars_pack([s % BSN_N for s in scalars]), ctypes.c_size_t(n), ctypes.c_int(window_bits), o)
    return te_unpack(o.raw)[0]


# ---------------------------------------------------------------- NTT
def ntt(values, omega: int, scale: int | None = None):
    n = len(values)
    buf = ctypes.create_string_buffer(b"".join(_b32(v % FR_P) for v in values), 32 * n)
    rc = lib().orc_ntt(buf, ctypes.c_size_t(n), _b32(omega), _b32(scale) if scale is not None else None)
    if rc != 0:
        raise ValueError("orc_ntt: size must be a power of two >= 2")
    raw = buf.raw
    return [int.from_bytes(raw[i : i + 32], "little") for i in range(0, 32 * n, 32)]


def ntt_raw(data: bytes, n: int, omega: int, scale: int | None = None) -> bytes:
    buf = ctypes.create_string_buffer(data, 32 * n)
    rc = lib().orc_ntt(buf, ctypes.c_size_t(n), _b32(omega), _b32(scale) if scale is not None else None)
    if rc != 0:
        raise ValueError("orc_ntt: size must be a power of two >= 2")
    return buf.raw


# ---------------------------------------------------------------- G1 (affine ints in/out; None = infinity)
def g1_pack(points) -> bytes:
    return b"".join((b"\0" * 96) if p is None else _b48(p[0]) + _b48(p[1]) for p in points)


def g1_unpack1(raw: bytes):
    if raw == b"\0" * 96:
        return None
    return int.from_bytes(raw[:48], "little"), int.from_bytes(raw[48:96], "little")


def g1_add(p, q):
    o = _out(96)
    lib().orc_g1_add(g1_pack([p]), g1_pack([q]), o)
    return g1_unpack1(o.raw)


def g1_mul(p, k: int):
    o = _out(96)
    lib().orc_g1_mul(g1_pack([p]), _b32(k), o)
    return g1_unpack1(o.raw)


def g1_on_curve(p) -> bool:
    return bool(lib().orc_g1_on_curve(g1_pack([p])))


def g1_recover_y(x: int, larger: bool) -> int | None:
    o = _out(48)
    ok = lib().orc_g1_recover_y(_b48(x), ctypes.c_int(1 if larger else 0), o)
    return int.from_bytes(o.raw, "little") if ok else None


def g1_msm_raw(pts: bytes, ks: bytes, n: int, window_bits: int = 0) -> bytes:
    o = _out(96)
    inf = ctypes.c_int(0)
    lib().orc_g1_msm(pts, ks, ctypes.c_size_t(n), ctypes.c_int(window_bits), o, ctypes.byref(inf))
    return o.raw


def g1_msm(points, scalars, window_bits: int = 0):
    return g1_unpack1(g1_msm_raw(g1_pack(points), scalars_pack(scalars), len(points), window_bits))


def g1_msm_naive(points, scalars):
    o = _out(96)
    lib().orc_g1_msm_naive(g1_pack(points), scalars_pack(scalars), ctypes.c_size_t(len(points)), o)
    return g1_unpack1(o.raw)
