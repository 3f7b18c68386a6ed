"""`dot_ring.curve.native_field.bandersnatch_te` over the HIP kernels — same names, same argument order, same return shapes.

The reference module (Cython, bandersnatch_te.pyx) is imported by `dot_ring/curve/glv.py:7-21` and
`dot_ring/curve/specs/bandersnatch.py:9`; its functions take Python ints (projective coordinates X, Y, Z, T of extended
twisted Edwards points, curve coefficients a, d and the field modulus p) and return projective tuples that the callers
normalise (`glv.py:243-248`), so results need only be projectively equal to the reference's: these return (x, y, 1, x*y).
Every call is one launch of `dr_bsn_msm_groups` / `dr_bsn_msm` (kernels K3 / K4); for throughput the better seam is one
level up — whole batches through `dr_bsn_scalar_mul_batch` (dot_ring_amd.curve.scalar_mul_batch).
"""
from __future__ import annotations

from .. import _native, runtime

_P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
_N = 0x1CFB69D4CA675F520CCE760202687600FF8F87007419047174FD06B52876E7E1       # Bandersnatch prime-order subgroup
_A = _P - 5
_D = 0x6389C12633C267CBC66E3BF86BE3B6D8CB66677177E54F92B369F2F5188D58E7


def _check_curve(a_coeff, d_coeff, p) -> None:
    # the kernels are compiled for Bandersnatch (a = -5, d as bandersnatch.py:63-64) over the BLS12-381 scalar field —
    # the only parameters the reference ever passes (glv.py:236-243)
    if int(p) != _P or int(a_coeff) % _P != _A or int(d_coeff) % _P != _D:
        raise ValueError("bandersnatch_te_hip serves the Bandersnatch curve over the BLS12-381 scalar field only")


def _affine(x, y, z):
    z = int(z) % _P
    if z == 0:
        return 0, 1                                     # projective_to_affine_cy's convention for Z = 0
    zi = pow(z, -1, _P)
    return int(x) * zi % _P, int(y) * zi % _P


def _xy(pt) -> bytes:
    return pt[0].to_bytes(32, "little") + pt[1].to_bytes(32, "little")


def _msm(points_xy, scalars):
    """sum k_i P_i for affine points and NON-NEGATIVE scalars of up to 256 bits -> extended projective tuple with Z = 1.
    Points must lie in the prime-order subgroup (every caller in the reference passes generators, public keys or hash-to-curve
    outputs, glv.py:191-472): scalars are reduced mod n here and by the kernels, so for a point with a torsion component the
    result would differ from the reference's plain integer multiplication by a torsion point.  Not checked (a subgroup test costs
    a scalar multiplication per point)."""
    ks = [int(k) for k in scalars]
    if any(k < 0 or k >> 256 for k in ks):
        raise OverflowError("scalar out of range")      # as the reference's 4-limb conversion (bandersnatch_te.pyx:55-66)
    raw_p = b"".join(_xy(p) for p in points_xy)
    raw_k = b"".join((k % _N).to_bytes(32, "little") for k in ks)     # points of the subgroup: k and k mod n give the same point
    ctx = runtime.context()
    out = ctx.bsn_msm_groups(raw_p, raw_k, len(ks)) if len(ks) <= 64 else ctx.bsn_msm(raw_p, raw_k)
    x, y = int.from_bytes(out[:32], "little"), int.from_bytes(out[32:64], "little")
    return x, y, 1, x * y % _P


def projective_to_affine_cy(x, y, z, p):
    """bandersnatch_te.pyx:244 — (X/Z, Y/Z), (0, 1) for Z = 0."""
    if z == 0:
        return (0, 1)
    p = int(p)
    zi = pow(int(z), -1, p)
    return (int(x) * zi % p, int(y) * zi % p)


def scalar_mult_windowed_native_w2_cy(k1, k2, p1_x, p1_y, p1_z, p1_t, p2_x, p2_y, p2_z, p2_t, a_coeff, d_coeff, p):
    """bandersnatch_te.pyx:480 — k1*P1 + k2*P2."""
    _check_curve(a_coeff, d_coeff, p)
    return _msm([_affine(p1_x, p1_y, p1_z), _affine(p2_x, p2_y, p2_z)], [k1, k2])


def scalar_mult_4_native_w2_cy(k1, k2, k3, k4,
                               p1_x, p1_y, p1_z, p1_t, p2_x, p2_y, p2_z, p2_t, p3_x, p3_y, p3_z, p3_t, p4_x, p4_y, p4_z, p4_t,
                               a_coeff, d_coeff, p):
    """bandersnatch_te.pyx:557 — k1*P1 + k2*P2 + k3*P3 + k4*P4."""
    _check_curve(a_coeff, d_coeff, p)
    return _msm([_affine(p1_x, p1_y, p1_z), _affine(p2_x, p2_y, p2_z), _affine(p3_x, p3_y, p3_z), _affine(p4_x, p4_y, p4_z)],
                [k1, k2, k3, k4])


def scalar_mult_6_native_w2_cy(k1, k2, k3, k4, k5, k6,
                               p1_x, p1_y, p1_z, p1_t, p2_x, p2_y, p2_z, p2_t, p3_x, p3_y, p3_z, p3_t,
                               p4_x, p4_y, p4_z, p4_t, p5_x, p5_y, p5_z, p5_t, p6_x, p6_y, p6_z, p6_t,
                               a_coeff, d_coeff, p):
    """bandersnatch_te.pyx:669 — the 6-term MSM behind GLV-split 3-point MSMs."""
    _check_curve(a_coeff, d_coeff, p)
    return _msm([_affine(p1_x, p1_y, p1_z), _affine(p2_x, p2_y, p2_z), _affine(p3_x, p3_y, p3_z),
                 _affine(p4_x, p4_y, p4_z), _affine(p5_x, p5_y, p5_z), _affine(p6_x, p6_y, p6_z)], [k1, k2, k3, k4, k5, k6])


def msm_pippenger_signed_native_cy(points, scalars, a_coeff, d_coeff, p, window_bits=7, affine=False):
    """bandersnatch_te.pyx:257 — variable-base MSM; `points` carry affine .x / .y, `scalars` are signed Python ints (the
    caller centres them into (-n/2, n/2], bandersnatch.py:270-284): a negative scalar multiplies the negated point."""
    n = len(points)
    if n != len(scalars):
        raise ValueError("Points and scalars must have same length")
    if n == 0:
        return (0, 1) if affine else (0, 1, 1, 0)
    if window_bits < 2 or window_bits > 8:
        raise ValueError("window_bits must be between 2 and 8")
    _check_curve(a_coeff, d_coeff, p)
    pts, ks = [], []
    for pt, k in zip(points, scalars):
        k = int(k)
        x, y = int(pt.x) % _P, int(pt.y) % _P
        if k < 0:
            x, k = (_P - x) % _P, -k                    # -(x, y) = (-x, y)
        pts.append((x, y))
        ks.append(k)
    if not any(ks):
        return (0, 1) if affine else (0, 1, 1, 0)
    x, y, z, t = _msm(pts, ks)
    return (x, y) if affine else (x, y, z, t)


def sqrt_mod_bls_scalar_cy(x):
    """bandersnatch_te.pyx:421 — a square root in the Bandersnatch base field (Tonelli-Shanks); ValueError for non-squares."""
    try:
        return _native.fr_sqrt(int(x) % _P)
    except ValueError:
        raise ValueError("sqrt_mod_bls_scalar_cy received a non-square") from None
