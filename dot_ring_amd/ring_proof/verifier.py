"""Ring-proof verifier scalar pass (dot_ring/ring_proof/verify.py:51-210): replays the transcript, evaluates the
constraint system at zeta in closed form and returns the two linear KZG claims that the PCS folds on the GPU."""
from __future__ import annotations

from .pcs import LinearPcsVerification
from .transcript import derive_challenges_after_vk

_INV_N: dict = {}


def batch_inverse(values, prime):
    """Montgomery's trick: all inverses with ONE modular inversion (zeros map to 0)."""
    prefix, run = [], 1
    for v in values:
        prefix.append(run)
        if v:
            run = run * v % prime
    inv_run = pow(run, -1, prime)
    out = [0] * len(values)
    for i in range(len(values) - 1, -1, -1):
        v = values[i]
        if v:
            out[i] = inv_run * prefix[i] % prime
            inv_run = inv_run * v % prime
    return out


def zeta_denominators(zeta, domain, prime):
    """The three values the scalar pass inverts: zeta - 1, zeta - w^(n-4), zeta^n - 1."""
    return (zeta - 1) % prime, (zeta - domain[-4]) % prime, (pow(zeta, len(domain), prime) - 1) % prime


def quotient_and_linearization_terms(alphas, nus, zeta, evals, l_zeta_omega, seed, result_seed, domain, edwards_a, omega, prime,
                                     inverses=None):
    """inverses = (1/(zeta-1), 1/(zeta-w^(n-4)), 1/(zeta^n-1)) when the caller batch-inverted them."""
    if len(alphas) < 7 or len(nus) < 8:
        raise ValueError("expected at least 7 alpha values and 8 aggregation values")
    p = prime
    pxz, pyz, sz, bz, ipz, axz, ayz = evals
    n = len(domain)
    z1, d4, zn1 = zeta_denominators(zeta, domain, p)
    if inverses is None:
        inverses = batch_inverse([z1, d4, zn1], p)
    inv_z1, inv_d4, inv_zn1 = inverses
    inv_n = pow(n, -1, p) if n not in _INV_N else _INV_N[n]
    _INV_N[n] = inv_n
    # L_0(zeta) = (zeta^n - 1) / (n (zeta - 1)),  L_{n-4}(zeta) = w^{n-4} (zeta^n - 1) / (n (zeta - w^{n-4}))
    l0 = 1 if z1 == 0 else inv_n * zn1 % p * inv_z1 % p
    ln = 1 if d4 == 0 else domain[-4] * inv_n % p * zn1 % p * inv_d4 % p
    one_b = (1 - bz) % p
    c_values = [
        -(ipz + bz * sz) * d4,
        (bz * -(axz * ayz + pxz * pyz) + one_b * -axz) * d4,
        (bz * -(axz * ayz - pxz * pyz) + one_b * -ayz) * d4,
        bz * one_b,
        (axz - seed.x) * l0 + (axz - result_seed.x) * ln,
        (ayz - seed.y) * l0 + (ayz - result_seed.y) * ln,
        ipz * l0 + (ipz - 1) * ln,
    ]
    lin = sum(a * c for a, c in zip(alphas, c_values)) % p
    tail = (zeta - domain[-1]) * (zeta - domain[-2]) % p * (zeta - domain[-3]) % p
    if zn1 == 0:
        raise ValueError("evaluation point lies in the domain")
    q_zeta = (lin + l_zeta_omega) * tail % p * inv_zn1 % p
    agg_zeta = sum(nu * v for nu, v in zip(nus, (*evals, q_zeta))) % p
    fx = (bz * (ayz * pyz + edwards_a * axz * pxz) + one_b) % p
    fy = (bz * (axz * pyz - pxz * ayz) + one_b) % p
    return (agg_zeta, alphas[0] * d4 % p, alphas[1] * fx % p * d4 % p, alphas[2] * fy % p * d4 % p,
            zeta * omega % p, l_zeta_omega)


def replay_challenges(proof, relation, params, transcript_prefix):
    """Transcript replay of one proof -> (witness commitments, evals, alphas, zeta, nus)."""
    pcs = params.pcs
    witness = (proof.c_b.commitment, proof.c_accip.commitment, proof.c_accx.commitment, proof.c_accy.commitment)
    wit_ser = b"".join(pcs.serialize_g1_uncompressed(c) for c in witness)
    evals = (proof.px_zeta, proof.py_zeta, proof.s_zeta, proof.b_zeta, proof.accip_zeta, proof.accx_zeta, proof.accy_zeta)
    _, alphas, zeta, nus = derive_challenges_after_vk(
        transcript_prefix, relation, wit_ser, pcs.serialize_g1_uncompressed(proof.c_q.commitment), evals, proof.l_zeta_omega)
    return witness, evals, alphas, zeta, nus


def linear_pcs_verifications(proof, fixed_commitments, relation, result_plus_seed, seed_point, params, transcript_prefix,
                             replay=None, inverses=None, domain=None):
    """proof: RingVRF-like object with the fifteen ring-proof fields.  replay / inverses let a batch verifier hoist the
    transcript replay and batch the modular inversions of all proofs."""
    p = params.prime
    witness, evals, alphas, zeta, nus = replay if replay is not None else replay_challenges(proof, relation, params, transcript_prefix)
    agg_zeta, k_ip, k_x, k_y, zeta_omega, l_zw = quotient_and_linearization_terms(
        alphas, nus, zeta, evals, proof.l_zeta_omega, seed_point, result_plus_seed, domain if domain is not None else params.domain,
        params.cv.curve.params.a, params.omega, p, inverses)
    c_px, c_py, c_s = fixed_commitments
    c_b, c_accip, c_accx, c_accy = witness
    quotient_terms = tuple(zip((c_px, c_py, c_s, c_b, c_accip, c_accx, c_accy, proof.c_q.commitment), nus))
    lin_terms = ((c_accip, k_ip), (c_accx, k_x), (c_accy, k_y))
    return (LinearPcsVerification(quotient_terms, proof.open_agg_zeta, zeta, agg_zeta),
            LinearPcsVerification(lin_terms, proof.open_l_zeta_omega, zeta_omega, l_zw))
