"""Ring-proof verifier scalar pass (dot_ring/ring_proof/verify.py:51-210): replays the transcript, evaluates the
constraint system at zeta in closed form and returns the two linear KZG claims that the PCS folds on the GPU."""
from __future__ import annotations

from .pcs import LinearPcsVerification
from .transcript import derive_challenges_after_vk


def quotient_and_linearization_terms(alphas, nus, zeta, evals, l_zeta_omega, seed, result_seed, domain, edwards_a, omega, prime):
    if len(alphas) < 7 or len(nus) < 8:
        raise ValueError("expected at least 7 alpha values and 8 aggregation values")
    p = prime
    pxz, pyz, sz, bz, ipz, axz, ayz = evals
    n = len(domain)
    zn1 = (pow(zeta, n, p) - 1) % p
    d4 = (zeta - domain[-4]) % p
    z1 = (zeta - 1) % p
    inv_n = pow(n, -1, p)
    # L_0(zeta) = (zeta^n - 1) / (n (zeta - 1)),  L_{n-4}(zeta) = w^{n-4} (zeta^n - 1) / (n (zeta - w^{n-4}))
    l0 = 1 if z1 == 0 else inv_n * zn1 % p * pow(z1, -1, p) % p
    ln = 1 if d4 == 0 else domain[-4] * inv_n % p * zn1 % p * pow(d4, -1, p) % p
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
    q_zeta = (lin + l_zeta_omega) * tail % p * pow(zn1, -1, p) % p
    agg_zeta = sum(nu * v for nu, v in zip(nus, (*evals, q_zeta))) % p
    fx = (bz * (ayz * pyz + edwards_a * axz * pxz) + one_b) % p
    fy = (bz * (axz * pyz - pxz * ayz) + one_b) % p
    return (agg_zeta, alphas[0] * d4 % p, alphas[1] * fx % p * d4 % p, alphas[2] * fy % p * d4 % p,
            zeta * omega % p, l_zeta_omega)


def linear_pcs_verifications(proof, fixed_commitments, relation, result_plus_seed, seed_point, params, transcript_prefix):
    """proof: RingVRF-like object with the fifteen ring-proof fields."""
    pcs, p = params.pcs, params.prime
    witness = (proof.c_b.commitment, proof.c_accip.commitment, proof.c_accx.commitment, proof.c_accy.commitment)
    wit_ser = b"".join(pcs.serialize_g1_uncompressed(c) for c in witness)
    evals = (proof.px_zeta, proof.py_zeta, proof.s_zeta, proof.b_zeta, proof.accip_zeta, proof.accx_zeta, proof.accy_zeta)
    _, alphas, zeta, nus = derive_challenges_after_vk(
        transcript_prefix, relation, wit_ser, pcs.serialize_g1_uncompressed(proof.c_q.commitment), evals, proof.l_zeta_omega)
    agg_zeta, k_ip, k_x, k_y, zeta_omega, l_zw = quotient_and_linearization_terms(
        alphas, nus, zeta, evals, proof.l_zeta_omega, seed_point, result_plus_seed, params.domain,
        params.cv.curve.params.a, params.omega, p)
    c_px, c_py, c_s = fixed_commitments
    c_b, c_accip, c_accx, c_accy = witness
    quotient_terms = tuple(zip((c_px, c_py, c_s, c_b, c_accip, c_accx, c_accy, proof.c_q.commitment), nus))
    lin_terms = ((c_accip, k_ip), (c_accx, k_x), (c_accy, k_y))
    return (LinearPcsVerification(quotient_terms, proof.open_agg_zeta, zeta, agg_zeta),
            LinearPcsVerification(lin_terms, proof.open_l_zeta_omega, zeta_omega, l_zw))
