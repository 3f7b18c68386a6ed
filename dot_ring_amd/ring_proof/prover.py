"""Ring-proof prover (dot_ring/ring_proof/proof_builder.py:38-315, columns/columns.py:69-167,
constraints/constraints.py:43-151), organised in PHASES so that a batch of proofs shares every GPU launch:

  phase A  witness columns  -> ONE batched iNTT (4 per proof) + ONE batched MSM (4 commitments per proof)
  phase B  alphas (host hash), 4N-domain evaluations -> ONE batched NTT, constraint aggregation, ONE batched
           iNTT(4N), quotient -> batched MSM of the quotients
  phase C  zeta (host hash), register evaluations, linearisation, nu (host hash), two openings per proof ->
           batched MSMs
The field arithmetic between launches (pointwise constraint evaluation, linear combinations, Horner) is host
big-int code in this round, as it is in the reference; moving it into kernels K7/K8 is the next step.
"""
from __future__ import annotations

from .columns import Column
from .pcs import synthetic_div_with_eval
from .poly import (evaluate_polys_fft, inverse_fft_batch, poly_divide_by_vanishing, poly_evaluate_single, poly_mul_small)
from .transcript import phase1_alphas_after_vk, phase2_eval_point, phase3_nu_vector


def _group_commit(pcs, polys):
    """Commit polynomials of mixed lengths: equal lengths share one batched MSM."""
    out = [None] * len(polys)
    by_len = {}
    for i, p in enumerate(polys):
        by_len.setdefault(len(p), []).append(i)
    for n, idxs in by_len.items():
        if hasattr(pcs, "commit_batch") and len(idxs) > 1:
            for i, c in zip(idxs, pcs.commit_batch([polys[i] for i in idxs])):
                out[i] = c
        else:
            for i in idxs:
                out[i] = pcs.commit(polys[i])
    return out


class _Job:
    """Prover state of one proof."""
    __slots__ = ("k", "t", "relation", "rps", "cols", "transcript", "alphas", "q", "c_q", "zeta", "evals", "lin", "l_zw", "nus")


def build_ring_proofs(ring, ring_root, producer_keys, blinding_factors, transcript_challenge=None):
    """Payload field tuples for several (producer_key, blinding_factor) pairs over ONE ring / ring root."""
    params = ring.params
    cv, pcs, p, n = params.cv, params.pcs, params.prime, params.domain_size
    aux = cv.curve.params.auxiliary_points
    seed = cv.point_type(*aux.accumulator_base)
    rows = n - params.padding_rows
    domain = params.domain
    challenge_label = transcript_challenge or cv.curve.params.suite_id

    # ---------------- phase A: witness columns (columns.py:111-146)
    jobs = []
    blinds = cv.point_type(*aux.blinding_base)
    from ..curve import scalar_mul_batch

    blinded = scalar_mul_batch([blinds] * len(producer_keys), list(blinding_factors))
    for key, t, tb in zip(producer_keys, blinding_factors, blinded):
        job = _Job()
        job.k, job.t = ring.index_of(key), t
        bits = [1 if i == job.k else 0 for i in range(params.max_ring_size)]
        bits += [int(ch) for ch in bin(t)[2:][::-1]]
        if len(bits) > rows:
            raise ValueError(
                "b vector length exceeds available rows: "
                f"{len(bits)} > {rows} (ring_size={params.max_ring_size}, secret_t_bits={t.bit_length()}, padding_rows={params.padding_rows})")
        bits += [0] * (rows - len(bits)) + [0]
        acc = [seed]
        for i in range(1, rows + 1):
            acc.append(acc[-1] + cv.point_type(*ring.nm_points[i - 1]) if bits[i - 1] else acc[-1])
        accip = [0]
        for i in range(1, rows + 1):
            accip.append(accip[-1] + bits[i - 1] * ring_root.s.evals[i - 1])
        job.cols = [Column("b", bits, size=n), Column("accx", [pt.x for pt in acc], size=n),
                    Column("accy", [pt.y for pt in acc], size=n), Column("accip", accip, size=n)]
        for col in job.cols:
            col.pad(p, hidden=True, test_vectors=params.test_vectors)
        job.relation = cv.point_type(*ring.nm_points[job.k]) + tb
        job.rps = job.relation + seed
        jobs.append(job)
    coeffs = inverse_fft_batch([col.evals for job in jobs for col in job.cols], params.omega, p)
    for j, job in enumerate(jobs):
        for c, col in enumerate(job.cols):
            col.coeffs = coeffs[4 * j + c]
    for (job_i, col_i), cm in zip([(j, c) for j in range(len(jobs)) for c in range(4)],
                                  _group_commit(pcs, [col.coeffs for job in jobs for col in job.cols])):
        jobs[job_i].cols[col_i].set_commitment(cm)

    # ---------------- phase B: alphas, constraints on the 4N domain, quotient
    prefix = ring_root.verifier_transcript_prefix(challenge_label)
    m, w4 = params.radix_domain_size, params.radix_omega
    fixed4 = evaluate_polys_fft([ring_root.px.coeffs, ring_root.py.coeffs, ring_root.s.coeffs], m, w4, p)
    inv_n = pow(n, -1, p)

    def lagrange(i):
        inv_xi, out, cur = pow(domain[i], -1, p), [], inv_n
        for _ in range(n):
            out.append(cur)
            cur = cur * inv_xi % p
        return out

    l0_4, ln_4 = evaluate_polys_fft([lagrange(0), lagrange(params.last_index)], m, w4, p)
    wit4 = evaluate_polys_fft([col.coeffs for job in jobs for col in job.cols], m, w4, p)
    radix_domain = params.radix_domain
    last_root = pow(params.omega, params.last_index, p)
    not_last = [(x - last_root) % p for x in radix_domain]
    shift = params.radix_shift
    a_coeff = cv.curve.params.a
    sx, sy = seed.x, seed.y
    px4, py4, s4 = fixed4
    agg_evals = []
    for j, job in enumerate(jobs):
        c_b, c_accx, c_accy, c_accip = job.cols
        wit_ser = b"".join(pcs.serialize_g1_uncompressed(c.commitment) for c in (c_b, c_accip, c_accx, c_accy))
        job.transcript, job.alphas = phase1_alphas_after_vk(prefix.copy(), job.relation, wit_ser)
        b4, ax4, ay4, ip4 = wit4[4 * j : 4 * j + 4]
        al = job.alphas
        rx, ry = job.rps.x, job.rps.y
        out = []
        for i in range(m):
            k = i + shift
            if k >= m:
                k -= m
            x1, y1, x2, y2, x3, y3, b, nl = ax4[i], ay4[i], px4[i], py4[i], ax4[k], ay4[k], b4[i], not_last[i]
            c1 = (ip4[k] - ip4[i] - b * s4[i]) * nl
            c2 = (b * (x3 * (y1 * y2 + a_coeff * x1 * x2) - (x1 * y1 + x2 * y2)) + (1 - b) * (x3 - x1)) * nl
            c3 = (b * (y3 * (x1 * y2 - x2 * y1) - (x1 * y1 - x2 * y2)) + (1 - b) * (y3 - y1)) * nl
            c4 = b * (1 - b)
            c5 = (x1 - sx) * l0_4[i] + (x1 - rx) * ln_4[i]
            c6 = (y1 - sy) * l0_4[i] + (y1 - ry) * ln_4[i]
            c7 = ip4[i] * l0_4[i] + (ip4[i] - 1) * ln_4[i]
            out.append((al[0] * c1 + al[1] * c2 + al[2] * c3 + al[3] * c4 + al[4] * c5 + al[5] * c6 + al[6] * c7) % p)
        agg_evals.append(out)
    agg_polys = inverse_fft_batch(agg_evals, w4, p)
    tail = [1]
    for off in range(1, 4):
        tail = poly_mul_small(tail, [-domain[-off] % p, 1], p)
    for job, poly in zip(jobs, agg_polys):
        c_agg = poly_mul_small(tail, poly, p)
        while c_agg and c_agg[-1] == 0:
            c_agg.pop()
        job.q = poly_divide_by_vanishing(c_agg, n, p)
    for job, cm in zip(jobs, _group_commit(pcs, [job.q for job in jobs])):
        job.c_q = cm

    # ---------------- phase C: zeta, register evaluations, linearisation, nu, openings
    fixed = (ring_root.px.coeffs, ring_root.py.coeffs, ring_root.s.coeffs)
    quotients = []
    for job in jobs:
        c_b, c_accx, c_accy, c_accip = job.cols
        job.transcript, job.zeta = phase2_eval_point(job.transcript, pcs.serialize_g1_uncompressed(job.c_q))
        zeta = job.zeta
        zeta_w = zeta * params.omega % p
        term = (zeta - domain[params.last_index]) % p
        job.evals = [poly_evaluate_single(c, zeta, p)
                     for c in (*fixed, c_b.coeffs, c_accip.coeffs, c_accx.coeffs, c_accy.coeffs)]
        pxz, pyz, _sz, bz, _ipz, axz, ayz = job.evals
        fx = (bz * (ayz * pyz + a_coeff * axz * pxz) + (1 - bz)) * term % p
        fy = (bz * (axz * pyz - pxz * ayz) + (1 - bz)) * term % p
        k0, k1, k2 = job.alphas[0] * term % p, job.alphas[1] * fx % p, job.alphas[2] * fy % p
        job.lin = [(k0 * ci + k1 * cx + k2 * cy) % p for ci, cx, cy in zip(c_accip.coeffs, c_accx.coeffs, c_accy.coeffs)]
        job.l_zw = poly_evaluate_single(job.lin, zeta_w, p)
        job.nus = phase3_nu_vector(job.transcript, job.evals, job.l_zw)
        polys = [*fixed, c_b.coeffs, c_accip.coeffs, c_accx.coeffs, c_accy.coeffs, job.q]
        width = max(len(q) for q in polys)
        agg = [sum(nu * (poly[i] if i < len(poly) else 0) for nu, poly in zip(job.nus, polys)) % p for i in range(width)]
        quotients.append(synthetic_div_with_eval(agg, zeta)[0])
        quotients.append(synthetic_div_with_eval(job.lin, zeta_w)[0])
    openings = _group_commit(pcs, quotients)

    payloads = []
    for j, job in enumerate(jobs):
        c_b, c_accx, c_accy, c_accip = job.cols
        c_q_col = Column(name="C_q", evals=[], _commitment=job.c_q)
        c_q_col._has_commitment = True
        payloads.append((c_b, c_accip, c_accx, c_accy, *job.evals, c_q_col, job.l_zw, openings[2 * j], openings[2 * j + 1]))
    return payloads
