"""Batched ring prover on the GPU: the four device phases of dr_ring_prover_* with the Fiat-Shamir transcript
hashed on the host between them.  Produces exactly the payload tuples of prover.build_ring_proofs (and of the
reference's RingProofBuilder.build, proof_builder.py:38-142) for a batch of proofs over one ring."""
from __future__ import annotations

import secrets

from .. import _native, runtime
from .columns import Column
from .params import ZK_ROWS
from .pcs import KZG
from .transcript import phase1_alphas_after_vk, phase2_eval_point, phase3_nu_vector

MAX_DEVICE_BATCH = 4096      # proofs per device pass (workspace ~2.5 MB + MSM scratch per proof at N = 2048)


class _RawPoint:
    """Minimal stand-in with .x/.y for transcript serialisation of the relation point."""
    __slots__ = ("x", "y")

    def __init__(self, raw64: bytes):
        self.x = int.from_bytes(raw64[:32], "little")
        self.y = int.from_bytes(raw64[32:], "little")


def supported(params) -> bool:
    return isinstance(params.pcs, type) and issubclass(params.pcs, KZG) and params.padding_rows == 4 and params.radix_domain_size == 4 * params.domain_size


def get_device_prover(ring, slot: int = 0) -> _native.RingProver:
    """Device prover for a Ring object: per-ring tables + the phase state of ONE batch in flight.  Pipelined slices
    of a batch use different slots (each slot owns its own phase state; the tables are ~15 MB per slot)."""
    ctx = runtime.context()
    provers = ring.__dict__.setdefault("_device_provers", {})
    cached = provers.get((id(ctx), slot))
    if cached is not None and cached.ctx is ctx and cached.handle:
        return cached
    params = ring.params
    pcs = params.pcs
    pcs.ensure_srs_size(3 * params.domain_size)
    srs = pcs._srs().device()
    pts = b"".join(int(x).to_bytes(32, "little") + int(y).to_bytes(32, "little") for x, y in ring.nm_points)
    seed = params.cv.curve.params.auxiliary_points.accumulator_base
    prover = _native.RingProver(ctx, srs, params.domain_size.bit_length() - 1, params.max_ring_size, params.omega, params.radix_omega,
                                pts, int(seed[0]).to_bytes(32, "little") + int(seed[1]).to_bytes(32, "little"), params.cv.curve.params.curve_id)
    provers[(id(ctx), slot)] = prover
    return prover


def ring_proofs_gen(ring, ring_root, producer_indices, blinding_factors, transcript_challenge=None, slot: int = 0):
    """Generator form (dot_ring_amd/pipeline.py): yields the four device phases, returns the payload tuples."""
    params = ring.params
    pcs, p = params.pcs, params.prime
    prefix = ring_root.verifier_transcript_prefix(transcript_challenge or params.cv.curve.params.suite_id)
    payloads = []
    ser = pcs.serialize_g1_uncompressed
    for start in range(0, len(producer_indices), MAX_DEVICE_BATCH):
        idx = list(producer_indices[start : start + MAX_DEVICE_BATCH])
        blinds = b"".join(int(t).to_bytes(32, "little") for t in blinding_factors[start : start + MAX_DEVICE_BATCH])
        batch = len(idx)
        zk = None
        if not params.test_vectors:
            # hidden rows: uniform field elements from the OS CSPRNG (48 random bytes reduced mod p: bias < 2^-128;
            # the reference draws them with secrets.randbelow, columns.py:43-48)
            raw = secrets.token_bytes(48 * batch * 4 * ZK_ROWS)
            zk = b"".join((int.from_bytes(raw[i : i + 48], "little") % p).to_bytes(32, "little") for i in range(0, len(raw), 48))
        relation_raw, wit = yield (lambda: get_device_prover(ring, slot).witness(idx, blinds, zk))
        transcripts, alphas = [], []
        for j in range(batch):
            wit_ser = b"".join(ser(c) for c in wit[4 * j : 4 * j + 4])
            t, al = phase1_alphas_after_vk(prefix.copy(), _RawPoint(relation_raw[64 * j : 64 * j + 64]), wit_ser)
            transcripts.append(t)
            alphas.append(al)
        alpha_raw = b"".join(a.to_bytes(32, "little") for al in alphas for a in al)
        c_qs = yield (lambda: get_device_prover(ring, slot).quotient(batch, alpha_raw))
        zetas = []
        for j in range(batch):
            transcripts[j], zeta = phase2_eval_point(transcripts[j], ser(c_qs[j]))
            zetas.append(zeta)
        zeta_raw = b"".join(z.to_bytes(32, "little") for z in zetas)
        ev_raw = yield (lambda: get_device_prover(ring, slot).evals(batch, zeta_raw))
        evals, nus = [], []
        for j in range(batch):
            vals = [int.from_bytes(ev_raw[256 * j + 32 * i : 256 * j + 32 * i + 32], "little") for i in range(8)]
            evals.append(vals)
            nus.append(phase3_nu_vector(transcripts[j], vals[:7], vals[7]))
        nu_raw = b"".join(v.to_bytes(32, "little") for nu in nus for v in nu)
        opens = yield (lambda: get_device_prover(ring, slot).openings(batch, nu_raw))
        for j in range(batch):
            cols = [Column(name, [], _commitment=wit[4 * j + i], _has_commitment=True) for i, name in enumerate(("c_b", "c_accip", "c_accx", "c_accy"))]
            c_q = Column("C_q", [], _commitment=c_qs[j], _has_commitment=True)
            payloads.append((*cols, *evals[j][:7], c_q, evals[j][7], opens[2 * j], opens[2 * j + 1]))
    return payloads


def build_ring_proofs_device(ring, ring_root, producer_indices, blinding_factors, transcript_challenge=None):
    from ..pipeline import drive

    return drive(ring_proofs_gen(ring, ring_root, producer_indices, blinding_factors, transcript_challenge))
