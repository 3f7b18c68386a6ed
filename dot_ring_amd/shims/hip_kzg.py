"""`HipKZG` — a `PCS` (dot_ring/ring_proof/pcs/protocol.py:10-40) whose G1 work runs on the GPU; pass it as
`RingProofParams(pcs=HipKZG)` (dot_ring/ring_proof/params.py:126) in place of `dot_ring.ring_proof.pcs.kzg.KZG`.

Every protocol member keeps the reference's signature and error behaviour (pcs/kzg.py:123-338):

    normalize_g1(point) -> (x, y)                      compress_g1(point) -> 48 bytes        kzg.py:123, 129
    serialize_g1_uncompressed(point) -> 96 bytes        decompress_g1(data) -> commitment     kzg.py:133, 137
    msm_g1(points, scalars) -> commitment               commit(coeffs) -> commitment          kzg.py:147, 152
    open(coeffs, x) -> Opening(proof, y)                verify(commitment, proof, point, value) -> bool      kzg.py:178, 195
    batch_verify(verifications) -> bool                 batch_verify_linear_preconverted(...)               kzg.py:232, 304
    commitment_size = 48, scalar_modulus = r, srs (.g1_points / .g2_points for the transcript, root.py:61-66), ensure_srs_size

Commitment objects are opaque to the layers above (they are only handed back to PCS methods), so they are 96-byte affine
records (None = infinity) instead of `blst.P1`; `normalize_g1` / `compress_g1` / `serialize_g1_uncompressed` give the
reference's integers and bytes.  The implementation is `dot_ring_amd.ring_proof.pcs.KZG`; this module adds the adapter a
maintainer needs to bind it to the REFERENCE's own SRS object (lists of integer pairs, pcs/srs.py:98-114).
"""
from __future__ import annotations

from ..ring_proof.pcs import KZG, SRS, LinearPcsVerification, Opening, PcsVerification   # noqa: F401  (re-exported)


class HipKZG(KZG):
    """The default instance serves the shipped 2^11 SRS, exactly like the reference's module-level `srs`."""


def _g2_record(pt) -> bytes:
    """((x1, x0), (y1, y0)) as pcs/srs.py:78-88 builds it -> the 192-byte zcash layout x.c1 || x.c0 || y.c1 || y.c0"""
    (x1, x0), (y1, y0) = pt
    return b"".join(int(v).to_bytes(48, "big") for v in (x0, x1, y0, y1))


def bind_reference_srs(srs_like) -> type:
    """A HipKZG class over a reference-style SRS object: `srs_like.g1_points` = [(x, y), ...] affine integers,
    `srs_like.g2_points` = [((x1, x0), (y1, y0)), ...] (at least two: [1]G2 and [tau]G2) — the attributes
    `dot_ring.ring_proof.pcs.srs.SRS` carries (pcs/srs.py:98-114).  The G1 points go to HBM once (dr_srs_load)."""
    g1_raw = b"".join(int(x).to_bytes(48, "big") + int(y).to_bytes(48, "big") for x, y in srs_like.g1_points)
    g2_raw = [_g2_record(p) for p in list(srs_like.g2_points)[:2]]
    if len(g2_raw) < 2:
        raise ValueError("SRS file must contain at least two G2 points")
    return HipKZG.with_srs(SRS(g1_raw, g2_raw))
