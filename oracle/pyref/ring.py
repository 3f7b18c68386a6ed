"""ORACLE (test infrastructure only) — ring-proof prover and verifier algebra, plain Python ints.

Restates:
  parameters / domains / root extension   dot_ring/ring_proof/params.py:12-287
  ring member preprocessing               dot_ring/vrf/ring/members.py:22-91
  ring root (3 fixed columns)             dot_ring/vrf/ring/root.py:21-173
  Fiat-Shamir transcript + phases         dot_ring/ring_proof/transcript/transcript.py:21-136, phases.py:18-128
  witness columns                         dot_ring/ring_proof/columns/columns.py:29-167
  constraints c1..c7                      dot_ring/ring_proof/constraints/constraints.py:43-151
  prover pipeline                         dot_ring/ring_proof/proof_builder.py:38-315
  polynomial helpers                      dot_ring/ring_proof/polynomial/ops.py:170-224, fft.py:87-144
  payload codec                           dot_ring/ring_proof/proof_payload.py:68-117
  verifier scalar pass                    dot_ring/ring_proof/verify.py:51-144
Everything is written the slow obvious way (lists of ints); NTTs and MSMs use the C oracle.
"""
from __future__ import annotations

import hashlib
import secrets
import struct

from .. import coracle
from . import bandersnatch as bsn
from . import kzg
from . import vrf

P = bsn.P
ROOT_OF_UNITY_2048 = 49307615728544765012166121802278658070711169839041683575071795236746050763237
PADDING_ROWS = 4
ZK_ROWS = 3
MAX_DOMAIN = 4096


def _sqrt_mod_prime(n: int) -> int:
    """params.py:63 — Tonelli-Shanks exactly as the reference runs it (the root it returns fixes omega at N=4096)."""
    if n == 0:
        return 0
    if pow(n, (P - 1) // 2, P) != 1:
        raise ValueError("No square root exists for provided value")
    q, s = P - 1, 0
    while q % 2 == 0:
        s += 1
        q //= 2
    z = 2
    while pow(z, (P - 1) // 2, P) != P - 1:
        z += 1
    m, c, x, t = s, pow(z, q, P), pow(n, (q + 1) // 2, P), pow(n, q, P)
    while t != 1:
        i, t2i = 1, t * t % P
        while i < m:
            if t2i == 1:
                break
            t2i = t2i * t2i % P
            i += 1
        b = pow(c, 1 << (m - i - 1), P)
        x = x * b % P
        t = t * b * b % P
        c = b * b % P
        m = i
    return x


class Params:
    """params.py:119 RingProofParams (Bandersnatch suites only)."""

    def __init__(self, domain_size: int = 512, max_ring_size: int | None = None, test_vectors: bool = False,
                 suite: bsn.Suite = bsn.SHA512, srs: kzg.SRS | None = None):
        if domain_size & (domain_size - 1) or domain_size <= 0:
            raise ValueError(f"domain_size must be a power of two, got {domain_size}")
        if domain_size > MAX_DOMAIN:
            raise ValueError(f"domain_size {domain_size} exceeds supported SRS domain size {MAX_DOMAIN}")
        self.N = domain_size
        self.radix = 4 * domain_size
        cap = domain_size - bsn.N.bit_length() - PADDING_ROWS
        if cap <= 0:
            raise ValueError("domain_size is too small for the scalar bit decomposition")
        if max_ring_size is None:
            max_ring_size = cap
        if max_ring_size > cap:
            raise ValueError(f"max_ring_size {max_ring_size} exceeds supported size {cap}")
        self.max_ring = max_ring_size
        self.test_vectors = test_vectors
        self.suite = suite
        self.srs = srs
        root, size = ROOT_OF_UNITY_2048, 2048
        while size < self.radix:           # params.py:108 _extend_root_to_size
            root = _sqrt_mod_prime(root)
            size *= 2
        self.omega = pow(root, size // self.N, P)
        self.radix_omega = pow(root, size // self.radix, P)
        self.domain = [pow(self.omega, i, P) for i in range(self.N)]
        self.last_index = self.N - PADDING_ROWS

    @classmethod
    def from_ring_size(cls, ring_size: int, **kw) -> "Params":
        if ring_size <= 0:
            raise ValueError(f"ring_size must be positive, got {ring_size}")
        need = ring_size + bsn.N.bit_length() + PADDING_ROWS
        n = 1
        while n < need:
            n *= 2
        return cls(domain_size=n, **kw)


# ------------------------------------------------------------------ NTT glue (fft.py)
def intt(values, omega):
    n = len(values)
    return coracle.ntt(values, pow(omega, -1, P), pow(n, -1, P))


def eval_on_domain(poly, size, omega):
    """fft.py:104 evaluate_poly_fft: fold mod X^size - 1, then NTT."""
    folded = [0] * size
    for i, c in enumerate(poly):
        folded[i % size] = (folded[i % size] + c) % P
    return coracle.ntt(folded, omega)


def horner(poly, x):
    acc = 0
    for c in reversed(poly):
        acc = (acc * x + c) % P
    return acc


# ------------------------------------------------------------------ ring + root
class Ring:
    """members.py:18 — public vector PK || padding || 2^i*B powers || 4 x (0,0)."""

    def __init__(self, keys, params: Params | None = None):
        self.params = params or Params.from_ring_size(len(keys))
        pr = self.params
        if len(keys) > pr.max_ring:
            raise ValueError(f"ring size {len(keys)} exceeds max supported size {pr.max_ring}")
        pad = pr.suite.padding_point
        pts = []
        for key in keys:
            pt = self._decode(key)
            pts.append(pad if pt is None else pt)
        pts += [pad] * (pr.max_ring - len(pts))
        fill = pr.N - PADDING_ROWS - len(pts)
        cur = pr.suite.blinding_base
        for _ in range(fill):
            pts.append(cur)
            cur = bsn.add(cur, cur)
        pts += [(0, 0)] * PADDING_ROWS
        self.points = pts

    @staticmethod
    def _decode(key: bytes):
        try:
            pt = bsn.dec_point(key)
        except ValueError:
            return None
        return None if pt == bsn.IDENTITY else pt

    def index_of(self, key: bytes) -> int:
        pt = self._decode(key)
        if pt is None:
            raise ValueError("invalid ring key")
        if pt == self.params.suite.padding_point:
            raise ValueError("producer key is not in ring")
        try:
            return self.points[: self.params.max_ring].index(pt)
        except ValueError as exc:
            raise ValueError("producer key is not in ring") from exc


class RingRoot:
    """root.py:14 — columns px, py, s: evaluations, coefficients, commitments."""

    def __init__(self, ring: Ring):
        pr = ring.params
        self.params = pr
        self.s_evals = [1 if i < pr.max_ring else 0 for i in range(pr.N)]
        self.px_evals = [pt[0] for pt in ring.points]
        self.py_evals = [pt[1] for pt in ring.points]
        self.s = intt(self.s_evals, pr.omega)
        self.px = intt(self.px_evals, pr.omega)
        self.py = intt(self.py_evals, pr.omega)
        self.c_px = kzg.commit(self.px, pr.srs)
        self.c_py = kzg.commit(self.py, pr.srs)
        self.c_s = kzg.commit(self.s, pr.srs)

    def encode(self) -> bytes:
        return kzg.compress(self.c_px) + kzg.compress(self.c_py) + kzg.compress(self.c_s)

    def vk_bytes(self) -> bytes:
        # root.py:54 + phases.py:72: G1[0] || G2[0] || G2[1] (file byte order) || 3 serialized commitments
        srs = self.params.srs or kzg.default_srs()
        return (kzg.serialize(srs.g1[0]) + srs.g2_raw[0] + srs.g2_raw[1]
                + kzg.serialize(self.c_px) + kzg.serialize(self.c_py) + kzg.serialize(self.c_s))


# ------------------------------------------------------------------ Fiat-Shamir transcript
class FsTranscript:
    """transcript.py:21 — SHAKE128; every item framed label || BE32(len label) || data || BE32(len data)."""

    def __init__(self, initial: bytes):
        self.h = hashlib.shake_128()
        self.h.update(initial + struct.pack(">I", len(initial)))

    def fork(self) -> "FsTranscript":
        t = FsTranscript.__new__(FsTranscript)
        t.h = self.h.copy()
        return t

    def absorb(self, label: bytes, data: bytes) -> None:
        self.h.update(label + struct.pack(">I", len(label)) + data + struct.pack(">I", len(data)))

    def challenges(self, label: bytes, n: int):
        prefix = label + struct.pack(">I", len(label)) + b"challenge"
        footer = b"\x00\x00\x00\x09"
        out = []
        self.h.update(prefix)
        for i in range(n):
            out.append(int.from_bytes(self.h.digest(48), "big") % P)
            self.h.update(footer if i == n - 1 else footer + prefix)
        return out


def _le32(v: int) -> bytes:
    return int(v).to_bytes(32, "little")


def derive_challenges(prefix: FsTranscript, relation, witness_ser: bytes, cq_ser: bytes, evals, l_zw):
    """phases.py:49 derive_challenges_after_vk."""
    t = prefix.fork()
    t.absorb(b"instance", _le32(relation[0]) + _le32(relation[1]))
    t.absorb(b"committed_cols", witness_ser)
    alphas = t.challenges(b"constraints_aggregation", 7)
    t.absorb(b"quotient", cq_ser)
    (zeta,) = t.challenges(b"evaluation_point", 1)
    t.absorb(b"register_evaluations", b"".join(_le32(e) for e in evals))
    t.absorb(b"shifted_linearization_evaluation", _le32(l_zw))
    return alphas, zeta, t.challenges(b"kzg_aggregation", 8)


# ------------------------------------------------------------------ prover
def _poly_mul_small(a, b):
    out = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            out[i + j] = (out[i + j] + x * y) % P
    return out


def prove_ring(ring: Ring, root: RingRoot, producer_key: bytes, blinding: int, zk_rows=None):
    """proof_builder.py:38 build() -> 592-byte payload.  zk_rows: optional {name: [3 ints]} to pin the
    hidden rows (columns.py:43 draws them with secrets.randbelow when test_vectors is False)."""
    pr = ring.params
    n, suite = pr.N, pr.suite
    k = ring.index_of(producer_key)

    # --- witness columns (columns.py:111-146)
    bits = [1 if i == k else 0 for i in range(pr.max_ring)]
    bits += [int(c) for c in bin(blinding)[2:][::-1]]
    if len(bits) > n - PADDING_ROWS:
        raise ValueError("b vector length exceeds available rows")
    bits += [0] * (n - PADDING_ROWS - len(bits)) + [0]
    acc = [suite.accumulator_base]
    for i in range(1, n - PADDING_ROWS + 1):
        acc.append(bsn._te_add_ref(acc[-1], ring.points[i - 1]) if bits[i - 1] else acc[-1])
    accip = [0]
    for i in range(1, n - PADDING_ROWS + 1):
        accip.append(accip[-1] + bits[i - 1] * root.s_evals[i - 1])

    def column(name, evals):
        evals = list(evals)
        if pr.test_vectors:
            evals += [0] * (n - len(evals))
        else:
            evals += [0] * (n - ZK_ROWS - len(evals))
            evals += list(zk_rows[name]) if zk_rows else [secrets.randbelow(P) for _ in range(ZK_ROWS)]
        coeffs = intt(evals, pr.omega)
        return coeffs, kzg.commit(coeffs, pr.srs)

    b_c, c_b = column("b", bits)
    accx_c, c_accx = column("accx", [pt[0] for pt in acc])
    accy_c, c_accy = column("accy", [pt[1] for pt in acc])
    accip_c, c_accip = column("accip", accip)

    relation = bsn._te_add_ref(ring.points[k], bsn.mul(suite.blinding_base, blinding))
    rx, ry = bsn._te_add_ref(relation, suite.accumulator_base)
    sx, sy = suite.accumulator_base

    # --- phase 1 challenges
    prefix = FsTranscript(suite.suite_id)
    prefix.absorb(b"vk", root.vk_bytes())
    t = prefix.fork()
    t.absorb(b"instance", _le32(relation[0]) + _le32(relation[1]))
    t.absorb(b"committed_cols", b"".join(kzg.serialize(c) for c in (c_b, c_accip, c_accx, c_accy)))
    alphas = t.challenges(b"constraints_aggregation", 7)

    # --- constraints on the 4N domain (constraints.py:43-151)
    m, w4 = pr.radix, pr.radix_omega
    px4, py4, s4 = (eval_on_domain(c, m, w4) for c in (root.px, root.py, root.s))
    b4, ax4, ay4, ip4 = (eval_on_domain(c, m, w4) for c in (b_c, accx_c, accy_c, accip_c))
    inv_n = pow(n, -1, P)
    def lagrange(i):
        inv_xi = pow(pr.domain[i], -1, P)
        return [inv_n * pow(inv_xi, j, P) % P for j in range(n)]
    l0 = eval_on_domain(lagrange(0), m, w4)
    ln = eval_on_domain(lagrange(pr.last_index), m, w4)
    last_root = pow(pr.omega, pr.last_index, P)
    shift = m // n
    agg = []
    x = 1
    for i in range(m):
        j = (i + shift) % m
        nl = (x - last_root) % P
        x1, y1, x2, y2, x3, y3, b = ax4[i], ay4[i], px4[i], py4[i], ax4[j], ay4[j], b4[i]
        c1 = (ip4[j] - ip4[i] - b * s4[i]) * nl
        c2 = (b * (x3 * (y1 * y2 + bsn.A * x1 * x2) - (x1 * y1 + x2 * y2)) + (1 - b) * (x3 - x1)) * nl
        c3 = (b * (y3 * (x1 * y2 - x2 * y1) - (x1 * y1 - x2 * y2)) + (1 - b) * (y3 - y1)) * nl
        c4 = b * (1 - b)
        c5 = (x1 - sx) * l0[i] + (x1 - rx) * ln[i]
        c6 = (y1 - sy) * l0[i] + (y1 - ry) * ln[i]
        c7 = ip4[i] * l0[i] + (ip4[i] - 1) * ln[i]
        agg.append(sum(a * c for a, c in zip(alphas, (c1, c2, c3, c4, c5, c6, c7))) % P)
        x = x * w4 % P

    # --- quotient (proof_builder.py:165-195, ops.py:207)
    agg_poly = intt(agg, w4)
    tail = [1]
    for off in range(1, 4):
        tail = _poly_mul_small(tail, [(-pr.domain[-off]) % P, 1])
    c_agg = _poly_mul_small(tail, agg_poly)
    while c_agg and c_agg[-1] == 0:
        c_agg.pop()
    q = [0] if len(c_agg) < n else [sum(c_agg[j + i * n] for i in range(1, len(c_agg) // n + 1) if j + i * n < len(c_agg)) % P
                                    for j in range(len(c_agg) - n)]
    while q and q[-1] == 0:
        q.pop()
    c_q = kzg.commit(q, pr.srs)

    # --- evaluation point, register evaluations, linearisation (proof_builder.py:197-286)
    t.absorb(b"quotient", kzg.serialize(c_q))
    (zeta,) = t.challenges(b"evaluation_point", 1)
    zeta_w = zeta * pr.omega % P
    term = (zeta - pr.domain[pr.last_index]) % P
    evals = [horner(c, zeta) for c in (root.px, root.py, root.s, b_c, accip_c, accx_c, accy_c)]
    pxz, pyz, _sz, bz, _ipz, axz, ayz = evals
    fx = (bz * (ayz * pyz + bsn.A * axz * pxz) + (1 - bz)) * term % P
    fy = (bz * (axz * pyz - pxz * ayz) + (1 - bz)) * term % P
    lin = [(alphas[0] * term % P * ci + alphas[1] * fx % P * cx + alphas[2] * fy % P * cy) % P
           for ci, cx, cy in zip(accip_c, accx_c, accy_c)]
    l_zw = horner(lin, zeta_w)

    # --- aggregation challenges + openings (proof_builder.py:288-315, kzg.py:178)
    t.absorb(b"register_evaluations", b"".join(_le32(e) for e in evals))
    t.absorb(b"shifted_linearization_evaluation", _le32(l_zw))
    nus = t.challenges(b"kzg_aggregation", 8)
    polys = [root.px, root.py, root.s, b_c, accip_c, accx_c, accy_c, q]
    width = max(len(p) for p in polys)
    agg_open = [sum(nu * (p[i] if i < len(p) else 0) for nu, p in zip(nus, polys)) % P for i in range(width)]
    phi_z, _ = kzg.open_at(agg_open, zeta, pr.srs)
    phi_zw, _ = kzg.open_at(lin, zeta_w, pr.srs)

    # --- payload (proof_payload.py:68)
    return (b"".join(kzg.compress(c) for c in (c_b, c_accip, c_accx, c_accy))
            + b"".join(_le32(e) for e in evals)
            + kzg.compress(c_q) + _le32(l_zw) + kzg.compress(phi_z) + kzg.compress(phi_zw))


def ring_vrf_prove(ring: Ring, root: RingRoot, alpha: bytes, ad: bytes, sk: bytes, zk_rows=None) -> bytes:
    """vrf/ring/vrf.py:185 — Pedersen proof (192) || ring payload (592)."""
    pk = bsn.public_key_from_secret(sk)
    ped, blinding = vrf.pedersen_prove(ring.params.suite, alpha, sk, ad)
    return ped + prove_ring(ring, root, pk, blinding, zk_rows)


# ------------------------------------------------------------------ verifier (scalar pass; pairing in pairing.py)
def decode_payload(data: bytes):
    if len(data) != 592:
        raise ValueError(f"invalid Ring VRF proof length: expected 592, got {len(data)}")
    pts = [kzg.decompress(data[48 * i : 48 * i + 48]) for i in range(4)]
    evals = []
    for i in range(7):
        v = int.from_bytes(data[192 + 32 * i : 224 + 32 * i], "little")
        if v >= P:
            raise ValueError("scalar is not canonical")
        evals.append(v)
    c_q = kzg.decompress(data[416:464])
    l_zw = int.from_bytes(data[464:496], "little")
    if l_zw >= P:
        raise ValueError("scalar is not canonical")
    return pts, evals, c_q, l_zw, kzg.decompress(data[496:544]), kzg.decompress(data[544:592])


def verifier_equations(root: RingRoot, relation, payload: bytes):
    """verify.py:51-210 — returns the two linear KZG claims as
    [(list[(commitment, scalar)], proof, point, value), ...]."""
    pr = root.params
    (c_b, c_accip, c_accx, c_accy), evals, c_q, l_zw, phi_z, phi_zw = decode_payload(payload)
    prefix = FsTranscript(pr.suite.suite_id)
    prefix.absorb(b"vk", root.vk_bytes())
    wit = b"".join(kzg.serialize(c) for c in (c_b, c_accip, c_accx, c_accy))
    alphas, zeta, nus = derive_challenges(prefix, relation, wit, kzg.serialize(c_q), evals, l_zw)
    pxz, pyz, sz, bz, ipz, axz, ayz = evals
    sx, sy = pr.suite.accumulator_base
    rx, ry = bsn._te_add_ref(pr.suite.accumulator_base, relation)
    n, dom = pr.N, pr.domain
    zn1 = (pow(zeta, n, P) - 1) % P
    d4 = (zeta - dom[-4]) % P
    # Lagrange values at zeta for rows 0 and N-4
    l0 = zn1 * pow(n * (zeta - 1) % P, -1, P) % P if zeta != 1 else 1
    ln = dom[-4] * zn1 % P * pow(n * d4 % P, -1, P) % P if d4 else 1
    one_b = (1 - bz) % P
    cs = [
        -(ipz + bz * sz) * d4,
        (bz * -(axz * ayz + pxz * pyz) + one_b * -axz) * d4,
        (bz * -(axz * ayz - pxz * pyz) + one_b * -ayz) * d4,
        bz * one_b,
        (axz - sx) * l0 + (axz - rx) * ln,
        (ayz - sy) * l0 + (ayz - ry) * ln,
        ipz * l0 + (ipz - 1) * ln,
    ]
    lin = sum(a * c for a, c in zip(alphas, cs)) % P
    tail = (zeta - dom[-1]) * (zeta - dom[-2]) % P * (zeta - dom[-3]) % P
    q_zeta = (lin + l_zw) * tail % P * pow(zn1, -1, P) % P
    agg = sum(nu * v for nu, v in zip(nus, evals + [q_zeta])) % P
    fx = (bz * (ayz * pyz + bsn.A * axz * pxz) + one_b) % P
    fy = (bz * (axz * pyz - pxz * ayz) + one_b) % P
    eq1 = (list(zip((root.c_px, root.c_py, root.c_s, c_b, c_accip, c_accx, c_accy, c_q), nus)), phi_z, zeta, agg)
    eq2 = ([(c_accip, alphas[0] * d4 % P), (c_accx, alphas[1] * fx % P * d4 % P), (c_accy, alphas[2] * fy % P * d4 % P)],
           phi_zw, zeta * pr.omega % P, l_zw)
    return eq1, eq2
