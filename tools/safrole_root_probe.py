"""Which padding rule reproduces the px / py thirds of tests/golden/full/safrole-ring-root.json?  (CPU only, uses the test oracle.)

The file (unused by the reference's own tests) carries 1023 keys, domain 2048, max_ring_size 1791 and two 144-byte roots
(`ring_root_hex`, `pre_gamma_z_hex`).  The selector third of both equals the commitment the reference's rule gives; the
px / py thirds equal NEITHER what the reference's Ring() / RingRoot.from_ring rule produces (members.py:35-55, root.py:21-44 —
the rule every N = 512 ring KAT pins) nor any of the variants below.  Result on this tree: all `False` — the file was generated
with other constants (another blinding-base / padding point generation of ark-vrf / ring-proof), so only its selector
commitment can pin anything (tests/test_oracle_kats.py::test_safrole_selector_commitment_n2048).
    python tools/safrole_root_probe.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.pyref import bandersnatch as bsn, kzg, ring  # noqa: E402

d = json.load(open(os.path.join(ROOT, "tests", "golden", "full", "safrole-ring-root.json")))
keys = [bsn.dec_point(bytes.fromhex(k)) for k in d["pubkeys_hex"]]
params = ring.Params(domain_size=d["domain_size"], max_ring_size=d["max_ring_size"])
N, M = params.N, params.max_ring
pad, base = params.suite.padding_point, params.suite.blinding_base
want_x = {d["ring_root_hex"][:96], d["pre_gamma_z_hex"][:96]}
want_y = {d["ring_root_hex"][96:192], d["pre_gamma_z_hex"][96:192]}


def powers(count):
    out, cur = [], base
    for _ in range(count):
        out.append(cur)
        cur = bsn.add(cur, cur)
    return out


def roots(points):
    cx = kzg.compress(kzg.commit(ring.intt([p[0] for p in points], params.omega))).hex()
    cy = kzg.compress(kzg.commit(ring.intt([p[1] for p in points], params.omega))).hex()
    return cx in want_x, cy in want_y


fill, pw = [pad] * (M - len(keys)), powers(N - 4 - M)
variants = {
    "reference rule: keys | padding point | 2^i B | 4 x (0,0)": keys + fill + pw + [(0, 0)] * 4,
    "last 4 rows = padding point": keys + fill + pw + [pad] * 4,
    "last 4 rows = identity (0,1)": keys + fill + pw + [(0, 1)] * 4,
    "last 4 rows = further powers of B": keys + fill + powers(N - M),
    "unused key rows = (0,0)": keys + [(0, 0)] * (M - len(keys)) + pw + [(0, 0)] * 4,
    "unused key rows = identity": keys + [(0, 1)] * (M - len(keys)) + pw + [(0, 0)] * 4,
    "keys in reverse order": keys[::-1] + fill + pw + [(0, 0)] * 4,
    "padding point, then 3 x (0,0)": keys + fill + pw + [pad] + [(0, 0)] * 3,
    "(0,0), then 3 x padding point": keys + fill + pw + [(0, 0)] + [pad] * 3,
}
for name, pts in variants.items():
    assert len(pts) == N
    print(f"{name:60s} px, py reproduced: {roots(pts)}")
