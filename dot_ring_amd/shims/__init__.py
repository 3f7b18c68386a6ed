"""Drop-in modules with the EXACT signatures of the reference's three native seams, over libdotring_hip.so:

* `bandersnatch_te_hip` — `dot_ring.curve.native_field.bandersnatch_te` (bandersnatch_te.pyx:244-811)
* `hip_kzg.HipKZG`      — a `PCS` (dot_ring/ring_proof/pcs/protocol.py:10-40) for `RingProofParams(pcs=...)`
* `ntt_plan.BlsScalarNTTPlan` — dot_ring/ring_proof/polynomial/ntt.pyx:29-163

A maintainer of the reference switches three import lines; tests/test_gpu_shims.py calls every function
with the reference's own argument shapes and compares with the CPU restatement of the reference (tests only).
"""
