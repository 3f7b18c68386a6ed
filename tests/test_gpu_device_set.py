"""GPU: ONE batch over a SET of devices inside one process (SURVEY 8(b)'s additive row; dr_ringvrf_prove_batch_multi /
dr_ringvrf_verify_batch_multi behind RingVRF.prove_batch / batch_verify when DOTRING_DEVICES names several devices).

The test box has one GPU, so the set is "0,0,0": three contexts — three streams, three copies of the SRS tables and of the prover
state — on the same card, each shard on a host thread of its own.  Everything but the device ordinal is what an 8-GPU process runs.
Deterministic proofs must equal the one-device batch byte for byte, whatever the split."""
import hashlib

import pytest

pytestmark = pytest.mark.gpu


def _case(count, keys_n=20):
    import dot_ring_amd as d

    cv = d.Bandersnatch
    sks = [(int.from_bytes(hashlib.sha256(b"set-member-%d" % i).digest(), "little") % cv.curve.params.subgroup_order).to_bytes(32, "little")
           for i in range(keys_n)]
    keys = [cv.public_key_from_secret(sk) for sk in sks]
    alphas = [b"device-set-" + i.to_bytes(3, "little") * (1 + i % 4) for i in range(count)]
    ads = [b"ad" * (i % 3) for i in range(count)]
    return d, cv, keys, alphas, ads, [sks[i % keys_n] for i in range(count)], [keys[i % keys_n] for i in range(count)]


@pytest.mark.parametrize("devices,count", [("0,0", 37), ("0,0,0", 7), ("0,0,0", 2), ("0,0,0,0", 130)])
def test_device_set_gives_the_one_device_bytes(ctx, monkeypatch, devices, count):
    """ragged splits (37 = 19 + 18, 7 = 3 + 2 + 2), empty shards (2 proofs over 3 devices), more proofs per shard than the prover's host
    head takes (130 over 4: kernels for Elligator and x * I on two shards' worth... 33 / 33 / 32 / 32 stay on the host route; the
    first case's 19 too) — and the verifier over the same set: accepts, rejects a wrong input in the LAST shard"""
    from dot_ring_amd import runtime

    d, cv, keys, alphas, ads, sks, pks = _case(count)
    vrf = d.RingVRF[cv]
    params = d.RingProofParams.from_ring_size(len(keys), test_vectors=True)
    ring = d.Ring(keys, params)
    root = d.RingRoot.from_ring(ring, params)
    monkeypatch.delenv("DOTRING_DEVICES", raising=False)
    want = vrf.encode_batch(vrf.prove_batch(alphas, ads, sks, pks, ring, root))
    monkeypatch.setenv("DOTRING_DEVICES", devices)
    assert len(runtime.device_contexts()) == devices.count(",") + 1
    proofs = vrf.prove_batch(alphas, ads, sks, pks, ring, root)
    assert vrf.encode_batch(proofs) == want
    assert vrf.batch_verify(proofs, alphas, ads, ring, root)
    wrong = list(alphas)
    wrong[-1] = b"not the input"
    assert not vrf.batch_verify(proofs, wrong, ads, ring, root)
    # production mode over the set: random hidden rows, so only the verdicts can be checked — here by ONE device
    zparams = d.RingProofParams.from_ring_size(len(keys))
    zring = d.Ring(keys, zparams)
    zroot = d.RingRoot.from_ring(zring, zparams)
    zproofs = vrf.prove_batch(alphas, ads, sks, pks, zring, zroot)
    monkeypatch.delenv("DOTRING_DEVICES")
    assert vrf.batch_verify(zproofs, alphas, ads, zring, zroot)
    assert [p.encode() for p in zproofs] != [want[784 * i : 784 * i + 784] for i in range(count)]


def test_device_set_entry_points_refuse_shared_contexts(ctx):
    """a context cannot serve two shards at once: the same prover / context twice in the set is an argument error"""
    import ctypes

    from dot_ring_amd import _native
    from dot_ring_amd.ring_proof import device_prover

    d, cv, keys, alphas, ads, sks, pks = _case(4)
    vrf = d.RingVRF[cv]
    params = d.RingProofParams.from_ring_size(len(keys), test_vectors=True)
    ring = d.Ring(keys, params)
    root = d.RingRoot.from_ring(ring, params)
    prover = device_prover.get_device_prover(ring, 0)
    with pytest.raises(ValueError):
        _native.ringvrf_prove_batch_multi([prover, prover], vrf._suite_struct(), alphas, ads, None, b"".join(sks), ring.indices_of(pks),
                                          root.verifier_transcript_prefix_bytes(), None)
    proofs = vrf.prove_batch(alphas, ads, sks, pks, ring, root)
    assert vrf.batch_verify(proofs, alphas, ads, ring, root)
    vk = root.__dict__["_native_vk"]
    c = d.runtime.context() if hasattr(d, "runtime") else None
    from dot_ring_amd import runtime

    with pytest.raises(ValueError):
        _native.ringvrf_verify_batch_multi([runtime.context(), runtime.context()], vrf._suite_struct(), vk, vrf.encode_batch(proofs), alphas, ads, None,
                                           bytes(32))
