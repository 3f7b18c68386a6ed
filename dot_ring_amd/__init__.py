"""dot_ring_amd — MI355X-native (gfx950) Ring-VRF hot path behind dot-ring's Python API.

The arithmetic lives in libdotring_hip.so (hand-written HIP, see dot_ring_amd/csrc and include/dotring_hip.h);
this package mirrors the reference's public names (dot_ring/__init__.py:3-19) for the Bandersnatch suites and JubJub:
    TinyVRF, ThinVRF, PedersenVRF, RingVRF, Ring, RingRoot, RingProofParams, Bandersnatch, Bandersnatch_SHAKE128, JubJub
plus the additive prove_batch() entry points.  There is no CPU fallback for the kernels.
"""
from . import _native  # noqa: F401
from .curve import Bandersnatch, Bandersnatch_SHAKE128, JubJub
from .ring_proof.params import RingProofParams
from .ring_proof.pcs import KZG
from .vrf.pedersen import PedersenVRF
from .vrf.ring_vrf import Ring, RingRoot, RingVRF
from .vrf.thin import ThinVRF
from .vrf.tiny import TinyVRF

__all__ = ["TinyVRF", "ThinVRF", "PedersenVRF", "RingVRF", "Ring", "RingRoot", "RingProofParams", "KZG",
           "Bandersnatch", "Bandersnatch_SHAKE128", "JubJub"]
