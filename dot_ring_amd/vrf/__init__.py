from .pedersen import PedersenVRF
from .ring_vrf import Ring, RingRoot, RingVRF
from .tiny import TinyVRF

__all__ = ["TinyVRF", "PedersenVRF", "RingVRF", "Ring", "RingRoot"]
