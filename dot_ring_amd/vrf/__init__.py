from .pedersen import PedersenVRF
from .ring_vrf import Ring, RingRoot, RingVRF
from .thin import ThinVRF
from .tiny import TinyVRF

__all__ = ["TinyVRF", "ThinVRF", "PedersenVRF", "RingVRF", "Ring", "RingRoot"]
