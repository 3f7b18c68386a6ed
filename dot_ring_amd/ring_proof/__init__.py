from .params import RingProofParams
from .pcs import KZG, SRS, Opening

__all__ = ["RingProofParams", "KZG", "SRS", "Opening"]
