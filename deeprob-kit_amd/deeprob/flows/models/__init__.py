from .base import NormalizingFlow
from .realnvp import RealNVP1d, RealNVP2d
