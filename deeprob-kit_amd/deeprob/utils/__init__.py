from .random import RandomState, check_random_state
from .region import RegionGraph
