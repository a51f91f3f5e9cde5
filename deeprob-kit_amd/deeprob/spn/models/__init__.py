from .dgcspn import DgcSpn
from .ratspn import RatSpn, GaussianRatSpn, BernoulliRatSpn
