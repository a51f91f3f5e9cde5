from .dgcspn import SpatialGaussianLayer, SpatialProductLayer, SpatialSumLayer, SpatialRootLayer
from .ratspn import RegionGraphLayer, GaussianLayer, BernoulliLayer, ProductLayer, SumLayer, RootLayer
