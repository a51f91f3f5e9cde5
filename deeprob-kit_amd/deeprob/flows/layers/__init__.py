from .coupling import CouplingLayer1d, CouplingLayer2d, CouplingBlock2d
from .densenet import DenseLayer, DenseBlock, Transition, DenseNetwork
from .resnet import ResidualBlock, ResidualNetwork
