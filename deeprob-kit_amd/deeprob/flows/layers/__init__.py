from .coupling import CouplingLayer1d
