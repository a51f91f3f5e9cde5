"""Running averages (interface of deeprob/torch/metrics.py:17-51).  The FID helpers of the reference are image-GAN
utilities outside the density-evaluation path."""
from typing import Union

import torch


class RunningAverageMetric:
    """Sample-weighted running average.  Accepts python floats like the reference, or 0-d device tensors: those
    are accumulated ON THE DEVICE (no host synchronisation per batch) and read back once by ``average()``."""

    def __init__(self):
        self.reset()

    def reset(self):
        self._total = 0.0       # float, or a 0-d device tensor once a tensor has been fed
        self._count = 0

    def __call__(self, metric: Union[float, torch.Tensor], num_samples: int):
        if torch.is_tensor(metric):
            metric = metric.detach().to(torch.float64)
        self._total = self._total + metric * num_samples
        self._count += num_samples

    def average(self) -> float:
        total = self._total.item() if torch.is_tensor(self._total) else self._total
        return total / self._count
