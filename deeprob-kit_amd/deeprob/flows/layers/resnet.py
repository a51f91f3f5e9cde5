"""ResNet conditioner of the 2-D coupling layers behind the reference interface (deeprob/flows/layers/resnet.py:9-90).

The modules keep the reference's structure (``nn.Sequential`` of BatchNorm2d / ReLU / WeightNormConv2d) so that
``state_dict`` names match; neither evaluation nor training calls them one by one: every BatchNorm2d + ReLU is folded into the operand
load of the convolution that follows it and every residual / skip addition into the convolution that produces the
addend (csrc/flows2d.hip), so a residual block is two launches and the network 3 + 3 * n_blocks.  In training mode the
same folded kernels run with the scale / shift vectors of the batch statistics (deeprob/hip/ops_flows2d_train.py).
"""
import torch
from torch import nn

from deeprob.torch.utils import WeightNormConv2d
from deeprob.hip import ops_flows2d


class ResidualBlock(nn.Module):
    def __init__(self, n_channels: int):
        """BN-ReLU-conv3x3-BN-ReLU-conv3x3 with identity shortcut (reference :9-36)."""
        super().__init__()
        self.block = nn.Sequential(
            nn.BatchNorm2d(n_channels),
            nn.ReLU(inplace=True),
            WeightNormConv2d(n_channels, n_channels, kernel_size=3, padding=1, bias=False),
            nn.BatchNorm2d(n_channels),
            nn.ReLU(inplace=True),
            WeightNormConv2d(n_channels, n_channels, kernel_size=3, padding=1, bias=False)
        )

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        h = ops_flows2d.conv2d(x, self.block[2], bn=self.block[0])
        return ops_flows2d.conv2d(h, self.block[5], bn=self.block[3], res=x)


class ResidualNetwork(nn.Module):
    def __init__(self, in_channels: int, mid_channels: int, out_channels: int, n_blocks: int):
        """Residual network with skip connections (reference :39-90).

        :raises ValueError: if n_blocks is not positive."""
        if n_blocks <= 0:
            raise ValueError("n_blocks must be at least 1, got {}".format(n_blocks))
        super().__init__()
        self.blocks = nn.ModuleList()
        self.skips = nn.ModuleList()
        self.in_conv = WeightNormConv2d(in_channels, mid_channels, kernel_size=3, padding=1, bias=False)
        self.in_skip = WeightNormConv2d(mid_channels, mid_channels, kernel_size=1, padding=0, bias=True)
        for _ in range(n_blocks):
            self.blocks.append(ResidualBlock(mid_channels))
            self.skips.append(WeightNormConv2d(mid_channels, mid_channels, kernel_size=1, padding=0, bias=True))
        self.out_network = nn.Sequential(
            nn.BatchNorm2d(mid_channels),
            nn.ReLU(inplace=True),
            WeightNormConv2d(mid_channels, out_channels, kernel_size=1, padding=0, bias=True)
        )

    def forward(self, x: torch.Tensor, in_mask=None) -> torch.Tensor:
        """`in_mask` [H, W]: evaluate the network on ``in_mask * x`` (the checkerboard coupling's masked input) without
        materialising the product."""
        x = ops_flows2d.conv2d(x, self.in_conv, in_mask=in_mask)
        z = ops_flows2d.conv2d(x, self.in_skip)
        for block, skip in zip(self.blocks, self.skips):
            x = block(x)
            z = ops_flows2d.conv2d(x, skip, res=z, out=z)
        return ops_flows2d.conv2d(z, self.out_network[2], bn=self.out_network[0])
