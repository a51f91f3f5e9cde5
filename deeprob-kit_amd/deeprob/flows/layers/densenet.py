"""DenseNet conditioner of the 2-D coupling layers behind the reference interface
(deeprob/flows/layers/densenet.py).  Evaluation: the concatenation of a dense block is one preallocated
[B, 5 * mid, H, W] tensor whose channel slices the convolutions read and write directly (no torch.cat), every
BatchNorm2d + ReLU folded into the operand load of the convolution that follows it (csrc/flows2d.hip).
`use_checkpoint` only matters for training memory and is accepted for interface parity."""
from typing import List

import torch
from torch import nn

from deeprob.torch.utils import WeightNormConv2d
from deeprob.hip import ops_flows2d


class DenseLayer(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, use_checkpoint: bool = False):
        """BN-ReLU-conv1x1 bottleneck (4 * out_channels) then BN-ReLU-conv3x3 (reference :11-40)."""
        super().__init__()
        self.use_checkpoint = use_checkpoint
        mid_channels = 4 * out_channels
        self.bottleneck_network = nn.Sequential(
            nn.BatchNorm2d(in_channels),
            nn.ReLU(inplace=True),
            WeightNormConv2d(in_channels, mid_channels, kernel_size=1, padding=0, bias=False)
        )
        self.network = nn.Sequential(
            nn.BatchNorm2d(mid_channels),
            nn.ReLU(inplace=True),
            WeightNormConv2d(mid_channels, out_channels, kernel_size=3, padding=1, bias=False)
        )

    def evaluate(self, stacked: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
        """`stacked` = the channel concatenation of the inputs (a slice of the block's buffer)."""
        h = ops_flows2d.conv2d(stacked, self.bottleneck_network[2], bn=self.bottleneck_network[0])
        return ops_flows2d.conv2d(h, self.network[2], bn=self.network[0], out=out)

    def forward(self, inputs: List[torch.Tensor]) -> torch.Tensor:
        return self.evaluate(inputs[0] if len(inputs) == 1 else torch.cat(list(inputs), dim=1))


class DenseBlock(nn.Module):
    def __init__(self, n_layers: int, in_channels: int, out_channels: int, use_checkpoint: bool = False):
        """`n_layers` dense layers, each fed the concatenation of everything before it (reference :83-117)."""
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.layers = nn.ModuleList()
        for i in range(n_layers):
            self.layers.append(DenseLayer(in_channels + i * out_channels, out_channels, use_checkpoint=use_checkpoint))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if ops_flows2d.graph_route(x, self):
            # graph-building route: the concatenations are torch.cat nodes (deeprob/hip/ops_flows2d_train.py)
            features = [x]
            for layer in self.layers:
                features.append(layer.evaluate(features[0] if len(features) == 1 else torch.cat(features, dim=1)))
            return torch.cat(features, dim=1)
        B, C, H, W = x.shape
        total = C + len(self.layers) * self.out_channels
        buf = torch.empty((B, total, H, W), dtype=torch.float32, device=x.device)
        buf[:, :C].copy_(x)
        for i, layer in enumerate(self.layers):
            lo = C + i * self.out_channels
            layer.evaluate(buf[:, :lo], out=buf[:, lo:lo + self.out_channels])
        return buf


class Transition(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, bias: bool = True):
        """BN-ReLU-conv1x1 (reference :120-149)."""
        super().__init__()
        self.network = torch.nn.Sequential(
            nn.BatchNorm2d(in_channels),
            nn.ReLU(inplace=True),
            WeightNormConv2d(in_channels, out_channels, kernel_size=1, padding=0, bias=bias)
        )

    def forward(self, x):
        return ops_flows2d.conv2d(x, self.network[2], bn=self.network[0])


class DenseNetwork(nn.Module):
    def __init__(self, in_channels: int, mid_channels: int, out_channels: int, n_blocks: int,
                 use_checkpoint: bool = False):
        """Input convolution, then `n_blocks` x (dense block of 4 layers + transition) (reference :152-189)."""
        super().__init__()
        self.blocks = nn.ModuleList()
        self.in_conv = WeightNormConv2d(in_channels, mid_channels, kernel_size=3, padding=1, bias=False)
        for i in range(n_blocks):
            self.blocks.append(DenseBlock(4, mid_channels, mid_channels, use_checkpoint=use_checkpoint))
            if i == n_blocks - 1:
                self.blocks.append(Transition(5 * mid_channels, out_channels, bias=True))
            else:
                self.blocks.append(Transition(5 * mid_channels, mid_channels, bias=False))

    def forward(self, x: torch.Tensor, in_mask=None) -> torch.Tensor:
        x = ops_flows2d.conv2d(x, self.in_conv, in_mask=in_mask)
        for block in self.blocks:
            x = block(x)
        return x
