"""RAT-SPN models behind the reference interface (deeprob/spn/models/ratspn.py), evaluated on MI355X.

``RatSpn.forward`` has two routes, both made of hand-written HIP kernels only:
  * inference (no autograd graph needed, Gaussian leaves, shape inside the built set): ONE fused
    launch for the whole model (``dpk_ratspn_forward``);
  * otherwise: the per-layer operators chained exactly like the reference's python loop, each with
    its own backward kernel.
"""
from typing import Optional, Tuple, Type

import torch

from deeprob.utils.random import RandomState
from deeprob.utils.region import RegionGraph
from deeprob.torch.base import ProbabilisticModel
from deeprob.torch.constraints import ScaleClipper
from deeprob.spn.layers.ratspn import RegionGraphLayer, GaussianLayer, BernoulliLayer
from deeprob.spn.layers.ratspn import SumLayer, ProductLayer, RootLayer
from deeprob.hip import ops


class RatSpn(ProbabilisticModel):
    def __init__(
        self,
        in_features: int,
        base_cls: Type[RegionGraphLayer],
        base_kwargs: Optional[dict] = None,
        out_classes: int = 1,
        rg_depth: int = 2,
        rg_repetitions: int = 1,
        rg_batch: int = 2,
        rg_sum: int = 2,
        in_dropout: Optional[float] = None,
        sum_dropout: Optional[float] = None,
        random_state: Optional[RandomState] = None
    ):
        """
        Randomized-and-tensorized SPN (constructor contract of the reference, ratspn.py:17-103).

        :param in_features: number of input features.
        :param base_cls: leaf layer class, a sub-class of RegionGraphLayer.
        :param base_kwargs: extra keyword arguments of the leaf layer.
        :param out_classes: number of root nodes (1 = plain density estimation).
        :param rg_depth: region graph depth.
        :param rg_repetitions: number of random repetitions of the region graph.
        :param rg_batch: leaf distributions per region.
        :param rg_sum: sum nodes per inner region.
        :param in_dropout: leaf dropout rate or None.
        :param sum_dropout: sum layer dropout rate or None.
        :param random_state: None, a seed or a NumPy RandomState.
        :raises ValueError: if a parameter is out of domain.
        """
        if not issubclass(base_cls, RegionGraphLayer):
            raise ValueError("The base distribution's class must be a sub-class of RegionGraphLayer")
        if in_features <= 0:
            raise ValueError("in_features must be at least 1")
        if out_classes <= 0:
            raise ValueError("The number of output classes must be positive")
        if rg_batch <= 0:
            raise ValueError("The number of base distribution batches must be positive")
        if rg_sum <= 0:
            raise ValueError("The number of sum nodes per region must be positive")
        if in_dropout is not None and not 0.0 < in_dropout < 1.0:
            raise ValueError("The dropout rate at base distribution must be in (0, 1)")
        if sum_dropout is not None and not 0.0 < sum_dropout < 1.0:
            raise ValueError("The dropout rate at sum layers must be in (0, 1)")

        super().__init__()
        self.in_features = in_features
        self.out_classes = out_classes
        self.rg_depth = rg_depth
        self.rg_batch = rg_batch
        self.rg_sum = rg_sum
        self.in_dropout = in_dropout
        self.sum_dropout = sum_dropout
        self.layers = torch.nn.ModuleList()

        # leaves first: [leaf regions], [partitions], [regions], ..., [root]
        graph = RegionGraph(self.in_features, self.rg_depth, random_state)
        self.rg_layers = list(reversed(graph.make_layers(rg_repetitions)))
        self.rg_repetitions = rg_repetitions

        self.base_layer = base_cls(
            self.in_features, self.rg_batch,
            regions=self.rg_layers[0], rg_depth=self.rg_depth, dropout=self.in_dropout,
            **(base_kwargs or {})
        )

        # odd levels are partitions (Product), even levels are regions (Sum)
        groups, nodes = self.base_layer.in_regions, self.base_layer.out_channels
        for level in range(1, len(self.rg_layers) - 1):
            if level % 2 == 1:
                layer = ProductLayer(groups, nodes)
                groups, nodes = layer.out_partitions, layer.out_nodes
            else:
                layer = SumLayer(groups, nodes, self.rg_sum, self.sum_dropout)
                groups, nodes = layer.out_regions, layer.out_nodes
            self.layers.append(layer)
        self.root_layer = RootLayer(groups, nodes, self.out_classes)

        self._fused_declined = False
        self._train_fused_declined = False
        self._train_fused_max_batch = 1 << 62
        self._fused_ctx = ops.LeafContext(
            self.in_features, self.base_layer.in_regions, self.rg_batch, self.base_layer.dimension,
            depth=self.rg_depth, reps=rg_repetitions, sums=self.rg_sum, classes=self.out_classes
        )

    def _needs_graph(self, x: torch.Tensor) -> bool:
        if not torch.is_grad_enabled():
            return False
        return x.requires_grad or any(p.requires_grad for p in self.parameters())

    def _forward_fused(self, x: torch.Tensor, ll_acc: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
        if not isinstance(self.base_layer, GaussianLayer):
            return None
        if self.training and (self.in_dropout is not None or self.sum_dropout is not None):
            return None
        base = self.base_layer
        sum_weights = [layer.weight for layer in self.layers if isinstance(layer, SumLayer)]
        return ops.ratspn_forward_fused(
            x, base.mask, base._pad_mask_or_none(), base.loc, base.scale, sum_weights,
            self.root_layer.weight, self._fused_ctx, ll_acc
        )

    def _forward_train_fused(self, x: torch.Tensor) -> Optional[torch.Tensor]:
        base = self.base_layer
        if self._train_fused_declined or not isinstance(base, GaussianLayer) or self.rg_depth != 2:
            return None
        if base.scale.requires_grad or x.requires_grad or not x.is_cuda or x.shape[0] > self._train_fused_max_batch:
            return None
        if self.training and (self.in_dropout is not None or self.sum_dropout is not None):
            return None
        layers = list(self.layers)
        if not (len(layers) == 3 and isinstance(layers[0], ProductLayer) and isinstance(layers[1], SumLayer)
                and isinstance(layers[2], ProductLayer)):
            return None
        out = ops.ratspn_forward_train(x, base.mask, base._pad_mask_or_none(), base.loc, base.scale, layers[1].weight,
                                       self.root_layer.weight, self._fused_ctx, base._leaf_ctx, layers[1]._ws,
                                       self.root_layer._ws)
        if out is None:
            # (8 channels: declined on the model's constants -- sums, classes, repetitions: do not ask again; the 2 / 4
            # channel kernels stop at dpk_ratspn_small_batch_max samples: do not ask again for batches this large)
            if self.rg_batch == 8:
                self._train_fused_declined = True
            else:
                self._train_fused_max_batch = min(self._train_fused_max_batch, x.shape[0] - 1)
        return out

    def _prefer_folded(self, x: torch.Tensor) -> bool:
        """8-channel unit-scale models: the leaf layer and the folded product / sum layers on the matrix cores beat the
        single-launch VALU kernel (measured 0.065 vs 0.194 ms at 4096 samples, 0.209 vs 0.243 ms at 65536;
        tools/bench_wide.py)."""
        base = self.base_layer
        if not (isinstance(base, GaussianLayer) and self.rg_batch == 8 and self.rg_sum in (8, 16)
                and not base.scale.requires_grad and self.in_features % 4 == 0):
            return False
        # round 3: depth-2 models with up to 8 repetitions and 32 classes run as ONE launch (a wave per repetition,
        # csrc/ratspn_gemm_wide.hip) -- the single-launch route then comes first.  Whether THIS call is inside that
        # kernel's envelope (shape, LDS budget, unit-scale hint, 16-byte aligned x) is the library's answer, not a
        # re-derivation here: outside it the fused entry point would fall to the VALU kernel, ~3x slower than the
        # folded MFMA route.
        if self.rg_sum != 8 or not x.is_cuda:
            return True
        ctx = self._fused_ctx
        # (scale.requires_grad is False here: LeafContext.workspace sets the same hint; a tensor the operator would
        # convert / compact first arrives 16-byte aligned from the allocator: address 0 stands for it)
        plain = x.dtype == torch.float32 and x.is_contiguous()
        one_launch = ops.load_library().dpk_ratspn_forward_on_mfma(
            x.data_ptr() if plain else 0, ctx.D, ctx.depth, ctx.reps, ctx.I, ctx.S, ctx.C, 0, ops.DPK_FLAG_UNIT_SCALE)
        return not one_launch

    def _forward_folded(self, x: torch.Tensor) -> Optional[torch.Tensor]:
        """Evaluation outside the single-launch kernel's envelope (e.g. rg_batch = rg_sum = 16): leaf kernel, then
        every ProductLayer folded into the Sum / Root layer above it (the product tensors are never written)."""
        if self.training and (self.in_dropout is not None or self.sum_dropout is not None):
            return None
        with torch.no_grad():
            h = self.base_layer(x)
            layers = list(self.layers)
            # depth 2: the sum layer's and the root layer's tables in one launch (each layer rebuilt its own per call)
            current = False
            if (len(layers) == 3 and isinstance(layers[0], ProductLayer) and isinstance(layers[1], SumLayer)
                    and isinstance(layers[2], ProductLayer) and h.dim() == 3 and h.is_cuda):
                R0, N0 = h.shape[1], h.shape[2]
                current = ops.upper_tables_pair(layers[1].weight, layers[1]._ws, R0, N0, self.root_layer.weight,
                                                self.root_layer._ws, R0 // 2, layers[1].weight.shape[1], h.device)
            i = 0
            while i < len(layers):
                if not isinstance(layers[i], ProductLayer):
                    return None
                if i + 1 < len(layers):
                    nxt = layers[i + 1]
                    if not isinstance(nxt, SumLayer):
                        return None
                    h = ops.prodsum_forward(h, nxt.weight, nxt._ws, tables_current=current)
                    i += 2
                else:
                    h = ops.prodroot_forward(h, self.root_layer.weight, self.root_layer._ws, tables_current=current)
                    i += 1
                    if h is None:
                        return None
                    return h
                if h is None:
                    return None
            return None

    def fused_plan(self, x: torch.Tensor, static_params: bool = False) -> Optional['ops.FusedForwardPlan']:
        """A pre-bound fused forward for a resident input buffer (see ``ops.FusedForwardPlan``); None when the
        model is outside the fused kernel's envelope.  ``static_params``: the caller guarantees a frozen model (the
        plan then skips the device-side check of the cached parameter tables)."""
        if not isinstance(self.base_layer, GaussianLayer):
            return None
        if self.training and (self.in_dropout is not None or self.sum_dropout is not None):
            return None
        base = self.base_layer
        sum_weights = [layer.weight for layer in self.layers if isinstance(layer, SumLayer)]
        plan = ops.FusedForwardPlan(x, base.mask, base._pad_mask_or_none(), base.loc, base.scale, sum_weights,
                                    self.root_layer.weight, self._fused_ctx, static_params=static_params)
        return plan if plan.supported else None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """
        Log-likelihood ``[B, out_classes]`` of the evidence ``x [B, D]``; NaN entries are marginalised
        (reference: ratspn.py:105-122).
        """
        if not self._needs_graph(x):
            if self._prefer_folded(x):
                out = self._forward_folded(x)
                if out is not None:
                    return out
            # (the single-launch kernel declines on (depth, channels, sums, classes) alone -- constants of the model: once
            # it has, later calls do not ask again; the attempt was a quarter of a wide model's host time per call)
            if not self._fused_declined:
                out = self._forward_fused(x)
                if out is not None:
                    return out
                self._fused_declined = isinstance(self.base_layer, GaussianLayer) and not self.training
            out = self._forward_folded(x)
            if out is not None:
                return out
        # training / gradients.  Depth-2 models inside the single-launch kernel's envelope: that kernel, writing what the
        # backward needs on its way up (ops.RatSpnTrainFn: one launch instead of leaf + 2 folded levels + 3 table builds)
        out = self._forward_train_fused(x)
        if out is not None:
            return out
        # otherwise the layer chain, every ProductLayer folded with the Sum / Root layer above it into one
        # autograd node (ops.ProdSumFn: no [B, P, N^2] product tensor in the forward)
        x = self.base_layer(x)
        layers, i = list(self.layers), 0
        while i < len(layers):
            layer = layers[i]
            if isinstance(layer, ProductLayer) and not (self.training and self.sum_dropout is not None):
                if i + 1 < len(layers) and isinstance(layers[i + 1], SumLayer):
                    y = ops.prodsum_autograd(x, layers[i + 1].weight, layers[i + 1]._ws)
                    if y is not None:
                        x, i = y, i + 2
                        continue
                elif i + 1 == len(layers):
                    y = ops.prodsum_autograd(x, self.root_layer.weight, self.root_layer._ws, root=True)
                    if y is not None:
                        return y
            x = layer(x)
            i += 1
        return self.root_layer(x)

    # ---- top-down passes: one launch (csrc/ratspn_topdown.hip) ------------------------------------------------------------
    def _topdown_src(self) -> torch.Tensor:
        """``[reps, D]`` int32: where variable f sits in a repetition's region-major row (the reference's ``inv_mask`` with the
        dummy variables dropped -- what ``unpad_samples``, ratspn.py:68-85, is meant to select; its own ``samples[inv_pad_mask]``
        keeps the DUMMY entries instead and raises on every padded region graph, a defect this mirror does not reproduce)."""
        base = self.base_layer
        key = (base.inv_mask.device, base.inv_mask.data_ptr())
        if getattr(self, '_td_src_key', None) != key:
            inv = base.inv_mask
            if base.pad > 0:
                inv = inv[~base.inv_pad_mask].view(inv.shape[0], self.in_features)
            self._td_src = inv.to(torch.int32).contiguous()
            self._td_src_key = key
        return self._td_src

    def _topdown_logw(self):
        """log_softmax of every sum level's weight (bottom to top) and of the root's: the tables of ratspn.py:397 / :415 /
        :470 / :488, formed by the same torch call as the reference forms them."""
        out = [torch.log_softmax(layer.weight, dim=2) for layer in self.layers if isinstance(layer, SumLayer)]
        out.append(torch.log_softmax(self.root_layer.weight, dim=1))
        return out

    def _leaf_params(self):
        base = self.base_layer
        if isinstance(base, GaussianLayer):
            return 0, base.loc, base.scale
        if isinstance(base, BernoulliLayer):
            return 1, base.logits, None
        return None

    def _upward_for_mpe(self, x: torch.Tensor):
        """Leaf and sum-level outputs (no product tensors) + the input of the root as (regions, nodes)."""
        h = self.base_layer(x)
        acts = [h]
        layers = list(self.layers)
        i = 0
        while i + 1 < len(layers):
            # (Product, Sum) pairs; the last ProductLayer belongs to the root
            y = ops.prodsum_forward(h, layers[i + 1].weight, layers[i + 1]._ws)
            if y is None:
                y = layers[i + 1](layers[i](h))
            h = y
            acts.append(h)
            i += 2
        return acts

    @torch.no_grad()
    def mpe(self, x: torch.Tensor, y: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Most probable completion of the NaN entries of ``x`` (reference: ratspn.py:124-162): the bottom-up pass through
        the leaf and folded product+sum kernels, the whole top-down pass in one launch."""
        leaf = self._leaf_params()
        if leaf is None or (self.training and (self.in_dropout is not None or self.sum_dropout is not None)):
            return self._mpe_layerwise(x, y)
        acts = self._upward_for_mpe(x)
        if self.out_classes == 1:
            y = None
        elif y is None:
            top = ops.prodroot_forward(acts[-1], self.root_layer.weight, self.root_layer._ws)
            if top is None:
                top = self.root_layer(self.layers[-1](acts[-1]))
            y = torch.argmax(top, dim=1)
        dist, p0, p1 = leaf
        return ops.ratspn_topdown(0, dist, x.shape[0], self._fused_ctx, x, y, acts, self._topdown_logw(),
                                  self._topdown_src(), p0, p1)

    @torch.no_grad()
    def sample(self, n_samples: int, y: Optional[torch.Tensor] = None, seed: Optional[int] = None) -> torch.Tensor:
        """Ancestral sampling, top-down (reference: ratspn.py:164-182), one launch.  The draws come from the library's
        counter-based hash seeded from torch's generator (``seed``: fix them, e.g. to replay a batch)."""
        leaf = self._leaf_params()
        if leaf is None:
            return self._sample_layerwise(n_samples, y)
        device = self.root_layer.weight.device
        if self.out_classes == 1:
            y = None
        elif y is None:
            y = torch.randint(self.out_classes, [n_samples], device=device)
        dist, p0, p1 = leaf
        return ops.ratspn_topdown(1, dist, n_samples, self._fused_ctx, None, y, None, self._topdown_logw(),
                                  self._topdown_src(), p0, p1, seed=ops.draw_seed() if seed is None else int(seed))

    @torch.no_grad()
    def _mpe_layerwise(self, x: torch.Tensor, y: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The layer-by-layer form (user-defined leaf layers, training-mode dropout): the layers' own ``mpe`` methods."""
        evidence = x
        n_samples = x.shape[0]
        x = self.base_layer(x)
        lls = []
        for layer in self.layers:
            lls.append(x)
            x = layer(x)
        if self.out_classes == 1:
            y = torch.zeros(n_samples, dtype=torch.long, device=x.device)
        elif y is None:
            y = torch.argmax(self.root_layer(x), dim=1)
        idx_group, idx_offset = self.root_layer.mpe(x, y)
        for i in reversed(range(len(self.layers))):
            idx_group, idx_offset = self.layers[i].mpe(lls[i], idx_group, idx_offset)
        return self.base_layer.mpe(evidence, idx_group, idx_offset)

    @torch.no_grad()
    def _sample_layerwise(self, n_samples: int, y: Optional[torch.Tensor] = None) -> torch.Tensor:
        device = self.root_layer.weight.device
        if self.out_classes == 1:
            y = torch.zeros(n_samples, dtype=torch.long, device=device)
        elif y is None:
            y = torch.randint(self.out_classes, [n_samples], device=device)
        idx_group, idx_offset = self.root_layer.sample(y)
        for i in reversed(range(len(self.layers))):
            idx_group, idx_offset = self.layers[i].sample(idx_group, idx_offset)
        return self.base_layer.sample(idx_group, idx_offset)

    def loss(self, x: torch.Tensor, y: Optional[torch.Tensor] = None) -> torch.Tensor:
        """-mean LL (generative) or cross entropy over classes (reference: ratspn.py:184-191)."""
        if self.out_classes == 1:
            return ops.neg_mean(x)
        return torch.nn.functional.nll_loss(torch.log_softmax(x, dim=1), y)


class GaussianRatSpn(RatSpn):
    def __init__(
        self,
        in_features: int,
        out_classes: int = 1,
        rg_depth: int = 2,
        rg_repetitions: int = 1,
        rg_batch: int = 2,
        rg_sum: int = 2,
        in_dropout: Optional[float] = None,
        sum_dropout: Optional[float] = None,
        random_state: Optional[RandomState] = None,
        uniform_loc: Optional[Tuple[float, float]] = None,
        optimize_scale: bool = False
    ):
        """RAT-SPN with Gaussian leaves (reference: ratspn.py:194-239)."""
        super().__init__(
            in_features, GaussianLayer, {'uniform_loc': uniform_loc, 'optimize_scale': optimize_scale},
            out_classes, rg_depth, rg_repetitions, rg_batch, rg_sum, in_dropout, sum_dropout, random_state
        )
        self.optimize_scale = optimize_scale
        if self.optimize_scale:
            self.scale_clipper = ScaleClipper()

    def apply_constraints(self):
        if self.optimize_scale:
            self.scale_clipper(self.base_layer)


class BernoulliRatSpn(RatSpn):
    def __init__(
        self,
        in_features: int,
        out_classes: int = 1,
        rg_depth: int = 2,
        rg_repetitions: int = 1,
        rg_batch: int = 2,
        rg_sum: int = 2,
        in_dropout: Optional[float] = None,
        sum_dropout: Optional[float] = None,
        random_state: Optional[RandomState] = None
    ):
        """RAT-SPN with Bernoulli leaves (reference: ratspn.py:242-273)."""
        super().__init__(
            in_features, BernoulliLayer, None,
            out_classes, rg_depth, rg_repetitions, rg_batch, rg_sum, in_dropout, sum_dropout, random_state
        )
