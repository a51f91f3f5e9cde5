"""Adam with the update of every parameter tensor in one HIP launch (``dpk_adam_step``).

``train_model(optimizer='adam')`` builds this instead of ``torch.optim.Adam(fused=True)`` for models of up to 96
parameter tensors on a HIP device: torch's fused kernel deals a tensor out in chunks of 65 536 elements, which leaves the
models of this path (3 .. 30 tensors, 50 k .. 1.5 M parameters) on a handful of work-groups -- 30 us of a 200 us RAT-SPN
step.  Same update rule and state names (``exp_avg``, ``exp_avg_sq``, ``step``) as ``torch.optim.Adam`` (non-amsgrad);
the step count lives on the device, so the optimiser is always capturable in a HIP graph.

Restriction (checked, not silent): ONE step count per parameter group, where ``torch.optim.Adam`` keeps one per parameter.
The two agree as long as every parameter of a group is updated from the group's first step on.  A parameter that first
receives a gradient later (frozen then unfrozen, a conditionally used branch) would get the bias corrections of step
k + 1 instead of step 1: ``step`` raises for it, and ``load_state_dict`` raises for a state whose per-parameter steps
differ -- use ``torch.optim.Adam`` for such schedules."""
import ctypes
from typing import Iterable

import torch

from deeprob.hip import load_library, check, HipError

MAX_TENSORS = 96


class _AdamTensor(ctypes.Structure):       # dpk_adam_tensor
    _fields_ = [('param', ctypes.c_void_p), ('grad', ctypes.c_void_p), ('exp_avg', ctypes.c_void_p),
                ('exp_avg_sq', ctypes.c_void_p), ('numel', ctypes.c_int64)]


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params: Iterable, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 maximize: bool = False, capturable: bool = True, fused: bool = True):
        if lr < 0.0 or eps < 0.0 or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or weight_decay < 0.0:
            raise ValueError("Invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, maximize=maximize,
                                      capturable=True))

    @staticmethod
    def supports(params) -> bool:
        params = [p for p in params if p.requires_grad]
        return (0 < len(params) <= MAX_TENSORS and
                all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in params))

    def _group_state(self, group, params):
        """The group's device-side step count (kept under torch.optim.Adam's state name ``step`` in every parameter's state,
        one shared tensor: it travels with ``state_dict()`` / ``load_state_dict()``) and its ticket word (scratch, not state)."""
        dev = params[0].device
        step_t = None
        for p in params:
            st = self.state[p].get('step')
            if torch.is_tensor(st):
                step_t = st.to(device=dev, dtype=torch.float32).reshape(1)
                break
        if step_t is None:
            step_t = torch.zeros(1, dtype=torch.float32, device=dev)
        for p in params:
            self.state[p]['step'] = step_t
        tickets = self.__dict__.setdefault('_dpk_tickets', {})
        key = (id(group), dev)
        if key not in tickets:
            tickets[key] = torch.zeros(1, dtype=torch.int32, device=dev)
        return step_t, tickets[key]

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        for gi, group in enumerate(self.param_groups):
            steps = set()
            for p in group['params']:
                st = self.state.get(p, {}).get('step')
                if st is not None:
                    steps.add(float(st.item()) if torch.is_tensor(st) else float(st))
            if len(steps) > 1:
                raise ValueError("FusedAdam.load_state_dict: the parameters of group {} carry different step counts {}; "
                                 "FusedAdam keeps one per group -- load this state into torch.optim.Adam".format(gi, sorted(steps)))
            # (parameters that already have moments count as members of the group's schedule)
            self.__dict__.setdefault('_dpk_seen', {})[gi] = {id(p) for p in group['params'] if 'exp_avg' in self.state.get(p, {})}

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = load_library()
        cache = self.__dict__.setdefault('_dpk_tables', {})
        for gi, group in enumerate(self.param_groups):
            keep, updated = [], []
            for p in group['params']:
                if p.grad is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                    raise HipError("FusedAdam: parameters must be contiguous fp32 tensors on a HIP device")
                g = p.grad if (p.grad.is_contiguous() and p.grad.dtype == torch.float32) else p.grad.float().contiguous()
                state = self.state[p]
                if 'exp_avg' not in state:
                    state['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                keep.append(g)
                updated.append(p)
            if not updated:
                continue
            if len(updated) > MAX_TENSORS:
                raise HipError("FusedAdam: {} parameter tensors in a group (at most {})".format(len(updated), MAX_TENSORS))
            seen = self.__dict__.setdefault('_dpk_seen', {}).setdefault(gi, set())
            late = [p for p in updated if id(p) not in seen]
            if late and seen and len(late) != len(updated):
                raise HipError("FusedAdam: {} parameter tensor(s) of group {} receive their first gradient after the group's "
                               "first step; FusedAdam keeps one step count per group (torch.optim.Adam: one per parameter) -- "
                               "use torch.optim.Adam for this schedule".format(len(late), gi))
            seen.update(id(p) for p in updated)
            step_t, ticket = self._group_state(group, updated)
            # the tensor table handed to the kernel: rebuilt only when the set of updated tensors (or their storage) changes;
            # per step only the gradient addresses move (zero_grad(set_to_none=True) gives every step fresh gradients)
            key = tuple((p.data_ptr(), id(self.state[p]['exp_avg']), id(self.state[p]['exp_avg_sq'])) for p in updated)
            hit = cache.get(gi)
            if hit is None or hit[0] != key:
                arr = (_AdamTensor * len(updated))(*[
                    _AdamTensor(p.data_ptr(), 0, self.state[p]['exp_avg'].data_ptr(), self.state[p]['exp_avg_sq'].data_ptr(),
                                p.numel()) for p in updated])
                touched = []
                for p in updated:
                    touched += [p, self.state[p]['exp_avg'], self.state[p]['exp_avg_sq']]
                hit = cache[gi] = (key, arr, touched)
            _, arr, touched = hit
            for i, g in enumerate(keep):
                arr[i].grad = g.data_ptr()
            entries = updated
            b1, b2 = group['betas']
            dev = group['params'][0].device
            check(lib.dpk_adam_step(len(entries), ctypes.cast(arr, ctypes.c_void_p), float(group['lr']), float(b1), float(b2),
                                    float(group['eps']), float(group['weight_decay']), int(bool(group['maximize'])),
                                    step_t.data_ptr(), ticket.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
                  'dpk_adam_step')
            # (the kernel wrote through raw pointers: tell autograd -- and the table caches keyed on version counters)
            torch.autograd.graph.increment_version(touched + [step_t])
            del keep
        return loss
