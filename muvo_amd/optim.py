"""Fused AdamW (torch.optim.AdamW semantics, muvo/trainer.py:1031-1060) on the flat ParamStore.

It is a torch.optim.Optimizer so that torch's OneCycleLR (host-side scalar schedule, incl. its beta1 cycling)
and Lightning can drive it; the arithmetic is muvo_adamw_step (one launch per contiguous range)."""
import torch

from muvo_amd import ops


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, store, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, extra_unused=()):
        self.store = store
        groups = [{'params': store.nodecay_params + [p for p in extra_unused if p.dim() == 1], 'weight_decay': 0.0},
                  {'params': store.decay_params + [p for p in extra_unused if p.dim() != 1],
                   'weight_decay': weight_decay}]
        super().__init__(groups, dict(lr=lr, betas=betas, eps=eps, weight_decay=0.0))
        self._step = 0
        self.grad_scale = 1.0  # 1/world_size after a sum all-reduce

    def zero_grad(self, set_to_none=False):
        self.store.zero_grad()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._step += 1
        st = self.store
        for _, decay, start, end in st.ranges:
            g = self.param_groups[1 if decay else 0]
            b1, b2 = g['betas']
            ops.adamw_step(st.flat_param[start:end], st.flat_grad[start:end], st.exp_avg[start:end],
                           st.exp_avg_sq[start:end], g['lr'], b1, b2, g['eps'], g['weight_decay'], self._step,
                           self.grad_scale)
        ops.bump_weight_epoch()
        return loss
