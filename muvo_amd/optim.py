"""Fused AdamW (torch.optim.AdamW semantics, muvo/trainer.py:1031-1060) on the flat ParamStore.

It is a torch.optim.Optimizer so that torch's OneCycleLR (host-side scalar schedule, incl. its beta1 cycling)
and Lightning can drive it; the arithmetic is muvo_adamw_step (one launch per contiguous range).

Checkpoints: `state_dict()` / `load_state_dict()` speak torch.optim.AdamW's format — `state[i] = {'step', 'exp_avg',
'exp_avg_sq'}` keyed by the parameter's index in the two param groups, which list the parameters in the reference's
order (trainer.py:1031-1051) — so a Lightning checkpoint written by either implementation restores the moments and the
bias-correction step count of the other.  The moments themselves live in the store's flat buffers."""
import torch

from muvo_amd import ops


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, store, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, extra_unused=()):
        self.store = store
        groups = [{'params': store.nodecay_params, 'weight_decay': 0.0},
                  {'params': store.decay_params, 'weight_decay': weight_decay}]
        super().__init__(groups, dict(lr=lr, betas=betas, eps=eps, weight_decay=0.0))
        self._step = 0
        self.grad_scale = 1.0  # 1/world_size after a sum all-reduce

    def zero_grad(self, set_to_none=False):
        """One memset of the flat gradient buffer.  `set_to_none` is accepted for torch / Lightning callers; the gradients
        stay views of the flat buffer either way (the kernels accumulate into it)."""
        self.store.zero_grad()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._step += 1
        st = self.store
        ops.join_side_streams()      # branches on side streams (ops.branch) wrote their parameter gradients there
        st.settle_grads()
        for _, decay, start, end in st.ranges:
            g = self.param_groups[1 if decay else 0]
            b1, b2 = g['betas']
            ops.adamw_step(st.flat_param[start:end], st.flat_grad[start:end], st.exp_avg[start:end],
                           st.exp_avg_sq[start:end], g['lr'], b1, b2, g['eps'], g['weight_decay'], self._step,
                           self.grad_scale)
        ops.bump_weight_epoch()
        return loss

    # ------------------------------------------------------------------ checkpointing (torch.optim.AdamW format)
    def _indexed_params(self):
        i = 0
        for g in self.param_groups:
            for p in g['params']:
                yield i, p
                i += 1

    def state_dict(self):
        st = self.store
        state = {}
        if self._step > 0:
            for i, p in self._indexed_params():
                if id(p) not in st.used_ids:
                    continue                      # never receives a gradient: torch.optim.AdamW keeps no state for it
                o, n = st._off[id(p)], p.numel()
                state[i] = {'step': torch.tensor(float(self._step)),
                            'exp_avg': st.exp_avg[o:o + n].view(p.shape).clone(),
                            'exp_avg_sq': st.exp_avg_sq[o:o + n].view(p.shape).clone()}
        groups, i = [], 0
        for g in self.param_groups:
            d = {k: v for k, v in g.items() if k != 'params'}
            d['params'] = list(range(i, i + len(g['params'])))
            i += len(g['params'])
            groups.append(d)
        return {'state': state, 'param_groups': groups}

    @torch.no_grad()
    def load_state_dict(self, state_dict):
        groups = state_dict['param_groups']
        if [len(g['params']) for g in groups] != [len(g['params']) for g in self.param_groups]:
            raise ValueError('optimizer state_dict has different parameter groups: '
                             f'{[len(g["params"]) for g in groups]} vs {[len(g["params"]) for g in self.param_groups]}')
        for g, src in zip(self.param_groups, groups):
            for k, v in src.items():
                if k != 'params':
                    g[k] = v
        st = self.store
        st.exp_avg.zero_()
        st.exp_avg_sq.zero_()
        steps = set()
        state = {int(k): v for k, v in state_dict['state'].items()}
        for i, p in self._indexed_params():
            s = state.get(i)
            if s is None:
                continue
            if id(p) not in st.used_ids:
                raise ValueError(f'optimizer state for parameter {i}, which never receives a gradient here')
            o, n = st._off[id(p)], p.numel()
            st.exp_avg[o:o + n].copy_(s['exp_avg'].reshape(-1))
            st.exp_avg_sq[o:o + n].copy_(s['exp_avg_sq'].reshape(-1))
            steps.add(int(float(s['step'])))
        if len(steps) > 1:
            raise ValueError(f'per-parameter step counts differ ({sorted(steps)}): the fused kernel keeps one count')
        self._step = steps.pop() if steps else 0
