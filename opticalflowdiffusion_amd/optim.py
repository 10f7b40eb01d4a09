"""`FusedAdam`: the optimiser FlowDiffuser configures (flow_diffuser.py:131-134,
`torch.optim.Adam(lr=cfg.lr, weight_decay=cfg.weight_decay)`) with the trainer's gradient clipping
(`gradient_clip_val`, experiments/exp_base.py:192,205) folded in, as three HIP launches per step
over all parameters (`ofd_adam_step`).  State-dict compatible with torch.optim.Adam
(`exp_avg`, `exp_avg_sq`, `step`)."""
import struct

import torch

from . import _lib as L


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_grad_norm=0.0):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, max_grad_norm=max_grad_norm)
        super().__init__(params, defaults)
        self._tables = {}
        self.last_grad_norm = None

    def _table(self, gi, group, params):
        key = (gi, tuple((p.data_ptr(), p.grad.data_ptr()) for p in params))
        tab = self._tables.get(gi)
        if tab is not None and tab["key"] == key:
            return tab
        dev = params[0].device
        chunk = L.lib().ofd_adam_chunk()
        rows, tt, tc = [], [], []
        for i, p in enumerate(params):
            st = self.state[p]
            rows.append(struct.pack("<QQQQQ", p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel()))
            for c in range((p.numel() + chunk - 1) // chunk):
                tt.append(i)
                tc.append(c)
        tab = dict(key=key,
                   table=torch.frombuffer(bytearray(b"".join(rows)), dtype=torch.uint8).to(dev),
                   tt=torch.tensor(tt, dtype=torch.int32, device=dev), tc=torch.tensor(tc, dtype=torch.int32, device=dev),
                   acc=torch.zeros(len(tt) + 1, dtype=torch.float64, device=dev), coef=torch.ones(1, dtype=torch.float32, device=dev),
                   norm=torch.zeros(1, dtype=torch.float32, device=dev), n=len(tt))
        self._tables[gi] = tab
        return tab

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            L.require_gpu(*params)
            for p in params:
                if p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
                    raise L.OfdError("FusedAdam needs contiguous fp32 parameters and gradients")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
            steps = {self.state[p]["step"] for p in params}
            if len(steps) != 1:
                raise L.OfdError("FusedAdam: parameters of one group must share the step count")
            step = steps.pop() + 1
            tab = self._table(gi, group, params)
            b1, b2 = group["betas"]
            L.check(L.lib().ofd_adam_step(L.ptr(tab["table"]), L.ptr(tab["tt"]), L.ptr(tab["tc"]), tab["n"], L.ptr(tab["acc"]),
                                          L.ptr(tab["coef"]), L.ptr(tab["norm"]), float(group["max_grad_norm"]), float(group["lr"]),
                                          float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]), step, L.stream()))
            for p in params:
                self.state[p]["step"] = step
            torch.autograd.graph.increment_version(params)      # written through raw pointers: tell torch (and Unet's cache)
            self.last_grad_norm = tab["norm"]
        return loss
