"""Adam over ONE flat parameter buffer (exp_runner.py:115 `torch.optim.Adam(params_to_train, lr=...)`, same update
rule, one launch per step instead of a multi-tensor kernel chain).

`FlatAdam(params)` re-homes the given parameters into a single contiguous fp32 buffer (each `p.data` becomes a
view of it, the modules keep working unchanged) and keeps `exp_avg` / `exp_avg_sq` flat as well.  The renderer's
backward already produces all parameter gradients as views of one flat buffer in `NeuSRenderer._leaves()` order;
when the parameters are given in that order (`list(sdf.parameters()) + list(deviation.parameters()) +
list(color.parameters())`, the order of exp_runner.py:105-108 without the unused NeRF) `step()` consumes that
buffer in place; otherwise the gradients are gathered first.  `param_groups[0]["lr"]` may be changed between
steps like with any torch optimizer (exp_runner.py:327-337).  Device parameters only."""
from __future__ import annotations

import ctypes as C

import torch

from . import native


class FlatAdam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.params = [p for p in params]
        if not self.params:
            raise ValueError("FlatAdam: empty parameter list")
        dev = self.params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatAdam: parameters must live on the GPU (there is no CPU path)")
        for p in self.params:
            if p.device != dev or p.dtype != torch.float32:
                raise ValueError("FlatAdam: all parameters must be fp32 on one device")
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += p.numel()
        self.numel = off
        self.flat = torch.empty(off, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                view = self.flat[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.step_count = 0
        self.param_groups = [dict(params=self.params, lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)]
        self._gather = None

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def _flat_grad(self):
        g0 = self.params[0].grad
        if any(p.grad is None for p in self.params):
            raise RuntimeError("FlatAdam.step(): every parameter needs a gradient (build the optimizer from the "
                               "parameters the step actually trains)")
        base, in_place = g0.data_ptr(), g0.dtype == torch.float32
        if in_place:
            for p, o in zip(self.params, self.offsets):
                g = p.grad
                if g.data_ptr() != base + 4 * o or not g.is_contiguous() or g.dtype != torch.float32:
                    in_place = False
                    break
        if in_place:
            st = g0.untyped_storage()
            if g0.storage_offset() * 4 + self.numel * 4 <= st.nbytes():
                return torch.empty(0, dtype=torch.float32, device=g0.device).set_(st, g0.storage_offset(),
                                                                                  (self.numel,), (1,))
        if self._gather is None:
            self._gather = torch.empty_like(self.flat)
        torch._foreach_copy_([self._gather[o:o + p.numel()] for p, o in zip(self.params, self.offsets)],
                             [p.grad.reshape(-1).to(torch.float32) for p in self.params])
        return self._gather

    def _check_homes(self):
        base = self.flat.data_ptr()
        for p, o in zip(self.params, self.offsets):
            if p.data_ptr() != base + 4 * o:
                raise RuntimeError("FlatAdam: a parameter no longer lives in the flat buffer (was the module moved "
                                   "or its .data replaced after the optimizer was built?); rebuild the optimizer")

    @torch.no_grad()
    def step(self):
        self._check_homes()
        g = self._flat_grad()
        grp = self.param_groups[0]
        self.step_count += 1
        native.check(native.load().rnb_adam_step(
            native.ptr(self.flat), native.ptr(g), native.ptr(self.exp_avg), native.ptr(self.exp_avg_sq), self.numel,
            float(grp["lr"]), float(grp["betas"][0]), float(grp["betas"][1]), float(grp["eps"]),
            float(grp["weight_decay"]), self.step_count, C.c_void_p(torch.cuda.current_stream().cuda_stream)))

    # torch.optim.Adam-shaped state (per-parameter views), so checkpoints interchange with exp_runner.py:373-386
    def state_dict(self):
        state = {}
        if self.step_count > 0:
            for i, (p, o) in enumerate(zip(self.params, self.offsets)):
                n = p.numel()
                state[i] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": self.exp_avg[o:o + n].view(p.shape).clone(),
                            "exp_avg_sq": self.exp_avg_sq[o:o + n].view(p.shape).clone()}
        g = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        g["params"] = list(range(len(self.params)))
        return {"state": state, "param_groups": [g]}

    def load_state_dict(self, sd):
        g = sd["param_groups"][0]
        for k in ("lr", "betas", "eps", "weight_decay"):
            if k in g:
                self.param_groups[0][k] = tuple(g[k]) if k == "betas" else g[k]
        steps = set()
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            st = sd["state"].get(i)
            if st is None:
                continue
            n = p.numel()
            self.exp_avg[o:o + n].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError("FlatAdam.load_state_dict: parameters with different step counts")
        self.step_count = steps.pop() if steps else 0
