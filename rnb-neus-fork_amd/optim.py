"""Adam over ONE flat parameter buffer (exp_runner.py:115 `torch.optim.Adam(params_to_train, lr=...)`, same update
rule, one launch per step instead of a multi-tensor kernel chain).

`FlatAdam(params)` takes the SAME list the reference hands to `torch.optim.Adam` — `nerf + sdf + variance + color`
(exp_runner.py:105-112) — or any sub-list.  Like torch's Adam it skips parameters that have no gradient: with
`n_outside = 0` the NeRF parameters never receive one, so they are carried by position only.  The
parameters that do train are re-homed into a single contiguous fp32 buffer (each `p.data` becomes a view of it, the
modules keep working unchanged) with flat `exp_avg` / `exp_avg_sq`; this happens lazily at the first `step()` (or
`load_state_dict`), when the trained set is known.  The renderer's backward already produces all parameter
gradients as views of one flat buffer in `NeuSRenderer._leaves()` order; when the trained parameters come in that
order `step()` consumes that buffer in place, otherwise the gradients are gathered first.

`state_dict()` / `load_state_dict()` use torch.optim.Adam's layout with the reference's positional indices: state
key i is the i-th entry of the list the optimizer was built from, `param_groups[0]["params"] = [0..n)`, and the group
carries every hyper-parameter key torch's Adam emits — so the `optimizer` entry of a reference checkpoint
(exp_runner.py:355-386) loads here and a checkpoint written here loads into `torch.optim.Adam` over the same list.
`param_groups[0]["lr"]` may be changed between steps like with any torch optimizer (exp_runner.py:327-337).
Device parameters only."""
from __future__ import annotations

import torch

from . import native


def _torch_adam_group_defaults():
    """Hyper-parameter keys torch.optim.Adam puts into a param group (version dependent), with their defaults."""
    return dict(torch.optim.Adam([torch.zeros(1, requires_grad=True)]).defaults)


class FlatAdam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, frozen=()):
        self.params = [p for p in params]
        if not self.params:
            raise ValueError("FlatAdam: empty parameter list")
        if len({id(p) for p in self.params}) != len(self.params):
            raise ValueError("FlatAdam: a parameter appears twice")
        frozen_ids = {id(p) for p in frozen}
        self._frozen_hint = [i for i, p in enumerate(self.params) if id(p) in frozen_ids]
        group = _torch_adam_group_defaults()
        group.update(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)
        group["params"] = self.params
        self.param_groups = [group]
        self.step_count = 0
        self.active = None          # indices (into self.params) of the parameters that train; set by _build
        self.offsets = None
        self.numel = 0
        self.flat = self.exp_avg = self.exp_avg_sq = None
        self._gather = None

    # ------------------------------------------------------------------ lazy construction of the flat buffers
    def _build(self, active):
        active = sorted(int(i) for i in active)
        if not active:
            raise RuntimeError("FlatAdam: no parameter has a gradient")
        ps = [self.params[i] for i in active]
        dev = ps[0].device      # (host parameters can hold / exchange state; only step() needs the GPU)
        for p in ps:
            if p.device != dev or p.dtype != torch.float32:
                raise ValueError("FlatAdam: all trained parameters must be fp32 on one device")
        self.active = active
        self.offsets, off = {}, 0
        for i, p in zip(active, ps):
            self.offsets[i] = off
            off += p.numel()
        self.numel = off
        self.flat = torch.empty(off, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for i, p in zip(active, ps):
                o = self.offsets[i]
                view = self.flat[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)

    def _trained(self):
        return [(i, self.params[i], self.offsets[i]) for i in self.active]

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def _flat_grad(self):
        trained = self._trained()
        g0 = trained[0][1].grad
        base, in_place = g0.data_ptr(), g0.dtype == torch.float32
        if in_place:
            for _, p, o in trained:
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
        torch._foreach_copy_([self._gather[o:o + p.numel()] for _, p, o in trained],
                             [p.grad.reshape(-1).to(torch.float32) for _, p, _ in trained])
        return self._gather

    def _check_homes(self):
        base = self.flat.data_ptr()
        for _, p, o in self._trained():
            if p.data_ptr() != base + 4 * o:
                raise RuntimeError("FlatAdam: a parameter no longer lives in the flat buffer (was the module moved "
                                   "or its .data replaced after the optimizer was built?); rebuild the optimizer")

    @torch.no_grad()
    def step(self):
        with_grad = [i for i, p in enumerate(self.params) if p.grad is not None]
        if self.active is None:
            bad = [i for i in with_grad if i in self._frozen_hint]
            if bad:
                raise RuntimeError(f"FlatAdam: parameters {bad} were declared frozen but received gradients")
            self._build(with_grad)
        elif with_grad != self.active:
            missing = sorted(set(self.active) - set(with_grad))
            extra = sorted(set(with_grad) - set(self.active))
            raise RuntimeError("FlatAdam.step(): the set of parameters with gradients changed since the flat buffers "
                               f"were built (now missing: {missing}, new: {extra}); build one optimizer per trained set "
                               "(e.g. with / without the albedo network, exp_runner.py:105-112)")
        if not self.flat.is_cuda:
            raise RuntimeError("FlatAdam.step(): parameters must live on the GPU (there is no CPU path)")
        self._check_homes()
        g = self._flat_grad()
        grp = self.param_groups[0]
        self.step_count += 1
        with native.on_device(self.flat) as stream:
            native.check(native.load().rnb_adam_step(
                native.ptr(self.flat), native.ptr(g), native.ptr(self.exp_avg), native.ptr(self.exp_avg_sq),
                self.numel, float(grp["lr"]), float(grp["betas"][0]), float(grp["betas"][1]), float(grp["eps"]),
                float(grp["weight_decay"]), self.step_count, stream))

    # ------------------------------------------------------------------ torch.optim.Adam-shaped state
    def state_dict(self):
        """Same structure as torch.optim.Adam(self.params).state_dict(): per-parameter entries (only for the
        parameters that have been stepped) keyed by their position in the constructor's list."""
        state = {}
        if self.step_count > 0:
            for i, p, o in self._trained():
                n = p.numel()
                state[i] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": self.exp_avg[o:o + n].view(p.shape).clone(),
                            "exp_avg_sq": self.exp_avg_sq[o:o + n].view(p.shape).clone()}
        g = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        g["params"] = list(range(len(self.params)))
        return {"state": state, "param_groups": [g]}

    def load_state_dict(self, sd):
        """Accepts what torch.optim.Adam.state_dict() over the same parameter list produces (the `optimizer` entry
        of a reference checkpoint) as well as FlatAdam's own."""
        groups = sd["param_groups"]
        if len(groups) != 1:
            raise ValueError(f"FlatAdam.load_state_dict: expected one param group, got {len(groups)}")
        g = groups[0]
        n_saved = len(g["params"])
        if n_saved != len(self.params):
            raise ValueError(f"FlatAdam.load_state_dict: the saved optimizer covers {n_saved} parameters, this one "
                             f"{len(self.params)} — build FlatAdam from the same list the checkpoint's optimizer was "
                             "built from (the reference: nerf + sdf + variance + color, exp_runner.py:105-112)")
        if g.get("amsgrad") or g.get("maximize"):
            raise ValueError("FlatAdam.load_state_dict: amsgrad / maximize are not supported")
        # saved param ids -> positions (torch numbers them 0..n-1 in list order)
        pos = {pid: k for k, pid in enumerate(g["params"])}
        entries = {}
        for key, st in sd["state"].items():
            if key not in pos:
                raise KeyError(f"FlatAdam.load_state_dict: state key {key!r} is not in param_groups[0]['params']")
            i = pos[key]
            p = self.params[i]
            for name in ("exp_avg", "exp_avg_sq"):
                if tuple(st[name].shape) != tuple(p.shape):
                    raise ValueError(f"FlatAdam.load_state_dict: state[{key}].{name} has shape "
                                     f"{tuple(st[name].shape)} but parameter {i} has {tuple(p.shape)}")
            entries[i] = st
        for k in ("lr", "betas", "eps", "weight_decay"):
            if k in g:
                self.param_groups[0][k] = tuple(g[k]) if k == "betas" else g[k]
        if not entries:
            self.step_count = 0
            if self.active is not None:
                self.exp_avg.zero_()
                self.exp_avg_sq.zero_()
            return
        if self.active is None:
            self._build(entries.keys())
        elif sorted(entries.keys()) != self.active:
            raise ValueError("FlatAdam.load_state_dict: the saved state covers parameters "
                             f"{sorted(entries.keys())} but this optimizer trains {self.active}")
        steps = set()
        with torch.no_grad():
            for i, p, o in self._trained():
                st = entries[i]
                n = p.numel()
                self.exp_avg[o:o + n].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
                steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError("FlatAdam.load_state_dict: parameters with different step counts")
        self.step_count = steps.pop()
