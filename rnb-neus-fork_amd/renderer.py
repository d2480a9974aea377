"""Drop-in NeuSRenderer (models/renderer.py:72-1224) backed by librnbneus_hip.so.

Same constructor, same `render`, `render_rnb`, `render_rnb_warmup`, `extract_geometry` signatures and
the same dict of tensors (models/renderer.py:638-648, :920-930, :1023-1033).  The returned tensors are
attached to autograd through one torch.autograd.Function whose backward runs the native explicit
backward (rnb_render_bwd + rnb_weightnorm_bwd) and returns `.grad`s for every trainable leaf, so the
reference's `loss.backward(); optimizer.step()` (exp_runner.py:259-263) works unchanged.

Differences that are deliberate and documented in DESIGN.md:
  * n_outside must be 0 (every shipped config); the NeRF background path is not implemented.
  * the [B,1] perturbation draw (renderer.py:572) is made here with torch.rand on the rays' device and
    may be supplied explicitly (`t_rand=`) so that tests can feed the oracle the same randomness.
  * with data parallelism enabled (`set_data_parallel`) the backward all-reduces the flat gradient
    buffer over RCCL before returning.  Default `exact=True`: the ranks hold contiguous shards of ONE global
    batch; the render returns its shard's eikonal sums (models/renderer.py:538-540) on a token attached to
    `gradient_error`, `rnb_loss(..., group=)` all-reduces them TOGETHER with `mask_sum` / the ray count in one
    4-float collective, and the gradient all-reduce is a SUM, so a G-rank step equals the single-process step
    on the whole batch.  Forwards are never collective (a rank-0-only validation render cannot desynchronise
    the ranks); the collectives of a step are the loss's and the backward's, and the backward raises if the
    loss it is driven by was not `rnb_loss(..., group=)`.  `exact=False` is the DDP convention (per-rank
    normalisers, mean of the per-rank gradients, any loss).
  * kernel / arithmetic variants (`set_variant`): explicit bits of the model descriptor, no environment
    variables (bf16 sweeps for BASELINE config 5, deterministic reductions, A/B tuning knobs).
"""
from __future__ import annotations

import ctypes as C
import logging

import numpy as np
import torch
import torch.distributed as dist

from . import native, parallel, runtime
from .fields import _mlp_struct, model_desc
from .parallel import allreduce_mean_, allreduce_sum_

_MESH_BACKEND_LOGGED = False


class ExactDPToken:
    """Travels on `out["gradient_error"]` of a render made under grad in exact data-parallel mode.  Carries this
    shard's eikonal (numerator, count) to `rnb_loss(..., group=)`, which all-reduces them with the mask counts, writes
    the global denominator into `gerr_den_global` (read by the native backward) and marks the token paired."""

    def __init__(self, group, gerr_partial, gerr_den_global):
        self.group, self.gerr_partial, self.gerr_den_global = group, gerr_partial, gerr_den_global
        self.paired = False

_OUT_KEYS = ("color_fine", "s_val", "cdf_fine", "weight_sum", "weight_max", "gradients", "weights",
             "gradient_error", "inside_sphere")


class _FinePass(torch.autograd.Function):
    """forward: rnb_weightnorm_fwd (done by the caller) + rnb_render_fwd; backward: rnb_render_bwd +
    rnb_weightnorm_bwd (+ optional RCCL all-reduce of the flat gradient buffer)."""

    @staticmethod
    def forward(ctx, renderer, call, *leaves):
        lib = native.load()
        desc = renderer.desc
        dev = call["rays_o"].device
        B, S = call["z_vals"].shape
        flags = call["flags"]
        mvps = bool(flags & native.MODE_MVPS)
        L = call["lights"].shape[0] if mvps else 1
        Cd = desc.col_d_out
        f32 = dict(dtype=torch.float32, device=dev)
        out = {
            "color_fine": torch.empty((L, B, Cd) if mvps else (B, 3), **f32),
            "weights": torch.empty(B, S, **f32),
            "cdf_fine": torch.empty(B, S, **f32),
            "gradients": torch.empty(B, S, 3, **f32),
            "inside_sphere": torch.empty(B, S, **f32),
            "weight_sum": torch.empty(B, 1, **f32),
            "weight_max": torch.empty(B, 1, **f32),
            "s_val": torch.empty(B, 1, **f32),
            "gradient_error": torch.empty((), **f32),
        }
        extras = {}
        if call.get("want_extras"):
            extras["sdf"] = torch.empty(B * S, 1, **f32)
            if not (mvps and (flags & native.FLAG_NO_ALBEDO)):
                extras["sampled_albedo"] = torch.empty(B, S, Cd, **f32)
        nbytes = C.c_int64()
        native.check(lib.rnb_render_workspace_bytes(C.byref(desc), B, S, flags, C.byref(nbytes)))
        ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
        args = native.RenderArgs()
        args.B, args.S, args.n_lights, args.flags = B, S, L, flags
        args.cos_anneal_ratio = float(call["cos_anneal_ratio"])
        keep = dict(rays_o=call["rays_o"], rays_d=call["rays_d"], z_vals=call["z_vals"], lights=call["lights"],
                    bg=call["background_rgb"], variance=renderer.deviation_network.variance.detach().reshape(1))
        args.rays_o, args.rays_d = keep["rays_o"].data_ptr(), keep["rays_d"].data_ptr()
        args.z_vals = keep["z_vals"].data_ptr()
        args.lights_dir = keep["lights"].data_ptr() if keep["lights"] is not None else None
        args.background_rgb = keep["bg"].data_ptr() if keep["bg"] is not None else None
        args.variance = keep["variance"].data_ptr()
        for k in ("color_fine", "weights", "cdf_fine", "gradients", "inside_sphere", "weight_sum", "weight_max",
                  "s_val", "gradient_error"):
            setattr(args, k, out[k].data_ptr())
        args.sdf = extras["sdf"].data_ptr() if "sdf" in extras else None
        args.sampled_albedo = extras["sampled_albedo"].data_ptr() if "sampled_albedo" in extras else None
        # exact data parallel, training render: keep this shard's eikonal (numerator, count); the loss all-reduces them
        # (no collective here: forwards stay local, forward-only renders return the shard-local gradient_error)
        exact_dp = renderer.dp_group is not None and renderer.dp_exact and not (flags & native.FLAG_FORWARD_ONLY)
        ctx.dp_token = None
        if exact_dp:
            keep["gerr_partial"] = torch.empty(2, **f32)
            keep["gerr_den_global"] = torch.empty(1, **f32)
            args.gerr_partial = keep["gerr_partial"].data_ptr()
            args.gerr_den_global = keep["gerr_den_global"].data_ptr()
            ctx.dp_token = ExactDPToken(renderer.dp_group, keep["gerr_partial"], keep["gerr_den_global"])
        with native.on_device(dev) as stream:
            native.check(lib.rnb_render_fwd(C.byref(desc), native.ptr(call["packed"]), C.byref(args), native.ptr(ws),
                                            ws.numel(), stream))
        if renderer.track_range and (flags & native.FLAG_FORWARD_ONLY):
            renderer._collect_range(desc, call["packed"], ws, B, S, flags)
        ctx.renderer, ctx.call, ctx.args, ctx.keep, ctx.ws, ctx.out = renderer, call, args, keep, ws, out
        # the descriptor AS OF this forward: the backward must carve the workspace with the layout the forward wrote,
        # whatever set_variant() did in between
        ctx.desc = native.ModelDesc.from_buffer_copy(desc)
        ctx.dp = (renderer.dp_group, renderer.dp_exact)
        ctx.n_leaves = len(leaves)
        ctx.mark_non_differentiable(out["inside_sphere"])
        # outputs the loss does not touch arrive as None in backward (no zero tensors are materialised;
        # the native backward treats a NULL cotangent as zero)
        ctx.set_materialize_grads(False)
        renderer.last_extras = extras
        renderer._last_dp_token = ctx.dp_token
        return tuple(out[k] for k in _OUT_KEYS)

    @staticmethod
    def backward(ctx, *gouts):
        lib = native.load()
        renderer, call, desc = ctx.renderer, ctx.call, ctx.desc
        dp_group, dp_exact = ctx.dp
        if ctx.ws is None:
            raise RuntimeError("NeuSRenderer: backward called a second time on the same render (the saved per-point "
                               "state is released after the first backward, retain_graph is not supported); re-run "
                               "the forward")
        if ctx.dp_token is not None and not ctx.dp_token.paired:
            raise RuntimeError("NeuSRenderer: exact data-parallel mode (set_data_parallel(exact=True)) needs the loss to "
                               "be rnb_loss(..., group=<the data-parallel group>): it all-reduces the batch-global "
                               "normalisers this backward divides by.  Use set_data_parallel(exact=False) with any "
                               "other loss (DDP mean of per-rank gradients)")
        dev = ctx.ws.device
        g = dict(zip(_OUT_KEYS, gouts))
        keepalive = []

        def gp(name):
            t = g.get(name)
            if t is None:
                return None
            t = t.to(torch.float32).contiguous()
            keepalive.append(t)
            return t.data_ptr()

        rg = native.RenderGrads()
        for k in ("color_fine", "weights", "cdf_fine", "gradients", "weight_sum", "weight_max", "s_val",
                  "gradient_error"):
            setattr(rg, k, gp(k))
        packed_grad = torch.empty_like(call["packed"])
        flat, views, order = renderer._alloc_flat_grads(call["train_color"], dev)
        dvar = views[id(renderer.deviation_network.variance)]
        with native.on_device(dev) as stream:
            native.check(lib.rnb_render_bwd(C.byref(desc), native.ptr(call["packed"]), C.byref(ctx.args), C.byref(rg),
                                            native.ptr(packed_grad), C.c_void_p(dvar.data_ptr()), native.ptr(ctx.ws),
                                            ctx.ws.numel(), stream))
        if renderer.track_range:
            renderer._collect_range(desc, call["packed"], ctx.ws, ctx.args.B, ctx.args.S, ctx.args.flags)
        sdf_net, col_net = renderer.sdf_network, renderer.color_network
        sp = _mlp_struct(sdf_net.lins(), sdf_net.weight_norm)
        sg = _mlp_struct(sdf_net.lins(), sdf_net.weight_norm, grads=views)
        use_col = call["train_color"]
        cp = _mlp_struct(col_net.lins(), col_net.weight_norm) if use_col else None
        cg = _mlp_struct(col_net.lins(), col_net.weight_norm, grads=views) if use_col else None
        with native.on_device(dev) as stream:
            native.check(lib.rnb_weightnorm_bwd(C.byref(desc), C.byref(sp), C.byref(cp) if cp is not None else None,
                                                native.ptr(packed_grad), C.byref(sg),
                                                C.byref(cg) if cg is not None else None, stream))
        if dp_group is not None:
            # the one exchange step of the path: the flat gradient buffer over xGMI (RCCL).  exact: the per-rank
            # losses are additive shares of the whole batch's loss -> SUM; otherwise the DDP mean.
            if dp_exact:
                allreduce_sum_(flat, dp_group)
            else:
                allreduce_mean_(flat, dp_group)
        ctx.ws = None
        grads = []
        for leaf in call["leaves"]:
            v = views.get(id(leaf))
            grads.append(v.view_as(leaf) if v is not None else None)
        return (None, None) + tuple(grads)


class NeuSRenderer:
    def __init__(self, nerf, sdf_network, deviation_network, color_network, n_samples, n_importance, n_outside,
                 up_sample_steps, perturb):
        if n_outside != 0:
            raise NotImplementedError("n_outside > 0 (NeRF background) is outside the accelerated path; every "
                                      "shipped config uses n_outside = 0")
        self.nerf = nerf
        self.sdf_network = sdf_network
        self.deviation_network = deviation_network
        self.color_network = color_network
        self.n_samples = n_samples
        self.n_importance = n_importance
        self.n_outside = n_outside
        self.up_sample_steps = up_sample_steps
        self.perturb = perturb
        # the reference reads self.color_depth without ever setting it (renderer.py:226; patched from
        # outside at exp_runner.py:125) — default it from the albedo network
        self.color_depth = color_network.d_out
        self.desc = model_desc(sdf_network, color_network, n_samples, n_importance, up_sample_steps)
        self.dp_group = None
        self.dp_exact = True
        self.last_z_vals = None
        self.last_extras = {}
        self.want_extras = False
        # diagnostics: with track_range the largest operand magnitudes of every render are read back (rnb_render_range: one
        # more pass over the saved state per step — off by default); range_report() returns the last ones
        self.track_range = False
        self._range = None

    # ------------------------------------------------------------------ data parallel
    def set_data_parallel(self, group=None, enabled=True, exact=True):
        """One process per GPU, each rendering its contiguous shard of a global ray batch; the backward
        all-reduces the flat gradient buffer over `group` — BACKWARDS are collective, forwards never are (a
        forward-only / no_grad render on one rank alone is fine and returns the shard-local `gradient_error`).
        exact=True (default): large-batch semantics, see the module docstring; the loss MUST then be
        `rnb_loss(..., group=group)` — the pairing is checked both ways (the backward raises otherwise, and so does
        `rnb_loss(group=)` on a render that was not made in exact mode).  exact=False: DDP mean of per-rank
        gradients, per-rank normalisers, any loss (`rnb_loss` without a group, or the reference's torch ops)."""
        self.dp_group = (group if group is not None else dist.group.WORLD) if enabled else None
        self.dp_exact = bool(exact)

    def set_variant(self, **kw):
        """Kernel / arithmetic variant bits of the model descriptor (include/rnbneus.h RNB_VARIANT_*):
        bf16=, deterministic=, generic=, dw_lds=, bwd_ti=, bwd_nw=, fwd_ti=, fwd_nw=.  Returns self."""
        self.desc.variant = native.variant_bits(**kw)
        return self

    # ------------------------------------------------------------------ helpers
    def _leaves(self, train_color):
        leaves = list(self.sdf_network.leaves()) + [self.deviation_network.variance]
        if train_color:
            leaves += list(self.color_network.leaves())
        return leaves

    def _alloc_flat_grads(self, train_color, dev):
        leaves = self._leaves(train_color)
        total = sum(p.numel() for p in leaves)
        flat = torch.empty(total, dtype=torch.float32, device=dev)
        views, off = {}, 0
        for p in leaves:
            views[id(p)] = flat[off:off + p.numel()]
            off += p.numel()
        return flat, views, leaves

    def _pack(self, use_color):
        dev = self.sdf_network.lin0.bias.device
        return runtime.pack_weights(self.desc, self.sdf_network, self.color_network if use_color else None, dev)

    def sample_z_vals(self, rays_o, rays_d, near, far, packed, perturb, t_rand=None):
        """The no-grad prologue shared by the three wrappers (models/renderer.py:557-608)."""
        lib = native.load()
        B = rays_o.shape[0]
        dev = rays_o.device
        if perturb > 0:
            if t_rand is None:
                t_rand = torch.rand([B, 1], device=dev)
            t_rand = t_rand.to(torch.float32).reshape(B).contiguous()
        else:
            t_rand = None
        S = self.n_samples + self.n_importance
        z = torch.empty(B, S, dtype=torch.float32, device=dev)
        nbytes = C.c_int64()
        native.check(lib.rnb_sample_workspace_bytes(C.byref(self.desc), B, C.byref(nbytes)))
        ws = torch.empty(max(nbytes.value, 256), dtype=torch.uint8, device=dev)
        near = near.to(torch.float32).reshape(B).contiguous()
        far = far.to(torch.float32).reshape(B).contiguous()
        native.same_device(packed, rays_o, rays_d, near, far, t_rand)
        with native.on_device(dev) as stream:
            native.check(lib.rnb_sample_rays(C.byref(self.desc), native.ptr(packed), native.ptr(rays_o),
                                             native.ptr(rays_d), native.ptr(near), native.ptr(far), native.ptr(t_rand),
                                             B, native.ptr(z), native.ptr(ws), ws.numel(), stream))
        return z

    def _run(self, rays_o, rays_d, near, far, lights_dir, perturb_overwrite, background_rgb, cos_anneal_ratio,
             flags, t_rand, z_vals):
        if not rays_o.is_cuda:
            raise RuntimeError("NeuSRenderer: rays must be on the GPU (no CPU path; librnbneus_hip.so only)")
        dev = rays_o.device
        if self.sdf_network.lin0.bias.device != dev:
            raise RuntimeError(f"NeuSRenderer: rays live on {dev} but the networks on "
                               f"{self.sdf_network.lin0.bias.device}")
        B = rays_o.shape[0]
        rays_o = rays_o.detach().to(torch.float32).contiguous()
        rays_d = rays_d.detach().to(torch.float32).contiguous()
        mvps = bool(flags & native.MODE_MVPS)
        no_albedo = bool(flags & native.FLAG_NO_ALBEDO)
        use_color = not (mvps and no_albedo)
        packed = self._pack(use_color)
        perturb = self.perturb if perturb_overwrite < 0 else perturb_overwrite
        if z_vals is None:
            with torch.no_grad():
                z_vals = self.sample_z_vals(rays_o, rays_d, near, far, packed, perturb, t_rand)
        else:
            z_vals = z_vals.detach().to(torch.float32).contiguous()
        self.last_z_vals = z_vals
        lights = None
        if mvps:
            L = lights_dir.shape[0]
            lt = lights_dir.detach().to(torch.float32)
            if lt.numel() == L * 3:
                lights = lt.reshape(L, 3).contiguous()
            else:
                lights = lt.reshape(L, B, 3).contiguous()
                flags |= native.FLAG_LIGHT_PER_RAY
        bg = None
        if background_rgb is not None and not mvps:
            bg = background_rgb.detach().to(torch.float32).reshape(3).contiguous()
        # leaves that receive gradients: as in exp_runner.py:105-112 the albedo net is trained unless no_albedo
        train_color = use_color
        leaves = self._leaves(train_color)
        grad_on = torch.is_grad_enabled() and any(p.requires_grad for p in leaves)
        if not grad_on:
            flags |= native.FLAG_FORWARD_ONLY
        call = dict(rays_o=rays_o, rays_d=rays_d, z_vals=z_vals, lights=lights, background_rgb=bg,
                    cos_anneal_ratio=cos_anneal_ratio, flags=flags, packed=packed, leaves=leaves,
                    train_color=train_color, want_extras=self.want_extras)
        outs = _FinePass.apply(self, call, *leaves)
        out = dict(zip(_OUT_KEYS, outs))
        token = getattr(self, "_last_dp_token", None)
        self._last_dp_token = None
        if token is not None:
            out["gradient_error"].rnb_dp_token = token
        return out

    # ------------------------------------------------------------------ reference API
    def render(self, rays_o, rays_d, near, far, perturb_overwrite=-1, background_rgb=None, cos_anneal_ratio=0.0,
               t_rand=None, z_vals=None):
        """models/renderer.py:556-648."""
        return self._run(rays_o, rays_d, near, far, None, perturb_overwrite, background_rgb, cos_anneal_ratio,
                         native.MODE_CORE, t_rand, z_vals)

    def render_rnb_warmup(self, rays_o, rays_d, near, far, lights_dir, perturb_overwrite=-1, background_rgb=None,
                          cos_anneal_ratio=0.0, no_albedo=False, t_rand=None, z_vals=None):
        """models/renderer.py:828-930 (ReLU on the shading)."""
        flags = native.MODE_MVPS | native.FLAG_RELU_SHADING | (native.FLAG_NO_ALBEDO if no_albedo else 0)
        return self._run(rays_o, rays_d, near, far, lights_dir, perturb_overwrite, background_rgb, cos_anneal_ratio,
                         flags, t_rand, z_vals)

    def render_rnb(self, rays_o, rays_d, near, far, lights_dir, perturb_overwrite=-1, background_rgb=None,
                   cos_anneal_ratio=0.0, no_albedo=False, t_rand=None, z_vals=None):
        """models/renderer.py:932-1033."""
        flags = native.MODE_MVPS | (native.FLAG_NO_ALBEDO if no_albedo else 0)
        return self._run(rays_o, rays_d, near, far, lights_dir, perturb_overwrite, background_rgb, cos_anneal_ratio,
                         flags, t_rand, z_vals)

    def color(self, points, normals, view_dirs, feature_vectors):
        """RenderingNetwork.forward through this renderer's packed weights (view_dirs unused in
        no_view_dir mode, models/fields.py:190-192)."""
        packed = self._pack(True)
        return runtime.color_forward(self.desc, packed, points, normals, feature_vectors)

    def extract_fields(self, bound_min, bound_max, resolution, chunk=64, group=None, to_host=True):
        """SDF grid query of models/renderer.py:10-25 (values negated as at :1224): `resolution`^3 forward-only
        evaluations.  The grid points are generated inside the forward kernel (`rnb_sdf_grid`), the volume is
        assembled on the device and copied to the host once (`to_host=False` returns the device tensor).  With data
        parallelism enabled (`set_data_parallel`) or a `group` given, every rank evaluates one contiguous x-slab and
        the slabs are exchanged with one all-gather (512^3: 512 MB in total).  `chunk` is accepted for signature
        compatibility with the reference's block size N = 64 and has no effect: every point is evaluated once,
        whatever the blocking."""
        dev = self.sdf_network.lin0.bias.device
        packed = self._pack(False)
        lib = native.load()
        res = int(resolution)
        group = group if group is not None else self.dp_group
        rank, world = (dist.get_rank(group), dist.get_world_size(group)) if group is not None else (0, 1)
        per, x0, x1 = parallel.grid_slab(res, rank, world)   # x-planes per rank (the last slabs may be shorter, or empty)
        gd = native.GridDesc()
        for d in range(3):
            gd.bound_min[d] = float(bound_min[d])
            gd.bound_max[d] = float(bound_max[d])
        gd.resolution, gd.x_begin, gd.x_end, gd.out_scale = res, x0, x1, -1.0
        slab = torch.empty(per, res, res, dtype=torch.float32, device=dev)     # padded to `per` planes for the gather
        with torch.no_grad():
            if x1 > x0:
                nbytes = C.c_int64()
                native.check(lib.rnb_sdf_grid_workspace_bytes(C.byref(self.desc), C.byref(gd), C.byref(nbytes)))
                ws = torch.empty(max(nbytes.value, 256), dtype=torch.uint8, device=dev)
                with native.on_device(dev) as stream:
                    native.check(lib.rnb_sdf_grid(C.byref(self.desc), native.ptr(packed), C.byref(gd), native.ptr(slab),
                                                  native.ptr(ws), ws.numel(), stream))
            if x1 - x0 < per:
                slab[x1 - x0:].zero_()
            if world > 1:
                u = parallel.gather_grid_slabs(slab, res, group)
            else:
                u = slab[:res]
        return u.cpu().numpy() if to_host else u

    def _collect_range(self, desc, packed, ws, B, S, flags):
        if self._range is None or self._range.device != ws.device:
            self._range = torch.zeros(8, dtype=torch.float32, device=ws.device)
        with native.on_device(ws.device) as stream:
            native.check(native.load().rnb_render_range(C.byref(desc), native.ptr(packed), native.ptr(ws), ws.numel(), B, S,
                                                        flags, native.ptr(self._range), stream))

    def range_report(self):
        """Largest operand magnitudes of the last render made with `track_range = True` (one device-to-host copy):
        `max_abs_weight`, `max_abs_activation` (SDF network, incl. the encoded input), `max_abs_jacobian_row`,
        `max_abs_albedo_activation`, `max_abs_adjoint` (after a backward; 0 for forward-only renders).  The default arithmetic
        takes every operand scale from the data (include/rnbneus.h, RNB_VARIANT_X2H): none of these has a limit; the
        numbers say how far a model is from the range the fixed scales of ABI 4 assumed (weights 255, activations 1023)."""
        if self._range is None:
            raise RuntimeError("range_report(): set `track_range = True` and render first")
        r = [float(x) for x in self._range.cpu()]
        return {"max_abs_weight": r[0], "max_abs_activation": r[1], "max_abs_jacobian_row": r[2],
                "max_abs_albedo_activation": r[3], "max_abs_adjoint": r[4]}

    def x2h_range_report(self):
        """Host-side (plain torch) maximum of |g v / ||v||| over every layer of both networks:
        `{"max_abs_weight": m, "layer": name, "limit": 255.0, "ok": m < 255}`.  Kept from ABI 4, where a weight beyond 255
        overflowed the fixed fp16 scale of the default arithmetic; since ABI 5 the scale of every matrix is taken from its own
        maximum and `ok` is informative only.  `range_report()` gives the device-side maxima of a whole render."""
        worst, where = 0.0, None
        with torch.no_grad():
            for prefix, net in (("sdf", self.sdf_network), ("color", self.color_network)):
                for i, lin in enumerate(net.lins()):
                    if hasattr(lin, "weight_g"):
                        v = lin.weight_v.detach().double()
                        w = lin.weight_g.detach().double().reshape(-1, 1) * v / v.norm(dim=1, keepdim=True)
                    else:
                        w = lin.weight.detach().double()
                    m = float(w.abs().max())
                    if m > worst or where is None:
                        worst, where = m, f"{prefix}.lin{i}"
        return {"max_abs_weight": worst, "layer": where, "limit": 255.0, "ok": bool(worst < 255.0)}

    def extract_geometry(self, bound_min, bound_max, resolution, threshold=0.0, backend=None):
        """models/renderer.py:1219-1224 / :27-36: SDF grid + marching cubes + rescaling to the bounding box; returns
        numpy `(vertices [V,3] float64, triangles [T,3])` as the reference does.
        `backend`: "native" — the library's own marching cubes on the volume still resident in HBM
        (csrc/mcubes.hip); "mcubes" — PyMCubes on the host copy, exactly the reference's call (raises ImportError when
        PyMCubes is not installed); None (default) — "mcubes" when it is importable (the reference-faithful default),
        else "native".  The choice made for None is logged once per process (logger `rnb_neus_fork_amd`, INFO) and kept in
        `self.last_mesh_backend`, so the same script cannot silently produce different triangulations on two machines.
        The native mesh has the same vertices (one per crossed grid edge, same interpolation); its triangulation of
        ambiguous cells / quad diagonals may differ from PyMCubes' table (parity unpinned: DESIGN)."""
        if backend is None:
            try:
                import mcubes  # noqa: F401
                backend = "mcubes"
            except ImportError:
                backend = "native"
            global _MESH_BACKEND_LOGGED
            if not _MESH_BACKEND_LOGGED:
                _MESH_BACKEND_LOGGED = True
                logging.getLogger("rnb_neus_fork_amd").info(
                    "extract_geometry(backend=None): using %r (%s); pass backend= to fix the choice", backend,
                    "PyMCubes is importable" if backend == "mcubes" else "PyMCubes is not installed")
        self.last_mesh_backend = backend
        if backend not in ("native", "mcubes"):
            raise ValueError(f"extract_geometry: unknown backend {backend!r}")
        b_max = bound_max.detach().cpu().numpy()
        b_min = bound_min.detach().cpu().numpy()
        if backend == "mcubes":
            import mcubes
            u = self.extract_fields(bound_min, bound_max, resolution)
            vertices, triangles = mcubes.marching_cubes(u, threshold)
        else:
            from .mcubes import marching_cubes
            u = self.extract_fields(bound_min, bound_max, resolution, to_host=False)
            v, t = marching_cubes(u, threshold)
            vertices, triangles = v.cpu().numpy(), t.cpu().numpy()
        vertices = vertices / (resolution - 1.0) * (b_max - b_min)[None, :] + b_min[None, :]
        return vertices, triangles
