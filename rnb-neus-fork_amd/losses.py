"""The loss of the reference's `train_rnb` step (exp_runner.py:241-258) as one device launch.

    loss = L1(color_fine - true_rgb | mask) / (mask_sum * n_lights) + igr_weight * gradient_error
           + mask_weight * BCE(clip(weight_sum, 1e-3, 1 - 1e-3), mask)

`rnb_loss(render_out, true_rgb, mask)` takes the dict returned by `NeuSRenderer.render_rnb*` and returns
`(loss, parts)` like the inline code of the reference; `loss.backward()` feeds the renderer's backward.  The
forward launch also produces the three input gradients, so the backward is a single scaling.  The reference's
chain of PyTorch ops on the same tensors gives the same numbers (tests/test_gpu_parity.py) — using this
function instead is optional.  Device tensors only: there is no CPU path.

Data parallel (`group=`): rays are sharded over the ranks, and the loss couples them through batch-global scalars
only: `mask_sum` (exp_runner.py:194), the BCE mean over B (exp_runner.py:251) and the eikonal ratio's
numerator and count (models/renderer.py:538-540).  With a group ONE 4-float all-reduce ahead of the launch carries
all of them — the eikonal partial sums arrive on the `ExactDPToken` the renderer attached to `gradient_error`
(`NeuSRenderer.set_data_parallel(exact=True)`) — and every rank normalises its own numerators by the GLOBAL
denominators (the ray count is the all-reduced one, so unequal shards stay exact), so the SUM over ranks of the
per-rank gradients (what the renderer's backward all-reduces in its exact mode) is the gradient of the
single-process loss on the whole batch; the returned loss value and parts are all-reduced too (second, optional
collective), i.e. every rank reports the loss of the whole batch.  The pairing is checked: `group=` on a render that
was not made in exact mode raises, and so does a missing `group=` on one that was."""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import native
from .parallel import all_reduce_sum, local_mask_count


class _RnbLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, color_fine, weight_sum, gradient_error, true_rgb, mask, igr_weight, mask_weight, group,
                report_global=True, token=None):
        for name, t in (("color_fine", color_fine), ("weight_sum", weight_sum), ("true_rgb", true_rgb),
                        ("mask", mask)):
            if not t.is_cuda:
                raise RuntimeError(f"rnb_loss: `{name}` must live on the GPU (there is no CPU path)")
        native.same_device(color_fine, weight_sum, gradient_error, true_rgb, mask)
        lib = native.load()
        color = color_fine.detach().to(torch.float32).contiguous()
        rgb = true_rgb.to(torch.float32).contiguous()
        if color.dim() == 2:                      # single-light layout [B, 3]
            color, rgb = color[None], rgb[None]
        if rgb.shape != color.shape:
            raise ValueError(f"rnb_loss: color_fine {tuple(color.shape)} vs true_rgb {tuple(rgb.shape)}")
        L, B, Cd = color.shape
        ws = weight_sum.detach().to(torch.float32).contiguous().reshape(-1)
        mk = mask.to(torch.float32).contiguous().reshape(-1)
        if ws.numel() != B or mk.numel() != B:
            raise ValueError("rnb_loss: weight_sum and mask must hold one value per ray")
        ge = gradient_error.detach().to(torch.float32).reshape(1).contiguous()
        loss = torch.empty((), dtype=torch.float32, device=color.device)
        parts = torch.empty(3, dtype=torch.float32, device=color.device)
        d_color = torch.empty_like(color)
        d_ws = torch.empty_like(ws)
        d_ge = torch.empty_like(ge)
        if group is None and token is not None:
            raise RuntimeError("rnb_loss: this render was made in exact data-parallel mode "
                               "(NeuSRenderer.set_data_parallel(exact=True)): pass group=<the data-parallel group>, or "
                               "switch the renderer to exact=False for per-rank normalisers")
        if group is not None and token is None:
            raise RuntimeError("rnb_loss(group=...) normalises by the whole data-parallel batch and needs a render made "
                               "under grad by a renderer in exact data-parallel mode "
                               "(NeuSRenderer.set_data_parallel(group, exact=True)); without it the gradient all-reduce "
                               "would be a mean and every term scaled by 1/world")
        if group is not None:
            if token.group is not group and token.group != group:
                raise RuntimeError("rnb_loss: `group` differs from the renderer's data-parallel group")
            world = dist.get_world_size(group)
            # ONE collective for every batch-global normaliser: [eikonal numerator, eikonal count, sum(mask > 0.5), rays]
            glob = torch.cat([token.gerr_partial, local_mask_count(mk, mask_weight > 0.0)])
            all_reduce_sum(glob, group)
            token.gerr_den_global.copy_(glob[1:2] + 1e-5)       # what the renderer's backward divides by
            ge_global = (glob[0:1] / token.gerr_den_global).contiguous()
            batch_global = glob[2:4].contiguous()
            token.paired = True
            with native.on_device(color) as stream:
                native.check(lib.rnb_loss_rnb_shard(
                    native.ptr(color), native.ptr(rgb), native.ptr(mk), native.ptr(ws), native.ptr(ge_global), L, B, Cd,
                    float(igr_weight), float(mask_weight), native.ptr(batch_global), 1.0 / world, native.ptr(loss),
                    native.ptr(parts), native.ptr(d_color), native.ptr(d_ws), native.ptr(d_ge), stream))
            # the loss VALUE of the whole batch on every rank (reporting only; gradients are already global-normalised).
            # report_global=False keeps this rank's additive share instead and saves the collective.
            if report_global:
                rep = torch.cat([loss.reshape(1), parts])
                all_reduce_sum(rep, group)
                loss, parts = rep[0].clone(), rep[1:].clone()
        else:
            with native.on_device(color) as stream:
                native.check(lib.rnb_loss_rnb(native.ptr(color), native.ptr(rgb), native.ptr(mk), native.ptr(ws),
                                              native.ptr(ge), L, B, Cd, float(igr_weight), float(mask_weight),
                                              native.ptr(loss), native.ptr(parts), native.ptr(d_color),
                                              native.ptr(d_ws), native.ptr(d_ge), stream))
        ctx.grads = (d_color.view(color_fine.shape), d_ws.view(weight_sum.shape), d_ge.view(gradient_error.shape))
        ctx.mark_non_differentiable(parts)
        return loss, parts

    @staticmethod
    def backward(ctx, g_loss, _g_parts):
        grads = ctx.grads
        if grads is None:
            raise RuntimeError("rnb_loss: backward called twice (the input gradients were released after the first "
                               "backward; re-run the forward)")
        ctx.grads = None
        out = torch._foreach_mul(list(grads), g_loss)      # one multi-tensor launch
        return out[0], out[1], out[2], None, None, None, None, None, None, None


def rnb_loss(render_out, true_rgb, mask, igr_weight=0.1, mask_weight=0.1, group=None, report_global=True):
    """exp_runner.py:229-258 (`train_rnb`).  Returns `(loss, {"color_loss", "eikonal_loss", "mask_loss"})`.
    `group`: the data-parallel process group whose ranks share one global batch (see the module docstring).
    `report_global` (with a group): True returns the loss VALUE of the whole batch on every rank (one more 4-float
    all-reduce per step); False returns this rank's additive share of it — the gradients are identical either way.
    `render_out["gradient_error"]` stays the shard-local value; `parts["eikonal_loss"]` is (this rank's share of) the
    global one."""
    loss, parts = _RnbLoss.apply(render_out["color_fine"], render_out["weight_sum"], render_out["gradient_error"],
                                 true_rgb, mask, igr_weight, mask_weight, group, report_global,
                                 getattr(render_out["gradient_error"], "rnb_dp_token", None))
    return loss, {"color_loss": parts[0], "eikonal_loss": parts[1], "mask_loss": parts[2]}
