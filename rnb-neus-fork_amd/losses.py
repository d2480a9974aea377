"""The loss of the reference's `train_rnb` step (exp_runner.py:241-258) as one device launch.

    loss = L1(color_fine - true_rgb | mask) / (mask_sum * n_lights) + igr_weight * gradient_error
           + mask_weight * BCE(clip(weight_sum, 1e-3, 1 - 1e-3), mask)

`rnb_loss(render_out, true_rgb, mask)` takes the dict returned by `NeuSRenderer.render_rnb*` and returns
`(loss, parts)` like the inline code of the reference; `loss.backward()` feeds the renderer's backward.  The
forward launch also produces the three input gradients, so the backward is a single scaling.  The reference's
chain of PyTorch ops on the same tensors gives the same numbers (tests/test_gpu_parity.py) — using this
function instead is optional.  Device tensors only: there is no CPU path.

Data parallel (`group=`): rays are sharded over the ranks, and the loss couples them through three batch-global
scalars only: `mask_sum` (exp_runner.py:194), the BCE mean over B (exp_runner.py:251) and the eikonal
normaliser (models/renderer.py:540; the renderer handles that one, `NeuSRenderer.set_data_parallel`).  With a
group the mask count is all-reduced before the launch and every rank normalises its own numerators by the GLOBAL
denominators, so the SUM over ranks of the per-rank gradients (what the renderer's backward all-reduces in its
exact mode) is the gradient of the single-process loss on the whole batch; the returned loss value and parts are
all-reduced too, i.e. every rank reports the loss of the whole batch."""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import native
from .parallel import global_mask_count


class _RnbLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, color_fine, weight_sum, gradient_error, true_rgb, mask, igr_weight, mask_weight, group,
                report_global=True):
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
        world = dist.get_world_size(group) if group is not None else 1
        if world > 1:
            # [sum(mask > 0.5), B] of the whole batch: one 2-float all-reduce ahead of the launch
            cnt = global_mask_count(mk, mask_weight > 0.0, group)
            B_global = B * world      # shards are equal by construction (parallel.shard_range)
            with native.on_device(color) as stream:
                native.check(lib.rnb_loss_rnb_shard(
                    native.ptr(color), native.ptr(rgb), native.ptr(mk), native.ptr(ws), native.ptr(ge), L, B, Cd,
                    float(igr_weight), float(mask_weight), native.ptr(cnt), B_global, 1.0 / world, native.ptr(loss),
                    native.ptr(parts), native.ptr(d_color), native.ptr(d_ws), native.ptr(d_ge), stream))
            # the loss VALUE of the whole batch on every rank (reporting only; gradients are already global-normalised).
            # report_global=False keeps this rank's additive share instead and saves the collective.
            if report_global:
                rep = torch.cat([loss.reshape(1), parts])
                dist.all_reduce(rep, op=dist.ReduceOp.SUM, group=group)
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
        return out[0], out[1], out[2], None, None, None, None, None, None


def rnb_loss(render_out, true_rgb, mask, igr_weight=0.1, mask_weight=0.1, group=None, report_global=True):
    """exp_runner.py:229-258 (`train_rnb`).  Returns `(loss, {"color_loss", "eikonal_loss", "mask_loss"})`.
    `group`: the data-parallel process group whose ranks share one global batch (see the module docstring).
    `report_global` (with a group): True returns the loss VALUE of the whole batch on every rank (one more 4-float
    all-reduce per step); False returns this rank's additive share of it — the gradients are identical either way."""
    loss, parts = _RnbLoss.apply(render_out["color_fine"], render_out["weight_sum"], render_out["gradient_error"],
                                 true_rgb, mask, igr_weight, mask_weight, group, report_global)
    return loss, {"color_loss": parts[0], "eikonal_loss": parts[1], "mask_loss": parts[2]}
