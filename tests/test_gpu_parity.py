"""GPU parity tests: the HIP path (through the C ABI, via the drop-in classes) against the pinned CPU
oracle and the committed golden vectors.

Tolerances.  Integer sample indices: bit-exact given identical inputs.  Floating point: SURVEY.md 8c states
|d| <= 1e-5 + 1e-4 |ref| for outputs and rel-L2 <= 1e-4 for parameter gradients — but the fixtures show that the
fp32 REFERENCE itself is further than that from the exact result wherever the quantity is ill-conditioned (cdf_fine
4.9e-5 at inv_s = 403; d loss / d lin8.weight_g 1.4e-3).  Every fixture therefore also holds the reference's own code
run in fp64 on the same samples (oracle/gen_golden.py::reference_fp64), and the fine-pass tests bound the HIP path's
distance from fp64 by the fp32 reference's distance from fp64:
    outputs:    max|hip - ref64| <= K_OUT * max|ref32 - ref64| + FLOOR_OUT * max(1, max|ref64|)
    gradients:  relL2(hip, ref64) <= max(1e-4, K_GRAD * relL2(ref32, ref64))          per tensor
i.e. "as accurate as the reference's fp32, up to a stated factor", instead of hand-set absolute bounds."""
import ctypes as C
import zlib

import numpy as np
import json
import math
import os

import pytest
import torch

from oracle import rnb_oracle as O
from tests.golden_util import Golden, case_names

pytestmark = pytest.mark.gpu

K_OUT, FLOOR_OUT = 3.0, 2e-6     # outputs: factor over the fp32 reference's own max error + a few fp32 ulps
K_GRAD = 3.0                     # gradients: factor over the fp32 reference's own relative L2 error
GRAD_CAP = 1e-2                  # ... capped: a gradient the fp32 reference resolves to worse than GRAD_CAP / K_GRAD
                                 # is not a parity target (tests/golden_util.py refuses such a fixture at load time)


def _grad_bound(rel32s):
    return min(GRAD_CAP, max(1e-4, K_GRAD * rel32s))


# SURVEY 8c states |d| <= 1e-5 + 1e-4 |ref| (outputs) and rel-L2 <= 1e-4 per gradient tensor.  The calibrated bounds above
# replace them where the fp32 reference itself is further than that from fp64; how many tensors still meet the ORIGINAL
# bounds is counted, printed, and held to the floor measured on MI355X in round 4 (tests/golden/survey_tol_floor.json:
# a drift towards the calibrated bounds' 3 x would otherwise pass unseen).
def _survey_counts(got_all, ref32_all, grads_mine, grads_ref):
    n_out = ok_out = n_g = ok_g = 0
    missed = []
    for k, ref in ref32_all.items():
        if k == "inside_sphere":
            continue
        d = (got_all[k].double() - ref.double()).abs()
        n_out += 1
        ok = bool((d <= 1e-5 + 1e-4 * ref.double().abs()).all())
        ok_out += int(ok)
        if not ok:
            missed.append(f"{k} (max excess {float((d - 1e-5 - 1e-4 * ref.double().abs()).max()):.1e})")
    for k, ref in grads_ref.items():
        rn = float(ref.double().norm())
        if rn < 1e-10:
            continue
        n_g += 1
        rel = float((grads_mine[k].double() - ref.double()).norm()) / rn
        ok_g += int(rel <= 1e-4)
        if rel > 1e-4:
            missed.append(f"d {k} ({rel:.1e})")
    return ok_out, n_out, ok_g, n_g, missed


def _survey_floor(tag):
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "survey_tol_floor.json")
    if not os.path.exists(path):
        return None
    return json.load(open(path)).get(tag)


_SURVEY_SEEN = {}


def _check_survey(tag, counts):
    ok_out, n_out, ok_g, n_g, missed = counts
    _SURVEY_SEEN[tag] = (ok_out, ok_g)
    print(f"SURVEYTOL {tag}: outputs within 1e-5 + 1e-4|ref| of the fp32 reference: {ok_out}/{n_out}; "
          f"gradient tensors within rel-L2 1e-4: {ok_g}/{n_g}" + (f"; outside: {', '.join(missed)}" if missed else ""))
    # Whether a tensor meets the bound at EVERY element against another fp32 realisation (the reference's own run) is decided
    # by single elements of fp32-vs-fp32 noise (cdf_fine at inv_s = 403 misses by 6e-7 in one build and passes in the next of
    # equal accuracy), so a fixture may lose one output / two gradient tensors against the recorded run; the sum over all
    # fixtures is held tighter (test_survey_tolerance_counts_in_aggregate).
    floor = _survey_floor(tag)
    if floor is not None:
        assert ok_out >= floor["outputs_ok"] - 1, f"{tag}: {ok_out} outputs meet SURVEY 8c's bound, {floor['outputs_ok']} did in round 4"
        assert ok_g >= floor["grads_ok_measured"] - 2, \
            f"{tag}: {ok_g} gradient tensors meet SURVEY 8c's bound, {floor['grads_ok_measured']} did in round 4"


def _assert_has_surface(out, dvariance=None):
    """Non-degeneracy guard: the rendered scene has a surface (otherwise weights, CDFs and colours are ~0 and every
    absolute bound passes for zeros) and the variance gradient is resolved."""
    assert float(out["weight_sum"].mean()) > 0.3, "degenerate scene: rays do not hit a surface"
    assert float(out["weights"].max()) > 1e-2, "degenerate scene: no sample carries weight"
    if dvariance is not None:
        assert float(dvariance.abs().max()) > 1e-6, "degenerate scene: d loss / d variance vanishes"

CASES = case_names()
TINY = [c for c in CASES if c.startswith("tiny")]


@pytest.fixture(scope="module")
def R():
    assert torch.cuda.is_available(), "GPU tests need a device"
    import rnb_neus_fork_amd as pkg
    pkg.native.load()
    return pkg


def _dev():
    return torch.device("cuda:0")


def _build(R, g: Golden):
    p = g.params()
    sdf, dev, col, ren = R.build_from_named_params(g.mc, p, _dev())
    return p, sdf, dev, col, ren


def _fine_points(g: Golden):
    z = g.z_fine
    b = g.batch
    sd = 2.0 / g.mc.render.n_samples
    dists = torch.cat([z[:, 1:] - z[:, :-1], torch.full_like(z[:, :1], sd)], -1)
    mid = z + dists * 0.5
    return (b["rays_o"][:, None, :] + b["rays_d"][:, None, :] * mid[..., None]).reshape(-1, 3)


# ---------------------------------------------------------------------------------------------------
# point-wise network evaluation
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", CASES)
def test_sdf_forward_matches_oracle(R, name):
    g = Golden(name)
    p, sdf, dev, col, ren = _build(R, g)
    pts = _fine_points(g)
    ref = O.sdf_forward(p, g.mc.sdf, pts)
    with torch.no_grad():   # (the direct network calls are forward-only and raise under grad mode)
        out = sdf(pts.to(_dev())).cpu()
    assert out.shape == ref.shape
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=2e-5)
    with torch.no_grad():
        out1 = sdf.sdf(pts.to(_dev())).cpu()
    torch.testing.assert_close(out1, ref[:, :1], rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("name", CASES)
def test_sdf_gradient_matches_oracle(R, name):
    g = Golden(name)
    p, sdf, dev, col, ren = _build(R, g)
    pts = _fine_points(g)
    ref = O.sdf_gradient(p, g.mc.sdf, pts, create_graph=False)
    with torch.no_grad():
        out = sdf.gradient(pts.to(_dev())).cpu()
    assert out.shape == (pts.shape[0], 1, 3)
    torch.testing.assert_close(out[:, 0, :], ref, rtol=2e-4, atol=5e-5)


@pytest.mark.parametrize("name", ["tiny_main_sharp", "full_main_sharp"])
def test_color_forward_matches_oracle(R, name):
    g = Golden(name)
    p, sdf, dev, col, ren = _build(R, g)
    pts = _fine_points(g)
    gen = torch.Generator().manual_seed(3)
    normals = torch.randn(pts.shape[0], 3, generator=gen)
    feats = torch.randn(pts.shape[0], g.mc.color.d_feature, generator=gen) * 0.3
    ref = O.color_forward(p, g.mc.color, pts, normals, normals, feats)
    with torch.no_grad():
        out = col(pts.to(_dev()), normals.to(_dev()), normals.to(_dev()), feats.to(_dev())).cpu()
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=2e-5)
    out2 = ren.color(pts.to(_dev()), normals.to(_dev()), None, feats.to(_dev())).cpu()
    torch.testing.assert_close(out2, ref, rtol=1e-4, atol=2e-5)


# ---------------------------------------------------------------------------------------------------
# hierarchical sampling: integer outputs bit-exact given the reference's own per-step inputs
# ---------------------------------------------------------------------------------------------------
def _up_sample_step(R, g, st):
    lib = R.native.load()
    d = _dev()
    z_in = st["z_in"].to(d).contiguous()
    sdf_in = st["sdf_in"].to(d).contiguous()
    B, n = z_in.shape
    n_new = st["new_z"].shape[1]
    ro = g.batch["rays_o"].to(d).contiguous()
    rd = g.batch["rays_d"].to(d).contiguous()
    new_z = torch.empty(B, n_new, device=d)
    inds = torch.empty(B, n_new, dtype=torch.int32, device=d)
    z_out = torch.empty(B, n + n_new, device=d)
    sidx = torch.empty(B, n + n_new, dtype=torch.int32, device=d)
    R.native.check(lib.rnb_up_sample_step(R.native.ptr(ro), R.native.ptr(rd), R.native.ptr(z_in),
                                          R.native.ptr(sdf_in), B, n, n_new, float(st["inv_s"]),
                                          R.native.ptr(new_z), R.native.ptr(inds), R.native.ptr(z_out),
                                          R.native.ptr(sidx), None))
    torch.cuda.synchronize()
    return new_z.cpu(), inds.cpu(), z_out.cpu(), sidx.cpu()


def _assert_sort_index_equal_up_to_ties(sidx, new_z, z_out, st, what):
    """The merged order must equal the reference's traced `torch.sort` index — except inside a run of BIT-EQUAL depths.
    The reference calls torch.sort without stable=True (models/renderer.py:183); when a new depth equals an old one
    exactly (3 of 2,048 rows of the B = 512 fixture: the inverse-CDF lerp lands on a bin edge), the order of the two
    equal keys is whatever ATen's unstable CPU sort (a vectorised quicksort whose code path depends on the host ISA)
    happens to produce — there the traced index is an artefact of the build container's CPU, not a property of the
    reference.  The device merge is the stable order (old before new).  The two tied samples are the same point of
    the ray, so either order carries the same SDF value.  Returns the number of rows that hold such a tie."""
    ref = st["sort_index"]
    diff = sidx != ref
    tie = torch.zeros_like(diff)
    for zz in (st["z_out"], z_out):          # a tie in the reference's depths or in the device's own
        eq = zz[:, 1:] == zz[:, :-1]
        tie[:, 1:] |= eq
        tie[:, :-1] |= eq
    assert not bool((diff & ~tie).any()), f"{what}: sort index differs outside runs of equal depths"
    cat = torch.cat([st["z_in"], new_z], -1)  # what the device merged: the reference's old depths + its own new ones
    assert torch.equal(torch.gather(cat, 1, sidx), z_out), f"{what}: z_out must be the depths permuted by sort_index"
    assert torch.equal(sidx, torch.sort(cat, dim=1, stable=True).indices), f"{what}: the device merge is the stable order"
    return int(diff.any(dim=1).sum())


@pytest.mark.parametrize("name", CASES)
def test_up_sample_step_indices_bit_exact(R, name):
    g = Golden(name)
    worst, ties = 0.0, 0
    for i, st in enumerate(g.steps):
        new_z, inds, z_out, sidx = _up_sample_step(R, g, st)
        assert torch.equal(inds.long(), st["inds"]), f"step {i}: searchsorted indices differ"
        ties += _assert_sort_index_equal_up_to_ties(sidx.long(), new_z, z_out, st, f"step {i}")
        # depths: same arithmetic except torch.sum's vectorised summation order and the device expf
        dz = (new_z - st["new_z"]).abs().max().item()
        worst = max(worst, dz)
        torch.testing.assert_close(new_z, st["new_z"], rtol=0, atol=2e-5)
        torch.testing.assert_close(z_out, st["z_out"], rtol=0, atol=2e-5)
        assert bool((z_out[:, 1:] >= z_out[:, :-1]).all()), "merged depths must be sorted"
    print(f"{name}: max |new_z - ref| over steps = {worst:.3e}; rows with an exact old/new depth tie: {ties}")


def test_gather_sdf(R):
    lib = R.native.load()
    d = _dev()
    B, n, n_new = 5, 16, 4
    gen = torch.Generator().manual_seed(1)
    old = torch.randn(B, n, generator=gen)
    new = torch.randn(B, n_new, generator=gen)
    idx = torch.stack([torch.randperm(n + n_new, generator=gen) for _ in range(B)]).int()
    out = torch.empty(B, n + n_new, device=d)
    o_d, n_d, i_d = old.to(d), new.to(d), idx.to(d)
    R.native.check(lib.rnb_gather_sdf(R.native.ptr(o_d), R.native.ptr(n_d), R.native.ptr(i_d), B, n, n_new,
                                      R.native.ptr(out), None))
    ref = torch.gather(torch.cat([old, new], -1), 1, idx.long())
    assert torch.equal(out.cpu(), ref)


@torch.no_grad()
def _device_sampling_trace(R, g, sdf, b, z0):
    """The up-sampling loop of rnb_sample_rays composed from the public per-step entry points (rnb_up_sample_step,
    rnb_sdf_forward, rnb_gather_sdf), so that the integer outputs of every step are visible.  (no_grad, like the reference's
    loop, models/renderer.py:590: the direct network calls raise under grad mode.)"""
    lib = R.native.load()
    d = _dev()
    rc = g.mc.render
    ro, rd = b["rays_o"].contiguous(), b["rays_d"].contiguous()
    B = ro.shape[0]
    n_new = rc.n_importance // rc.up_sample_steps
    z = z0.contiguous()
    pts = ro[:, None, :] + rd[:, None, :] * z[..., None]
    sdfv = sdf.sdf(pts.reshape(-1, 3)).reshape(B, -1).contiguous()
    inds_all = []
    for i in range(rc.up_sample_steps):
        n = z.shape[1]
        new_z = torch.empty(B, n_new, device=d)
        inds = torch.empty(B, n_new, dtype=torch.int32, device=d)
        z_out = torch.empty(B, n + n_new, device=d)
        sidx = torch.empty(B, n + n_new, dtype=torch.int32, device=d)
        R.native.check(lib.rnb_up_sample_step(R.native.ptr(ro), R.native.ptr(rd), R.native.ptr(z), R.native.ptr(sdfv),
                                              B, n, n_new, float(64 * 2 ** i), R.native.ptr(new_z), R.native.ptr(inds),
                                              R.native.ptr(z_out), R.native.ptr(sidx), None))
        inds_all.append(inds.cpu().long())
        if i + 1 < rc.up_sample_steps:
            npts = ro[:, None, :] + rd[:, None, :] * new_z[..., None]
            new_sdf = sdf.sdf(npts.reshape(-1, 3)).reshape(B, n_new).contiguous()
            merged = torch.empty(B, n + n_new, device=d)
            R.native.check(lib.rnb_gather_sdf(R.native.ptr(sdfv), R.native.ptr(new_sdf), R.native.ptr(sidx), B, n,
                                              n_new, R.native.ptr(merged), None))
            sdfv = merged
        z = z_out
    return inds_all, z


# End-to-end sampling cannot be bit-exact across implementations: the up-sampling loop amplifies a last-bit difference of
# the coarse SDF into other searchsorted results two steps later.  The yardstick is the reference-pinned CPU oracle ITSELF
# under such a perturbation (tests/test_oracle_golden.py::test_weight_norm_rounding_is_amplified_by_the_up_sampling_loop:
# the mathematically equal weight-norm expression): 8 of 240 fixture rays change at least one index row, 98.4 % of the
# depths stay within 1e-4.  The device is held to that class — not to its own last measurement:
ORACLE_FLIP_RATE = 8.0 / 240.0     # rays with any differing index row, oracle vs re-associated oracle
ORACLE_CLOSE = 0.984               # fraction of depths within 1e-4, same experiment


def _max_flipped(n_rays):
    """Rays of one fixture that may differ: 3 x the oracle's own rate (small fixtures: at least one)."""
    return max(1, math.ceil(3.0 * ORACLE_FLIP_RATE * n_rays))


def _e2e_run(R, g):
    """Device sampling of one fixture: (flipped rays, rays, fraction of depths within 1e-4, max |dz|, z, z0)."""
    p, sdf, dev, col, ren = _build(R, g)
    b = {k: v.to(_dev()) for k, v in g.batch.items()}
    perturb = g.mc.render.perturb if g.perturb_overwrite < 0 else g.perturb_overwrite
    packed = ren._pack(False)
    z = ren.sample_z_vals(b["rays_o"], b["rays_d"], b["near"], b["far"], packed, perturb, b["t_rand"])
    ren0 = R.NeuSRenderer(None, sdf, dev, col, n_samples=g.mc.render.n_samples, n_importance=0, n_outside=0,
                          up_sample_steps=1, perturb=g.mc.render.perturb)
    z0 = ren0.sample_z_vals(b["rays_o"], b["rays_d"], b["near"], b["far"], packed, perturb, b["t_rand"])
    if g.n_steps == 0:
        return 0, z.shape[0], 1.0, 0.0, z, z0, None
    inds_all, z_composed = _device_sampling_trace(R, g, sdf, b, z0)
    same = torch.ones(z.shape[0], dtype=torch.bool)
    for mine, st in zip(inds_all, g.steps):
        same &= (mine == st["inds"]).all(dim=1)
    diff = (z.cpu() - g.z_fine).abs()
    return int((~same).sum()), same.numel(), (diff < 1e-4).float().mean().item(), float(diff.max()), z, z0, z_composed


@pytest.mark.parametrize("name", CASES)
def test_sample_rays_end_to_end(R, name):
    """Whole prologue on the device.  Coarse depths bit-exact; rnb_sample_rays equal to the loop composed from the per-step
    entry points; the final depths and index rows against the reference's within the class the oracle's own re-association
    experiment defines (above)."""
    g = Golden(name)
    flipped, n_rays, frac_close, dmax, z, z0, z_composed = _e2e_run(R, g)
    ref = g.z_fine
    assert z.shape == ref.shape
    assert bool((z[:, 1:] >= z[:, :-1]).all())
    z0_ref = g.steps[0]["z_in"] if g.n_steps else g.z_fine
    assert torch.equal(z0.cpu(), z0_ref), "initial depths must be bit-exact"
    if g.n_steps == 0:
        assert torch.equal(z.cpu(), ref)
        return
    assert torch.equal(z_composed, z), "rnb_sample_rays must equal the loop composed from the per-step entry points"
    print(f"E2E {name}: z_vals within 1e-4: {frac_close:.4f} (oracle experiment: {ORACLE_CLOSE}); rays with a differing index row: "
          f"{flipped}/{n_rays} (allowed {_max_flipped(n_rays)} = 3 x the oracle experiment's rate); max |dz| {dmax:.3e}")
    assert frac_close >= ORACLE_CLOSE
    assert flipped <= _max_flipped(n_rays)
    assert (z.cpu() - ref).abs().mean().item() < 5e-4


def test_sample_rays_end_to_end_aggregate_rate(R):
    """Over all full-size fixtures together (704 rays) the device changes index rows at no more than 1.5 x the rate of the
    oracle's own re-association experiment (3.3 %), and keeps at least its fraction of depths within 1e-4."""
    flipped = rays = 0
    close = []
    for name in [c for c in CASES if c.startswith("full") and Golden(c).n_steps > 0]:
        f, n, fc, _, _, _, _ = _e2e_run(R, Golden(name))
        flipped += f
        rays += n
        close.append(fc * n)
    rate = flipped / rays
    print(f"E2E aggregate: {flipped}/{rays} rays differ = {rate:.4f} (oracle experiment {ORACLE_FLIP_RATE:.4f}); depths within 1e-4: "
          f"{sum(close) / rays:.4f} (oracle experiment {ORACLE_CLOSE})")
    assert rays >= 600
    assert rate <= 1.5 * ORACLE_FLIP_RATE
    assert sum(close) / rays >= ORACLE_CLOSE


# ---------------------------------------------------------------------------------------------------
# fine pass forward + backward on the reference's own z_vals
# ---------------------------------------------------------------------------------------------------
def test_more_weight_gradient_jobs_than_one_launch_group_holds(R):
    """A 12-hidden-layer SDF network and a 4-hidden-layer albedo network, all 256 wide: 17 weight-gradient jobs of the
    256-row kernel, more than the 12 one launch group carries, so the group is flushed mid-way and its slab workspace
    must be handed to the next group (DwBatch::flush_staged).  Every parameter gradient against the oracle's autograd."""
    mc = O.ModelConf(sdf=O.SDFConf(n_layers=12, skip_in=(4,)), color=O.ColorConf(n_layers=4),
                     render=O.RenderConf(n_samples=32, n_importance=32, up_sample_steps=2))
    torch.manual_seed(6)
    p = O.init_params(mc)
    with torch.no_grad():
        p["dev.variance"].fill_(0.3)
    sdf, dev, col, ren = R.build_from_named_params(mc, p, _dev())
    batch = O.synthetic_batch(64, seed=37, step=1, warmup=False)
    b = {k: v.to(_dev()) for k, v in batch.items()}
    out = ren.render_rnb(b["rays_o"], b["rays_d"], b["near"], b["far"], b["lights_dir"], cos_anneal_ratio=1.0,
                         t_rand=b["t_rand"])
    O.rnb_loss(out, b["true_rgb"], b["mask"])[0].backward()
    z = ren.last_z_vals.cpu()
    pr = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    ref = O.render_rnb(pr, mc, batch["rays_o"], batch["rays_d"], batch["near"], batch["far"], batch["lights_dir"],
                       cos_anneal_ratio=1.0, warmup=False, z_vals=z)
    O.rnb_loss(ref, batch["true_rgb"], batch["mask"])[0].backward()
    for k in ("color_fine", "weights", "gradients"):
        torch.testing.assert_close(out[k].detach().cpu(), ref[k].detach(), rtol=1e-4, atol=2e-5, msg=lambda m: f"{k}: {m}")
    n_checked = 0
    for name, leaf in _named(sdf, dev, col).items():
        gr = pr[name].grad
        assert leaf.grad is not None and bool(torch.isfinite(leaf.grad).all()), name
        if float(gr.norm()) < 1e-12:
            continue
        rel = float((leaf.grad.cpu().reshape(gr.shape) - gr).norm() / gr.norm())
        assert rel < 2e-3, (name, rel)
        n_checked += 1
    assert n_checked >= 3 * (13 + 5)


def _render(ren, g, b, z_vals):
    kw = dict(perturb_overwrite=g.perturb_overwrite, cos_anneal_ratio=g.cos_anneal_ratio, z_vals=z_vals)
    if g.api == "render":
        bg = b.get("background_rgb")
        return ren.render(b["rays_o"], b["rays_d"], b["near"], b["far"], background_rgb=bg, **kw)
    fn = ren.render_rnb_warmup if g.api == "render_rnb_warmup" else ren.render_rnb
    return fn(b["rays_o"], b["rays_d"], b["near"], b["far"], b["lights_dir"], no_albedo=g.no_albedo, **kw)


def _loss(g, out, b):
    if g.api == "render":
        return (out["color_fine"] - b["true_rgb"][0]).abs().mean() + 0.1 * out["gradient_error"] \
            + 0.1 * torch.nn.functional.binary_cross_entropy(
                out["weight_sum"].clip(1e-3, 1 - 1e-3), (b["mask"] > 0.5).float())
    return O.rnb_loss(out, b["true_rgb"], b["mask"])[0]


def _named(sdf, dev, col):
    named = {("sdf." + k): v for k, v in sdf.named_parameters()}
    named["dev.variance"] = dev.variance
    named.update({("color." + k): v for k, v in col.named_parameters()})
    return named


@pytest.mark.parametrize("name", CASES)
def test_fine_pass_golden(R, name):
    """Fine pass forward + loss + backward on the reference's own z_vals, bounded by the fp32 reference's own
    distance from the fp64 reference (module docstring)."""
    g = Golden(name)
    p, sdf, dev, col, ren = _build(R, g)
    b = {k: v.to(_dev()) for k, v in g.batch.items()}
    z_vals = g.z_fine.to(_dev())
    out = _render(ren, g, b, z_vals)
    loss = _loss(g, out, b)
    got_all = {k: out[k].detach().cpu().double() for k in g.out if k != "loss"}
    got_all["loss"] = loss.detach().cpu().double()
    _assert_has_surface(g.out)       # the reference's own outputs: every fixture renders a surface
    _assert_has_surface(out)
    worst_out = ("", 0.0)
    for k, ref32 in g.out.items():
        got = got_all[k]
        assert got.shape == ref32.shape, k
        if k == "inside_sphere":
            assert torch.equal(got.float(), ref32), "inside_sphere is an exact predicate of the inputs"
            continue
        ref64 = g.out64[k]
        e_hip = float((got - ref64).abs().max())
        e_ref = float((ref32.double() - ref64).abs().max())
        bound = K_OUT * e_ref + FLOOR_OUT * max(1.0, float(ref64.abs().max()))
        ratio = e_hip / max(e_ref, 1e-30)
        if ratio > worst_out[1] and e_hip > FLOOR_OUT:
            worst_out = (k, ratio)
        assert e_hip <= bound, f"{k}: |hip - fp64| {e_hip:.3e} > {bound:.3e} (fp32 reference: {e_ref:.3e})"
    loss.backward()
    torch.cuda.synchronize()
    named = _named(sdf, dev, col)
    with_grad = {k for k, v in named.items() if v.grad is not None}
    assert with_grad == set(g.grads.keys())
    worst = ("", 0.0, 0.0)
    for k, g64 in g.grad64.items():
        st = g.grad64_stride[k]
        mine = named[k].grad.detach().cpu().reshape(-1)[::st].double()
        n64 = float(g64.norm())
        if n64 < 1e-12:
            assert float(mine.abs().max()) < 1e-8, k
            continue
        rel = float((mine - g64).norm()) / n64
        bound = _grad_bound(g.rel32s[k])
        if rel / bound > worst[1]:
            worst = (k, rel / bound, rel)
        assert rel <= bound, f"{k}: rel-L2 vs fp64 {rel:.3e} > {bound:.3e} (fp32 reference: {g.rel32s[k]:.3e})"
    # and directly against the fp32 reference's gradients (every stored element)
    for k, ref in g.grads.items():
        mine = named[k].grad.detach().cpu().reshape(-1)[:: g.grad_stride]
        if g.gradnorm[k] < 1e-10:
            assert float(mine.abs().max()) < 1e-8, k
            continue
        rel = float((mine - ref).double().norm() / ref.double().norm())
        # triangle inequality through the fp64 result: (K_GRAD + 1) x the fp32 reference's own error
        assert rel <= min(2.0 * GRAD_CAP, max(2e-4, (1.0 + K_GRAD) * max(g.rel32.get(k, 0.0), g.rel32s.get(k, 0.0)))), \
            f"{k}: rel-L2 vs the fp32 reference {rel:.3e}"
    print(f"FINE {name}: worst output error ratio hip/ref32 = {worst_out[1]:.2f} ({worst_out[0]}); worst gradient: "
          f"{worst[0]} rel-L2 vs fp64 {worst[2]:.2e} = {worst[1]:.2f} of its bound")
    mine_g = {k: named[k].grad.detach().cpu().reshape(-1)[:: g.grad_stride] for k in g.grads}
    _check_survey(name, _survey_counts(got_all, g.out, mine_g, g.grads))


def _step_against_fp64(R, mc, p, sdf, dev, col, ren, batch, tag, survey=True):
    """One END-TO-END train-shaped step on the device (sampling + fine pass + loss + backward) against the CPU oracle in fp64
    on the z_vals the device sampled; outputs and every parameter gradient bounded by the fp32 oracle's own distance from fp64
    (module docstring).  `p`: the named parameters (CPU tensors) the device modules were built from."""
    b = {k: v.to(_dev()) for k, v in batch.items()}
    out = ren.render_rnb(b["rays_o"], b["rays_d"], b["near"], b["far"], b["lights_dir"], cos_anneal_ratio=1.0,
                         t_rand=b["t_rand"])
    loss = O.rnb_loss(out, b["true_rgb"], b["mask"])[0]
    loss.backward()
    torch.cuda.synchronize()
    _assert_has_surface(out, dev.variance.grad)
    z = ren.last_z_vals.cpu()
    torch.set_num_threads(16)
    # ground truth in float64 (bias gradients are sums of 65,536 signed terms: an fp32 CPU sum is itself
    # only good to ~1e-3 there, so both fp32 implementations are measured against the fp64 oracle)
    pr = {k: v.detach().double().requires_grad_(True) for k, v in p.items()}
    b64 = {k: v.double() for k, v in batch.items()}
    ref = O.render_rnb(pr, mc, b64["rays_o"], b64["rays_d"], b64["near"], b64["far"], b64["lights_dir"],
                       cos_anneal_ratio=1.0, z_vals=z.double())
    ref_loss = O.rnb_loss(ref, b64["true_rgb"], b64["mask"])[0]
    ref_loss.backward()
    # the same step with the oracle in fp32 (the reference's own arithmetic) calibrates outputs and gradients
    p32 = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    ref32 = O.render_rnb(p32, mc, batch["rays_o"], batch["rays_d"], batch["near"], batch["far"],
                         batch["lights_dir"], cos_anneal_ratio=1.0, z_vals=z)
    O.rnb_loss(ref32, batch["true_rgb"], batch["mask"])[0].backward()
    for k in ("color_fine", "weights", "weight_sum", "gradients", "cdf_fine", "gradient_error"):
        assert bool(torch.isfinite(out[k]).all()), f"{tag}: {k} is not finite"
        r64 = ref[k].detach().double()
        e_hip = float((out[k].detach().cpu().double() - r64).abs().max())
        e_ref = float((ref32[k].detach().double() - r64).abs().max())
        bound = K_OUT * e_ref + FLOOR_OUT * max(1.0, float(r64.abs().max()))
        assert e_hip <= bound, f"{tag}: {k}: |hip - fp64| {e_hip:.3e} > {bound:.3e} (fp32 CPU oracle: {e_ref:.3e})"
    torch.testing.assert_close(loss.detach().cpu().double(), ref_loss.detach(), rtol=1e-5, atol=1e-6)
    named = _named(sdf, dev, col)
    worst = ("", 0.0, 0.0)
    for k, v in named.items():
        rg = pr[k].grad
        den = float(rg.norm())
        assert den > 1e-9, f"{k}: the fp64 gradient vanishes: not a parity target"
        assert bool(torch.isfinite(v.grad).all()), f"{tag}: gradient of {k} is not finite"
        rel = float((v.grad.cpu().double() - rg).norm()) / den
        rel32 = float((p32[k].grad.double() - rg).norm()) / den
        assert rel32 <= GRAD_CAP / K_GRAD, f"{k}: the fp32 oracle itself is {rel32:.2e} from fp64: not a parity target"
        bound = _grad_bound(rel32)
        if rel / bound > worst[1]:
            worst = (k, rel / bound, rel)
        assert rel <= bound, f"{tag}: {k}: rel-L2 {rel:.3e} > {bound:.3e} (fp32 CPU oracle: {rel32:.3e})"
    print(f"{tag} vs fp64 oracle: weight_sum mean {float(out['weight_sum'].mean()):.3f}; worst gradient {worst[0]}: "
          f"rel-L2 {worst[2]:.2e} = {worst[1]:.2f} of its bound")
    if survey:
        # SURVEY 8c's original bounds, against the fp32 oracle (the reference's arithmetic) on the same depths
        keys = ("color_fine", "weights", "weight_sum", "gradients", "cdf_fine", "gradient_error")
        _check_survey(tag, _survey_counts({k: out[k].detach().cpu() for k in keys}, {k: ref32[k].detach() for k in keys},
                                          {k: v.grad.cpu() for k, v in named.items()}, {k: p32[k].grad for k in named}))
    return out


def test_full_batch_512_matches_oracle(R):
    """BASELINE config 2 at its real shape END TO END (device sampling + fine pass + loss + backward): 512 rays x
    (64+64), full-size nets in the sharpened state of the reference-generated fixtures (a model that HAS a surface),
    against the CPU oracle in fp64 on the z_vals the device sampled.  (The reference's own run of this shape on its own
    z_vals is the fixture `full_main_b512`, covered by test_fine_pass_golden / test_up_sample_step_indices_bit_exact /
    test_sample_rays_end_to_end.)  Non-degeneracy is asserted first: a scene without a surface would pass any
    absolute bound."""
    g = Golden("full_main_b512")
    p = g.params()
    sdf, dev, col, ren = R.build_from_named_params(g.mc, p, _dev())
    _step_against_fp64(R, g.mc, p, sdf, dev, col, ren, O.synthetic_batch(512, seed=22, step=7, warmup=False), "b512_end_to_end")


def _max_effective_weight(net):
    m = 0.0
    for lin in net.lins():
        v = lin.weight_v.detach().double()
        m = max(m, float((lin.weight_g.detach().double().reshape(-1, 1) * v / v.norm(dim=1, keepdim=True)).abs().max()))
    return m


def test_operands_pushed_out_of_the_fp16_range_between_two_train_steps(R):
    """The reference evaluates ANY weights (models/fields.py:82-104, plain fp32).  The default arithmetic multiplies two-plane
    fp16 operands; through round 4 their scales were fixed (|weight| < 255, |activation| < 1023: beyond, inf / NaN).  They are
    now taken from the data (per matrix for the weights, per tile and layer for activations and Jacobian rows, per launch
    for the saved state the weight gradients read), so nothing is out of range.  Here: one normal train step, then — between
    two steps, as a checkpoint load or a diverging run would — one hidden unit of the SDF network and one of the albedo
    network are rescaled by K = 2e4 (row times K, the next layer's column divided by K: the function keeps its
    surface, the operands leave the old range: weights of several thousand, activations of those units beyond 1023), and the FOLLOWING step
    must match the oracle in fp64 at the same calibrated bounds as every golden fixture — outputs and all 37 gradients."""
    g = Golden("full_main_sharp")
    mc = g.mc
    p = g.params()
    sdf, dev, col, ren = R.build_from_named_params(mc, p, _dev())
    params = list(sdf.parameters()) + list(dev.parameters()) + list(col.parameters())
    opt = R.FlatAdam(params, lr=1e-4)
    b = {k: v.to(_dev()) for k, v in O.synthetic_batch(64, seed=23, step=8, warmup=False).items()}
    out = ren.render_rnb(b["rays_o"], b["rays_d"], b["near"], b["far"], b["lights_dir"], cos_anneal_ratio=1.0, t_rand=b["t_rand"])
    loss, _ = R.rnb_loss(out, b["true_rgb"], b["mask"])
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    assert max(_max_effective_weight(sdf), _max_effective_weight(col)) < 64.0
    K = 2.0e4
    with torch.no_grad():
        sdf.lin3.weight_g[7] *= K
        sdf.lin3.bias[7] *= K
        sdf.lin4.weight_v[:, 7] /= K
        col.lin0.weight_g[5] *= K
        col.lin0.bias[5] *= K
        col.lin1.weight_v[:, 5] /= K
    assert _max_effective_weight(sdf) > 255.0 and _max_effective_weight(col) > 255.0, "the push must leave the old fp16 range"
    for q in params:
        q.grad = None
    p_now = {k: v.detach().cpu().clone() for k, v in _named(sdf, dev, col).items()}
    out = _step_against_fp64(R, mc, p_now, sdf, dev, col, ren, O.synthetic_batch(64, seed=24, step=9, warmup=False),
                             "pushed_out_of_range", survey=False)
    # ... and the activations really were beyond the old limit (the per-tile scale was exercised, not just the weights')
    with torch.no_grad():
        z = ren.last_z_vals.cpu()
        bb = O.synthetic_batch(64, seed=24, step=9, warmup=False)
        pts = (bb["rays_o"][:, None, :] + bb["rays_d"][:, None, :] * z[:, :, None]).reshape(-1, 3)
        x = O.embed(pts * mc.sdf.scale, mc.sdf.multires)
        for l in range(4):     # layers 0..3 (the skip connection enters at 4)
            x = O.softplus100(torch.nn.functional.linear(x, O.effective_weight(p_now, f"sdf.lin{l}"), p_now[f"sdf.lin{l}.bias"]))
        assert float(x.abs().max()) > 1023.0, f"largest activation of layer 3: {float(x.abs().max()):.1f} (the old limit was 1023)"
    del out


def test_x2h_has_no_operand_range(R):
    """Forward-only SDF / normal queries on the full-size network with operands far outside what the fixed fp16 scales of
    round 4 could hold: a weight row of ~1500, coordinates of 3000, a last layer that makes the Jacobian rows ~1e4.  All must
    come back finite and equal to the oracle (the arithmetic is the reference's fp32 to 2^-22 per operand, whatever the
    magnitudes)."""
    mc = O.ModelConf()
    torch.manual_seed(6)
    p = O.init_params(mc)
    sdf, devn, col, ren = R.build_from_named_params(mc, p, _dev())
    gen = torch.Generator().manual_seed(2)
    pts = (torch.rand(4096, 3, generator=gen) * 2 - 1).to(_dev())

    def oracle(points, dt):
        named = {("sdf." + n): q.detach().cpu().to(dt) for n, q in sdf.named_parameters()}
        x = points.cpu().to(dt).requires_grad_(True)
        y = O.sdf_forward(named, mc.sdf, x)[:, :1]
        (n,) = torch.autograd.grad(y.sum(), x)
        return y.detach(), n

    def check(tag, points, k_out=K_OUT):
        with torch.no_grad():
            got, nrm = sdf.sdf(points), sdf.gradient(points).reshape(-1, 3)
        assert bool(torch.isfinite(got).all()) and bool(torch.isfinite(nrm).all()), tag
        # the usual calibration: distance from the fp64 oracle, bounded by the fp32 oracle's own (coordinates of 3000 make
        # sin(32 x) ill-conditioned for ANY fp32 implementation)
        (r64, n64), (r32, n32) = oracle(points, torch.float64), oracle(points, torch.float32)
        for name, mine, ref64, ref32 in (("sdf", got, r64, r32), ("normal", nrm, n64, n32)):
            e_hip = float((mine.cpu().double() - ref64).abs().max())
            e_ref = float((ref32.double() - ref64).abs().max())
            bound = k_out * e_ref + FLOOR_OUT * max(1.0, float(ref64.abs().max()))
            assert e_hip <= bound, f"{tag}: {name}: |hip - fp64| {e_hip:.3e} > {bound:.3e} (fp32 CPU oracle: {e_ref:.3e})"

    check("in range", pts)
    with torch.no_grad():
        sdf.lin3.weight_g[7] = 1.0e4            # one row of layer 3 with |w| up to 1e4 max|v| / ||v|| (~ 1500)
    assert _max_effective_weight(sdf) > 300.0
    check("weight row ~1500", pts)
    with torch.no_grad():
        sdf.lin8.weight_g[0] *= 3.0e4           # d sdf / d a_last: Jacobian rows of ~1e4 in the reverse sweep
        sdf.lin8.bias[0] *= 3.0e4
    check("Jacobian rows ~1e4", pts)
    # (sin / cos of 2^5 x 3000: the derivative of the encoding is ill-conditioned to 1e-3 relative in ANY fp32 — the fp32 CPU
    # oracle is 2e-4 from fp64 on the normal here — so one realisation of that noise is a loose yardstick: factor 10)
    check("coordinates of 3000", pts * 3000.0, k_out=10.0)


def test_size_independent_properties(R):
    """Properties that hold at any size: weights in [0,1] with sum <= 1; colour is linear in the light
    directions (render_rnb has no ReLU); no_albedo with unit lights reproduces sum_s w * (n.l)."""
    mc = O.ModelConf()
    torch.manual_seed(1)
    p = O.init_params(mc)
    sdf, dev, col, ren = R.build_from_named_params(mc, p, _dev())
    batch = O.synthetic_batch(256, seed=5, step=1, warmup=False)
    b = {k: v.to(_dev()) for k, v in batch.items()}
    with torch.no_grad():
        o1 = ren.render_rnb(b["rays_o"], b["rays_d"], b["near"], b["far"], b["lights_dir"], cos_anneal_ratio=1.0,
                            t_rand=b["t_rand"])
        z = ren.last_z_vals
        o2 = ren.render_rnb(b["rays_o"], b["rays_d"], b["near"], b["far"], 2.0 * b["lights_dir"],
                            cos_anneal_ratio=1.0, z_vals=z)
        o3 = ren.render_rnb(b["rays_o"], b["rays_d"], b["near"], b["far"], b["lights_dir"], cos_anneal_ratio=1.0,
                            z_vals=z, no_albedo=True)
    w = o1["weights"]
    assert bool((w >= 0).all()) and bool((w <= 1).all())
    assert bool((o1["weight_sum"] <= 1.0 + 1e-5).all())
    torch.testing.assert_close(o2["color_fine"], 2.0 * o1["color_fine"], rtol=1e-5, atol=1e-6)
    sh = (o3["gradients"][None] * b["lights_dir"]).sum(-1)
    expect = (o3["weights"][None] * sh).sum(-1, keepdim=True).expand(-1, -1, 3)
    torch.testing.assert_close(o3["color_fine"], expect, rtol=1e-4, atol=1e-5)
    assert torch.equal(o1["weights"], o3["weights"])


def test_missing_library_fails_loudly(R, monkeypatch, tmp_path):
    monkeypatch.setattr(R.native, "_lib", None)
    monkeypatch.setattr(R.native, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(R.native.NativeError):
        R.native.load()


def test_256_samples_per_ray_matches_oracle(R):
    """BASELINE config 5 sampling shape (128 coarse + 4x32 importance samples = 256 samples per ray),
    fp32, small networks: sampling statistics + fine pass + gradients against the oracle."""
    mc = O.ModelConf(sdf=O.SDFConf(d_out=65, d_hidden=64), color=O.ColorConf(d_feature=64, d_hidden=64),
                     render=O.RenderConf(n_samples=128, n_importance=128, up_sample_steps=4))
    torch.manual_seed(3)
    p = O.init_params(mc)
    sdf, dev, col, ren = R.build_from_named_params(mc, p, _dev())
    batch = O.synthetic_batch(24, seed=31, step=0, warmup=True)
    b = {k: v.to(_dev()) for k, v in batch.items()}
    out = ren.render_rnb_warmup(b["rays_o"], b["rays_d"], b["near"], b["far"], b["lights_dir"],
                                cos_anneal_ratio=1.0, t_rand=b["t_rand"])
    assert out["weights"].shape == (24, 256)
    loss = O.rnb_loss(out, b["true_rgb"], b["mask"])[0]
    loss.backward()
    z = ren.last_z_vals.cpu()
    assert bool((z[:, 1:] >= z[:, :-1]).all())
    z_ref = O.sample_rays(p, mc, batch["rays_o"], batch["rays_d"], batch["near"], batch["far"], batch["t_rand"], 1.0)
    assert ((z - z_ref).abs() < 1e-4).float().mean().item() > 0.95
    pr = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    ref = O.render_rnb(pr, mc, batch["rays_o"], batch["rays_d"], batch["near"], batch["far"], batch["lights_dir"],
                       cos_anneal_ratio=1.0, warmup=True, z_vals=z)
    O.rnb_loss(ref, batch["true_rgb"], batch["mask"])[0].backward()
    for k in ("color_fine", "weights", "gradients", "cdf_fine"):
        torch.testing.assert_close(out[k].detach().cpu(), ref[k].detach(), rtol=1e-4, atol=2e-5,
                                   msg=lambda m: f"{k}: {m}")
    g = sdf.lin2.weight_v.grad.cpu()
    gr = pr["sdf.lin2.weight_v"].grad
    assert float((g - gr).norm() / gr.norm()) < 1e-3


def test_direct_network_calls_are_loud_under_grad(R):
    """models/fields.py:82-127, :177-215 are autograd modules in the reference (`gradient` with create_graph=True); here the
    direct calls are native forward sweeps without a grad_fn.  Under grad mode with trainable parameters (or an input that
    requires grad) they must raise and name NeuSRenderer.render*, not detach silently (missing item 1); under
    no_grad, or with frozen parameters, they work."""
    mc = O.ModelConf(sdf=O.SDFConf(d_out=65, d_hidden=64), color=O.ColorConf(d_feature=64, d_hidden=64))
    torch.manual_seed(0)
    p = O.init_params(mc)
    sdf, devn, col, ren = R.build_from_named_params(mc, p, _dev())
    pts = (torch.rand(100, 3) - 0.5).to(_dev())
    nrm = torch.randn(100, 3, device=_dev())
    feat = torch.randn(100, 64, device=_dev())
    assert torch.is_grad_enabled() and sdf.lin0.bias.requires_grad
    for call in (lambda: sdf(pts), lambda: sdf.sdf(pts), lambda: sdf.gradient(pts), lambda: sdf.sdf_hidden_appearance(pts),
                 lambda: col(pts, nrm, nrm, feat)):
        with pytest.raises(RuntimeError, match="NeuSRenderer.render"):
            call()
    with torch.no_grad():
        out = sdf(pts)
        g = sdf.gradient(pts)
        c = col(pts, nrm, nrm, feat)
    assert out.shape == (100, 65) and g.shape == (100, 1, 3) and c.shape == (100, 3)
    assert out.grad_fn is None and not out.requires_grad
    # frozen parameters + plain inputs: nothing could receive a gradient, the call is allowed under grad mode
    for q in list(sdf.parameters()) + list(col.parameters()):
        q.requires_grad_(False)
    assert torch.equal(sdf(pts), out)
    with pytest.raises(RuntimeError, match="NeuSRenderer.render"):
        sdf.sdf(pts.clone().requires_grad_(True))


def test_mv_forward_variant_matches_the_default(R):
    """RNB_VARIANT_REG_TILE: the M/V kernels (csrc/sweep_mv.hip: matrix waves + vector waves, transposed products, weights
    through an LDS-DMA ring) for forward-only sweeps of >= 25,600 points, against the default LDS-tile kernels and the
    oracle — sdf alone (the sampling passes' call) and sdf + feature."""
    mc = O.ModelConf()
    torch.manual_seed(3)
    p = O.init_params(mc)
    gen = torch.Generator().manual_seed(4)
    for k in p:                       # no layer of the geometric init is special any more
        if k.endswith("weight_v"):
            p[k] = p[k] + 0.02 * torch.randn(p[k].shape, generator=gen)
    sdf, devn, col, ren = R.build_from_named_params(mc, p, _dev())
    n = 128 * 230 + 57               # ragged: the last workgroup is partly padding
    pts = (torch.rand(n, 3, generator=gen) * 2 - 1) * 0.9
    ref = O.sdf_forward(p, mc.sdf, pts[:3000])
    from rnb_neus_fork_amd import runtime
    outs = {}
    for tag, kw in (("lds", dict(lds_tile=True)), ("mv", dict(reg_tile=True))):
        ren.set_variant(**kw)
        packed = ren._pack(True)
        outs[tag] = (runtime.sdf_forward(ren.desc, packed, pts.to(_dev()), True).cpu(),
                     runtime.sdf_forward(ren.desc, packed, pts.to(_dev()), False).cpu())
    ren.set_variant()
    for tag in outs:
        torch.testing.assert_close(outs[tag][0][:3000], ref, rtol=1e-4, atol=2e-5, msg=lambda m: f"{tag} vs oracle: {m}")
        assert torch.equal(outs[tag][0][:, :1], outs[tag][1]), f"{tag}: sdf-only and sdf+feature sweeps must agree bit for bit"
    torch.testing.assert_close(outs["mv"][0], outs["lds"][0], rtol=1e-5, atol=5e-6)


def test_fused_and_generic_paths_agree(R):
    """The 256-wide network runs through the fused sweep kernels; RNB_VARIANT_GENERIC (rnb_model_desc.variant) forces
    the per-layer GEMM path (the one other widths use), RNB_VARIANT_DW_LDS the LDS-staged weight-gradient GEMMs, the
    BWD_TI / BWD_NW bits the other tile shapes of the backward sweeps.  All must produce the same step."""
    mc = O.ModelConf()
    torch.manual_seed(1)
    p = O.init_params(mc)
    sdf, devn, col, ren = R.build_from_named_params(mc, p, _dev())
    b = {k: v.to(_dev()) for k, v in O.synthetic_batch(128, seed=5, step=1).items()}
    params = list(sdf.parameters()) + list(devn.parameters()) + list(col.parameters())
    res = {}
    z = None
    for tag, kw in (("fused", {}), ("generic", dict(generic=True)), ("dw_lds", dict(dw_lds=True)),
                    ("dw_staged", dict(dw_staged=True)), ("dw_staged_det", dict(dw_staged=True, deterministic=True)),
                    ("bwd_ti1", dict(bwd_ti=1, bwd_nw=4)), ("bwd_ti2", dict(bwd_ti=2, bwd_nw=4)),
                    ("bwd_ti2_nw8", dict(bwd_ti=2, bwd_nw=8)), ("fwd_ti1", dict(fwd_ti=1, fwd_nw=4)),
                    ("deterministic", dict(deterministic=True)), ("x3", dict(x3=True)),
                    ("fwd_ti2_nw8", dict(fwd_ti=2, fwd_nw=8)), ("f32_mfma", dict(f32_mfma=True)), ("f32_mfma_ti2", dict(f32_mfma=True, bwd_ti=2, bwd_nw=4)),
                    ("x3_ti1_nw4", dict(x3=True, bwd_ti=1, bwd_nw=4, fwd_ti=1, fwd_nw=4)),
                    ("x3_ti2_nw8", dict(x3=True, bwd_ti=2, bwd_nw=8, fwd_ti=2)),
                    ("no_x2h", dict(x2h=False)), ("no_x2h_det", dict(x2h=False, deterministic=True)),
                    ("x2h_ti1", dict(x2h=True, bwd_ti=1, bwd_nw=4, fwd_ti=1, fwd_nw=4)),
                    ("x2h_reg_tile", dict(x2h=True, reg_tile=True))):
        ren.set_variant(**kw)
        for q in params:
            q.grad = None
        out = ren.render_rnb(b["rays_o"], b["rays_d"], b["near"], b["far"], b["lights_dir"], cos_anneal_ratio=1.0,
                             t_rand=b["t_rand"], z_vals=z)
        if z is None:
            z = ren.last_z_vals     # every variant renders the same samples
        loss = O.rnb_loss(out, b["true_rgb"], b["mask"])[0]
        loss.backward()
        res[tag] = {"loss": float(loss), "wsum": float(out["weight_sum"].sum()),
                    "g": [q.grad.double().clone() for q in params]}
    ren.set_variant()
    for other in res:
        if other == "fused":
            continue
        assert abs(res["fused"]["loss"] - res[other]["loss"]) < 1e-5, other
        assert abs(res["fused"]["wsum"] - res[other]["wsum"]) < 1e-3, other
        for a_, c_ in zip(res["fused"]["g"], res[other]["g"]):
            rel = float((a_ - c_).norm() / c_.norm().clamp_min(1e-12))
            assert rel < 2e-4, f"{other}: gradient rel-L2 {rel:.2e}"


def test_gradients_scale_exactly_with_the_loss(R):
    """The x2h weight-gradient kernel and FB sweep scale their adjoint operands by powers of two taken from the data (the
    recorded maxima), so multiplying the loss by 2^k must multiply every gradient by exactly 2^k — no overflow at 2^+40
    (adjoints ~1e8: far outside fp16 without the scales), no loss of bits at 2^-40.  Deterministic variant (ordered
    reductions): the comparison is bit for bit."""
    mc = O.ModelConf()
    torch.manual_seed(5)
    p = O.init_params(mc)
    with torch.no_grad():
        p["dev.variance"].fill_(0.3)
    sdf, devn, col, ren = R.build_from_named_params(mc, p, _dev())
    ren.set_variant(deterministic=True)
    b = {k: v.to(_dev()) for k, v in O.synthetic_batch(128, seed=8, step=3).items()}
    params = list(sdf.parameters()) + list(devn.parameters()) + list(col.parameters())
    grads = {}
    z = None
    for k in (0, 40, -40):
        for q in params:
            q.grad = None
        out = ren.render_rnb(b["rays_o"], b["rays_d"], b["near"], b["far"], b["lights_dir"], cos_anneal_ratio=1.0,
                             t_rand=b["t_rand"], z_vals=z)
        z = ren.last_z_vals
        (O.rnb_loss(out, b["true_rgb"], b["mask"])[0] * (2.0 ** k)).backward()
        grads[k] = [q.grad.clone() for q in params]
        assert all(bool(torch.isfinite(x).all()) for x in grads[k]), k
    # an all-zero adjoint (recorded maxima 0: the scale falls back to its clamp) gives exactly zero gradients, not NaN
    for q in params:
        q.grad = None
    out = ren.render_rnb(b["rays_o"], b["rays_d"], b["near"], b["far"], b["lights_dir"], cos_anneal_ratio=1.0,
                         t_rand=b["t_rand"], z_vals=z)
    (O.rnb_loss(out, b["true_rgb"], b["mask"])[0] * 0.0).backward()
    assert all(bool((q.grad == 0).all()) for q in params)
    ren.set_variant()
    assert any(float(x.abs().max()) > 0 for x in grads[0])
    for k in (40, -40):
        for a_, c_ in zip(grads[0], grads[k]):
            assert torch.equal(a_, c_ * (2.0 ** -k)), f"loss x 2^{k}"


def test_deterministic_variant_is_bit_reproducible(R):
    """RNB_VARIANT_DETERMINISTIC: ordered reductions instead of fp32 atomics -> two runs give identical bits."""
    mc = O.ModelConf()
    torch.manual_seed(2)
    p = O.init_params(mc)
    sdf, devn, col, ren = R.build_from_named_params(mc, p, _dev())
    ren.set_variant(deterministic=True)
    b = {k: v.to(_dev()) for k, v in O.synthetic_batch(96, seed=6, step=2).items()}
    params = list(sdf.parameters()) + list(devn.parameters()) + list(col.parameters())
    runs = []
    z = None
    for _ in range(3):
        for q in params:
            q.grad = None
        out = ren.render_rnb(b["rays_o"], b["rays_d"], b["near"], b["far"], b["lights_dir"], cos_anneal_ratio=1.0,
                             t_rand=b["t_rand"], z_vals=z)
        z = ren.last_z_vals
        O.rnb_loss(out, b["true_rgb"], b["mask"])[0].backward()
        runs.append([q.grad.clone() for q in params])
    for r in runs[1:]:
        for a_, c_ in zip(runs[0], r):
            assert torch.equal(a_, c_), "deterministic variant must be bit-reproducible"


def test_second_backward_raises_clearly(R):
    mc = O.ModelConf(sdf=O.SDFConf(d_out=65, d_hidden=64), color=O.ColorConf(d_feature=64, d_hidden=64),
                     render=O.RenderConf(n_samples=16, n_importance=16))
    torch.manual_seed(0)
    p = O.init_params(mc)
    sdf, devn, col, ren = R.build_from_named_params(mc, p, _dev())
    b = {k: v.to(_dev()) for k, v in O.synthetic_batch(8, seed=3, step=0).items()}
    out = ren.render_rnb(b["rays_o"], b["rays_d"], b["near"], b["far"], b["lights_dir"], t_rand=b["t_rand"])
    loss = out["color_fine"].sum()
    loss.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="second time"):
        loss.backward()


def test_wrong_device_is_rejected(R):
    """Tensors of one call must live on one GPU, and a model on cuda:k works whatever the current device is."""
    mc = O.ModelConf(sdf=O.SDFConf(d_out=65, d_hidden=64), color=O.ColorConf(d_feature=64, d_hidden=64),
                     render=O.RenderConf(n_samples=16, n_importance=16))
    torch.manual_seed(0)
    p = O.init_params(mc)
    sdf, devn, col, ren = R.build_from_named_params(mc, p, _dev())
    b = O.synthetic_batch(8, seed=3, step=0)
    with pytest.raises(RuntimeError):
        ren.render_rnb(b["rays_o"], b["rays_d"], b["near"], b["far"], b["lights_dir"], t_rand=b["t_rand"])   # CPU rays
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs for the cross-device part")
    d1 = torch.device("cuda:1")
    b1 = {k: v.to(d1) for k, v in b.items()}
    with pytest.raises(RuntimeError, match="live on"):
        ren.render_rnb(b1["rays_o"], b1["rays_d"], b1["near"], b1["far"], b1["lights_dir"], t_rand=b1["t_rand"])
    sdf1, devn1, col1, ren1 = R.build_from_named_params(mc, p, d1)
    torch.cuda.set_device(0)     # current device != the model's device: the guard must switch
    o0 = ren.render_rnb(*(b[k].to(_dev()) for k in ("rays_o", "rays_d", "near", "far", "lights_dir")), t_rand=b["t_rand"].to(_dev()))
    o1 = ren1.render_rnb(b1["rays_o"], b1["rays_d"], b1["near"], b1["far"], b1["lights_dir"], t_rand=b1["t_rand"])
    torch.testing.assert_close(o0["color_fine"].cpu(), o1["color_fine"].cpu(), rtol=1e-5, atol=1e-6)


def test_nan_parameters_give_nan_outputs_not_a_fault(R):
    """A diverged model (NaN weights) must come back as NaNs exactly where the reference's come back as NaNs — not as
    a GPU fault, and not as whatever the output buffers held before.  With a NaN SDF every CDF of the up-sampling loop
    is NaN: ATen's upper bound then returns n (new depths NaN) and torch.sort puts NaNs last, so the reference's
    z_vals are [64 finite coarse depths, 64 NaNs]; the importance-sampling merge here is total under the same order
    (sampling.hip::lt_total), so every slot is written.  The output buffers are pre-filled with a finite sentinel
    through torch.empty to prove nothing stale survives."""
    mc = O.ModelConf()
    torch.manual_seed(4)
    p = O.init_params(mc)
    sdf, devn, col, ren = R.build_from_named_params(mc, p, _dev())
    with torch.no_grad():
        sdf.lin3.bias[7] = float("nan")
        p["sdf.lin3.bias"][7] = float("nan")
    batch = O.synthetic_batch(64, seed=8, step=3)
    b = {k: v.to(_dev()) for k, v in batch.items()}
    ref = O.render_rnb(p, mc, batch["rays_o"], batch["rays_d"], batch["near"], batch["far"], batch["lights_dir"],
                       cos_anneal_ratio=1.0, t_rand=batch["t_rand"])
    orig_empty = torch.empty

    def sentinel_empty(*a, **k):           # stale-looking finite content in every buffer the shim allocates
        t = orig_empty(*a, **k)
        if t.is_cuda and t.dtype == torch.float32:
            t.fill_(0.25)
        return t

    torch.empty = sentinel_empty
    try:
        out = ren.render_rnb(b["rays_o"], b["rays_d"], b["near"], b["far"], b["lights_dir"], cos_anneal_ratio=1.0,
                             t_rand=b["t_rand"])
    finally:
        torch.empty = orig_empty
    # the library's loss kernel: NaN in, NaN out.  (torch's own binary_cross_entropy device-asserts 0 <= input <= 1 and
    # would take the whole process down on a NaN weight_sum — as it raises on the CPU in the reference.)
    loss = R.rnb_loss(out, b["true_rgb"], b["mask"])[0]
    loss.backward()
    torch.cuda.synchronize()
    z, zr = ren.last_z_vals.cpu(), ref["z_vals"]
    assert torch.equal(torch.isnan(z), torch.isnan(zr)), "NaN depths exactly where the reference has them (sorted last)"
    assert torch.equal(z[~torch.isnan(z)], zr[~torch.isnan(zr)]), "the finite (coarse) depths stay bit-exact"
    for k in ("color_fine", "weights", "weight_sum", "weight_max", "cdf_fine", "gradients", "gradient_error"):
        assert bool(torch.isnan(ref[k]).all()), k                       # what the reference does
        assert bool(torch.isnan(out[k]).all()), f"{k}: every element must be NaN, as in the reference"
    assert torch.equal(out["inside_sphere"].cpu(), ref["inside_sphere"])
    assert bool(torch.isnan(loss))
    assert bool(torch.isnan(sdf.lin0.weight_v.grad).all())


def test_nan_rays_poison_only_themselves(R):
    """Rays are independent: NaN origins in two rays of a batch give NaN outputs for exactly those rays (as the
    reference does) and leave every other ray's outputs bit-identical to the clean batch."""
    g = Golden("full_main_sharp")
    p, sdf, devn, col, ren = _build(R, g)
    batch = O.synthetic_batch(64, seed=9, step=2)
    bad = torch.tensor([3, 40])
    poisoned = {k: v.clone() for k, v in batch.items()}
    poisoned["rays_o"][bad, 1] = float("nan")
    outs = []
    with torch.no_grad():
        for bt in (batch, poisoned):
            b = {k: v.to(_dev()) for k, v in bt.items()}
            outs.append(ren.render_rnb(b["rays_o"], b["rays_d"], b["near"], b["far"], b["lights_dir"],
                                       cos_anneal_ratio=1.0, t_rand=b["t_rand"]))
    clean, pois = outs
    ref = O.render_rnb(p, g.mc, poisoned["rays_o"], poisoned["rays_d"], poisoned["near"], poisoned["far"],
                       poisoned["lights_dir"], cos_anneal_ratio=1.0, t_rand=poisoned["t_rand"])
    good = torch.ones(64, dtype=torch.bool)
    good[bad] = False
    for k in ("weights", "weight_sum", "cdf_fine", "gradients"):
        assert torch.equal(torch.isnan(pois[k]).cpu(), torch.isnan(ref[k])), k
        assert bool(torch.isnan(pois[k][bad.to(_dev())]).all()), k
        assert torch.equal(pois[k][good.to(_dev())], clean[k][good.to(_dev())]), k
    assert torch.equal(torch.isnan(pois["color_fine"]).cpu(), torch.isnan(ref["color_fine"]))
    assert torch.equal(pois["color_fine"][:, good.to(_dev())], clean["color_fine"][:, good.to(_dev())])


@pytest.mark.parametrize("variant", [dict(deterministic=True), dict(deterministic=True, bf16=True),
                                     dict(deterministic=True, f32_mfma=True)])
def test_uninitialised_workspace_is_never_read(R, variant):
    """Every buffer the shim hands to the library comes from torch.empty: whatever it held before must not reach a
    result.  The same train step (deterministic variant: bit-reproducible) is run on buffers pre-filled with NaN bit
    patterns (fp32 NaN / 0xFF bytes: per-point state, split-K slabs, packed weights and gradients, outputs) and must
    reproduce the clean run bit for bit.  (Round 2 saw four GPU faults traced to NaN depths whose origin was never
    pinned down: a kernel reading a slab or a padded row it had not written is the class of bug this catches.)"""
    g = Golden("full_main_sharp")
    batch = O.synthetic_batch(200, seed=5, step=1)        # ragged: 25,600 points, padded rows in every tile family
    runs = []
    orig_empty, orig_empty_like = torch.empty, torch.empty_like

    def poison(t):
        if t.is_cuda and t.dtype == torch.float32:
            t.fill_(float("nan"))
        elif t.is_cuda and t.dtype == torch.uint8:
            t.fill_(255)
        return t

    for poisoned in (False, True):
        p, sdf, devn, col, ren = _build(R, g)
        ren.set_variant(**variant)
        b = {k: v.to(_dev()) for k, v in batch.items()}
        if poisoned:
            torch.empty = lambda *a, **k: poison(orig_empty(*a, **k))
            torch.empty_like = lambda *a, **k: poison(orig_empty_like(*a, **k))
        try:
            out = ren.render_rnb(b["rays_o"], b["rays_d"], b["near"], b["far"], b["lights_dir"], cos_anneal_ratio=1.0,
                                 t_rand=b["t_rand"])
            loss = O.rnb_loss(out, b["true_rgb"], b["mask"])[0]
            loss.backward()
            torch.cuda.synchronize()
        finally:
            torch.empty, torch.empty_like = orig_empty, orig_empty_like
        grads = {k: v.grad.clone() for k, v in _named(sdf, devn, col).items()}
        runs.append((ren.last_z_vals.clone(), {k: v.detach().clone() for k, v in out.items()}, grads))
    (z0, o0, g0), (z1, o1, g1) = runs
    assert torch.equal(z0, z1)
    for k in o0:
        assert torch.equal(o0[k], o1[k]), f"output {k} depends on what its buffers held before the call"
    for k in g0:
        assert torch.equal(g0[k], g1[k]), f"gradient of {k} depends on what the workspace held before the call"


def test_x3_weight_mirror_is_an_exact_three_way_split(R):
    """The default arithmetic multiplies fp32 operands as hi + mid + lo bf16 terms.  The mirror that
    rnb_weightnorm_fwd writes behind the fp32 weights must hold, in MFMA-fragment order, three round-to-nearest bf16 planes
    whose sum is the fp32 weight EXACTLY (8 + 8 + 8 mantissa bits) — checked on the first matrix (layer 0: 256 x 64, at
    float offset 0; fragment (32 rows, 16 k) = planes hi, mid, lo of 64 lanes x 8 values, lane (c, h) =
    W[32 nt + c][16 ks + 4 h + {0..3, 8..11}], the SDF network's k order).  Behind it the fp16 mirror of the forward-type
    sweeps (x2h: two planes, weights times 2^8): hi = fp16(256 w), lo = fp16(256 w - hi), |hi + lo - 256 w| <= 2^-22 |256 w|
    (two 11-bit roundings; rms 2^-23.6) + half a subnormal step."""
    mc = O.ModelConf()
    torch.manual_seed(9)
    p = O.init_params(mc)
    sdf, devn, col, ren = R.build_from_named_params(mc, p, _dev())
    packed = ren._pack(True)
    total = (packed.numel() - 256) * 2 // 7              # fp32 part: total + 1.5 total (bf16 planes) + total (fp16 planes) + scale table
    assert packed.numel() == total + total // 2 * 3 + total + 256
    W0 = packed[:256 * 64].reshape(256, 64).cpu()
    j = torch.arange(8)
    kk = torch.where(j < 4, j, j + 4)                    # j-th value of a lane -> k offset inside the 16-k step (+ 4 h)

    def unpermute(planes, n_planes):                     # [nt, ks, plane, h, c, j] -> [plane, row, k]
        rec = torch.empty(n_planes, 256, 64, dtype=planes.dtype)
        for h in range(2):
            for ks in range(4):
                cols = 16 * ks + 4 * h + kk
                rec[:, :, cols] = planes[:, ks, :, h].permute(1, 0, 2, 3).reshape(n_planes, 256, 8)
        return rec

    mirror = packed[total:].view(torch.int16)[:3 * 256 * 64].cpu().to(torch.int32) & 0xFFFF
    rec = unpermute((mirror << 16).view(torch.float32).reshape(8, 4, 3, 2, 32, 8), 3)
    hi, mid, lo = rec[0], rec[1], rec[2]
    assert torch.equal((hi + mid) + lo, W0), "hi + mid + lo must reproduce the fp32 weight bit for bit"
    assert torch.equal(hi, W0.to(torch.bfloat16).to(torch.float32)), "hi = bf16(w), round to nearest even"
    assert torch.equal(mid, (W0 - hi).to(torch.bfloat16).to(torch.float32))
    assert float((lo.abs() - W0.abs() * 2.0 ** -17).clamp_min(0).max()) == 0.0
    h16 = packed[total + total // 2 * 3:].view(torch.float16)[:2 * 256 * 64].cpu().reshape(8, 4, 2, 2, 32, 8)
    rec = unpermute(h16.to(torch.float32), 2)
    ws = W0 * 256.0
    assert torch.equal(rec[0], ws.to(torch.float16).to(torch.float32)), "hi = fp16(256 w), round to nearest even"
    assert torch.equal(rec[1], (ws - rec[0]).to(torch.float16).to(torch.float32))
    err = ((rec[0].double() + rec[1].double()) - ws.double()).abs()
    assert bool((err <= ws.double().abs() * 2.0 ** -22 + 2.0 ** -25).all())
    assert float((err / ws.double().abs().clamp_min(1e-3)).pow(2).mean().sqrt()) < 2.0 ** -23


def test_survey_tolerance_counts_in_aggregate():
    """Runs last in this module: over ALL fixtures and the 512-ray end-to-end case, at most 2 fewer output tensors and 4 fewer
    gradient tensors than in the recorded round-4 run meet SURVEY 8c's original tolerance (skipped when only part of the
    module ran)."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "survey_tol_floor.json")
    floor = {k: v for k, v in json.load(open(path)).items() if not k.startswith("_")}
    if set(floor) - set(_SURVEY_SEEN):
        pytest.skip("not every fixture ran in this session")
    out_now = sum(_SURVEY_SEEN[k][0] for k in floor)
    g_now = sum(_SURVEY_SEEN[k][1] for k in floor)
    out_then = sum(v["outputs_ok"] for v in floor.values())
    g_then = sum(v["grads_ok_measured"] for v in floor.values())
    print(f"SURVEYTOL total: outputs {out_now} (recorded {out_then} of {sum(v['outputs'] for v in floor.values())}), "
          f"gradient tensors {g_now} (recorded {g_then} of {sum(v['grads'] for v in floor.values())})")
    assert out_now >= out_then - 2 and g_now >= g_then - 4
