"""GPU parity of the two train-step helpers around the renderer: the one-launch loss (rnb_loss_rnb) against the
oracle's restatement of exp_runner.py:241-258, and the flat one-launch Adam (rnb_adam_step) against
torch.optim.Adam.  Tolerances: fp32, different summation order only (1e-6 relative)."""
import pytest
import torch

from oracle import rnb_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def R():
    assert torch.cuda.is_available(), "GPU tests need a device"
    import rnb_neus_fork_amd as pkg
    pkg.native.load()
    return pkg


def _fake_render_out(B, L, seed, edge=False):
    g = torch.Generator().manual_seed(seed)
    color = torch.rand(L, B, 3, generator=g)
    rgb = torch.rand(L, B, 3, generator=g)
    ws = torch.rand(B, 1, generator=g)
    if edge:                       # values on and outside the clip range, exact zeros of the L1 term
        ws[:4, 0] = torch.tensor([0.0, 1.0, 1e-3, 1.0 - 1e-3])
        color[:, 5] = rgb[:, 5]
    mask = (torch.rand(B, 1, generator=g) > 0.4).float()
    ge = torch.rand((), generator=g)
    return color, rgb, ws, mask, ge


@pytest.mark.parametrize("B,L,mask_weight,edge", [(512, 3, 0.1, False), (37, 1, 0.1, True), (2048, 3, 0.0, False),
                                                  (1, 2, 0.1, False)])
def test_loss_matches_oracle(R, B, L, mask_weight, edge):
    color, rgb, ws, mask, ge = _fake_render_out(B, L, seed=B + L, edge=edge)
    ref_in = [t.clone().requires_grad_(True) for t in (color, ws, ge)]
    ref, ref_parts = O.rnb_loss({"color_fine": ref_in[0], "weight_sum": ref_in[1], "gradient_error": ref_in[2]},
                                rgb, mask, igr_weight=0.1, mask_weight=mask_weight)
    (ref * 1.7).backward()
    dev = torch.device("cuda:0")
    gin = [t.clone().to(dev).requires_grad_(True) for t in (color, ws, ge)]
    out, parts = R.rnb_loss({"color_fine": gin[0], "weight_sum": gin[1], "gradient_error": gin[2]},
                            rgb.to(dev), mask.to(dev), igr_weight=0.1, mask_weight=mask_weight)
    (out * 1.7).backward()
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=2e-6, atol=1e-7)
    for k in ("color_loss", "eikonal_loss", "mask_loss"):
        torch.testing.assert_close(parts[k].cpu(), ref_parts[k].detach(), rtol=2e-6, atol=1e-7)
    for a, b in zip(gin, ref_in):
        assert a.grad.shape == b.grad.shape
        torch.testing.assert_close(a.grad.cpu(), b.grad, rtol=2e-6, atol=1e-9)


def test_loss_rejects_cpu_tensors(R):
    color, rgb, ws, mask, ge = _fake_render_out(8, 3, seed=0)
    with pytest.raises(RuntimeError, match="GPU"):
        R.rnb_loss({"color_fine": color, "weight_sum": ws, "gradient_error": ge}, rgb, mask)


def _param_set(dev, seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(256, 39), (256,), (256, 1), (1,), (), (3, 256), (17, 5)]
    return [torch.nn.Parameter(torch.randn(s, generator=g).to(dev)) for s in shapes]


@pytest.mark.parametrize("flat_grads", [True, False])
def test_flat_adam_matches_torch_adam(R, flat_grads):
    dev = torch.device("cuda:0")
    pa, pb = _param_set(dev, 1), _param_set(dev, 1)
    ref = torch.optim.Adam(pa, lr=5e-4)
    opt = R.FlatAdam(pb, lr=5e-4)
    assert all(torch.equal(a, b) for a, b in zip(pa, pb))          # re-homing keeps the values
    total = sum(p.numel() for p in pa)
    g = torch.Generator().manual_seed(7)
    for it in range(25):
        flat = (torch.randn(total, generator=g) * (10.0 if it % 7 == 0 else 1e-3)).to(dev)
        off = 0
        for a, b in zip(pa, pb):
            n = a.numel()
            a.grad = flat[off:off + n].view(a.shape).clone()
            # flat_grads: views of one buffer in parameter order (what the renderer's backward produces)
            b.grad = flat[off:off + n].view(b.shape) if flat_grads else flat[off:off + n].view(b.shape).clone()
            off += n
        lr = 5e-4 * (0.5 + 0.5 * it / 25)                            # schedule via param_groups, as exp_runner does
        ref.param_groups[0]["lr"] = lr
        opt.param_groups[0]["lr"] = lr
        ref.step()
        opt.step()
    for a, b in zip(pa, pb):
        torch.testing.assert_close(b.detach(), a.detach(), rtol=1e-5, atol=1e-7)
    sd = opt.state_dict()
    rsd = ref.state_dict()
    assert sorted(sd["state"].keys()) == sorted(rsd["state"].keys())
    torch.testing.assert_close(sd["state"][0]["exp_avg_sq"], rsd["state"][0]["exp_avg_sq"], rtol=1e-4, atol=1e-12)
    # state round trip into a torch optimizer and back
    opt2 = R.FlatAdam(_param_set(dev, 1), lr=1.0)
    opt2.load_state_dict(sd)
    assert opt2.step_count == 25 and opt2.param_groups[0]["lr"] == opt.param_groups[0]["lr"]
    assert torch.equal(opt2.exp_avg, opt.exp_avg)


def test_flat_adam_skips_gradless_parameters_like_torch(R):
    """exp_runner.py:105-115 hands Adam the NeRF parameters too; they never get a gradient.  Like torch's Adam,
    FlatAdam carries them by position and trains the rest; a later change of the trained set is an error."""
    dev = torch.device("cuda:0")
    ps = _param_set(dev, 2)
    before = [p.detach().clone() for p in ps]
    opt = R.FlatAdam(ps, lr=1e-2)
    ref = torch.optim.Adam([p.detach().clone().requires_grad_(True) for p in ps], lr=1e-2)
    rp = ref.param_groups[0]["params"]
    for i in (1, 3):
        ps[i].grad = torch.ones_like(ps[i])
        rp[i].grad = torch.ones_like(rp[i])
    opt.step()
    ref.step()
    assert opt.active == [1, 3]
    for i, p in enumerate(ps):
        torch.testing.assert_close(p.detach(), rp[i].detach(), rtol=1e-6, atol=1e-7)
        assert torch.equal(p.detach(), before[i]) == (i not in (1, 3))
    assert set(opt.state_dict()["state"].keys()) == set(ref.state_dict()["state"].keys()) == {1, 3}
    ps[0].grad = torch.zeros_like(ps[0])
    with pytest.raises(RuntimeError, match="changed"):
        opt.step()


def test_flat_adam_detects_rehomed_parameters(R):
    dev = torch.device("cuda:0")
    ps = _param_set(dev, 3)
    opt = R.FlatAdam(ps)
    for p in ps:
        p.grad = torch.zeros_like(p)
    opt.step()
    ps[2].data = ps[2].data.clone()          # what module.to(...) / load with assign would do
    with pytest.raises(RuntimeError, match="flat buffer"):
        opt.step()


def test_train_step_with_library_loss_and_flat_adam_tracks_torch_ops(R):
    """Three train steps of the tiny model: library loss + FlatAdam vs. the oracle's torch-op loss +
    torch.optim.Adam around the same renderer; parameters must stay together."""
    dev = torch.device("cuda:0")
    mc = O.ModelConf(sdf=O.SDFConf(d_out=65, d_hidden=64), color=O.ColorConf(d_feature=64, d_hidden=64),
                     render=O.RenderConf(n_samples=16, n_importance=16))
    runs = []
    for use_lib in (False, True):
        torch.manual_seed(0)
        p = O.init_params(mc)
        sdf, devn, col, ren = R.build_from_named_params(mc, p, dev)
        params = list(sdf.parameters()) + list(devn.parameters()) + list(col.parameters())
        opt = R.FlatAdam(params, lr=5e-4) if use_lib else torch.optim.Adam(params, lr=5e-4)
        for it in range(3):
            b = {k: v.to(dev) for k, v in O.synthetic_batch(32, seed=3, step=it).items()}
            out = ren.render_rnb(b["rays_o"], b["rays_d"], b["near"], b["far"], b["lights_dir"],
                                 cos_anneal_ratio=1.0, t_rand=b["t_rand"])
            loss = (R.rnb_loss if use_lib else O.rnb_loss)(out, b["true_rgb"], b["mask"])[0]
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
        runs.append((float(loss.detach()), [x.detach().clone() for x in params]))
    assert runs[0][0] == pytest.approx(runs[1][0], rel=1e-4)
    for a, b in zip(runs[0][1], runs[1][1]):
        torch.testing.assert_close(b, a, rtol=1e-3, atol=2e-5)


def test_three_train_steps_track_the_cpu_oracle(R):
    """End-to-end train_rnb parity over consecutive steps: HIP renderer + library loss + flat Adam on the device
    against the CPU oracle (autograd double backward) + torch.optim.Adam, both starting from the same
    parameters and fed the same rays; the oracle renders on the depths the device sampled in that step
    (sample indices amplify 1-ulp differences).  After three updates the parameters agree."""
    dev = torch.device("cuda:0")
    mc = O.ModelConf(sdf=O.SDFConf(d_out=65, d_hidden=64), color=O.ColorConf(d_feature=64, d_hidden=64),
                     render=O.RenderConf(n_samples=16, n_importance=16))
    torch.manual_seed(5)
    p0 = O.init_params(mc)
    with torch.no_grad():
        p0["dev.variance"].fill_(0.4)
    sdf, devn, col, ren = R.build_from_named_params(mc, p0, dev)
    params = list(sdf.parameters()) + list(devn.parameters()) + list(col.parameters())
    opt = R.FlatAdam(params, lr=1e-3)
    pr = {k: v.clone().requires_grad_(True) for k, v in p0.items()}
    names = O.param_order(mc)
    ref_opt = torch.optim.Adam([pr[k] for k in names], lr=1e-3)
    losses = []
    for it in range(3):
        batch = O.synthetic_batch(48, seed=23, step=it)
        b = {k: v.to(dev) for k, v in batch.items()}
        out = ren.render_rnb(b["rays_o"], b["rays_d"], b["near"], b["far"], b["lights_dir"], cos_anneal_ratio=0.5,
                             t_rand=b["t_rand"])
        loss = R.rnb_loss(out, b["true_rgb"], b["mask"])[0]
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        z = ren.last_z_vals.cpu()
        ref = O.render_rnb(pr, mc, batch["rays_o"], batch["rays_d"], batch["near"], batch["far"], batch["lights_dir"],
                           cos_anneal_ratio=0.5, z_vals=z)
        ref_loss = O.rnb_loss(ref, batch["true_rgb"], batch["mask"])[0]
        ref_opt.zero_grad()
        ref_loss.backward()
        ref_opt.step()
        losses.append((float(loss.detach()), float(ref_loss.detach())))
    for a, b in losses:
        assert a == pytest.approx(b, rel=2e-4, abs=1e-6)
    for name, leaf in zip(names, params):
        got, want = leaf.detach().cpu().double(), pr[name].detach().double()
        # Adam normalises the step: a gradient entry near zero can flip the sign of its 1e-3 update, so compare
        # against the size of the accumulated update rather than against the parameter
        assert float((got - want).abs().max()) <= 0.15 * 3e-3 + 1e-7, name
        assert float((got - want).norm()) <= 5e-3 * float((want - p0[name].double()).norm()) + 1e-6, name
