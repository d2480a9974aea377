"""CPU-side checks: the C-ABI library loads and exports every symbol of include/rnbneus.h, argument
validation / error reporting of entry points that do not touch the GPU, and the host-side drop-in
classes (parameter names, order, initial values, state_dict round trip).  No compute calls."""
import ctypes as C
import os
import re

import pytest
import torch

import rnb_neus_fork_amd as R
from oracle import rnb_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "rnbneus.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rnb_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = R.native.load()
    declared = _header_symbols()
    assert len(declared) >= 17
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/rnbneus.h but not exported"
    assert set(declared) == set(R.native.EXPORTED_SYMBOLS)
    assert lib.rnb_abi_version() == 5


def test_build_id_ties_stored_profiles_to_the_library(tmp_path, monkeypatch):
    """rnb_build_id() = hash of csrc/ + include/ + compiler flags (buildid.source_build_id).  bench.py quotes stored PMC
    traffic only for the build it was measured on: one byte of one source changed -> another id -> `traffic: null`."""
    import json
    import sys
    sys.path.insert(0, ROOT)
    import bench
    from rnb_neus_fork_amd import buildid
    lib_id = R.native.build_id()
    assert re.fullmatch(r"[0-9a-f]{16}", lib_id), lib_id
    assert lib_id == buildid.source_build_id(), "the library in the tree was not built from the sources in the tree"
    # one byte of one kernel source differs -> a different id
    name, path = [f for f in buildid.source_files() if f[0] == "csrc/fused.hip"][0]
    data = bytearray(open(path, "rb").read())
    data[len(data) // 2] ^= 1
    other = buildid.source_build_id(overrides={name: bytes(data)})
    assert other != lib_id
    prof = {"build_id": lib_id, "rays": 512, "samples": 128, "hbm_bytes_per_launch": 1.0, "hbm_bytes_per_step": 13.0,
            "launches_per_step": 13.0}
    (tmp_path / "hbm_traffic.json").write_text(json.dumps(prof))
    monkeypatch.setattr(bench, "PROFILES", str(tmp_path))
    t, note = bench.measured_traffic("hbm_traffic.json", 1, 512, 128, lib_id)
    assert t is not None and note is None and t["hbm_bytes_per_step"] == 13.0
    t, note = bench.measured_traffic("hbm_traffic.json", 1, 512, 128, other)          # the flipped build
    assert t is None and "not quoted" in note
    t, note = bench.measured_traffic("hbm_traffic.json", 1, 256, 128, lib_id)         # another workload
    assert t is None and note
    prof.pop("build_id")                                                               # a profile from before the ids
    (tmp_path / "hbm_traffic.json").write_text(json.dumps(prof))
    t, note = bench.measured_traffic("hbm_traffic.json", 1, 512, 128, lib_id)
    assert t is None and "none recorded" in note


def test_direct_network_calls_refuse_to_detach_silently():
    """The direct calls sdf_network(x) / .sdf / .gradient / color_network(...) are native forward sweeps without a grad_fn;
    the reference's are autograd modules (models/fields.py:82-127, :177-215).  With grad mode on and trainable parameters the
    call raises (before anything touches a device) and points at NeuSRenderer.render*."""
    sdf = R.SDFNetwork(d_in=3, d_out=33, d_hidden=32, n_layers=2, skip_in=[], multires=2)
    col = R.RenderingNetwork(d_feature=32, mode="no_view_dir", d_in=6, d_out=3, d_hidden=32, n_layers=1, multires_view=2)
    x = torch.zeros(4, 3)
    for call in (lambda: sdf(x), lambda: sdf.sdf(x), lambda: sdf.gradient(x), lambda: col(x, x, x, torch.zeros(4, 32))):
        with pytest.raises(RuntimeError, match="forward-only.*NeuSRenderer.render"):
            call()
    with torch.no_grad():   # allowed — and then fails only because there is no CPU path
        with pytest.raises(RuntimeError, match="GPU"):
            sdf.sdf(x)


def _desc(**over):
    mc = O.ModelConf()
    sdf = R.SDFNetwork(d_in=3, d_out=257, d_hidden=256, n_layers=8, skip_in=[4], multires=6)
    col = R.RenderingNetwork(d_feature=256, mode="no_view_dir", d_in=6, d_out=3, d_hidden=256, n_layers=2,
                             multires_view=4)
    d = R.model_desc(sdf, col)
    for k, v in over.items():
        setattr(d, k, v)
    return d


def test_packed_layout_size_and_workspace_queries():
    lib = R.native.load()
    d = _desc()
    n = C.c_int64()
    R.native.check(lib.rnb_packed_floats(C.byref(d), C.byref(n)))
    # 8 hidden layers + feature head (each with a transposed copy for the reverse-shaped sweeps) + sdf row
    # + albedo net, all padded to multiples of 32
    expect = (2 * 256 * 64 + 256) + 7 * (2 * 256 * 256 + 256) + (2 * 256 * 256 + 256) + 256 + 32 \
        + (2 * 256 * 320 + 256) + (2 * 256 * 256 + 256) + (32 * 256 + 32)
    # the default variant of this shape multiplies fp32 operands as three bf16 terms (x3): a split mirror of 1.5 x the
    # floats follows the fp32 weights, and behind it the two fp16 planes of the forward-type sweeps (x2h: 1 x); the
    # native-fp32-MFMA and generic variants carry no mirror, bf16 one of 0.5 x
    # (+ 256 floats behind the fp16 planes: the per-matrix scale table of that mirror)
    assert n.value == expect + expect // 2 * 3 + expect + 256
    for kw, extra in ((dict(f32_mfma=True), 0), (dict(generic=True), 0), (dict(bf16=True), expect // 2),
                      (dict(x2h=False), expect // 2 * 3), (dict(x2h=True), expect // 2 * 3 + expect + 256)):
        d2 = _desc()
        d2.variant = R.native.variant_bits(**kw)
        R.native.check(lib.rnb_packed_floats(C.byref(d2), C.byref(n)))
        assert n.value == expect + extra, kw
    b = C.c_int64()
    R.native.check(lib.rnb_render_workspace_bytes(C.byref(d), 512, 128, R.native.MODE_MVPS, C.byref(b)))
    fwd_bwd = b.value
    R.native.check(lib.rnb_render_workspace_bytes(C.byref(d), 512, 128,
                                                  R.native.MODE_MVPS | R.native.FLAG_FORWARD_ONLY, C.byref(b)))
    assert 0 < b.value < fwd_bwd < 8 << 30
    tf, ff = C.c_double(), C.c_double()
    R.native.check(lib.rnb_algorithmic_flops(C.byref(d), 512, R.native.MODE_MVPS, C.byref(tf), C.byref(ff)))
    assert abs(tf.value / 512 - 1.035e9) < 0.01e9      # ~1.035 GFLOP per training ray
    assert abs(ff.value / 512 - 0.423e9) < 0.01e9


def test_invalid_arguments_are_reported_not_fatal():
    lib = R.native.load()
    n = C.c_int64()
    d = _desc(sdf_d_in=2)
    rc = lib.rnb_packed_floats(C.byref(d), C.byref(n))
    assert rc == -1 and b"sdf_d_in" in lib.rnb_last_error_string()
    d = _desc(sdf_skip_in=0)
    assert lib.rnb_packed_floats(C.byref(d), C.byref(n)) == -1
    d = _desc()
    assert lib.rnb_packed_floats(C.byref(d), None) == -4
    with pytest.raises(R.native.NativeError):
        R.native.check(lib.rnb_packed_floats(None, C.byref(n)))
    b = C.c_int64()
    d = _desc(n_importance=63)
    assert lib.rnb_sample_workspace_bytes(C.byref(d), 512, C.byref(b)) == -1


def test_drop_in_modules_match_reference_names_order_and_init():
    mc = O.ModelConf()
    torch.manual_seed(0)
    p = O.init_params(mc)
    torch.manual_seed(0)
    sdf = R.SDFNetwork(d_in=3, d_out=257, d_hidden=256, n_layers=8, skip_in=[4], multires=6, bias=0.5, scale=1.0,
                       geometric_init=True, weight_norm=True)
    dev = R.SingleVarianceNetwork(0.3)
    col = R.RenderingNetwork(d_feature=256, mode="no_view_dir", d_in=6, d_out=3, d_hidden=256, n_layers=2,
                             weight_norm=True, multires_view=4, squeeze_out=True)
    names = ["sdf." + k for k, _ in sdf.named_parameters()] + ["dev." + k for k, _ in dev.named_parameters()] \
        + ["color." + k for k, _ in col.named_parameters()]
    assert names == O.param_order(mc)
    assert sum(v.numel() for v in sdf.parameters()) == 529076
    assert sum(v.numel() for v in col.parameters()) == 146694
    for k, v in list(sdf.state_dict().items()):
        assert torch.equal(v, p["sdf." + k]), k
    for k, v in list(col.state_dict().items()):
        assert torch.equal(v, p["color." + k]), k
    sd = sdf.state_dict()
    sdf2 = R.SDFNetwork(d_in=3, d_out=257, d_hidden=256, n_layers=8, skip_in=[4], multires=6)
    sdf2.load_state_dict(sd)
    assert all(torch.equal(a, b) for a, b in zip(sdf.parameters(), sdf2.parameters()))
    ren = R.NeuSRenderer(None, sdf, dev, col, n_samples=64, n_importance=64, n_outside=0, up_sample_steps=4,
                         perturb=1.0)
    assert ren.color_depth == 3 and ren.n_samples == 64 and ren.n_importance == 64
    assert [id(x) for x in ren._leaves(True)] == [id(x) for x in list(sdf.parameters()) + list(dev.parameters())
                                                  + list(col.parameters())]


def test_unsupported_configurations_raise():
    with pytest.raises(NotImplementedError):
        R.RenderingNetwork(d_feature=256, mode="idr", d_in=9, d_out=3, d_hidden=256, n_layers=2)
    sdf = R.SDFNetwork(d_in=3, d_out=65, d_hidden=64, n_layers=8, skip_in=[4], multires=6)
    col = R.RenderingNetwork(d_feature=64, mode="no_view_dir", d_in=6, d_out=3, d_hidden=64, n_layers=2,
                             multires_view=4)
    with pytest.raises(NotImplementedError):
        R.NeuSRenderer(None, sdf, R.SingleVarianceNetwork(0.3), col, 64, 64, 32, 4, 1.0)
    ren = R.NeuSRenderer(None, sdf, R.SingleVarianceNetwork(0.3), col, 16, 16, 0, 4, 1.0)
    o = torch.zeros(4, 3)
    with pytest.raises(RuntimeError, match="GPU"):
        ren.render(o, o, o[:, :1], o[:, :1])      # CPU tensors: there is no CPU path


def test_embedder_matches_oracle():
    fn, dim = R.get_embedder(6, 3)
    x = torch.randn(10, 3)
    assert dim == 39
    assert torch.equal(fn(x), O.embed(x, 6))


def test_checkpoint_layout_round_trip(tmp_path):
    """Reference checkpoint layout (exp_runner.py:373-386): same keys, parameter names and Adam state order."""
    from rnb_neus_fork_amd import checkpoint as CK
    torch.manual_seed(0)
    nerf = R.NeRF(D=8, d_in=4, d_in_view=3, W=256, multires=10, multires_view=4, output_ch=4, skips=[4],
                  use_viewdirs=True)
    sdf = R.SDFNetwork(d_in=3, d_out=65, d_hidden=64, n_layers=8, skip_in=[4], multires=6)
    dev = R.SingleVarianceNetwork(0.3)
    col = R.RenderingNetwork(d_feature=64, mode="no_view_dir", d_in=6, d_out=3, d_hidden=64, n_layers=2,
                             multires_view=4)
    params = list(nerf.parameters()) + list(sdf.parameters()) + list(dev.parameters()) + list(col.parameters())
    opt = torch.optim.Adam(params, lr=5e-4)
    for p in list(sdf.parameters())[:3]:
        p.grad = torch.ones_like(p)
    opt.step()
    path = CK.save_checkpoint(str(tmp_path / "checkpoints" / "ckpt_000123.pth"), nerf, sdf, dev, col, opt, 123)
    raw = torch.load(path, weights_only=True)
    assert tuple(raw.keys()) == CK.KEYS
    assert list(raw["sdf_network_fine"].keys())[:3] == ["lin0.bias", "lin0.weight_g", "lin0.weight_v"]
    assert list(raw["variance_network_fine"].keys()) == ["variance"]
    sdf2 = R.SDFNetwork(d_in=3, d_out=65, d_hidden=64, n_layers=8, skip_in=[4], multires=6)
    dev2 = R.SingleVarianceNetwork(0.1)
    col2 = R.RenderingNetwork(d_feature=64, mode="no_view_dir", d_in=6, d_out=3, d_hidden=64, n_layers=2,
                              multires_view=4)
    it = CK.load_checkpoint(path, None, sdf2, dev2, col2)
    assert it == 123 and float(dev2.variance) == pytest.approx(0.3)
    assert all(torch.equal(a, b) for a, b in zip(sdf.parameters(), sdf2.parameters()))
    assert CK.latest_checkpoint(str(tmp_path / "checkpoints")) == path
    assert CK.latest_checkpoint(str(tmp_path / "checkpoints"), end_iter=100) is None


def _ref_layout_modules(seed):
    torch.manual_seed(seed)
    nerf = R.NeRF(D=2, d_in=4, d_in_view=3, W=16, multires=2, multires_view=2, output_ch=4, skips=[1],
                  use_viewdirs=True)
    sdf = R.SDFNetwork(d_in=3, d_out=65, d_hidden=64, n_layers=8, skip_in=[4], multires=6)
    dev = R.SingleVarianceNetwork(0.3)
    col = R.RenderingNetwork(d_feature=64, mode="no_view_dir", d_in=6, d_out=3, d_hidden=64, n_layers=2,
                             multires_view=4)
    return nerf, sdf, dev, col


def test_flat_adam_state_interchanges_with_torch_adam_over_the_reference_list():
    """exp_runner.py:105-115 builds Adam over nerf + sdf + variance + color; the NeRF parameters never get a
    gradient (n_outside = 0), so torch keeps no state for them and the trained parameters' state keys start at
    len(nerf.parameters()).  FlatAdam built from the same list must read and write exactly that layout."""
    nerf, sdf, dev, col = _ref_layout_modules(0)
    params = list(nerf.parameters()) + list(sdf.parameters()) + list(dev.parameters()) + list(col.parameters())
    n_nerf = len(list(nerf.parameters()))
    trained = params[n_nerf:]
    opt = torch.optim.Adam(params, lr=5e-4)
    g = torch.Generator().manual_seed(1)
    for _ in range(2):
        for p in trained:
            p.grad = torch.randn(p.shape, generator=g)
        opt.step()
    sd = opt.state_dict()
    assert min(sd["state"].keys()) == n_nerf and len(sd["state"]) == len(trained)

    nerf2, sdf2, dev2, col2 = _ref_layout_modules(0)
    params2 = list(nerf2.parameters()) + list(sdf2.parameters()) + list(dev2.parameters()) + list(col2.parameters())
    fa = R.FlatAdam(params2, lr=1e-3)
    fa.load_state_dict(sd)
    assert fa.active == list(range(n_nerf, len(params2))) and fa.step_count == 2
    assert fa.param_groups[0]["lr"] == 5e-4
    out = fa.state_dict()
    assert out["param_groups"][0]["params"] == list(range(len(params2)))
    assert set(out["state"].keys()) == set(sd["state"].keys())
    for i, st in sd["state"].items():
        assert torch.equal(out["state"][i]["exp_avg"], st["exp_avg"]), i
        assert torch.equal(out["state"][i]["exp_avg_sq"], st["exp_avg_sq"]), i
        assert float(out["state"][i]["step"]) == float(st["step"]) == 2.0
    # ... and torch's Adam over the same list accepts FlatAdam's dict and continues from it
    opt3 = torch.optim.Adam(params2, lr=1.0)
    opt3.load_state_dict(out)
    for p in params2[n_nerf:]:
        p.grad = torch.zeros_like(p)
    opt3.step()
    assert opt3.param_groups[0]["lr"] == 5e-4
    assert float(opt3.state_dict()["state"][n_nerf]["step"]) == 3.0

    # mismatching lists are refused with a clear message instead of loading shifted
    with pytest.raises(ValueError, match="same list"):
        R.FlatAdam(params2[n_nerf:]).load_state_dict(sd)
    swapped = list(params2)
    swapped[n_nerf], swapped[n_nerf + 2] = swapped[n_nerf + 2], swapped[n_nerf]
    with pytest.raises(ValueError, match="shape"):
        R.FlatAdam(swapped).load_state_dict(sd)


def test_reference_layout_checkpoint_fixture_loads():
    """tests/golden/ref_ckpt_tiny.pth was written by the REFERENCE's modules and torch.optim.Adam in the reference's
    layout (oracle/gen_golden.py::checkpoint_case).  It must load (weights_only) into the drop-in modules and FlatAdam."""
    from rnb_neus_fork_amd import checkpoint as CK
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", "ref_ckpt_tiny.pth")
    nerf, sdf, dev, col = _ref_layout_modules(123)
    params = list(nerf.parameters()) + list(sdf.parameters()) + list(dev.parameters()) + list(col.parameters())
    fa = R.FlatAdam(params, lr=1.0)
    it = CK.load_checkpoint(path, nerf, sdf, dev, col, fa)
    assert it == 2 and fa.step_count == 2 and fa.param_groups[0]["lr"] == 5e-4
    raw = torch.load(path, weights_only=True)
    assert list(raw["sdf_network_fine"].keys()) == list(sdf.state_dict().keys())
    assert all(torch.equal(raw["sdf_network_fine"][k], v) for k, v in sdf.state_dict().items())
    assert all(torch.equal(raw["nerf"][k], v) for k, v in nerf.state_dict().items())
    n_nerf = len(list(nerf.parameters()))
    assert fa.active == list(range(n_nerf, len(params)))
    st = raw["optimizer"]["state"]
    mine = fa.state_dict()["state"]
    assert set(mine.keys()) == set(st.keys())
    assert all(torch.equal(mine[i]["exp_avg_sq"], st[i]["exp_avg_sq"]) for i in st)


def test_package_synthetic_batches_equal_the_oracle_generator():
    """bench.py's measured leg draws its rays from the package, the CPU baseline from the oracle: same workload."""
    from rnb_neus_fork_amd import synthetic as S
    for kw in (dict(step=0), dict(step=5, warmup=True), dict(step=3, n_lights=2)):
        a = S.synthetic_batch(64, seed=0, **kw)
        b = O.synthetic_batch(64, seed=0, **kw)
        assert a.keys() == b.keys()
        assert all(torch.equal(a[k], b[k]) for k in a)
    n, f = S.near_far_from_sphere(torch.tensor([[0.0, 0.0, -3.0]]), torch.tensor([[0.0, 0.0, 1.0]]))
    assert float(n) == pytest.approx(2.0) and float(f) == pytest.approx(4.0)


def test_cpu_tensors_are_rejected_by_the_train_helpers():
    """No CPU path anywhere: the loss, the flat optimizer and the ray generator refuse host tensors."""
    q = torch.nn.Parameter(torch.zeros(3))
    q.grad = torch.ones(3)
    with pytest.raises(RuntimeError, match="GPU"):
        R.FlatAdam([q]).step()
    with pytest.raises(RuntimeError, match="GPU"):
        R.rnb_loss({"color_fine": torch.zeros(3, 4, 3), "weight_sum": torch.zeros(4, 1),
                    "gradient_error": torch.zeros(())}, torch.zeros(3, 4, 3), torch.zeros(4, 1))
    with pytest.raises(RuntimeError, match="GPU"):
        R.DeviceRays(torch.zeros(1, 1, 2, 2, 3), None, torch.zeros(1, 2, 2, 1), None, None, torch.eye(4)[None],
                     torch.eye(4)[None], "cpu")


def test_x2h_range_report_flags_weights_beyond_the_fp16_scale():
    """NeuSRenderer.x2h_range_report (host-side torch): the geometric init is far inside the operand range of the default
    arithmetic; a weight row scaled to 1e4 is reported with its layer."""
    torch.manual_seed(0)
    sdf = R.SDFNetwork(d_out=257, d_in=3, d_hidden=256, n_layers=8, skip_in=[4], multires=6, bias=0.5, scale=1.0,
                       geometric_init=True, weight_norm=True)
    dev = R.SingleVarianceNetwork(0.3)
    col = R.RenderingNetwork(d_feature=256, mode="no_view_dir", d_in=6, d_out=3, d_hidden=256, n_layers=2,
                             weight_norm=True, multires_view=4, squeeze_out=True)
    ren = R.NeuSRenderer(None, sdf, dev, col, n_samples=64, n_importance=64, n_outside=0, up_sample_steps=4, perturb=1.0)
    rep = ren.x2h_range_report()
    assert rep["ok"] and rep["max_abs_weight"] < 2.0 and rep["limit"] == 255.0
    with torch.no_grad():
        sdf.lin3.weight_g[7] = 1.0e4
    rep = ren.x2h_range_report()
    assert not rep["ok"] and rep["layer"] == "sdf.lin3" and rep["max_abs_weight"] > 255.0

