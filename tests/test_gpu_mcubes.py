"""Device marching cubes (csrc/mcubes.hip, through the C ABI) against the numpy oracle (oracle/mc_oracle.py) and the
properties an iso-surface must have.  Parity with PyMCubes itself is UNPINNED (not importable here, no reference mesh
fixture) — see oracle/mc_oracle.py; the oracle and the device share the derived tables and the output order, so the
comparison below is EXACT (integer triangles bit-equal, double vertices bit-equal)."""
import numpy as np
import pytest
import torch

from oracle import mc_oracle as M
from oracle import rnb_oracle as O
from tests.test_mc_oracle import _check_on_surface, _sphere, _torus

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def R():
    import rnb_neus_fork_amd as pkg
    pkg.native.load()
    return pkg


def _device(R, u, thr=0.0):
    v, t = R.marching_cubes(torch.from_numpy(u).cuda(), thr)
    torch.cuda.synchronize()
    return v.cpu().numpy(), t.cpu().numpy()


@pytest.mark.parametrize("shape,chi,n", [("sphere", 2, 40), ("torus", 0, 40), ("sphere", 2, 131)])
def test_analytic_volumes_match_the_oracle_and_are_closed_manifolds(R, shape, chi, n):
    u = -(_sphere(n) if shape == "sphere" else _torus(n))
    v, t = _device(R, u)
    vo, to = M.marching_cubes(u, 0.0)
    assert t.dtype == np.int32 and v.dtype == np.float64
    assert np.array_equal(t, to), "triangles: bit-equal to the oracle (same tables, same order)"
    assert np.array_equal(v, vo), "vertices: bit-equal (double interpolation of the same two fp32 samples)"
    V, E, F, euler, closed = M.mesh_report(v, t)
    assert closed and euler == chi and V == len(v)
    _check_on_surface(u, v, 0.0)


def test_noise_volume_non_cubic_grid_and_threshold(R):
    rng = np.random.default_rng(5)
    u = rng.standard_normal((37, 21, 50)).astype(np.float32)         # every case, ambiguous faces, ragged tiles
    u[0], u[-1], u[:, 0], u[:, -1], u[:, :, 0], u[:, :, -1] = 2, 2, 2, 2, 2, 2
    for thr in (0.0, 0.37):
        v, t = _device(R, u, thr)
        vo, to = M.marching_cubes(u, thr)
        assert np.array_equal(t, to) and np.array_equal(v, vo)
        assert M.mesh_report(v, t)[4], "closed, consistently oriented manifold on all 256 cases"
    # open surfaces (the iso-surface leaves the grid) still agree with the oracle
    u2 = rng.standard_normal((9, 8, 7)).astype(np.float32)
    v, t = _device(R, u2)
    vo, to = M.marching_cubes(u2, 0.0)
    assert np.array_equal(t, to) and np.array_equal(v, vo)


def test_empty_full_and_nan_volumes(R):
    for val in (1.0, -1.0):
        v, t = _device(R, np.full((5, 6, 7), val, dtype=np.float32))
        assert v.shape == (0, 3) and t.shape == (0, 3)
    u = -_sphere(24)
    u[3, 4, 5] = np.nan                                   # NaN counts as outside: no crossing against outside neighbours
    v, t = _device(R, u)
    vo, to = M.marching_cubes(u, 0.0)
    assert np.array_equal(t, to) and np.array_equal(v, vo, equal_nan=True)
    with pytest.raises(RuntimeError):
        R.marching_cubes(torch.zeros(4, 4, 4))            # CPU tensor: no CPU path
    with pytest.raises(ValueError):
        R.marching_cubes(torch.zeros(4, 4, device="cuda"))


def test_extract_geometry_native_on_the_geometric_init(R):
    """validate_mesh's path end to end (exp_runner.py:561-581): SDF grid of the geometric-init network (a sphere of
    radius ~0.5) -> native marching cubes -> bounding-box rescale; the mesh is a closed genus-0 surface at that radius,
    and equals the oracle run on the very same volume."""
    mc = O.ModelConf()
    torch.manual_seed(0)
    p = O.init_params(mc)
    sdf, dev, col, ren = R.build_from_named_params(mc, p, torch.device("cuda:0"))
    lo, hi = torch.tensor([-1.01, -1.01, -1.01]), torch.tensor([1.01, 1.01, 1.01])
    res = 96
    verts, tris = ren.extract_geometry(lo, hi, res, threshold=0.0, backend="native")
    assert verts.dtype == np.float64 and verts.shape[1] == 3
    u = ren.extract_fields(lo, hi, res)
    vo, to = M.marching_cubes(u, 0.0)
    vo = vo / (res - 1.0) * (hi - lo).numpy()[None] + lo.numpy()[None]
    assert np.array_equal(tris, to) and np.array_equal(verts, vo)
    V, E, F, euler, closed = M.mesh_report(verts, tris)
    assert closed and euler == 2
    r = np.linalg.norm(verts, axis=1)
    assert 0.3 < r.min() and r.max() < 0.7          # the geometric init is a rough sphere of radius ~0.5
    # orientation: outward (u = -sdf decreases outwards)
    pts = verts[tris.astype(np.int64)]
    nrm = np.cross(pts[:, 1] - pts[:, 0], pts[:, 2] - pts[:, 0])
    assert ((nrm * pts.mean(axis=1)).sum(-1) > 0).mean() > 0.999
    with pytest.raises(ImportError):
        ren.extract_geometry(lo, hi, 8, backend="mcubes")      # PyMCubes is not installed on the GPU box
