"""Marching cubes, CPU side: the derived case tables (tools/gen_mc_tables.py) and the numpy oracle
(oracle/mc_oracle.py) on analytic volumes.  Parity with PyMCubes is UNPINNED (not importable; the reference holds no
mesh fixture): what can be checked are properties — every vertex on the iso-surface along its grid edge, a closed,
consistently oriented 2-manifold, the Euler characteristic of the analytic shape, outward orientation."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_mc_tables as G  # noqa: E402
from oracle import mc_oracle as M  # noqa: E402


def test_committed_tables_are_the_generated_ones():
    with open(G.INC_PATH) as f:
        assert f.read() == G.render_inc(G.tables()), "rerun tools/gen_mc_tables.py"


def test_every_case_is_face_consistent_and_covers_its_crossed_edges():
    tri = G.tables()
    for case in range(256):
        active = {e for e, (a, b) in enumerate(G.EDGES) if ((case >> a) ^ (case >> b)) & 1}
        used = {e for t in tri[case] for e in t}
        assert used == active, case
        # the boundary of the cell's patch (directed edges used once) is exactly the face segments, as oriented
        d = {}
        for a, b, c in tri[case]:
            for p, q in ((a, b), (b, c), (c, a)):
                d[(p, q)] = d.get((p, q), 0) + 1
        boundary = {k for k, n in d.items() if (k[1], k[0]) not in d}
        assert all(n == 1 for n in d.values()), case
        assert boundary == set(G.face_segments(case)), case
    assert max(len(t) for t in tri) == 5
    assert [len(t) for t in tri][:9] == [0, 1, 1, 2, 1, 2, 2, 3, 1]      # the classic table's counts
    assert tri[1] == [(0, 8, 3)] and tri[2] == [(0, 1, 9)] and tri[4] == [(1, 2, 10)]   # ... and its first rows


def _grid(n, lo=-1.0, hi=1.0):
    ax = np.linspace(lo, hi, n, dtype=np.float32)
    return np.meshgrid(ax, ax, ax, indexing="ij")


def _sphere(n, r=0.6, c=(0.05, -0.02, 0.03)):
    x, y, z = _grid(n)
    return (np.sqrt((x - c[0]) ** 2 + (y - c[1]) ** 2 + (z - c[2]) ** 2) - r).astype(np.float32)


def _torus(n, R=0.55, r=0.22):
    x, y, z = _grid(n)
    return (np.sqrt((np.sqrt(x * x + y * y) - R) ** 2 + z * z) - r).astype(np.float32)


def _check_on_surface(u, verts, threshold):
    """Each vertex lies on ONE grid edge, where the linear interpolant of the two samples equals the threshold."""
    base = np.floor(verts + 1e-12).astype(np.int64)
    frac = verts - base
    axis = np.argmax(frac, axis=1)
    assert ((frac > 0).sum(axis=1) <= 1).all(), "a vertex moves along one axis only"
    i0 = tuple(base.T)
    nb = base.copy()
    nb[np.arange(len(nb)), axis] += 1
    nb = np.minimum(nb, np.array(u.shape) - 1)
    f0, f1 = u[i0].astype(np.float64), u[tuple(nb.T)].astype(np.float64)
    t = frac[np.arange(len(frac)), axis]
    val = f0 + t * (f1 - f0)
    assert np.abs(val - threshold).max() < 1e-6


@pytest.mark.parametrize("shape,chi", [("sphere", 2), ("torus", 0)])
def test_oracle_on_analytic_volumes(shape, chi):
    n = 40
    u = -(_sphere(n) if shape == "sphere" else _torus(n))          # the reference meshes u = -sdf at threshold 0
    verts, tris = M.marching_cubes(u, 0.0)
    assert len(verts) > 500 and tris.max() == len(verts) - 1
    V, E, F, euler, closed = M.mesh_report(verts, tris)
    assert closed, "every edge must be shared by exactly two triangles, traversed in opposite directions"
    assert V == len(verts) and euler == chi
    _check_on_surface(u, verts, 0.0)
    # orientation: the right-hand normal points towards smaller u = towards growing SDF = out of the object
    p = verts[tris.astype(np.int64)]
    nrm = np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0])
    cen = p.mean(axis=1) / (n - 1) * 2.0 - 1.0
    if shape == "sphere":
        out = cen - np.array([0.05, -0.02, 0.03])
    else:
        rho = np.sqrt(cen[:, 0] ** 2 + cen[:, 1] ** 2)
        ring = np.stack([cen[:, 0] / rho * 0.55, cen[:, 1] / rho * 0.55, np.zeros_like(rho)], -1)
        out = cen - ring
    big = np.linalg.norm(nrm, axis=1) > 1e-9
    assert ((nrm[big] * out[big]).sum(-1) > 0).all()
    # signed volume of the closed mesh against the analytic one
    vol = (p[:, 0] * np.cross(p[:, 1], p[:, 2])).sum() / 6.0 * (2.0 / (n - 1)) ** 3
    exact = 4.0 / 3.0 * np.pi * 0.6 ** 3 if shape == "sphere" else 2 * np.pi ** 2 * 0.55 * 0.22 ** 2
    assert vol == pytest.approx(exact, rel=2e-2)


def test_oracle_watertight_on_random_volumes():
    """Noise volumes exercise all 256 cases, ambiguous faces included: still a closed, oriented manifold (the volume
    is padded with 'outside' so the surface cannot leave the grid)."""
    rng = np.random.default_rng(3)
    u = rng.standard_normal((14, 13, 12)).astype(np.float32)
    u[0], u[-1], u[:, 0], u[:, -1], u[:, :, 0], u[:, :, -1] = 1, 1, 1, 1, 1, 1
    verts, tris = M.marching_cubes(u, 0.0)
    _, _, _, _, closed = M.mesh_report(verts, tris)
    assert closed
    _check_on_surface(u, verts, 0.0)
    # non-cubic grid and a non-zero threshold on a ramp: one plane of quads
    x = np.broadcast_to(np.arange(5, dtype=np.float32)[:, None, None], (5, 4, 3)).copy()
    v, t = M.marching_cubes(x, 1.25)
    assert len(v) == 12 and np.allclose(v[:, 0], 1.25) and len(t) == 2 * 3 * 2


def test_empty_and_full_volumes():
    for val in (1.0, -1.0):
        v, t = M.marching_cubes(np.full((4, 4, 4), val, dtype=np.float32), 0.0)
        assert v.shape == (0, 3) and t.shape == (0, 3)
