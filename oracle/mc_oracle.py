"""CPU oracle of the device marching cubes (rnb-neus-fork_amd/csrc/mcubes.hip).

TEST INFRASTRUCTURE ONLY (imported by tests/ alone).  numpy restatement of the algorithm behind
`mcubes.marching_cubes(u, threshold)` as the reference calls it (models/renderer.py:31).

Parity status: UNPINNED.  PyMCubes (pinned `PyMCubes==0.1.6` at README.md:36 of the reference) is a third-party C++
extension, not vendored by the reference and not importable in this image, and the reference holds no mesh fixture.
What is restated from its published algorithm: classic corner / edge numbering, inside := value <= isovalue, one
shared vertex per crossed grid edge, linear interpolation in double, grid-index coordinates.  The case tables are the
derived ones of tools/gen_mc_tables.py (face-consistent, hence watertight), not PyMCubes' literal table: ambiguous
configurations may be triangulated differently, quads may be split along the other diagonal; the surface as a set
of crossed edges — the vertices — is the same.  The checks available are therefore properties on analytic volumes
(tests/test_mc_oracle.py) and exact agreement of the device kernels with this restatement (tests/test_gpu_mcubes.py).
Output order is the device's: vertices by owning grid point (x slowest) then axis, triangles by cell then table."""
from __future__ import annotations

import os
import sys

import numpy as np

_TOOLS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
if _TOOLS not in sys.path:
    sys.path.insert(0, _TOOLS)
import gen_mc_tables as G  # noqa: E402

_TRI = G.tables()
_OWNER = [G.edge_owner(e) for e in range(12)]


def marching_cubes(u: np.ndarray, threshold: float = 0.0):
    u = np.ascontiguousarray(u, dtype=np.float32)
    nx, ny, nz = u.shape
    iso32 = np.float32(threshold)
    inside = u <= iso32
    # ---- vertices: one per crossed grid edge, owned by the edge's lower end point ------------------
    act = np.zeros((nx, ny, nz, 3), dtype=bool)
    act[:-1, :, :, 0] = inside[:-1] != inside[1:]
    act[:, :-1, :, 1] = inside[:, :-1] != inside[:, 1:]
    act[:, :, :-1, 2] = inside[:, :, :-1] != inside[:, :, 1:]
    vid = np.cumsum(act.reshape(-1)).reshape(act.shape) - 1          # id of the vertex on (point, axis) where active
    xs, ys, zs, ax = np.nonzero(act)                                   # C order == (point, axis) order
    f0 = u[xs, ys, zs].astype(np.float64)
    f1 = u[xs + (ax == 0), ys + (ax == 1), zs + (ax == 2)].astype(np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        t = (np.float64(iso32) - f0) / (f1 - f0)
    verts = np.stack([xs, ys, zs], -1).astype(np.float64)
    verts[np.arange(len(t)), ax] += t
    # ---- triangles: per cell in C order, table order inside the cell ------------------------------
    c = inside
    cube = (c[:-1, :-1, :-1].astype(np.int32) | (c[1:, :-1, :-1] << 1) | (c[1:, 1:, :-1] << 2) | (c[:-1, 1:, :-1] << 3)
            | (c[:-1, :-1, 1:] << 4) | (c[1:, :-1, 1:] << 5) | (c[1:, 1:, 1:] << 6) | (c[:-1, 1:, 1:] << 7))
    ntab = np.array([len(t_) for t_ in _TRI])
    cx, cy, cz = np.nonzero(ntab[cube] > 0)
    tris = []
    order = []
    cases = cube[cx, cy, cz]
    cell_rank = np.arange(len(cases))
    for case in np.unique(cases):
        sel = cases == case
        x, y, z = cx[sel], cy[sel], cz[sel]
        for k, tri in enumerate(_TRI[case]):
            ids = []
            for e in tri:
                (dx, dy, dz), axis = _OWNER[e]
                ids.append(vid[x + dx, y + dy, z + dz, axis])
            tris.append(np.stack(ids, -1))
            order.append(cell_rank[sel] * 8 + k)
    if tris:
        tris = np.concatenate(tris)
        tris = tris[np.argsort(np.concatenate(order), kind="stable")]
    else:
        tris = np.zeros((0, 3), dtype=np.int64)
    return verts, tris.astype(np.int32)


def mesh_report(verts, tris):
    """Topology of a triangle mesh: (#vertices used, #edges, #faces, Euler characteristic, closed-manifold flag:
    every undirected edge in exactly two triangles with opposite directions)."""
    t = np.asarray(tris, dtype=np.int64)
    e = np.concatenate([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]])
    und = np.sort(e, axis=1)
    key = und[:, 0] * (int(t.max()) + 1 if len(t) else 1) + und[:, 1]
    uniq, cnt = np.unique(key, return_counts=True)
    dirsum = np.zeros(len(uniq), dtype=np.int64)
    np.add.at(dirsum, np.searchsorted(uniq, key), np.where(e[:, 0] < e[:, 1], 1, -1))
    closed = bool((cnt == 2).all() and (dirsum == 0).all())
    V = len(np.unique(t))
    return V, len(uniq), len(t), V - len(uniq) + len(t), closed
