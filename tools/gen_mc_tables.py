"""Generator of the marching-cubes case tables used by csrc/mcubes.hip and oracle/mc_oracle.py.

The reference calls PyMCubes (`mcubes.marching_cubes`, models/renderer.py:31), a third-party C++ extension that is
neither vendored nor importable here: parity with it is UNPINNED.  The tables are therefore not typed in
from memory but derived, so that their one essential property — a watertight, consistently oriented surface — holds
by construction and is checked by tests:

  * corner / edge numbering of the classic algorithm (corner m at ((m&1)^((m>>1)&1), (m>>1)&1, m>>2), i.e.
    0:(0,0,0) 1:(1,0,0) 2:(1,1,0) 3:(0,1,0) 4..7 the same at z = 1; edges 0-3 bottom ring, 4-7 top ring, 8-11 verticals);
  * bit m of the case index is set when corner m is "set" (value <= isovalue);
  * on every cube face the crossing points are joined by segments that depend only on that face's four corner
    bits: 2 crossings -> one segment; 4 crossings (diagonal corners alike) -> each SET corner is cut off by its own
    segment.  Two cells sharing a face see the same bits, draw the same segments: no cracks, in any configuration;
  * every segment is oriented with the set region on its left seen from outside the cube, so the segments chain
    into closed loops whose right-hand normal points towards the set corners (towards smaller values);
  * every loop is triangulated as a fan; the apex is the one that creates the fewest diagonals between two points
    of the same cube face (such a diagonal could coincide with the neighbour's).

`python tools/gen_mc_tables.py` rewrites rnb-neus-fork_amd/csrc/mc_tables.inc; tests/test_mc_tables.py checks the
committed file against this generator and the tables' properties for all 256 cases.
"""
from __future__ import annotations

import itertools
import os

CORNERS = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]
EDGES = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]
# faces as corner cycles, counter-clockwise seen from OUTSIDE the cube, with their outward normals
FACES = [((0, 3, 2, 1), (0, 0, -1)), ((4, 5, 6, 7), (0, 0, 1)), ((0, 1, 5, 4), (0, -1, 0)),
         ((3, 7, 6, 2), (0, 1, 0)), ((0, 4, 7, 3), (-1, 0, 0)), ((1, 2, 6, 5), (1, 0, 0))]


def _edge_id(a, b):
    for e, (p, q) in enumerate(EDGES):
        if (p, q) == (a, b) or (p, q) == (b, a):
            return e
    raise KeyError((a, b))


def _mid(e):
    a, b = EDGES[e]
    return tuple((CORNERS[a][d] + CORNERS[b][d]) / 2.0 for d in range(3))


def _sub(a, b):
    return tuple(x - y for x, y in zip(a, b))


def _cross(a, b):
    return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


def _dot(a, b):
    return sum(x * y for x, y in zip(a, b))


def _faces_of_edge(e):
    a, b = EDGES[e]
    return {f for f, (cyc, _) in enumerate(FACES) if a in cyc and b in cyc}


EDGE_FACES = [_faces_of_edge(e) for e in range(12)]


def face_segments(case):
    """Oriented segments (from_edge, to_edge) of one case, face by face."""
    segs = []
    for cyc, normal in FACES:
        bits = [(case >> c) & 1 for c in cyc]
        crossing = [i for i in range(4) if bits[i] != bits[(i + 1) % 4]]      # face edge i joins cyc[i], cyc[i+1]
        if not crossing:
            continue
        eid = [_edge_id(cyc[i], cyc[(i + 1) % 4]) for i in range(4)]
        pairs = []
        if len(crossing) == 2:
            s = next(cyc[i] for i in range(4) if bits[i])                     # any set corner: all lie on one side
            pairs.append((eid[crossing[0]], eid[crossing[1]], s))
        else:                                                                 # 4 crossings: cut off every set corner
            for i in range(4):
                if bits[i]:
                    pairs.append((eid[(i - 1) % 4], eid[i], cyc[i]))
        for e1, e2, s in pairs:
            p, q = _mid(e1), _mid(e2)
            left = _dot(normal, _cross(_sub(q, p), _sub(CORNERS[s], p)))
            assert left != 0
            segs.append((e1, e2) if left > 0 else (e2, e1))
    return segs


def loops_of(case):
    segs = face_segments(case)
    nxt = {}
    for a, b in segs:
        assert a not in nxt, "two outgoing segments at one crossing: orientation rule violated"
        nxt[a] = b
    active = {e for e, (a, b) in enumerate(EDGES) if ((case >> a) ^ (case >> b)) & 1}
    assert set(nxt) == active and set(nxt.values()) == active
    loops, seen = [], set()
    for start in sorted(active):
        if start in seen:
            continue
        loop, e = [], start
        while e not in seen:
            seen.add(e)
            loop.append(e)
            e = nxt[e]
        assert e == start
        loops.append(loop)
    return loops


def triangulate(loop):
    n = len(loop)
    best = None
    for r in range(n):
        rot = loop[r:] + loop[:r]
        cost = sum(1 for i in range(2, n - 1) if EDGE_FACES[rot[0]] & EDGE_FACES[rot[i]])
        if best is None or cost < best[0]:
            best = (cost, rot)
    rot = best[1]
    return [(rot[0], rot[i], rot[i + 1]) for i in range(1, n - 1)]


def edge_owner(e):
    """(offset of the owning grid point from the cell's corner 0, axis): the edge runs from that point in +axis."""
    a, b = (CORNERS[c] for c in EDGES[e])
    lo = tuple(min(x, y) for x, y in zip(a, b))
    axis = next(d for d in range(3) if a[d] != b[d])
    return lo, axis


def tables():
    tri = [[t for loop in loops_of(c) for t in triangulate(loop)] for c in range(256)]
    return tri


def render_inc(tri):
    max_t = max(len(t) for t in tri)
    lines = ["// GENERATED by tools/gen_mc_tables.py — do not edit.  Marching-cubes case tables (see the generator for",
             "// the construction: face-consistent segments, set region on the left, fan triangulation).",
             "// RNB_MC_TABLE is the storage qualifier the including file wants (e.g. `static __constant__ const`).",
             f"#define RNB_MC_MAX_TRIS {max_t}",
             "// the grid point that owns cube edge e (offset from the cell's corner 0) and the edge's axis (0 x, 1 y, 2 z)",
             "RNB_MC_TABLE unsigned char kMcEdgeOwner[12][3] = {"
             + ", ".join("{%d, %d, %d}" % edge_owner(e)[0] for e in range(12)) + "};",
             "RNB_MC_TABLE unsigned char kMcEdgeAxis[12] = {" + ", ".join(str(edge_owner(e)[1]) for e in range(12)) + "};",
             "// triangles per case",
             "RNB_MC_TABLE unsigned char kMcNumTris[256] = {"]
    for r in range(0, 256, 32):
        lines.append("  " + ", ".join(str(len(tri[c])) for c in range(r, r + 32)) + ",")
    lines.append("};")
    lines.append("// cube edges of the triangles' corners, 3 per triangle, padded with 255")
    lines.append(f"RNB_MC_TABLE unsigned char kMcTriEdges[256][{3 * max_t}] = {{")
    for c in range(256):
        flat = [e for t in tri[c] for e in t]
        flat += [255] * (3 * max_t - len(flat))
        lines.append("  {" + ", ".join(f"{e:3d}" for e in flat) + "},")
    lines.append("};")
    return "\n".join(lines) + "\n"


INC_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rnb-neus-fork_amd", "csrc",
                        "mc_tables.inc")

if __name__ == "__main__":
    t = tables()
    with open(INC_PATH, "w") as f:
        f.write(render_inc(t))
    print("wrote", INC_PATH, "max triangles per cell:", max(len(x) for x in t),
          "total triangles over the 256 cases:", sum(len(x) for x in t))
