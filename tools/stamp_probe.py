"""Development probe (GPU box, with the library built from tools/patches/r05_stamp_instrumentation.diff and -DRNB_STAMP=1 in place
of the in-tree one): where a layer of the forward-only sweep spends its clocks — matrix loop, wait at the barrier behind it,
epilogue, wait at the barrier behind that — per wave, with one and with two workgroups per CU, and how the two workgroups of a
CU sit relative to each other.   usage: python tools/stamp_probe.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import rnb_neus_fork_amd as R  # noqa: E402
from rnb_neus_fork_amd import native, runtime  # noqa: E402
from tests.golden_util import Golden  # noqa: E402

NB, NW, NL = 4096, 4, 10


def stamps(lib):
    st = np.zeros(NB * NW * NL * 4, dtype=np.int64)
    hw = np.zeros(NB, dtype=np.uint32)
    rc = lib.rnb_debug_stamps(st.ctypes.data_as(C.c_void_p), hw.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return st.reshape(NB, NW, NL, 4), hw


def main():
    dev = torch.device("cuda:0")
    g = Golden("full_main_b512")
    sdf, devn, col, ren = R.build_from_named_params(g.mc, g.params(), dev)
    ren.set_variant(fwd_ti=2)
    packed = ren._pack(True)
    lib = native.load()
    lib.rnb_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
    lib.rnb_debug_stamps.restype = C.c_int
    big = (torch.rand(1 << 18, 3, device=dev) * 2 - 1) * 0.9
    for tiles in (128, 256, 512, 1024):
        pts = big[: tiles * 64].contiguous()
        for _ in range(3):
            runtime.sdf_forward(ren.desc, packed, pts, False)
        torch.cuda.synchronize()
        st, hw = stamps(lib)
        st = st[:tiles]
        hw = hw[:tiles]
        L = slice(1, 7)                       # layers 1 .. 6 (K = 256, a following layer exists)
        M = (st[:, :, L, 1] - st[:, :, L, 0]).astype(np.float64)
        B1 = (st[:, :, L, 2] - st[:, :, L, 1]).astype(np.float64)
        E = (st[:, :, L, 3] - st[:, :, L, 2]).astype(np.float64)
        nxt = st[:, :, 2:8, 0]
        B2 = (nxt - st[:, :, L, 3]).astype(np.float64)
        tot = (st[:, :, 2:8, 0] - st[:, :, L, 0]).astype(np.float64)
        print(f"{tiles:5d} tiles: per layer and wave (clocks, mean over layers 1..6): matrix loop {M.mean():8.0f}  barrier {B1.mean():7.0f}  "
              f"epilogue {E.mean():8.0f}  barrier {B2.mean():7.0f}  | layer {tot.mean():8.0f}   (p10 / p90 of the layer: "
              f"{np.percentile(tot, 10):.0f} / {np.percentile(tot, 90):.0f})", flush=True)
        # the CU's partner: workgroups with the same (XCC, SE, SH, CU) id whose lifetimes overlap
        order = {}
        for b in range(tiles):
            order.setdefault(int(hw[b]), []).append(b)
        both_m, both_e, mixed, n = 0.0, 0.0, 0.0, 0
        for ids in order.values():
            for i in range(len(ids)):
                for j in range(i + 1, len(ids)):
                    a, b = st[ids[i], 0], st[ids[j], 0]          # wave 0 of each: [layer][4]
                    lo, hi = max(a[1, 0], b[1, 0]), min(a[7, 0], b[7, 0])
                    if hi - lo < 20000:
                        continue
                    # sample the common window: which phase is each workgroup in?
                    ts = np.linspace(lo, hi, 400)

                    def phase(s, t):
                        l = np.searchsorted(s[1:8, 0], t, side="right")          # index into layers 1..7
                        l = np.clip(l, 1, 7)
                        row = s[l]
                        return np.where(t < row[:, 1], 0, np.where(t < row[:, 2], 2, np.where(t < row[:, 3], 1, 2)))   # 0 M, 1 E, 2 barrier
                    pa, pb = phase(a, ts), phase(b, ts)
                    both_m += np.mean((pa == 0) & (pb == 0))
                    both_e += np.mean((pa == 1) & (pb == 1))
                    mixed += np.mean(((pa == 0) & (pb == 1)) | ((pa == 1) & (pb == 0)))
                    n += 1
        if n:
            print(f"        {n} co-resident pairs: both in the matrix loop {both_m / n:.2f} of the common time, both in the epilogue {both_e / n:.2f}, "
                  f"one in each {mixed / n:.2f}", flush=True)
        print(f"        distinct CU ids {len(order)}, workgroups per id: " + str(sorted({len(v) for v in order.values()})), flush=True)


if __name__ == "__main__":
    main()
