"""Development aid: accuracy of RNB_VARIANT_X3 (fp32 products as six bf16 MFMA terms) against the native fp32 MFMA path,
both measured against an fp64 evaluation of the same network.  Run on the GPU box from the repo root."""
import sys
import torch
sys.path.insert(0, ".")
from oracle import rnb_oracle as O
import rnb_neus_fork_amd as pkg

dev = torch.device("cuda:0")
mc = O.ModelConf()
torch.manual_seed(1)
p = O.init_params(mc)
sdf, devn, col, ren = pkg.build_from_named_params(mc, p, dev)
res = 48
bmin, bmax = [-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]
xs = torch.linspace(-1, 1, res, dtype=torch.float64)
pts = torch.stack(torch.meshgrid(xs, xs, xs, indexing="ij"), -1).reshape(-1, 3)
p64 = {k: v.double() for k, v in p.items()}
with torch.no_grad():
    ref = -O.sdf_only(p64, mc.sdf, pts).reshape(res, res, res)
out = {}
for tag, kw in (("f32", dict(f32_mfma=True)), ("x3", dict(x3=True)), ("p3", dict(x3=True, fwd_ti=2, fwd_nw=8))):
    ren.set_variant(**kw)
    u = ren.extract_fields(bmin, bmax, res)
    out[tag] = torch.as_tensor(u).double()
    e = (out[tag] - ref).abs()
    print(f"{tag}: max abs err vs fp64 {e.max():.3e}  rms {e.pow(2).mean().sqrt():.3e}   (|sdf| max {ref.abs().max():.3f})")
print("x3 vs f32 max abs", float((out["x3"] - out["f32"]).abs().max()), " p3 vs x3 max abs", float((out["p3"] - out["x3"]).abs().max()))
