"""Explicit-math statement of the fine pass (forward AND backward), no autograd.

TEST INFRASTRUCTURE ONLY (same rules as oracle/rnb_oracle.py).  This is the
specification the HIP kernels implement: the reference obtains the SDF normal
and every parameter gradient through autograd (models/fields.py:114-127 with
create_graph=True, exp_runner.py:261); here the same quantities are written as
explicit sweeps so that each kernel stage has a named intermediate that tests
can compare:

  F   forward sweep        a_l = softplus(W_l in_l + b_l)                      (fields.py:82-104)
  R   reverse sweep        gz_l = g_l * D_l, g_{l-1} = gz_l W_l  -> normal     (fields.py:114-127)
  C   albedo MLP           fields.py:177-215
  K   per-ray composite    renderer.py:506-540, :910-914 / :1014-1017
  K'  composite backward
  C'  albedo MLP backward
  RA  adjoint of R         u_{l+1} = (W_l u_l) * D_l,  zR_l = 100 (W_l u_l) gz_l (1-D_l)
  FB  backward of F        zb_{l-1} = (zb_l W_l) * D_{l-1} + zR_{l-1}
  dW  dW_l = gz_l^T u_l + zb_l^T in_l,  db_l = colsum(zb_l)

It is validated against autograd of oracle/rnb_oracle.py in tests/test_explicit_spec.py
(float64: agreement to ~1e-10).
"""
from __future__ import annotations

import math
from typing import Dict

import torch

from . import rnb_oracle as O


def pe_freqs(multires: int, dtype):
    return [float(2.0 ** k) for k in range(multires)]


def pe_forward(x, multires):
    parts = [x]
    for f in pe_freqs(multires, x.dtype):
        parts += [torch.sin(x * f), torch.cos(x * f)]
    return torch.cat(parts, -1)


def pe_jt(x, g_e, multires):
    """J^T g_e  (3-vector per point) for e = pe(x)."""
    n = g_e[:, 0:3].clone()
    for k, f in enumerate(pe_freqs(multires, x.dtype)):
        gs = g_e[:, 3 + 6 * k: 6 + 6 * k]
        gc = g_e[:, 6 + 6 * k: 9 + 6 * k]
        n = n + f * (gs * torch.cos(x * f) - gc * torch.sin(x * f))
    return n


def pe_j(x, nbar, multires):
    """J nbar  (pe-dim vector per point)."""
    parts = [nbar]
    for f in pe_freqs(multires, x.dtype):
        parts += [f * torch.cos(x * f) * nbar, -f * torch.sin(x * f) * nbar]
    return torch.cat(parts, -1)


def softplus100(z):
    return torch.where(z * 100.0 > 20.0, z, torch.log1p(torch.exp(torch.clamp(z * 100.0, max=30.0))) / 100.0)


def eff_weights(p: O.Params, prefix: str, n_lin: int):
    """Weight-norm forward: returns lists W_l, b_l (fields.py:72-74)."""
    Ws, bs = [], []
    for l in range(n_lin):
        if f"{prefix}.lin{l}.weight" in p:
            Ws.append(p[f"{prefix}.lin{l}.weight"])
        else:
            g, v = p[f"{prefix}.lin{l}.weight_g"], p[f"{prefix}.lin{l}.weight_v"]
            Ws.append(v * (g / v.norm(dim=1, keepdim=True)))
        bs.append(p[f"{prefix}.lin{l}.bias"])
    return Ws, bs


def weightnorm_backward(p: O.Params, prefix: str, l: int, dW, db, grads: Dict[str, torch.Tensor]):
    grads[f"{prefix}.lin{l}.bias"] = db
    if f"{prefix}.lin{l}.weight" in p:
        grads[f"{prefix}.lin{l}.weight"] = dW
        return
    g, v = p[f"{prefix}.lin{l}.weight_g"], p[f"{prefix}.lin{l}.weight_v"]
    nrm = v.norm(dim=1, keepdim=True)
    vh = v / nrm
    dg = (dW * vh).sum(dim=1, keepdim=True)
    grads[f"{prefix}.lin{l}.weight_g"] = dg
    grads[f"{prefix}.lin{l}.weight_v"] = (g / nrm) * (dW - vh * dg)


class FinePass:
    """Explicit forward with saved state, then explicit backward."""

    def __init__(self, p: O.Params, mc: O.ModelConf):
        self.p, self.mc = p, mc
        self.sc, self.cc = mc.sdf, mc.color
        self.n_lin = self.sc.n_layers + 1
        self.W, self.b = eff_weights(p, "sdf", self.n_lin)
        self.Wc, self.bc = eff_weights(p, "color", self.cc.n_layers + 1)
        self.skip = self.sc.skip_in[0] if len(self.sc.skip_in) else -1
        self.rs2 = 1.0 / math.sqrt(2.0)

    # ------------------------------------------------------------------ F + R + C
    def forward_points(self, pts, use_color=True):
        sc, n_lin = self.sc, self.n_lin
        x = pts * sc.scale
        e = pe_forward(x, sc.multires) if sc.multires > 0 else x
        self.x, self.e = x, e
        ins, acts = [], []
        h = e
        for l in range(n_lin):
            if l == self.skip:
                h = torch.cat([h, e], 1) * self.rs2
            ins.append(h)
            z = h @ self.W[l].t() + self.b[l]
            if l < n_lin - 1:
                h = softplus100(z)
                acts.append(h)
            else:
                out = z
        self.ins, self.acts = ins, acts
        sdf = out[:, :1] / sc.scale
        feat = out[:, 1:]
        # R: reverse sweep for d sdf / d x ; D_l = sigmoid(100 z_l) = 1 - exp(-100 a_l)
        # (above PyTorch's softplus threshold, 100 z > 20, a == z and the derivative is exactly 1)
        self.D = [torch.where(100.0 * a > 20.0, torch.ones_like(a), -torch.expm1(-100.0 * a)) for a in acts]
        g = self.W[n_lin - 1][0:1, :].expand(pts.shape[0], -1)
        gz = [None] * (n_lin - 1)
        g_e = torch.zeros_like(e)
        for l in range(n_lin - 2, -1, -1):
            gz[l] = g * self.D[l]
            u = gz[l] @ self.W[l]
            if l == self.skip:
                k = u.shape[1] - e.shape[1]
                g_e = g_e + u[:, k:] * self.rs2
                u = u[:, :k] * self.rs2
            g = u
        g_e = g_e + g
        self.gz = gz
        # sdf = out0/scale and x = scale*pts  =>  d sdf/d pts = J^T g_e
        normal = pe_jt(x, g_e, sc.multires) if sc.multires > 0 else g_e
        self.sdf, self.feat, self.normal = sdf, feat, normal
        if use_color:
            self.color_forward(pts, normal, feat)
        return sdf, feat, normal

    def color_forward(self, pts, normal, feat):
        cc = self.cc
        m = cc.multires_view
        assert cc.mode == "no_view_dir"
        cin = torch.cat([pe_forward(pts, m), pe_forward(normal, m), feat], -1)
        self.cin = cin
        h = cin
        self.cacts = []
        nl = cc.n_layers + 1
        for l in range(nl):
            z = h @ self.Wc[l].t() + self.bc[l]
            if l < nl - 1:
                h = torch.relu(z)
                self.cacts.append(h)
        self.albedo = torch.sigmoid(z) if cc.squeeze_out else z
        return self.albedo

    # ------------------------------------------------------------------ K
    def composite(self, rays_d, z_vals, lights_dir, *, cos_anneal_ratio, relu_shading, no_albedo,
                  mvps=True, background_rgb=None):
        B, S = z_vals.shape
        sample_dist = 2.0 / self.mc.render.n_samples
        dists = torch.cat([z_vals[:, 1:] - z_vals[:, :-1], torch.full_like(z_vals[:, :1], sample_dist)], -1)
        self.dists = dists
        n = self.normal.reshape(B, S, 3)
        s = self.sdf.reshape(B, S)
        inv_s = torch.exp(self.p["dev.variance"] * 10.0).clip(1e-6, 1e6)
        self.inv_s = inv_s
        c = cos_anneal_ratio
        tc = (rays_d[:, None, :] * n).sum(-1)
        ic = -(torch.relu(-tc * 0.5 + 0.5) * (1.0 - c) + torch.relu(-tc) * c)
        e_next = s + ic * dists * 0.5
        e_prev = s - ic * dists * 0.5
        pc = torch.sigmoid(e_prev * inv_s)
        nc = torch.sigmoid(e_next * inv_s)
        raw = (pc - nc + 1e-5) / (pc + 1e-5)
        alpha = raw.clip(0.0, 1.0)
        xs = torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha + 1e-7], -1)
        P = torch.cumprod(xs, -1)
        w = alpha * P[:, :-1]
        nn_ = n.norm(dim=-1)
        pts_norm = self.pts.reshape(B, S, 3).norm(dim=-1)
        relax = (pts_norm < 1.2).to(n.dtype)
        # the reference keeps the mask in fp32 (`.float()`), so its count + 1e-5 is an fp32 sum
        self.relax_den = (relax.float().sum() + 1e-5).to(n.dtype)
        gerr = (relax * (nn_ - 1.0) ** 2).sum() / self.relax_den
        alb = self.albedo.reshape(B, S, -1)
        if mvps:
            if no_albedo:
                alb = torch.ones_like(alb)
            sh = (n[None] * lights_dir).sum(-1)                  # [L,B,S]
            sh_raw = sh
            if relu_shading:
                sh = torch.relu(sh)
            color = (alb[None] * (w[None] * sh)[..., None]).sum(2)
        else:
            sh = sh_raw = None
            color = (alb[:, :, :3] * w[:, :, None]).sum(1)
            if background_rgb is not None:
                color = color + background_rgb * (1.0 - w.sum(-1, keepdim=True))
        self.k = dict(tc=tc, ic=ic, e_next=e_next, e_prev=e_prev, pc=pc, nc=nc, raw=raw, alpha=alpha,
                      xs=xs, P=P, w=w, nn=nn_, relax=relax, sh=sh, sh_raw=sh_raw, alb=alb, c=c,
                      relu=relu_shading, no_albedo=no_albedo, mvps=mvps, bg=background_rgb)
        return {"color_fine": color, "weights": w, "weight_sum": w.sum(-1, keepdim=True),
                "weight_max": w.max(-1, keepdim=True)[0], "gradients": n, "gradient_error": gerr,
                "cdf_fine": pc, "s_val": (1.0 / inv_s).expand(B, 1),
                "inside_sphere": (pts_norm < 1.0).to(n.dtype)}

    def forward(self, rays_o, rays_d, z_vals, lights_dir, **kw):
        B, S = z_vals.shape
        sample_dist = 2.0 / self.mc.render.n_samples
        dists = torch.cat([z_vals[:, 1:] - z_vals[:, :-1], torch.full_like(z_vals[:, :1], sample_dist)], -1)
        mid = z_vals + dists * 0.5
        pts = (rays_o[:, None, :] + rays_d[:, None, :] * mid[..., None]).reshape(-1, 3)
        self.pts = pts
        self.rays_d = rays_d
        self.lights = lights_dir
        self.forward_points(pts)
        return self.composite(rays_d, z_vals, lights_dir, **kw)

    # ------------------------------------------------------------------ K'
    def composite_backward(self, gout: Dict[str, torch.Tensor]):
        """gout: grads w.r.t. color_fine, weights, weight_sum, weight_max, gradients, gradient_error,
        cdf_fine, s_val (any subset).  Returns per-point sbar [P], nbar [P,3], albbar [P,C], dvariance."""
        k = self.k
        w, alpha, P, xs = k["w"], k["alpha"], k["P"], k["xs"]
        B, S = w.shape
        dt = w.dtype
        n = self.normal.reshape(B, S, 3)
        alb = k["alb"]
        Cb = gout.get("color_fine")
        wbar = torch.zeros_like(w)
        nbar = torch.zeros_like(n)
        albbar = torch.zeros_like(alb)
        if Cb is not None:
            if k["mvps"]:
                sh = k["sh"]
                wbar = wbar + (Cb[:, :, None, :] * alb[None] * sh[..., None]).sum((0, 3))
                if not k["no_albedo"]:
                    albbar = albbar + (Cb[:, :, None, :] * (w[None] * sh)[..., None]).sum(0)
                shbar = (Cb[:, :, None, :] * alb[None]).sum(-1) * w[None]
                if k["relu"]:
                    shbar = shbar * (k["sh_raw"] > 0).to(dt)
                nbar = nbar + (shbar[..., None] * self.lights.expand(-1, B, S, 3)).sum(0)
            else:
                wbar = wbar + (Cb[:, None, :] * alb[:, :, :3]).sum(-1)
                albbar[:, :, :3] = albbar[:, :, :3] + Cb[:, None, :] * w[:, :, None]
                if k["bg"] is not None:
                    wbar = wbar - (Cb * k["bg"]).sum(-1, keepdim=True)
        if "weights" in gout:
            wbar = wbar + gout["weights"]
        if "weight_sum" in gout:
            wbar = wbar + gout["weight_sum"]
        if "weight_max" in gout:
            idx = w.argmax(-1, keepdim=True)
            wbar = wbar.scatter_add(1, idx, gout["weight_max"])
        # w = alpha * P[:, :-1];  P = cumprod(xs)
        alphabar = wbar * P[:, :-1]
        Pbar = torch.cat([wbar * alpha, torch.zeros_like(w[:, :1])], -1)
        suffix = torch.flip(torch.cumsum(torch.flip(Pbar * P, [-1]), -1), [-1])     # sum_{k>=j}
        xsbar = suffix / xs
        alphabar = alphabar - xsbar[:, 1:]
        rawbar = alphabar * ((k["raw"] >= 0) & (k["raw"] <= 1)).to(dt)
        pc, nc = k["pc"], k["nc"]
        den = pc + 1e-5
        pcbar = rawbar * (1.0 / den - (pc - nc + 1e-5) / (den * den))
        ncbar = -rawbar / den
        if "cdf_fine" in gout:
            pcbar = pcbar + gout["cdf_fine"]
        inv_s = self.inv_s
        epb = pcbar * pc * (1 - pc)
        enb = ncbar * nc * (1 - nc)
        inv_s_bar = (epb * k["e_prev"]).sum() + (enb * k["e_next"]).sum()
        epb = epb * inv_s
        enb = enb * inv_s
        sbar = epb + enb
        icbar = (enb - epb) * self.dists * 0.5
        c, tc = k["c"], k["tc"]
        tcbar = icbar * (0.5 * (1.0 - c) * ((-tc * 0.5 + 0.5) > 0).to(dt) + c * ((-tc) > 0).to(dt))
        nbar = nbar + tcbar[..., None] * self.rays_d[:, None, :]
        if "gradient_error" in gout:
            ge = gout["gradient_error"]
            nn_ = k["nn"]
            coef = ge * k["relax"] * 2.0 * (nn_ - 1.0) / self.relax_den
            nbar = nbar + coef[..., None] * n / nn_.clamp_min(1e-30)[..., None]
        if "gradients" in gout:
            nbar = nbar + gout["gradients"]
        if "s_val" in gout:
            inv_s_bar = inv_s_bar - gout["s_val"].sum() / (inv_s * inv_s)
        v = self.p["dev.variance"]
        raw_inv_s = torch.exp(v * 10.0)
        dvar = inv_s_bar * 10.0 * inv_s * ((raw_inv_s >= 1e-6) & (raw_inv_s <= 1e6)).to(dt)
        return sbar.reshape(-1), nbar.reshape(-1, 3), albbar.reshape(B * S, -1), dvar

    # ------------------------------------------------------------------ C' RA FB dW
    def backward(self, gout: Dict[str, torch.Tensor]):
        sbar, nbar, albbar, dvar = self.composite_backward(gout)
        grads: Dict[str, torch.Tensor] = {"dev.variance": dvar}
        sc, cc, n_lin = self.sc, self.cc, self.n_lin
        self.dbg = {}
        fbar = torch.zeros_like(self.feat)
        if not self.k["no_albedo"]:
            # C': albedo MLP backward
            zb = albbar * self.albedo * (1 - self.albedo) if cc.squeeze_out else albbar
            nl = cc.n_layers + 1
            for l in range(nl - 1, -1, -1):
                inp = self.cin if l == 0 else self.cacts[l - 1]
                dW = zb.t() @ inp
                db = zb.sum(0)
                weightnorm_backward(self.p, "color", l, dW, db, grads)
                inb = zb @ self.Wc[l]
                if l > 0:
                    zb = inb * (self.cacts[l - 1] > 0).to(inb.dtype)
            m = cc.multires_view
            pe_d = 3 * (1 + 2 * m) if m > 0 else 3
            pen_bar = inb[:, pe_d:2 * pe_d]
            fbar = inb[:, 2 * pe_d:]
            nbar = nbar + (pe_jt(self.normal, pen_bar, m) if m > 0 else pen_bar)
        self.dbg["nbar"] = nbar
        self.dbg["fbar"] = fbar
        # RA: adjoint of the reverse sweep (runs in forward layer order)
        ge_bar = pe_j(self.x, nbar, sc.multires) if sc.multires > 0 else nbar
        u = ge_bar
        us, zR = [], []
        for l in range(n_lin - 1):
            if l == self.skip:
                u = torch.cat([u, ge_bar], 1) * self.rs2
            us.append(u)
            gzb = u @ self.W[l].t()
            D = self.D[l]
            zR.append(100.0 * gzb * self.gz[l] * (1.0 - D))
            u = gzb * D
        g_last_bar = u                                   # adjoint of the constant row W_last[0]
        self.dbg["us"], self.dbg["zR"] = us, zR
        # FB: ordinary backward of the forward sweep
        outbar = torch.cat([sbar[:, None] / sc.scale, fbar], 1)
        dW_last = outbar.t() @ self.ins[n_lin - 1]
        dW_last[0] = dW_last[0] + g_last_bar.sum(0)
        weightnorm_backward(self.p, "sdf", n_lin - 1, dW_last, outbar.sum(0), grads)
        ab = outbar @ self.W[n_lin - 1]
        zbs = [None] * (n_lin - 1)
        for l in range(n_lin - 2, -1, -1):
            zb = ab * self.D[l] + zR[l]
            zbs[l] = zb
            dW = zb.t() @ self.ins[l] + self.gz[l].t() @ us[l]
            weightnorm_backward(self.p, "sdf", l, dW, zb.sum(0), grads)
            if l > 0:
                ab = zb @ self.W[l]
                if l == self.skip:
                    ab = ab[:, : ab.shape[1] - self.e.shape[1]] * self.rs2
        self.dbg["zb"] = zbs
        return grads
