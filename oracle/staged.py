"""ORACLE (part 2) -- the path restated in the STAGED, AUTOGRAD-FREE form the HIP kernels use.
TEST INFRASTRUCTURE ONLY (see oracle/m3gnet_oracle.py header for who may import this).

`m3gnet_oracle.energy_forces` restates the reference module by module and differentiates with
autograd.  The engine instead runs the algebraically restructured pipeline below with a hand-derived
reverse pass (DESIGN.md "Stages").  This file spells that pipeline out in plain torch so that
  (1) tests/test_staged_oracle.py proves on the CPU (fp64) that it equals the module-by-module oracle,
  (2) GPU parity tests can compare every engine stage buffer with the tensor of the same name here.

Restructuring facts used (all follow from the reference formulas):
  * chi_ln(d_ik) fc(d_ik) is a per-EDGE quantity q[e,c]; per triplet only Y_l(cos) remains
    (nn/interaction.py:188-210).
  * W1 [x_i | x_j | e] = W1a x_i + W1b x_j + W1c e: the x parts are per-NODE tables TA/TB
    (nn/conv.py:91-97 concat followed by nn/core.py Linear).
  * cos(theta) = u_ij . u_ik with u = r/|r| (nn/invariant.py:37).
Index c = l*n_max + n everywhere (nn/interaction.py:206).
"""
from __future__ import annotations

import math

import torch

from .m3gnet_oracle import OracleConfig, OracleConstants, cutoff_function, radial_basis


# Matrix-product hook for the edge-block products (the ones the MFMA kernels execute); tests/checkers/split_precision_study.py
# swaps it for an emulation of split-precision MFMA arithmetic.  Default: plain matmul.
def MM(a, b):
    return a @ b


def _silu(p):
    return p * torch.sigmoid(p)


def _dsilu(p):
    s = torch.sigmoid(p)
    return s * (1 + p * (1 - s))


def _sph_j_and_derivative(x, l_max):
    """j_l(x), j_l'(x) for l < l_max with the reference's small-x branch (nn/interaction.py:293-348)."""
    big = x > 1e-8
    s, c = torch.sin(x) / x, torch.cos(x)
    j = [torch.where(big, s, torch.ones_like(x))]
    jn1 = torch.where(big, (s - c) / x, x / 3)  # j_1 (needed for j_0')
    seq = [j[0], jn1]
    dfact = 3
    for n in range(1, l_max):
        dfact *= 2 * n + 3
        seq.append(torch.where(big, (2 * n + 1) / x * seq[n] - seq[n - 1], x / dfact))
    js = seq[:l_max]
    djs = []
    for l in range(l_max):
        if l == 0:
            djs.append(torch.where(big, -seq[1], torch.zeros_like(x)))
        else:
            below = torch.full_like(x, 1.0 / 3) if l == 1 else torch.zeros_like(x)
            djs.append(torch.where(big, seq[l - 1] - (l + 1) / x * seq[l], below))
    return js, djs


def _legendre_and_derivative(x, l_max):
    p, dp = [torch.ones_like(x)], [torch.zeros_like(x)]
    if l_max > 1:
        p.append(x)
        dp.append(torch.ones_like(x))
    for n in range(1, l_max - 1):
        p.append(((2 * n + 1) * x * p[n] - n * p[n - 1]) / (n + 1))
        dp.append(((2 * n + 1) * (p[n] + x * dp[n]) - n * dp[n - 1]) / (n + 1))
    return p, dp


def split_conv_weights(p, prefix, D):
    """Split the two GatedMLPs of one M3GNetConv into the per-stage matrices.
    Returns dict m -> {w1a,w1b,w1c [2D,D] (rows: dense then gate), b1 [2D], w2d,w2g [D,D], b2d,b2g, wl [D,R]}."""
    out = {}
    for m, mlp, lin in (("e", "concat_edge_update", "edge_linear"), ("n", "concat_node_update", "node_linear")):
        w1 = torch.cat([p[f"{prefix}.{mlp}.dense.0.weight"], p[f"{prefix}.{mlp}.gate.0.weight"]], dim=0)  # [2D, 3D]
        out[m] = dict(
            w1a=w1[:, :D], w1b=w1[:, D:2 * D], w1c=w1[:, 2 * D:],
            b1=torch.cat([p[f"{prefix}.{mlp}.dense.0.bias"], p[f"{prefix}.{mlp}.gate.0.bias"]]),
            w2d=p[f"{prefix}.{mlp}.dense.2.weight"], b2d=p[f"{prefix}.{mlp}.dense.2.bias"],
            w2g=p[f"{prefix}.{mlp}.gate.2.weight"], b2g=p[f"{prefix}.{mlp}.gate.2.bias"],
            wl=p[f"{prefix}.{lin}.weight"],
        )
    return out


def _mlp2_forward(table_sum, e_in, w, h):
    """One conv GatedMLP given the gathered layer-1 table contribution.  Returns output and saved pre-acts."""
    D = e_in.size(1)
    p1 = table_sum + MM(e_in, w["w1c"].T)  # [E, 2D] (bias already in TA)
    hd, hg = _silu(p1[:, :D]), _silu(p1[:, D:])
    p2d = MM(hd, w["w2d"].T) + w["b2d"]
    p2g = MM(hg, w["w2g"].T) + w["b2g"]
    out = _silu(p2d) * torch.sigmoid(p2g)
    s = h @ w["wl"].T
    return out * s, dict(p1=p1, p2d=p2d, p2g=p2g, out=out, s=s)


def _mlp2_backward(d_upd, sv, w, D):
    """Reverse of `_mlp2_forward`: returns (d_e_in contribution, d_p1 [E,2D], d_h [E,R])."""
    d_out = d_upd * sv["s"]
    d_s = d_upd * sv["out"]
    d_h = d_s @ w["wl"]
    sg = torch.sigmoid(sv["p2g"])
    d_p2d = d_out * sg * _dsilu(sv["p2d"])
    d_p2g = d_out * _silu(sv["p2d"]) * sg * (1 - sg)
    d_hd, d_hg = MM(d_p2d, w["w2d"]), MM(d_p2g, w["w2g"])
    p1 = sv["p1"]
    d_p1 = torch.cat([d_hd * _dsilu(p1[:, :D]), d_hg * _dsilu(p1[:, D:])], dim=1)
    return MM(d_p1, w["w1c"]), d_p1, d_h


def forward_backward(p: dict, cfg: OracleConfig, c: OracleConstants, graph: dict) -> dict:
    """Energy + analytic forces.  Returns all stage buffers (names = DESIGN.md / csrc workspace names)."""
    dt = p["model.3.linear.weight"].dtype
    D, R, L, C, B = cfg.embedding_dim, cfg.n_max, cfg.l_max, cfg.l_max * cfg.n_max, cfg.num_blocks
    pos, lattice = graph["pos"].to(dt), graph["lattice"].to(dt)
    batch, ei, tei = graph["batch"].long(), graph["edge_index"].long(), graph["triplet_edge_index"].long()
    shift, types = graph["edge_cell_shift"].to(dt), graph["atom_types"].long()
    src, dst = ei[0], ei[1]
    t1, t2 = tei[0], tei[1]
    N, E, S = pos.size(0), src.numel(), lattice.size(0)
    rc, rc3, ls = cfg.scaled_cutoff, cfg.scaled_threebody_cutoff, cfg.length_scale
    st = {}

    # ---- S0 geometry + bases (per edge) -----------------------------------------------------
    spos, slat = pos / ls, lattice / ls
    rvec = spos[dst] + torch.einsum("ep,epa->ea", shift, slat[batch[src]]) - spos[src]
    d = torch.sqrt((rvec * rvec).sum(1))
    u = rvec / d[:, None]
    h = radial_basis(d, cfg, c)  # [E,R]
    # dh/dd: d/dd sinc_t(a d) = (cos(pi a d) - sinc_t(a d)) / d,  sinc_t = torch.sinc
    iota = torch.arange(R)
    a1 = ((iota[:, None] + 1) * torch.pi / rc).to(dt)
    a2 = ((iota[:, None] + 2) * torch.pi / rc).to(dt)
    dfm = c.coeff[:, None] * ((torch.cos(torch.pi * a1 * d) - torch.sinc(a1 * d)) + (torch.cos(torch.pi * a2 * d) - torch.sinc(a2 * d))) / d
    hp = [dfm[0]]
    for m in range(1, R):
        hp.append((dfm[m] + torch.sqrt(c.em[m] / c.dm[m - 1]) * hp[m - 1]) / torch.sqrt(c.dm[m]))
    hp = torch.stack(hp, dim=1)
    fc3 = cutoff_function(d, rc3)
    rho = d / rc3
    fc3p = torch.where(rho <= 1, (-30 * rho**4 + 60 * rho**3 - 30 * rho**2) / rc3, torch.zeros_like(d))
    zl = c.zeros[:L, :R].to(dt)  # [L,R]
    xarg = zl[:, :, None] * d[None, None, :] / rc  # [L,R,E]
    js, djs = _sph_j_and_derivative(xarg.reshape(L * R, E), L)  # evaluate all orders on all args
    chi = torch.stack([js[l].reshape(L, R, E)[l] for l in range(L)]) / c.factors[:, :, None]
    dchi = torch.stack([djs[l].reshape(L, R, E)[l] for l in range(L)]) * (zl[:, :, None] / rc) / c.factors[:, :, None]
    q = (chi * fc3).reshape(C, E).T.contiguous()  # [E,C]
    qp = (dchi * fc3 + chi * fc3p).reshape(C, E).T.contiguous()
    st.update(rvec=rvec, d=d, u=u, h=h, hp=hp, fc3=fc3, fc3p=fc3p, q=q, qp=qp)

    # ---- S1 embeddings ----------------------------------------------------------------------
    x = p["model.3.linear.weight"].T[types]  # row gather == one_hot @ W^T
    w_adj = p["model.5.linear.weight"]  # [D,R]
    pe0 = h @ w_adj.T
    e = _silu(pe0)
    st.update(x0=x, e0=e)

    # per-triplet geometry (depends only on u)
    cos_raw = (u[t1] * u[t2]).sum(1)
    cosang = torch.clamp(cos_raw, -1, 1)
    inside = ((cos_raw >= -1) & (cos_raw <= 1)).to(dt)
    P, dP = _legendre_and_derivative(cosang, L)
    ynorm = [math.sqrt((2 * l + 1) / (4.0 * math.pi)) for l in range(L)]
    Y = torch.stack([ynorm[l] * P[l] for l in range(L)], dim=1)  # [T,L]
    dY = torch.stack([ynorm[l] * dP[l] for l in range(L)], dim=1)
    Yc = Y.repeat_interleave(R, dim=1)  # [T,C]
    dYc = dY.repeat_interleave(R, dim=1)
    st.update(triplet_angles=cosang)

    saved = []
    for b in range(B):
        tb, cv = f"model.{6 + 2 * b}", f"model.{7 + 2 * b}"
        w = split_conv_weights(p, cv, D)
        # ---- S2 node pre-pass -------------------------------------------------------------
        w1, b1 = p[f"{tb}.linear_sigmoid1.weight"], p[f"{tb}.linear_sigmoid1.bias"]
        v = torch.sigmoid(x @ w1.T + b1)  # [N,C]
        TA = torch.cat([x @ w["e"]["w1a"].T + w["e"]["b1"], x @ w["n"]["w1a"].T + w["n"]["b1"]], dim=1)  # [N,4D]
        TB = torch.cat([x @ w["e"]["w1b"].T, x @ w["n"]["w1b"].T], dim=1)
        # ---- S3 three-body aggregate ---------------------------------------------------------
        g = q * v[dst]  # [E,C]
        Ssum = torch.zeros(E, C, dtype=dt).index_add(0, t1, Yc * g[t2])
        m = fc3[:, None] * Ssum
        # ---- S4 edge block -----------------------------------------------------------------
        wd, wg = p[f"{tb}.gated_mlp.dense.0.weight"], p[f"{tb}.gated_mlp.gate.0.weight"]  # [D,C]
        pd, pg = MM(m, wd.T), MM(m, wg.T)
        e1 = e + _silu(pd) * torch.sigmoid(pg)
        tab = TA[src] + TB[dst]
        upd_e, sv_e = _mlp2_forward(tab[:, : 2 * D], e1, w["e"], h)
        e2 = e1 + upd_e
        msg, sv_n = _mlp2_forward(tab[:, 2 * D:], e2, w["n"], h)
        x_new = x + torch.zeros(N, D, dtype=dt).index_add(0, src, msg)
        saved.append(dict(x=x, v=v, g=g, Ssum=Ssum, m=m, pd=pd, pg=pg, sv_e=sv_e, sv_n=sv_n, w=w, wd=wd, wg=wg, w1=w1))
        st[f"v_{b}"], st[f"TA_{b}"], st[f"TB_{b}"], st[f"m_{b}"] = v, TA, TB, m
        st[f"e1_{b}"], st[f"e2_{b}"], st[f"x_{b}"] = e1, e2, x_new
        x, e = x_new, e2

    # ---- S5 readout ---------------------------------------------------------------------------
    ro = f"model.{6 + 2 * B}.gated"
    Wd = [p[f"{ro}.dense.{2 * i}.weight"] for i in range(3)]
    Bd = [p[f"{ro}.dense.{2 * i}.bias"] for i in range(3)]
    Wg = [p[f"{ro}.gate.{2 * i}.weight"] for i in range(3)]
    Bg = [p[f"{ro}.gate.{2 * i}.bias"] for i in range(3)]
    pd1, pg1 = x @ Wd[0].T + Bd[0], x @ Wg[0].T + Bg[0]
    hd1, hg1 = _silu(pd1), _silu(pg1)
    pd2, pg2 = hd1 @ Wd[1].T + Bd[1], hg1 @ Wg[1].T + Bg[1]
    hd2, hg2 = _silu(pd2), _silu(pg2)
    od, og = hd2 @ Wd[2].T + Bd[2], hg2 @ Wg[2].T + Bg[2]  # [N,1]
    sg = torch.sigmoid(og)
    eps = (od * sg)[:, 0]
    scaled_atomic = c.elemental_energies[types] / cfg.energy_scale + eps
    scaled_total = torch.zeros(S, dtype=dt).index_add(0, batch, scaled_atomic)
    total = cfg.energy_scale * scaled_total
    st.update(x=x, edge_attr=e, scaled_atomic_energies=scaled_atomic, scaled_total_energy=scaled_total, total_energy=total)

    # ================= reverse pass: dL/d* for L = sum(total) =================================
    d_eps = torch.full((N, 1), cfg.energy_scale, dtype=dt)
    d_od, d_og = d_eps * sg, d_eps * od * sg * (1 - sg)
    d_hd2, d_hg2 = d_od @ Wd[2], d_og @ Wg[2]
    d_pd2, d_pg2 = d_hd2 * _dsilu(pd2), d_hg2 * _dsilu(pg2)
    d_hd1, d_hg1 = d_pd2 @ Wd[1], d_pg2 @ Wg[1]
    d_pd1, d_pg1 = d_hd1 * _dsilu(pd1), d_hg1 * _dsilu(pg1)
    d_x = d_pd1 @ Wd[0] + d_pg1 @ Wg[0]  # [N,D]
    st["dx_readout"] = d_x
    d_e = torch.zeros(E, D, dtype=dt)
    d_h = torch.zeros(E, R, dtype=dt)
    d_d = torch.zeros(E, dtype=dt)
    d_u = torch.zeros(E, 3, dtype=dt)
    for b in reversed(range(B)):
        sv = saved[b]
        w = sv["w"]
        # ---- B4 edge block reverse ---------------------------------------------------------
        d_msg = d_x[src]
        de_n, dp1_n, dh_n = _mlp2_backward(d_msg, sv["sv_n"], w["n"], D)
        d_e2 = d_e + de_n
        de_e, dp1_e, dh_e = _mlp2_backward(d_e2, sv["sv_e"], w["e"], D)
        d_e1 = d_e2 + de_e
        d_h = d_h + dh_n + dh_e
        sgg = torch.sigmoid(sv["pg"])
        d_pd = d_e1 * sgg * _dsilu(sv["pd"])
        d_pg = d_e1 * _silu(sv["pd"]) * sgg * (1 - sgg)
        d_m = MM(d_pd, sv["wd"]) + MM(d_pg, sv["wg"])  # [E,C]
        d_e = d_e1
        dp1 = torch.cat([dp1_e, dp1_n], dim=1)  # [E,4D]
        d_TA = torch.zeros(N, 4 * D, dtype=dt).index_add(0, src, dp1)
        d_TB = torch.zeros(N, 4 * D, dtype=dt).index_add(0, dst, dp1)
        # ---- B3 three-body reverse ---------------------------------------------------------
        d_fc1 = (d_m * sv["Ssum"]).sum(1)
        d_d = d_d + fc3p * d_fc1
        d_S = fc3[:, None] * d_m  # [E,C]
        g = sv["g"]
        d_cos = (d_S[t1] * dYc * g[t2]).sum(1) * inside  # [T]
        d_u = d_u.index_add(0, t1, d_cos[:, None] * u[t2]).index_add(0, t2, d_cos[:, None] * u[t1])
        d_g = torch.zeros(E, C, dtype=dt).index_add(0, t2, d_S[t1] * Yc)
        v = sv["v"]
        d_d = d_d + (d_g * v[dst] * qp).sum(1)
        d_v = torch.zeros(N, C, dtype=dt).index_add(0, dst, d_g * q)
        # ---- B2 node reverse -----------------------------------------------------------------
        d_x = (d_x
               + d_TA[:, : 2 * D] @ w["e"]["w1a"] + d_TA[:, 2 * D:] @ w["n"]["w1a"]
               + d_TB[:, : 2 * D] @ w["e"]["w1b"] + d_TB[:, 2 * D:] @ w["n"]["w1b"]
               + (d_v * v * (1 - v)) @ sv["w1"])
        st[f"dm_{b}"], st[f"dx_in_{b}"], st[f"de_in_{b}"] = d_m, d_x, d_e
    # ---- B1 edge embedding reverse -----------------------------------------------------------
    d_h = d_h + (d_e * _dsilu(pe0)) @ w_adj
    # ---- B0 geometry reverse -----------------------------------------------------------------
    d_d = d_d + (d_h * hp).sum(1)
    d_r = d_d[:, None] * u + (d_u - (d_u * u).sum(1, keepdim=True) * u) / d[:, None]
    grad_spos = torch.zeros(N, 3, dtype=dt).index_add(0, dst, d_r).index_add(0, src, -d_r)
    forces = -grad_spos / ls
    vir = torch.zeros(S, 3, 3, dtype=dt).index_add(0, batch, pos[:, :, None] * forces[:, None, :])
    vol = torch.abs(torch.sum(lattice[:, 0] * torch.linalg.cross(lattice[:, 1], lattice[:, 2]), dim=1))
    voigt = torch.stack([vir[:, 0, 0], vir[:, 1, 1], vir[:, 2, 2], vir[:, 1, 2], vir[:, 2, 0], vir[:, 0, 1]], dim=1)
    st.update(dh=d_h, dd=d_d, du=d_u, dr=d_r, forces=forces, stresses=voigt / vol[:, None])
    return st
