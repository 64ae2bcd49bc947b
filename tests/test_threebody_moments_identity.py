"""CPU check of the identity behind the three-body moment kernels (csrc/m3g_threebody.hip, k_threebody_moments): for the
complete partner list of one centre atom the Legendre-weighted sums over partners (reference: nn/interaction.py:187-217,
353-365) equal a contraction of the row's own direction with per-atom moments, minus the row's own term -- forward sums and the
direction gradients the reverse pass needs.  fp64 numpy, no GPU."""
import numpy as np


def _legendre(x):
    return np.stack([np.ones_like(x), x, 1.5 * x * x - 0.5], -1), np.stack([np.zeros_like(x), np.ones_like(x), 3.0 * x], -1)


def test_moment_contraction_equals_the_partner_sums():
    rng = np.random.default_rng(0)
    n, R = 13, 3                      # 13 active edges of one centre, n_max = 3, l_max = 3
    u = rng.normal(size=(n, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    g = rng.normal(size=(n, 3, R))    # payload g[e, l, n] = q v[dst]
    ds = rng.normal(size=(n, 3, R))   # incoming gradient rows dS = fc dm

    # ---- the list walk: every ordered pair e1 != e2
    cos = u @ u.T
    P, dP = _legendre(cos)            # [e1, e2, l]
    off = 1.0 - np.eye(n)
    S_list = np.einsum("ab,abl,bln->aln", off, P, g)                       # forward aggregate per row (before fc, ynorm)
    T_list = np.einsum("ab,abl,aln->bln", off, P, ds)                      # dL/dg[e2] = sum_e1 dS[e1] P(u1.u2)
    # gradient of  sum_{e1 != e2} dS[e1] . P(u1.u2) g[e2]  with respect to the direction of row r (as first and as second edge)
    A_list = np.einsum("ab,abl,aln,bln,bx->ax", off, dP, ds, g, u) + np.einsum("ab,abl,aln,bln,ax->bx", off, dP, ds, g, u)

    # ---- the moments (one set per payload)
    def moments(p):
        return (p[:, 0].sum(0), np.einsum("ex,en->nx", u, p[:, 1]), np.einsum("ex,ey,en->nxy", u, u, p[:, 2]), p[:, 2].sum(0))

    def contract(M, own, r):
        M0, M1, M2, M2s = M
        ur, s = u[r], u[r] @ u[r]
        S0 = M0 - own[0]
        S1 = M1 @ ur - s * own[1]
        G2u = M2 @ ur                                  # [n, 3]
        S2 = 1.5 * (G2u @ ur - s * s * own[2]) - 0.5 * (M2s - own[2])
        V1 = M1 - np.outer(own[1], ur)                 # sum over the OTHER rows of P1'(u.u') u' p'
        V2 = 3.0 * (G2u - s * np.outer(own[2], ur))
        return np.stack([S0, S1, S2]), V1, V2

    MG, MH = moments(g), moments(ds)
    for r in range(n):
        Sg, V1g, V2g = contract(MG, g[r], r)
        Sh, V1h, V2h = contract(MH, ds[r], r)
        np.testing.assert_allclose(Sg, S_list[r], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(Sh, T_list[r], rtol=1e-12, atol=1e-12)
        A = np.einsum("n,nx->x", ds[r, 1], V1g) + np.einsum("n,nx->x", g[r, 1], V1h) \
            + np.einsum("n,nx->x", ds[r, 2], V2g) + np.einsum("n,nx->x", g[r, 2], V2h)
        np.testing.assert_allclose(A, A_list[r], rtol=1e-11, atol=1e-11)
