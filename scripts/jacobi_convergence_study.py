"""Study (GPU box): convergence of the one-sided Jacobi polar factor on real X_l = I_l^+ D^2 V_l matrices taken from a
running reconstruction -- per-sweep max relative off-diagonal for cold / warm / sorted / QR-preconditioned starts."""
import sys, os
import numpy as np
import scipy.linalg as sla
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
np.seterr(all='ignore')
from xframe_amd.fxs import hostsetup as hs, synthetic as S
from xframe_amd.fxs.engine import Engine

cfg = 3
N, L = S._SIZES[cfg]
ed = Engine({'grid': {'n_radial_points': N, 'max_order': L}}, None, n_batch=1, max_q=S.data_cutoff(N))
data, _ = S.make_invariants(ed, N, L); ed.close()
e = Engine(S.config_overrides(cfg), data, n_batch=1, fused=True)
e.set_density(0, hs.bump_density(e.rs, e.shape, S.PARTICLE_RADIUS, 0.3, 2, np.random.default_rng(1000),
                                 e.rsetup.integrated_intensity, e.int_wr, e.int_wt))
e.init_state()
ramp = hs.ExponentialRamp(0.5, 0.4, -1 / 250, 500)


def pairs_of_round(r, Cp):
    M = Cp - 1
    a = [r]; b = [M]
    for pi in range(1, Cp // 2):
        a.append((r + pi) % M); b.append((r - pi) % M)
    return np.array(a), np.array(b)


def jacobi(X, V=None, max_sweeps=20, tol=1e-14):
    X = X.copy(); n, k = X.shape
    V = np.eye(k, dtype=complex) if V is None else V.copy()
    Cp = k + (k & 1)
    hist = []
    for sweep in range(max_sweeps):
        m = 0.0
        for r in range(Cp - 1):
            I, J = pairs_of_round(r, Cp)
            ok = (I < k) & (J < k); I, J = I[ok], J[ok]
            A, Bc = X[:, I], X[:, J]
            al = (abs(A) ** 2).sum(0); be = (abs(Bc) ** 2).sum(0); ga = (A.conj() * Bc).sum(0)
            g2 = abs(ga) ** 2
            rot = (g2 > tol * tol * al * be) & (g2 > 0)
            if not rot.any():
                continue
            m = max(m, (g2[rot] / (al[rot] * be[rot])).max())
            g = np.sqrt(g2[rot]); I, J = I[rot], J[rot]
            zeta = 0.5 * (be[rot] - al[rot]) / g
            t = np.sign(zeta + (zeta == 0)) / (abs(zeta) + np.sqrt(1 + zeta ** 2))
            cs = 1 / np.sqrt(1 + t * t); sn = cs * t
            em = ga[rot].conj() / g
            for M_ in (X, V):
                a = M_[:, I].copy(); bj = em * M_[:, J]
                M_[:, I] = cs * a - sn * bj
                M_[:, J] = sn * a + cs * bj
        hist.append(np.sqrt(m))
        if m <= 1e-16:
            break
    return X, V, hist


def fmt(h):
    return ' '.join('%.0e' % x for x in h)


step = 0
Vprev = {}
for target in (5, 6, 30, 31, 59, 60):
    while step < target:
        e.run('HIO', True, [ramp.eval(step)]); step += 1
    rho = e.density(0)
    F = e.fourier_transform(rho)[0]
    Ilm = e.sht_forward(np.abs(F) ** 2 + 0j)[0]
    for l in (32, 16):
        Vl = e.rsetup.projection_matrices[l]
        Il = Ilm[:, l * l:(l + 1) ** 2]
        X = Il.conj().T @ (e.qs[:, None] ** 2 * Vl)
        sv = np.linalg.svd(X, compute_uv=False)
        print('step', step, 'l', l, 'shape', X.shape, 'sigma ratio min/max %.1e' % (sv[-1] / sv[0]), 'median %.1e' % (np.median(sv) / sv[0]))
        _, Vc, h = jacobi(X)
        print('   cold            ', len(h), fmt(h))
        if l in Vprev:
            _, Vw, h = jacobi(X @ Vprev[l], Vprev[l])
            print('   warm            ', len(h), fmt(h))
            Xw = X @ Vprev[l]
            order = np.argsort(-(abs(Xw) ** 2).sum(0))
            _, _, h = jacobi(Xw[:, order])
            print('   warm+sorted     ', len(h), fmt(h))
            Vprev[l] = Vw
        else:
            Vprev[l] = Vc
        order = np.argsort(-(abs(X) ** 2).sum(0))
        _, _, h = jacobi(X[:, order])
        print('   cold sorted     ', len(h), fmt(h))
        Q, R, P = sla.qr(X, pivoting=True, mode='economic')
        _, _, h = jacobi(R.conj().T)
        print('   QRP, jacobi(R^H)', len(h), fmt(h))
        Q2, R2 = np.linalg.qr(R.conj().T)
        _, _, h = jacobi(R2.conj().T)
        print('   QRP+QR, jac(R2^H)', len(h), fmt(h))
