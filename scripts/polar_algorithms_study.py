"""Study (CPU only): candidate polar-factor algorithms for the GPU kernel on real X_l of an oracle run
(scripts/polar_capture.py), compared the way the parity tests compare them: through V_l U_l against numpy's
SVD-based u @ vh (fxs_Projections.py:752-767)."""
import sys
import numpy as np

f = np.load(sys.argv[1] if len(sys.argv) > 1 else '/tmp/polar_capture_cfg3.npz')
steps = list(f['steps'])
orders = sorted({int(k.split('_l')[1]) for k in f.files if k.startswith('V_l')})


def qdwh_params(l):
    l2 = l * l
    dd = (4 * (1 - l2) / (l2 * l2)) ** (1 / 3)
    sq = np.sqrt(1 + dd)
    a = sq + 0.5 * np.sqrt(8 - 4 * dd + 8 * (2 - l2) / (l2 * sq))
    b = (a - 1) ** 2 / 4
    c = a + b - 1
    return a, b, c


def qdwh(A, l0=1e-16, cmax=np.inf, max_it=30, qr_above=100.0, tol=1e-15):
    """QDWH; QR-based iteration while c > qr_above, Cholesky-based otherwise.  cmax caps c (and thus a, b) so that
    a Cholesky-only variant stays well conditioned (more iterations)."""
    m, n = A.shape
    alpha = np.linalg.norm(A, 'fro')
    X = A / alpha
    l = l0
    nqr = nch = 0
    for it in range(max_it):
        a, b, c = qdwh_params(min(l, 1 - 1e-16)) if l < 1 else (3.0, 1.0, 3.0)
        if c > cmax:
            # capped weights: choose a from c = a + (a-1)^2/4 - 1 = cmax
            a = 2 * np.sqrt(cmax + 1) - 1    # solves (a+1)^2/4 = cmax + 1
            b = (a - 1) ** 2 / 4
            c = a + b - 1
        if c > qr_above:
            Q, _ = np.linalg.qr(np.vstack([np.sqrt(c) * X, np.eye(n)]))
            Xn = (b / c) * X + (1 / np.sqrt(c)) * (a - b / c) * (Q[:m] @ Q[m:].conj().T)
            nqr += 1
        else:
            Z = np.eye(n) + c * (X.conj().T @ X)
            W = np.linalg.cholesky(Z)              # Z = W W^H
            T = np.linalg.solve(W, X.conj().T)     # W^-1 X^H
            T = np.linalg.solve(W.conj().T, T)     # Z^-1 X^H
            Xn = (b / c) * X + (a - b / c) * T.conj().T
            nch += 1
        l = l * (a + b * l * l) / (1 + c * l * l)
        d = np.linalg.norm(Xn - X, 'fro') / np.linalg.norm(Xn, 'fro')
        X = Xn
        if d < tol or (l >= 1 - 1e-15 and d < 1e-8):
            break
    return X, (nqr, nch)


def newton(A, max_it=30, tol=1e-15):
    """scaled Newton with LU inverse, (1, inf)-norm scaling (Higham)."""
    X = A.copy()
    its = 0
    for its in range(1, max_it + 1):
        Xi = np.linalg.inv(X)
        g = ((np.linalg.norm(Xi, 1) * np.linalg.norm(Xi, np.inf)) / (np.linalg.norm(X, 1) * np.linalg.norm(X, np.inf))) ** 0.25
        Xn = 0.5 * (g * X + Xi.conj().T / g)
        d = np.linalg.norm(Xn - X, 'fro') / np.linalg.norm(Xn, 'fro')
        X = Xn
        if d < tol:
            break
    return X, its


def newton_fixed(A, lo=1e-16, n_it=9):
    """Newton with the a-priori optimal (Byers-Xu style) scaling from a lower bound on sigma_min/alpha: no norms."""
    alpha = np.linalg.norm(A, 'fro')
    X = A / alpha
    a, b = lo, 1.0
    for it in range(n_it):
        mu = 1.0 / np.sqrt(a * b)
        Xi = np.linalg.inv(X)
        X = 0.5 * (mu * X + Xi.conj().T / mu)
        # interval of singular values after the step
        hi = 0.5 * (mu * b + 1 / (mu * b))
        a, b = 1.0, hi
        if it == 0:
            a = 1.0
    return X, n_it


algs = {
    'qdwh QR>100': lambda A: qdwh(A, 1e-16),
    'qdwh chol cap1e2': lambda A: qdwh(A, 1e-16, cmax=1e2, qr_above=np.inf),
    'qdwh chol cap1e4': lambda A: qdwh(A, 1e-16, cmax=1e4, qr_above=np.inf),
    'qdwh chol cap1e6': lambda A: qdwh(A, 1e-16, cmax=1e6, qr_above=np.inf),
    'qdwh chol nocap': lambda A: qdwh(A, 1e-16, qr_above=np.inf),
    'newton LU': newton,
}
worst = {k: 0.0 for k in algs}
print('step   l    cond  rank | ' + ' | '.join('%-18s' % k for k in algs))
for st in steps:
    for l in orders:
        key = 'X_s%d_l%d' % (st, l)
        if key not in f.files or l == 0:
            continue
        X = f[key]
        V = f['V_l%d' % l]
        if not np.any(V):
            continue
        s = np.linalg.svd(X, compute_uv=False)
        u, _, vh = np.linalg.svd(X, full_matrices=False)
        ref = V @ (u @ vh)
        row = []
        for name, fn in algs.items():
            try:
                U, its = fn(X)
                d = np.linalg.norm(V @ U - ref) / np.linalg.norm(ref)
            except Exception as e:      # singular matrix etc.
                d, its = np.inf, type(e).__name__
            worst[name] = max(worst[name], d if l in (8, 16, 24, 32) or True else 0)
            row.append('%8.1e %-9s' % (d, its))
        if l in (2, 8, 20, 32):
            print('%4d %3d %8.1e %2d/%2d | ' % (st, l, s[0] / max(s[-1], 1e-300), int((s > 1e-15 * s[0]).sum()), len(s)) + ' | '.join(row))
print('worst |V U - V U_svd| / |V U_svd|:')
for k, v in worst.items():
    print('   %-20s %.2e' % (k, v))
