"""Numerics of the upstream `extract` step: B_l -> eigenpairs -> projection matrices V_l, and the
positive-semidefinite projection of B_l.  Host mirror of

    xframe/projects/fxs/projectLibrary/fxs_invariant_tools.py:1079-1207   deg2_invariant_to_projection_matrices(_3d),
                                                                           deg2_invariant_eigenvalues
    xframe/library/mathLibrary.py:872-892                                  nearest_positive_semidefinite_matrix
    xframe/projects/fxs/extract.py:418-430                                 apply_invariant_constraints (bl_enforce_psd)

with the eigen-decompositions on the device (``Engine.hermitian_eig``: real symmetric matrices up to 128 x 128 in the LDS-
resident solver k_sym_eig, anything else in the general Hermitian kernel); sorting, cuts and clipping are the reference's rules.
All orders of a B_l stack go to the device in one batch (the reference spreads them over worker processes, 1106)."""
import numpy as np


def _batched_eigh(engine, mats):
    """eigenvalues ASCENDING (K, n) and eigenvectors (K, n, n) in columns, like numpy.linalg.eigh, from the device solver"""
    if engine is None or not hasattr(engine, 'hermitian_eig'):
        raise TypeError('the extract rules need an eigensolver object with `hermitian_eig` (an Engine: the device solvers); there is no '
                        'CPU fallback in the product')
    vals, vecs = engine.hermitian_eig(np.asarray(mats))          # descending
    vecs = vecs[:, :, ::-1]
    if not np.iscomplexobj(np.asarray(mats)):
        vecs = vecs.real
    return vals[:, ::-1].copy(), np.ascontiguousarray(vecs)


def deg2_invariant_eigenvalues(engine, b_matrices, sort_mode=0):
    """fxs_invariant_tools.py:1114-1141 for a stack (K, n, n): returns (eigenvalues (K, n), eigenvectors (K, n, n)) sorted in
    descending order of the eigenvalue (sort_mode 0) or of median_q |sqrt|lambda| v(q)| sign(lambda) (sort_mode 1); a matrix that
    is zero to numpy.isclose gets zero eigenpairs (1123-1130)."""
    b = np.asarray(b_matrices)
    b = (b + np.conj(np.swapaxes(b, -1, -2))) / 2                                  # 1122
    K, n = b.shape[0], b.shape[1]
    if n == 0:
        return np.zeros((K, 0)), np.zeros((K, 0, 0))
    zero = np.array([np.isclose(m, 0).all() for m in b])
    w = np.zeros((K, n))
    v = np.zeros((K, n, n), dtype=b.dtype)
    if (~zero).any():
        w[~zero], v[~zero] = _batched_eigh(engine, b[~zero])
    signs = np.sign(w)
    if sort_mode == 0:
        metric = w
    else:
        metric = np.median(np.abs(np.sqrt(np.abs(w[:, None, :])) * v), axis=1) * signs
    ids = np.argsort(metric, axis=1)[:, ::-1]
    return np.take_along_axis(w, ids, axis=1).real, np.take_along_axis(v, ids[:, None, :], axis=2)


def deg2_invariant_to_projection_matrices(engine, b_coeff, q_id_limits=False, sort_mode=0):
    """fxs_invariant_tools.py:1079-1112 (dim 3) with 1171-1207 per order: V_l = the first min(block, 2l+1) eigenvectors of the
    block q_id_limits[l, 0] of B_l times sqrt(eigenvalue), negative eigenvalues zeroed, embedded in Nq rows.  Returns
    (tuple of V_l (complex, like 1207), tuple of eigenvalues)."""
    b_coeff = np.asarray(b_coeff)
    n_orders, nq = b_coeff.shape[0], b_coeff.shape[-1]
    if isinstance(q_id_limits, bool):                                              # 1092-1094
        lim = np.zeros((n_orders, 2, 2), dtype=int)
        lim[..., 1] = nq
    else:
        lim = np.array(q_id_limits)
    if not (lim[:, 0, :] == lim[:, 1, :]).all():                                   # 1095-1099
        lim[:, 1] = lim[:, 0]
    # orders with the same block go to the device together
    pairs = [None] * n_orders
    blocks = {}
    for o in range(n_orders):
        blocks.setdefault((int(lim[o, 0, 0]), int(lim[o, 0, 1])), []).append(o)
    for (lo, hi), orders in blocks.items():
        sub = b_coeff[orders][:, lo:hi, lo:hi]
        w, v = deg2_invariant_eigenvalues(engine, sub, sort_mode)
        for i, o in enumerate(orders):
            pairs[o] = (w[i], v[i], lo, hi)
    pms, evs = [], []
    for o in range(n_orders):
        w, v, lo, hi = pairs[o]
        n_full = min(nq, 2 * o + 1)
        full_v = np.zeros((nq, n_full), dtype=v.dtype)
        full_w = np.zeros(n_full)
        if len(w) != 0:
            n = min(len(v), 2 * o + 1)
            vv, ww = v[:, :n].copy(), w[:n].copy()
            neg = ww < 0
            ww[neg] = 0
            vv[:, neg] = 0
            full_v[lo:hi, :n] = vv
            full_w[:n] = ww
        pms.append((full_v * np.sqrt(full_w)[None, :]).astype(complex))
        evs.append(full_w)
    return tuple(pms), tuple(evs)


def nearest_positive_semidefinite_matrix(engine, A, low_positive_eigenvalues_to_zero=False):
    """mathLibrary.py:872-892 for one matrix or a stack: eigenvalues of the Hermitian part below the limit set to zero.  The
    eigen-decomposition runs on the device; the noise-floor variant needs the spectrum of the unsymmetrised A (numpy, host)."""
    A = np.asarray(A)
    single = A.ndim == 2
    stack = A[None] if single else A.reshape((-1,) + A.shape[-2:])
    B = (stack + np.conj(np.swapaxes(stack, -1, -2))) / 2
    w, v = _batched_eigh(engine, B)
    out = np.empty_like(B)
    for i in range(len(B)):
        limit = 0
        if low_positive_eigenvalues_to_zero:
            limit = np.abs(np.min(np.linalg.eig(stack[i])[0]))
        wi = w[i].copy()
        wi[wi < limit] = 0
        out[i] = (v[i] * wi[None, :]) @ np.conj(v[i]).T
    return out[0] if single else out.reshape(A.shape)


def apply_invariant_constraints(engine, b_coeff, q_id_limits, bl_enforce_psd=True):
    """xframe/projects/fxs/extract.py:418-430: the block q_id_limits[o, 0] of every order replaced by its nearest positive
    semidefinite matrix"""
    out = np.array(b_coeff, copy=True)
    if not bl_enforce_psd:
        return out
    lim = np.array(q_id_limits)
    if not (lim[:, 0, :] == lim[:, 1, :]).all():
        lim[:, 1] = lim[:, 0]
    for o in range(len(out)):
        lo, hi = int(lim[o, 0, 0]), int(lim[o, 0, 1])
        if hi > lo:
            out[o, lo:hi, lo:hi] = nearest_positive_semidefinite_matrix(engine, b_coeff[o, lo:hi, lo:hi])
    return out
