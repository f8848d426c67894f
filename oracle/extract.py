"""TEST INFRASTRUCTURE (CPU oracle, numpy / scipy): the numerics of the reference's `extract` step, B_l -> projection matrices.
Restated from the reference, pinned by tests/golden/extract_ops.npz (G15: outputs of the reference's own functions, written by
tests/golden/make_golden.py extract).  Imported only by tests/."""
import numpy as np
import scipy.linalg


def deg2_invariant_eigenvalues(b_matrix, sort_mode=0):
    """xframe/projects/fxs/projectLibrary/fxs_invariant_tools.py:1114-1141: Hermitian part, eigen-decomposition (zeros for a
    matrix that is zero to numpy.isclose), eigenpairs sorted in DESCENDING order of the eigenvalue (sort_mode 0) or of
    median_q |sqrt|lambda| v(q)| * sign(lambda) (sort_mode 1)."""
    b = (b_matrix + b_matrix.T.conj()) / 2                                        # 1122
    if not np.isclose(b, 0).all():                                                # 1123-1126
        w, v = scipy.linalg.eigh(b, driver='ev')
    else:                                                                         # 1128-1130
        v = np.zeros(b.shape)
        w = np.zeros(b.shape[0])
    signs = np.sign(w)                                                            # 1132
    if sort_mode == 0:
        metric = w                                                                # 1133-1135
    else:
        metric = np.median(np.abs(np.sqrt(np.abs(w[None, :])) * v), axis=0) * signs   # 1136-1137
    ids = np.argsort(metric)[::-1]                                                # 1138
    return w[ids].real, v[:, ids]                                                 # 1139-1141


def deg2_invariant_to_projection_matrices_3d(b_coeff, q_id_limits, order, sort_mode=0):
    """fxs_invariant_tools.py:1171-1207 for one order: eigenpairs of the square block q_id_limits[0] of B_l, the first
    min(block size, 2 order + 1) of them kept, negative eigenvalues and their vectors zeroed, V = vectors sqrt(values) embedded
    in len(b_coeff) rows and min(len(b_coeff), 2 order + 1) columns; returns (V as complex, eigenvalues)."""
    q_slice = slice(*q_id_limits[0])                                              # 1172
    w, v = deg2_invariant_eigenvalues(b_coeff[q_slice, q_slice], sort_mode=sort_mode)
    n_full = min(len(b_coeff), 2 * order + 1)                                     # 1196
    if len(w) != 0:                                                               # 1179-1194
        n = min(len(v), 2 * order + 1)
        v, w = v[:, :n].copy(), w[:n].copy()
        neg = w < 0
        w[neg] = 0
        v[:, neg] = 0
    full_v = np.zeros((len(b_coeff), n_full), dtype=v.dtype)                      # 1197-1201
    full_w = np.zeros(n_full, dtype=w.dtype)
    if len(w) != 0:
        full_v[q_slice, :n] = v
        full_w[:n] = w
    return (full_v @ np.diag(np.sqrt(full_w))).astype(complex), full_w            # 1202-1207


def default_q_id_limits(b_coeff):
    """fxs_invariant_tools.py:1092-1094: without limits every order uses the whole matrix; 1095-1099: limits of a non-square
    selection are replaced by the first axis' limits."""
    lim = np.zeros((b_coeff.shape[0], 2, 2), dtype=int)
    lim[..., 1] = b_coeff.shape[-1]
    return lim


def deg2_invariant_to_projection_matrices(b_coeff, q_id_limits=False, sort_mode=0):
    """fxs_invariant_tools.py:1079-1112, dim == 3: every order through deg2_invariant_to_projection_matrices_3d (the reference
    spreads the orders over worker processes, 1106); returns (tuple of V_l, tuple of eigenvalues)."""
    if isinstance(q_id_limits, bool):
        q_id_limits = default_q_id_limits(b_coeff)
    q_id_limits = np.array(q_id_limits)
    if not (q_id_limits[:, 0, :] == q_id_limits[:, 1, :]).all():                  # 1095-1099
        q_id_limits[:, 1] = q_id_limits[:, 0]
    res = [deg2_invariant_to_projection_matrices_3d(b_coeff[o], q_id_limits[o], o, sort_mode) for o in range(len(b_coeff))]
    return tuple(p for p, _ in res), tuple(e for _, e in res)


def nearest_positive_semidefinite_matrix(A, low_positive_eigenvalues_to_zero=False):
    """xframe/library/mathLibrary.py:872-892 (Higham 1988): eigenvalues of the Hermitian part below the limit (0, or |smallest
    eigenvalue of A itself| as a noise floor) set to zero; batched over leading axes."""
    B = (A + np.swapaxes(A, -1, -2).conj()) / 2                                   # 878
    w, v = np.linalg.eigh(B)                                                      # 879
    limit = 0
    if low_positive_eigenvalues_to_zero:                                          # 882-886
        limit = np.abs(np.min(np.linalg.eig(A)[0]))
    w[w < limit] = 0                                                              # 889
    return v * w[..., None, :] @ np.swapaxes(v, -1, -2).conj()                    # 890
