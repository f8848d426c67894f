"""Pin the oracle SHT (shtns stand-in; "parity unpinned" vs shtns itself) with analytic known answers."""
import numpy as np
import pytest

from oracle.sht import SHT, angular_grid_size


def test_grid_sizes_follow_plugin_formula():
    # shtns_plugin.py:94-101
    assert angular_grid_size(8) == (16, 32)
    assert angular_grid_size(16) == (32, 64)
    assert angular_grid_size(32) == (64, 128)
    assert angular_grid_size(48) == (128, 256)
    assert angular_grid_size(63) == (128, 256)


def test_known_answers_condon_shortley_orthonormal():
    s = SHT(4)
    th, ph = np.meshgrid(s.theta, s.phi, indexing='ij')
    ct, st = np.cos(th), np.sin(th)
    known = {
        (0, 0): np.full(th.shape, 0.5 / np.sqrt(np.pi)) + 0j,
        (1, 0): np.sqrt(3 / (4 * np.pi)) * ct + 0j,
        (1, 1): -np.sqrt(3 / (8 * np.pi)) * st * np.exp(1j * ph),
        (1, -1): np.sqrt(3 / (8 * np.pi)) * st * np.exp(-1j * ph),
        (2, 0): np.sqrt(5 / (16 * np.pi)) * (3 * ct ** 2 - 1) + 0j,
        (2, 1): -np.sqrt(15 / (8 * np.pi)) * st * ct * np.exp(1j * ph),
        (2, 2): np.sqrt(15 / (32 * np.pi)) * st ** 2 * np.exp(2j * ph),
        (2, -2): np.sqrt(15 / (32 * np.pi)) * st ** 2 * np.exp(-2j * ph),
        (3, -3): np.sqrt(35 / (64 * np.pi)) * st ** 3 * np.exp(-3j * ph),
    }
    for (l, m), f in known.items():
        c = s.forward_d(f[None])[0]
        e = np.zeros(s.n_coeff, complex)
        e[l * (l + 1) + m] = 1
        assert np.abs(c - e).max() < 1e-13, (l, m)
        assert np.abs(s.inverse_d(e[None])[0] - f).max() < 1e-13, (l, m)


@pytest.mark.parametrize('L', [4, 8, 16, 32])
def test_orthonormality_roundtrip_parseval(L):
    s = SHT(L)
    rng = np.random.default_rng(L)
    c = rng.normal(size=(2, s.n_coeff)) + 1j * rng.normal(size=(2, s.n_coeff))
    f = s.inverse_d(c)
    assert np.abs(s.forward_d(f) - c).max() < 1e-12
    # Parseval with the quadrature:  int |f|^2 dOmega = sum |c|^2
    quad = (2 * np.pi / s.n_phi) * np.einsum('t,stp->s', s.weights, np.abs(f) ** 2)
    assert np.allclose(quad, np.sum(np.abs(c) ** 2, axis=1), rtol=1e-12)
    # theta runs north -> south
    assert np.all(np.diff(s.theta) > 0)


def test_layout_views():
    s = SHT(5)
    rng = np.random.default_rng(0)
    f = rng.normal(size=(3, s.n_theta, s.n_phi)) + 0j
    d = s.forward_d(f)
    lm = s.forward_l(f)
    ml = s.forward_m(f)
    assert [x.shape for x in lm] == [(3, 2 * l + 1) for l in range(6)]
    assert [x.shape[1] for x in ml] == [6 - abs(m) for m in s.m]
    for l in range(6):
        assert np.array_equal(lm[l], d[:, l * l:(l + 1) ** 2])
    for i, m in enumerate(s.m):
        ls = np.arange(abs(m), 6)
        assert np.array_equal(ml[i], d[:, ls * (ls + 1) + m])
    assert np.allclose(s.inverse_l(lm), s.inverse_d(d)) and np.allclose(s.inverse_m(ml), s.inverse_d(d))
