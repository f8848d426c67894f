"""Resolved settings tree of the ``fxs reconstruct`` worker.

Key names are the reference's (``xframe/projects/fxs/settings/reconstruct/default_0.01.yaml:1-321``); the
reference's YAML defaults DSL (``_value`` / ``_if`` / ``command:`` eval, ``xframe/database/database.py:495-697``)
is not re-implemented: callers pass plain nested dicts that are merged over :func:`default_settings`.
"""
import copy

import numpy as np


class DictNamespace(dict):
    """dict with attribute access (``xframe/library/pythonLibrary.py:911-976`` look-alike)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def dict(self):
        return to_plain(self)

    @classmethod
    def dict_to_dictnamespace(cls, d):
        return to_namespace(d)


def to_namespace(d):
    if isinstance(d, dict):
        return DictNamespace({k: to_namespace(v) for k, v in d.items()})
    return d


def to_plain(d):
    if isinstance(d, dict):
        return {k: to_plain(v) for k, v in d.items()}
    return d


def deep_update(base, upd):
    out = copy.deepcopy(base)
    for k, v in upd.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict):
            out[k] = deep_update(out[k], v)
        else:
            out[k] = copy.deepcopy(v)
    return out


def default_settings():
    return {
        'dimensions': 3,
        'structure_name': 'default_structure',
        'particle_radius': 150,
        'grid': {'max_q': False, 'max_order': 63, 'n_phi': 0, 'n_theta': 0, 'n_radial_points': 128},
        'fourier_transform': {'type': 'midpoint', 'reciprocity_coefficient': 2.0,
                              'allow_weight_calculation': True, 'allow_weight_saving': True},
        'density_guess': {'type': 'bump', 'bump': {'slope': 0.3}, 'radius': 150,
                          'amplitude_function': 'random', 'random': {'SNR': 2}},
        'projections': {
            'real': {
                'projections': {
                    'apply': ['support', 'value_threshold', 'assert_real'],
                    'value_threshold': {'threshold': [0, False]},
                    'limit_imag': {'threshold': 2},
                    'support': {'initial_support': {'type': 'max_radius', 'max_radius': 150,
                                                    'auto_correlation': {'threshold': 0.01}},
                                'enforce_initial_support': {'apply': True, 'if_error_bigger_than': 6e-3}},
                },
                'shrink_wrap': {'sigmas': [[False, [False, False], False], [False, [False, False], False]],
                                'thresholds': [[0.08, [0, 0], 0], [0.08, [0, 0], 0]]},
                'HIO': {'beta': [[0.5, 0.4, -1 / 700, 1600], [0.01, 0.002, -1 / 200, 200]],
                        'considered_projections': ['all']},
            },
            'reciprocal': {
                'number_of_particles': {'initial': 1.0, 'estimate': False},
                'regrid': {'interpolation': 'cubic'},
                'used_order_ids': np.arange(64),
                'odd_orders_to_0': True,
                'use_averaged_intensity': True,
                'q_mask': {'type': 'none'},
                'SO_freedom': {'use': False},
            },
        },
        'output_density_modifiers': {'shift_to_center': False},
        'main_loop': {
            'error': {'methods': {
                'real': {'calculate': ['l2_projection_diff'],
                         'l2_projection_diff': {'inside_initial_support': True}},
                'reciprocal': {'calculate': [], 'deg2_invariant_l2_diff': {'order': 2}},
                'main': {'metrics': {'real': ['l2_projection_diff'], 'reciprocal': []}, 'type': 'mean'}}},
            'sub_loops': {
                'order': ['main', 'refinement'],
                'main': {'methods': {'HIO': {'iterations': 60, 'ft_stab': True},
                                     'ER': {'iterations': 40, 'ft_stab': True}, 'SW': 1},
                         'order': ['HIO', 'SW', 'ER'], 'iterations': 5,
                         'best_density_not_in_first_n_iterations': np.inf},
                'refinement': {'methods': {'ER': {'iterations': 100, 'ft_stab': True}, 'SW': 1},
                               'order': ['SW', 'ER'], 'iterations': 2,
                               'best_density_not_in_first_n_iterations': np.inf},
            },
        },
        'GPU': {'use': True, 'n_gpu_workers': 1},
        'multi_process': {'use': True, 'n_parallel_reconstructions': 1},
        'profiling': {'enable': False, 'reconstruction_process_id': 1, 'gpu_worker_id': -1},
        # xframe/settings/general.py:25-27 (decides which variant of the real error metric the reference uses)
        'general': {'cache_aware': True, 'L2_cache': 512},
    }


def resolve(overrides=None):
    o = default_settings()
    if overrides:
        o = deep_update(o, to_plain(overrides))
    return o


def reciprocity_coefficient(ft_opt):
    """``misk.py:387-394`` / ``mathLibrary.py:1178-1182``."""
    pi_in_q = ft_opt.get('pi_in_q', None)
    if isinstance(pi_in_q, bool):
        return np.pi if pi_in_q else 1 / 2
    return ft_opt.get('reciprocity_coefficient', np.pi)
