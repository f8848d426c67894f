"""Settings of the 2-D loop sub-variants recorded in mtip2d_variants_N12_M6.npz (data only: imported by make_golden.py, which runs the
reference on them, and by tests/parity_cases.py, which runs the oracle and the product on them)."""

# the 2-D loop's own sub-variants (reconstruct.py:598-613, 721-755, 886-904 with dimensions == 2), same data and rho0 as G20
VARIANTS_2D = {
    'nonfxs': {'main_loop': {'sub_loops': {'main': {
        'methods': {'HIO': {'iterations': 3, 'ft_stab': True}, 'HIO_non_FXS': {'iterations': 2, 'ft_stab': True},
                    'SW': 1, 'ER_non_FXS': {'iterations': 2, 'ft_stab': False}, 'ER': {'iterations': 2, 'ft_stab': True}},
        'order': ['HIO', 'HIO_non_FXS', 'SW', 'ER_non_FXS', 'ER'], 'iterations': 2}}}},
    'swcenter': {'main_loop': {'sub_loops': {'main': {
        'methods': {'HIO': {'iterations': 3, 'ft_stab': True}, 'SW': 1, 'ER': {'iterations': 2, 'ft_stab': True},
                    'SW_center': 2, 'HIO_non_FXS': {'iterations': 2, 'ft_stab': False}},
        'order': ['HIO', 'SW', 'ER', 'SW_center', 'HIO_non_FXS'], 'iterations': 2}}}},
    'shift': {'output_density_modifiers': {'shift_to_center': True}},
    'recip_deg2': {'main_loop': {'error': {'methods': {'reciprocal': {'calculate': ['deg2_invariant_l2_diff'],
                                                                      'deg2_invariant_l2_diff': {'order': 2}}}}}},
    'recip_l2': {'main_loop': {'error': {'methods': {'reciprocal': {'calculate': ['l2_projection_diff']}}}}},
    'autocorr_support': {'projections': {'real': {'projections': {'support': {'initial_support': {'type': 'auto_correlation'}}}}}},
    'so_freedom': {'projections': {'reciprocal': {'SO_freedom': {'use': True, 'radial_high_pass': 0.2}}}},
    'so_freedom_fix': {'projections': {'reciprocal': {'SO_freedom': {'use': True, 'radial_high_pass': 0.2}}},
                       'output_density_modifiers': {'fix_orientation': True, 'shift_to_center': True}},
    # radial_high_pass 0.5 ranks order 4 first: the ladder of generate_remaining_SO_projection_2D (1053-1080) is not empty
    'so_freedom_fix_hp': {'projections': {'reciprocal': {'SO_freedom': {'use': True, 'radial_high_pass': 0.5}}},
                          'output_density_modifiers': {'fix_orientation': True}},
    'autocorr_guess': {'density_guess': {'type': 'low_resolution_autocorrelation'}, '_reference_guess': True},
}
