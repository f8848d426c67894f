"""Operator registry with the reference's operator names, backed by the HIP engine.

The reference registers name -> callable pairs in a ``RecipeFactory`` (``xframe/library/pythonLibrary.py:575-905``;
names at ``xframe/projects/fxs/reconstruct.py:370, 391, 445, 457, 475, 485``) and strings them together with
"sketches".  The production loop here bypasses the registry (it runs fused on the device); the registry exists so
that the same names can be called operator-by-operator -- numpy in, numpy out -- for parity tests and for user
code written against the reference API.  `RecipeFactory.buildProcessFromSketch` understands the sketch
notation of the live sketches.
"""
import inspect

import numpy as np

from . import hostsetup as hs


def _id(x):
    return x


class Process:
    def __init__(self, factory, steps):
        self.factory = factory
        self.steps = steps                     # list of (mapping | None, [op specs])
        first_map = steps[0][0]
        self.n_inputs = len(first_map) if first_map is not None else None

    def _arity(self, fn, name):
        if isinstance(fn, Process):
            return fn.n_inputs if fn.n_inputs is not None else 1
        n = self.factory.number_of_arguments_per_operator.get(name)
        if isinstance(n, int):
            return n
        return len(inspect.signature(fn).parameters)

    def run(self, *args):
        prev = tuple(args)
        for mapping, ops in self.steps:
            specs = []
            for spec in ops:
                fixed = ()
                name = spec
                if isinstance(spec, (tuple, list)) and not isinstance(spec, str):
                    name, fixed = spec[0], tuple(spec[1])
                fn = self.factory.operatorDict[name]
                specs.append((name, fn, fixed, self._arity(fn, name) - len(fixed)))
            n_free = sum(s[3] for s in specs)
            if mapping is not None:
                ins = [prev[i] for i in mapping]
            elif len(prev) == 1:
                ins = [prev[0]] * n_free
            elif len(prev) == n_free:
                ins = list(prev)
            elif len(prev) < n_free:
                d = n_free - len(prev)
                ins = [prev[0]] * (d + 1) + list(prev[1:])
            else:
                raise AssertionError('Invalid process sketch: more outputs than inputs of the next step')
            outs = []
            pos = 0
            for name, fn, fixed, free in specs:
                a = tuple(fixed) + tuple(ins[pos:pos + free])
                pos += free
                r = fn.run(*a) if isinstance(fn, Process) else fn(*a)
                if r is not None:
                    outs.append(r)
            prev = tuple(outs)
        return prev[0] if len(prev) == 1 else prev


class RecipeFactory:
    def __init__(self, operatorDict=None):
        self.operatorDict = {}
        self.number_of_arguments_per_operator = {}
        self.addOperators({'id': _id})
        self.addOperators(operatorDict or {})

    def copy(self):
        f = RecipeFactory(dict(self.operatorDict))
        f.number_of_arguments_per_operator = dict(self.number_of_arguments_per_operator)
        return f

    def addOperators(self, operatorDict):
        for key, value in operatorDict.items():
            if isinstance(value, list):
                assert len(value) == 2 and isinstance(value[1], int)
                self.operatorDict[key] = value[0]
                self.number_of_arguments_per_operator[key] = value[1]
            else:
                self.operatorDict[key] = value

    def get_operator(self, name):
        return self.operatorDict[name]

    def buildProcessFromSketch(self, sketch):
        steps = []
        for step in sketch:
            mapping = None
            ops = step
            if isinstance(step, str):
                ops = [step]
            elif len(step) == 2 and not isinstance(step[0], str) and isinstance(step[1], (list, tuple)) and \
                    isinstance(step[0], (tuple, list, np.ndarray)) and all(isinstance(i, (int, np.integer)) for i in step[0]):
                mapping, ops = tuple(int(i) for i in step[0]), list(step[1])
            steps.append((mapping, list(ops)))
        return Process(self, steps)


def build_operators(engine):
    """name -> callable dict for ``RecipeFactory.addOperators`` (host numpy arrays in / out, batch element 0
    when an engine holds several restarts)."""
    e = engine
    L = e.L

    def _to_lm(c):
        return [np.array(c[:, l * l:(l + 1) ** 2]) for l in range(L + 1)]

    def _from_lm(lst):
        return np.concatenate(lst, axis=-1)

    def fourier_transform(data):
        return e.fourier_transform(data)[0]

    def inverse_fourier_transform(data):
        return e.fourier_transform(data, inverse=True)[0]

    def harmonic_transform(data):                       # 'lm' layout (shtns_plugin.py:218-222)
        return _to_lm(e.sht_forward(data)[0])

    def inverse_harmonic_transform(coeff):
        return e.sht_inverse(_from_lm(coeff))[0]

    def square_grid(data):                              # misk.py:159-168
        return data * data.conj()

    def abs_value(data):                                # misk.py:221-225
        return np.sqrt((data * data.conj()).real).astype(complex)

    state = {'unknowns': None}

    def approximate_unknowns(Ilm):
        e.project_coefficients(_from_lm(Ilm))
        state['unknowns'] = e.unknowns(0)
        return state['unknowns']

    def mtip_projection(Ilm, unknowns):
        # I'_l = V_l U_l with the caller's unknowns (fxs_Projections.py:832-849); None = solve and apply in one pass
        if unknowns is None:
            return _to_lm(e.project_coefficients(_from_lm(Ilm))[0])
        return _to_lm(e.apply_unknowns(_from_lm(Ilm), unknowns)[0])

    def project_to_modified_intensity(reciprocal_density, square, new_intensity):
        return e.modulus_replacement(reciprocal_density, new_intensity)[0]

    def hybrid_input_output(without_projection, projection_out, _input, beta=None):
        b = e.hio_beta if beta is None else beta
        return e.real_space_update(without_projection, _input, 'HIO', b)[0][0]

    def error_reduction(out_without_projection, out, _input):
        return np.array(out[0])

    def real_projection(data):
        proj, _ = e.real_space_update(data, data, 'ER', 0.0)
        data[...] = proj[0]                              # the reference projects in place (fxs_Projections.py:120-130)
        return [data, {'all': None}]

    def real_errors(values, projected):
        return {'l2_projection_diff': float(e.real_space_update(values, values, 'ER', 0.0)[1][0])}

    def calc_deg2_invariant(Ilm):
        return e.deg2_invariants(_from_lm(Ilm))[0]

    def copy(data):
        return np.array(data)

    def diff(a, b):
        return a - b

    def add_above_zero_index(a, b):                     # misk.py:326-329
        r = a + b
        r[0] = a[0]
        return r

    # ---- operators of the reference registry that are plain host logic there as well ------------------------------
    fixed = {'intensity': None}
    results = {'n_particles': []}
    sw = {'sigma': e.default_sigma, 'threshold': 0.06}

    def project_to_fixed_intensity(reciprocal_density, square):          # fxs_Projections.py:911-923
        if fixed['intensity'] is None:
            raise RuntimeError("project_to_fixed_intensity: set operators['set_fixed_intensity'](I) first")
        return e.modulus_replacement(reciprocal_density, fixed['intensity'])[0]

    def set_fixed_intensity(intensity):                                  # rp.fixed_intensity setter, reconstruct.py:899-904
        fixed['intensity'] = np.array(intensity, dtype=complex)

    def save_number_of_particles():                                      # reconstruct.py:380-389
        results['n_particles'].append(e.rsetup.number_of_particles)

    def multiply_ft_gaussian(data):                                      # fxs_Projections.py:294-298, mathLibrary.py:616-624
        a = 1.0 / (2.0 * sw['sigma'] ** 2)
        q2 = np.asarray(e.qs)[:, None, None] ** 2
        return data * (np.sqrt(np.pi / a) * np.exp(-np.pi ** 2 * q2 * q2 / a))

    def calculate_support_mask(convolution_data):                        # fxs_Projections.py:245-258
        c = np.array(convolution_data.real)
        c[c < 0] = 0
        lo, hi = c.min(), c.max()
        return c >= lo + sw['threshold'] * (hi - lo)

    def set_shrink_wrap(sigma=None, threshold=None):
        if sigma is not None:
            sw['sigma'] = sigma
        if threshold is not None:
            sw['threshold'] = threshold

    def calc_center(density):                                            # misk.py:295-312
        return hs.calc_center(e.rs, e.theta, e.phi, density)

    def negative_shift(reciprocal_density, vector):                      # fxs_Projections.py:1419-1444, opposite direction
        reciprocal_density *= hs.shift_phases(e.qs, e.theta, e.phi, vector, opposite_direction=True)
        return reciprocal_density

    def save_to_dict(dictionary, keys, mode, data):                      # misk.py:104-134
        keys = keys if isinstance(keys, list) else [keys]
        folder = dictionary
        for key in keys[:-1]:
            folder = folder[key]
        key = keys[-1]
        if mode == 'append':
            folder[key] = folder.get(key, []) + [data]
        elif mode == 'iterative_append':
            for dkey in data:
                folder[key][dkey] = folder[key].get(dkey, []) + [data[dkey]]
        elif mode == 'iterative_overwrite':
            for dkey in data:
                folder[key][dkey] = data[dkey]
        else:
            folder[key] = data
        return data

    def load_from_dict(dictionary, keys):                                # misk.py:136-145
        value = dictionary
        for key in (keys if isinstance(keys, list) else [keys]):
            value = value[key]
        return value

    e.hio_beta = e.opt['projections']['real']['HIO']['beta'][0][0]
    return {
        'fourier_transform': fourier_transform, 'inverse_fourier_transform': inverse_fourier_transform,
        'harmonic_transform': harmonic_transform, 'inverse_harmonic_transform': inverse_harmonic_transform,
        'complex_harmonic_transform': harmonic_transform, 'complex_inverse_harmonic_transform': inverse_harmonic_transform,
        'mtip_projection': mtip_projection, 'approximate_unknowns': approximate_unknowns,
        'project_to_modified_intensity': project_to_modified_intensity,
        'hybrid_input_output': [hybrid_input_output, 3], 'error_reduction': error_reduction,
        'real_projection': real_projection, 'real_errors': [real_errors, 2],
        'square_grid': square_grid, 'abs_value': abs_value, 'calc_deg2_invariant': calc_deg2_invariant,
        'copy': copy, 'add_above_zero_index': add_above_zero_index, 'diff': diff,
        'project_to_fixed_intensity': project_to_fixed_intensity, 'set_fixed_intensity': set_fixed_intensity,
        'save_number_of_particles': save_number_of_particles, 'multiply_ft_gaussian': multiply_ft_gaussian,
        'calculate_support_mask': calculate_support_mask, 'set_shrink_wrap': set_shrink_wrap,
        'calc_center': calc_center, 'negative_shift': negative_shift,
        'save_to_dict': [save_to_dict, 4], 'load_from_dict': [load_from_dict, 2],
    }
