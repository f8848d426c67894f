"""On-disk contract of the fxs project without h5py: the dict trees on both sides of the files.

  * ``reconstruction_tree``   what the reference's reconstruct worker hands to its database after a run
                              (xframe/projects/fxs/reconstruct.py:160-185, post_processing): the per-restart result dicts
                              keyed by restart id in order of their final error, the grid pair, the projection matrices, stats;
  * ``hdf5_layout``           what the reference's HDF5 plugin writes for such a tree (xframe/externalLibraries/hdf5_plugin.py:
                              53-140): one entry per group / dataset with path, dtype, shape and the `type` attribute that
                              marks strings, lists, tuples and NestedArrays; ``tree_from_hdf5_layout`` is its loader (90-112);
  * ``load_invariants``       what ProjectDB.load_invariants (xframe/projects/fxs/_database_.py:566-609) makes of the tree of an
                              invariants file -- the `data` dict the reconstruct worker starts from.

Writing the bytes needs an HDF5 library (absent from this image): ``write_hdf5`` / ``read_hdf5`` do it through h5py when it can
be imported and raise otherwise.  The layouts are pinned by tests/golden/io_contract.npz (G16), recorded from the reference's own
post_processing, plugin and loader running against a recording stand-in for h5py.File."""
import numpy as np


class GridPair:
    """the two grids of a reconstruction (reference: FTGridPair of NestedArrays, pythonLibrary.py:1045-1048)"""

    def __init__(self, real_grid, reciprocal_grid):
        self.realGrid, self.reciprocalGrid = np.asarray(real_grid), np.asarray(reciprocal_grid)


def reconstruction_tree(results, xray_wavelength, reciprocity_coefficient, stats=None):
    """reconstruct.py:160-185: `results` = sequence (or dict id -> dict) of per-restart result dicts as ``MTIP.phasing_loop``
    returns them (reconstruct.py:1003-1021).  'grid_pair' and 'projection_matrices' are taken out of every result dict (those
    of the last one are kept, 170-173), the dicts are keyed by str(id) in ascending order of their last main error (174-176)."""
    items = list(results.items()) if isinstance(results, dict) else list(enumerate(results))
    dicts = [dict(d) for _, d in items]
    errors, grid_pair, projection_matrices = [], None, None
    for d in dicts:
        grid_pair = d.pop('grid_pair')
        errors.append(d['error_dict']['main'][-1])
        projection_matrices = d.pop('projection_matrices')
    order = np.argsort(errors)
    return {'configuration': {'internal_grid': grid_pair, 'xray_wavelength': xray_wavelength,
                              'reciprocity_coefficient': reciprocity_coefficient},
            'reconstruction_results': {str(int(i)): dicts[int(i)] for i in order},
            'projection_matrices': projection_matrices, 'stats': dict(stats or {})}


def _array_entry(path, a, type_attr='', n_ndim=-1):
    a = np.asarray(a)
    if a.dtype == np.complex128 or a.dtype.kind == 'c':
        a = a.astype('<c16')                                     # hdf5_plugin.py:117-118
    elif a.dtype.kind == 'U':
        a = a.astype('S')                                        # 122-123
    return {'path': path, 'kind': 'dataset', 'dtype': str(a.dtype), 'shape': tuple(a.shape), 'type': type_attr, 'n_ndim': n_ndim,
            'value': a}


def hdf5_layout(tree, path=''):
    """hdf5_plugin.py:53-88 (recursively_save_dict_to_group): the nodes the plugin creates for `tree`, in its order.  Keys become
    str; scalars datasets; str utf-8 bytes with type 'str'; arrays datasets (complex as <c16, unicode as bytes); lists / tuples
    groups with type 'list' / 'tuple' and children '0', '1', ...; dicts groups; a grid pair a group of its two grids (138-141),
    whose NestedArrays carry type 'NestedArray' and n_ndim (134-136).  Anything else is an error, as there (84-85)."""
    out = []
    for key, item in tree.items():
        key = str(key)
        p = path + '/' + key
        if isinstance(item, (complex, float, int, bytes, bool, np.number, np.bool_)):
            out.append(_array_entry(p, item))
        elif isinstance(item, str):
            e = _array_entry(p, np.asarray(item.encode('utf-8')), 'str')
            out.append(e)
        elif isinstance(item, np.ndarray):
            out.append(_array_entry(p, item))
        elif isinstance(item, (list, tuple)):
            out.append({'path': p, 'kind': 'group', 'dtype': '', 'shape': (), 'type': 'list' if isinstance(item, list) else 'tuple',
                        'n_ndim': -1, 'value': None})
            out += hdf5_layout({str(i): x for i, x in enumerate(item)}, p)
        elif isinstance(item, dict):
            out.append({'path': p, 'kind': 'group', 'dtype': '', 'shape': (), 'type': '', 'n_ndim': -1, 'value': None})
            out += hdf5_layout(item, p)
        elif isinstance(item, GridPair):
            out.append({'path': p, 'kind': 'group', 'dtype': '', 'shape': (), 'type': '', 'n_ndim': -1, 'value': None})
            out.append(_array_entry(p + '/real_grid', item.realGrid, 'NestedArray', 1))
            out.append(_array_entry(p + '/reciprocal_grid', item.reciprocalGrid, 'NestedArray', 1))
        else:
            raise ValueError('Cannot save {} type for key {}'.format(type(item), key))
    return out


def tree_from_hdf5_layout(entries):
    """hdf5_plugin.py:90-112, 128-137 (recursively_load_dict_from_group): datasets become values (type 'str' decoded), groups
    dicts, groups of type 'list' / 'tuple' sequences of their children '0' .. 'n-1'"""
    root = {}
    groups = {'': (root, '')}
    for e in entries:
        parent, name = e['path'].rsplit('/', 1)
        node = groups[parent][0]
        if e['kind'] == 'group':
            child = {}
            node[name] = child
            groups[e['path']] = (child, e['type'])
        else:
            v = e['value']
            if e['type'] == 'str':
                v = bytes(v[()]).decode('utf-8')
            elif v.shape == ():
                v = v[()]
            node[name] = v

    def finish(d, path):
        for k in list(d):
            if isinstance(d[k], dict):
                p = path + '/' + k
                finish(d[k], p)
                t = groups[p][1]
                if t in ('list', 'tuple'):
                    seq = [d[k][str(i)] for i in range(len(d[k]))]
                    d[k] = seq if t == 'list' else tuple(seq)
        return d
    return finish(root, '')


def load_invariants(tree):
    """_database_.py:566-609 on the tree of an invariants file (as the HDF5 loader returns it).  Returns the `data` dict of the
    reconstruct worker: 'data_projection_matrices' an object array over the orders (dict-of-orders sorted by integer key; the
    'I1I1' member of a two-dataset file, the full dict kept as 'data_projection_matrices_2'; a one-dimensional l = 0 matrix of old
    files made a column), 'data_low_resolution_intensity_coefficients' an object array or False, 'b_coeff' = the file's
    'deg_2_invariant' or False.  'average_intensity' stays the sampled values (the reference wraps them with the radial points into
    a SampledFunction; the engine takes both arrays from this dict)."""
    data = dict(tree)
    pm = data['data_projection_matrices']
    if isinstance(pm, np.ndarray):
        matrices = pm
    elif 'I1I1' in pm:
        matrices = pm['I1I1']
        data['data_projection_matrices_2'] = pm
    else:
        matrices = pm
    low_res = data.get('data_low_resolution_intensity_coefficients', False)
    data['b_coeff'] = data.get('deg_2_invariant', False)
    if isinstance(matrices, dict):
        keys = np.sort(tuple(int(k) for k in matrices.keys())).astype(str)
        matrices = tuple(matrices[k] for k in keys)
    if data['dimensions'] == 3:
        tmp = np.empty(len(matrices), object)
        for i, m in enumerate(matrices):
            tmp[i] = m
        matrices = tmp
        data['data_projection_matrices'] = matrices
        if not isinstance(low_res, bool):
            tmp = np.empty(len(low_res), object)
            for i, m in enumerate(low_res):
                tmp[i] = m
            low_res = tmp
        if len(matrices[0].shape) < 2:
            matrices[0] = matrices[0][:, None]                   # legacy files
        data['data_low_resolution_intensity_coefficients'] = low_res
    elif data['dimensions'] == 2:
        data['data_projection_matrices'] = np.array(matrices)
    return data


def _h5py():
    try:
        import h5py
        return h5py
    except ImportError as e:
        raise ImportError('writing / reading HDF5 bytes needs h5py, which this environment does not have; the dict trees and their '
                          'HDF5 layout are available without it (reconstruction_tree, hdf5_layout, load_invariants)') from e


def write_hdf5(path, tree):
    """the file the reference's plugin would write for `tree` (hdf5_plugin.py:29-36)"""
    h5 = _h5py()
    with h5.File(path, 'w') as f:
        for e in hdf5_layout(tree):
            if e['kind'] == 'group':
                g = f.create_group(e['path'])
                if e['type']:
                    g.attrs['type'] = e['type']
            else:
                d = f.create_dataset(e['path'], data=e['value'])
                if e['type']:
                    d.attrs['type'] = e['type']
                if e['n_ndim'] >= 0:
                    d.attrs['n_ndim'] = e['n_ndim']


def read_hdf5(path):
    """hdf5_plugin.py:38-51"""
    h5 = _h5py()
    entries = []

    def visit(name, obj):
        t = obj.attrs.get('type', '')
        t = t.decode() if isinstance(t, bytes) else t
        if isinstance(obj, h5.Dataset):
            entries.append({'path': '/' + name, 'kind': 'dataset', 'type': t, 'value': np.asarray(obj[()]), 'n_ndim': int(obj.attrs.get('n_ndim', -1))})
        else:
            entries.append({'path': '/' + name, 'kind': 'group', 'type': t, 'value': None, 'n_ndim': -1})
    with h5.File(path, 'r') as f:
        f.visititems(visit)
    return tree_from_hdf5_layout(entries)
