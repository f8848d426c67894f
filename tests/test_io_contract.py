"""The fxs project's on-disk contract without h5py: xframe_amd/fxs/io.py against G16
(tests/golden/io_contract.npz: recorded from the reference's post_processing, HDF5 plugin and load_invariants)."""
import os

import numpy as np
import pytest

from xframe_amd.fxs import io as IO

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'io_contract.npz'))


def _result(rid):
    pre = f'G16_res{rid}/'
    d = {}
    for k in G.files:
        if not k.startswith(pre):
            continue
        parts = k[len(pre):].split('/')
        if len(parts) == 1:
            v = G[k]
            d[parts[0]] = v[()] if v.shape == () else v
    for name in ('fxs_unknowns', 'n_particles', 'n_particles_gradients', 'n_particles_fraction', 'projection_matrices'):
        n = int(G[pre + name + '/__len__'])
        d[name] = [(lambda v: v[()] if v.shape == () else v)(G[f'{pre}{name}/{i}']) for i in range(n)]
    d['final_error'] = float(d['final_error'])
    d['loop_iterations'] = int(d['loop_iterations'])
    d['n_particles'] = [int(x) for x in d['n_particles']]
    d['error_dict'] = {'main': G[pre + 'error_dict/main'], 'real': {'l2_projection_diff': G[pre + 'error_dict/real/l2_projection_diff']},
                       'reciprocal': {}}
    d['grid_pair'] = IO.GridPair(G[pre + 'grid_pair/real_grid'], G[pre + 'grid_pair/reciprocal_grid'])
    return d


def _tree():
    return IO.reconstruction_tree({0: _result(0), 1: _result(1)}, 1.23984, 2.0, {'run_time': 1.5})


def test_reconstruction_tree_matches_reference_post_processing():
    t = _tree()
    assert sorted(t) == list(G['G16_tree_keys'])
    assert list(t['reconstruction_results']) == list(G['G16_tree_result_order'])        # ascending final error
    assert 'grid_pair' not in t['reconstruction_results']['0'] and 'projection_matrices' not in t['reconstruction_results']['0']


def test_hdf5_layout_matches_reference_plugin():
    lay = {e['path']: e for e in IO.hdf5_layout(_tree())}
    # (the order of the members of a group follows the insertion order of the dicts and is no part of the contract: HDF5 groups
    # are unordered)
    assert sorted(lay) == sorted(G['G16_h5_paths'])
    for path, kind, dt, sh, ta, nn in zip(G['G16_h5_paths'], G['G16_h5_kinds'], G['G16_h5_dtypes'], G['G16_h5_shapes'], G['G16_h5_type_attr'], G['G16_h5_n_ndim_attr']):
        e = lay[str(path)]
        assert e['kind'] == kind, e['path']
        assert e['type'] == ta, e['path']
        assert e['n_ndim'] == nn, e['path']
        if kind == 'dataset':
            assert e['dtype'] == dt and str(e['shape']) == sh, (e['path'], e['dtype'], dt, e['shape'], sh)
            key = 'G16_h5_value' + e['path']
            if key in G.files:
                assert np.array_equal(e['value'], G[key]), e['path']


def test_layout_round_trip_matches_reference_loader():
    back = IO.tree_from_hdf5_layout(IO.hdf5_layout(_tree()))

    def flat(d, pre=''):
        r = {}
        for k, v in d.items():
            if isinstance(v, dict):
                r.update(flat(v, pre + k + '/'))
            elif isinstance(v, (list, tuple)):
                r[pre + k + '/__type__'] = np.array(type(v).__name__)
                r.update(flat({str(i): x for i, x in enumerate(v)}, pre + k + '/'))
            else:
                r[pre + k] = np.asarray(v)
        return r
    fb = flat(back)
    assert sorted(fb) == list(G['G16_back_paths'])
    # (the reference's loader wraps the two grids into its NestedArray class, dtype object here; the product returns the arrays)
    assert [str(fb[k].dtype) for k in sorted(fb)] == [('float64' if d == 'object' else d) for d in G['G16_back_dtypes']]
    assert isinstance(back['projection_matrices'], list) and isinstance(back['reconstruction_results']['1']['fxs_unknowns'], list)
    assert np.array_equal(back['reconstruction_results']['1']['real_density'], _result(1)['real_density'])


@pytest.mark.parametrize('name', ['orders_dict', 'I1I1', 'legacy_1d_l0'])
def test_load_invariants_matches_reference(name):
    pre = f'G16_inv_{name}_in/'
    tree = {}
    for k in G.files:
        if not k.startswith(pre):
            continue
        parts = k[len(pre):].split('/')
        node = tree
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        if parts[-1] != '__type__':
            v = G[k]
            node[parts[-1]] = v[()] if v.shape == () else v
    if 'data_low_resolution_intensity_coefficients' in tree:
        lr = tree['data_low_resolution_intensity_coefficients']
        tree['data_low_resolution_intensity_coefficients'] = [lr[str(i)] for i in range(len(lr))]
    tree['dimensions'] = int(tree['dimensions'])
    d = IO.load_invariants(tree)
    ref_keys = set(G[f'G16_inv_{name}_keys'])
    assert set(d) == ref_keys
    assert len(d['data_projection_matrices']) == int(G[f'G16_inv_{name}_n_pm'])
    for l, m in enumerate(d['data_projection_matrices']):
        assert np.array_equal(m, G[f'G16_inv_{name}_pm{l}']) and m.ndim == 2
    assert np.array_equal(d['average_intensity'], G[f'G16_inv_{name}_aint_data'])
    assert np.array_equal(d['data_radial_points'][:, None], G[f'G16_inv_{name}_aint_grid'])
    assert np.array_equal(np.asarray(d['b_coeff']), G[f'G16_inv_{name}_b_coeff'])
    lr = d['data_low_resolution_intensity_coefficients']
    assert isinstance(lr, bool) == bool(G[f'G16_inv_{name}_lowres_is_bool'])
    if not isinstance(lr, bool):
        for i, m in enumerate(lr):
            assert np.array_equal(m, G[f'G16_inv_{name}_lowres{i}'])


def _flat(d, pre=''):
    r = {}
    for k, v in d.items():
        if isinstance(v, dict):
            r.update(_flat(v, pre + k + '/'))
        elif isinstance(v, (list, tuple)):
            r[pre + k + '/__type__'] = np.array(type(v).__name__)
            r.update(_flat({str(i): x for i, x in enumerate(v)}, pre + k + '/'))
        else:
            r[pre + k] = np.asarray(v)
    return r


def test_hdf5_bytes_round_trip_when_h5py_exists(tmp_path):
    """The bytes themselves: write_hdf5 -> an actual file -> (a) every group / dataset / attribute where G16 says the reference's
    plugin puts it (hdf5_plugin.py:29-88), read back with plain h5py calls; (b) read_hdf5 returns the tree the reference's loader
    returns (G16_back_*).  This image has no h5py: the test is skipped here and runs on the first box that has it."""
    h5py = pytest.importorskip('h5py')
    path = str(tmp_path / 'data.h5')
    IO.write_hdf5(path, _tree())
    seen = {}
    with h5py.File(path, 'r') as f:
        def visit(name, obj):
            t = obj.attrs.get('type', '')
            t = t.decode() if isinstance(t, bytes) else t
            if isinstance(obj, h5py.Dataset):
                seen['/' + name] = ('dataset', str(obj.dtype), tuple(obj.shape), t, int(obj.attrs.get('n_ndim', -1)), np.asarray(obj[()]))
            else:
                seen['/' + name] = ('group', '', (), t, -1, None)
        f.visititems(visit)
    assert sorted(seen) == sorted(G['G16_h5_paths'])
    for path_, kind, dt, sh, ta, nn in zip(G['G16_h5_paths'], G['G16_h5_kinds'], G['G16_h5_dtypes'], G['G16_h5_shapes'],
                                           G['G16_h5_type_attr'], G['G16_h5_n_ndim_attr']):
        k, d, shp, t, n, v = seen[str(path_)]
        assert k == str(kind) and t == str(ta) and n == int(nn), path_
        if k == 'dataset':
            assert d == str(dt) and str(shp) == str(sh), (path_, d, dt, shp, sh)
            key = 'G16_h5_value' + str(path_)
            if key in G.files:
                assert np.array_equal(v, G[key]), path_
    back = IO.read_hdf5(path)
    fb = _flat(back)
    assert sorted(fb) == list(G['G16_back_paths'])
    assert [str(fb[k].dtype) for k in sorted(fb)] == [('float64' if d == 'object' else d) for d in G['G16_back_dtypes']]
    assert np.array_equal(back['reconstruction_results']['1']['real_density'], _result(1)['real_density'])


def test_hdf5_bytes_need_h5py():
    try:
        import h5py  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError):
            IO.write_hdf5('/tmp/never_written.h5', {})
