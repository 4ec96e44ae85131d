"""Weight files of the reference, read and written without TensorFlow.

The reference keeps model weights in HDF5: one group per layer; `NN.CNN` stores the datasets `Weight` and `Bias`
(NN.py:379-394, read back at :396-419 and :505-517), `NN_extended.CNN` one dataset per variable under the variable's own
name (NN_extended.py:670-693: 'Weight' / 'Bias' by its naming convention, NN_extended.py:397-411).  Arrays are in the TF
layouts (conv HWIO / DHWIO, conv_transpose [k.., out, in], fc [out, in], fc bias [out, 1], conv bias [O]) - the layouts
`DeviceModel.set_weights` takes.

h5py is an optional dependency: with it `.h5` / `.hdf5` files are read and written in that format; without it (the build
container has none) the `.npz` twin is used - keys '<layer>/Weight', '<layer>/Bias' - and asking for an HDF5 file raises a
clear error instead of guessing.
"""
import os

import numpy as np

H5_SUFFIXES = ('.h5', '.hdf5', '.hdf')


def have_h5py():
    try:
        import h5py  # noqa: F401
        return True
    except ImportError:
        return False


def is_h5(path):
    return str(path).lower().endswith(H5_SUFFIXES)


def _need_h5py(path):
    try:
        import h5py
        return h5py
    except ImportError:
        raise ImportError('%s is an HDF5 weight file (the reference\'s format, NN.py:379-419) but h5py is not installed; '
                          'install h5py or convert the file to the .npz twin (keys "<layer>/Weight", "<layer>/Bias")' % (path,))


def _pick(group, want, layer, path):
    """Dataset `want` ('Weight' / 'Bias') of a layer group; NN_extended names datasets after the variables, so accept a
    unique dataset whose name starts with it (e.g. 'Weight', 'Weight_1')."""
    if want in group:
        return np.array(group[want])
    hits = [k for k in group.keys() if k.lower().startswith(want.lower())]
    if len(hits) == 1:
        return np.array(group[hits[0]])
    raise KeyError('%s: layer group %r has no dataset %r (found %s)' % (path, layer, want, sorted(group.keys())))


def read_weights(path, layer_names):
    """{layer: [W, b]} for `layer_names` from an HDF5 file of the reference or its .npz twin."""
    if is_h5(path):
        h5py = _need_h5py(path)
        out = {}
        with h5py.File(path, 'r') as f:
            for n in layer_names:
                if n not in f:
                    raise KeyError('%s has no group for layer %r (groups: %s)' % (path, n, sorted(f.keys())))
                out[n] = [_pick(f[n], 'Weight', n, path), _pick(f[n], 'Bias', n, path)]
        return out
    p = path if os.path.exists(path) or str(path).endswith('.npz') else str(path) + '.npz'
    f = np.load(p)
    return {n: [f[n + '/Weight'], f[n + '/Bias']] for n in layer_names}


def write_weights(path, var_dict):
    """var_dict: {layer: (W, b)} in TF layouts -> HDF5 (groups per layer, datasets Weight / Bias) or .npz."""
    if is_h5(path):
        h5py = _need_h5py(path)
        with h5py.File(path, 'w') as f:
            for n, (W, b) in var_dict.items():
                g = f.create_group(n)
                g.create_dataset('Weight', data=np.asarray(W))
                g.create_dataset('Bias', data=np.asarray(b))
        return
    d = {}
    for n, (W, b) in var_dict.items():
        d[n + '/Weight'], d[n + '/Bias'] = np.asarray(W), np.asarray(b)
    np.savez(path, **d)
