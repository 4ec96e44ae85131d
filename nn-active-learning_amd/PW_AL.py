"""The pieces of the reference's patch-wise AL loop (PW_AL.py) that sit either side of the query-scoring path:
volume statistics and grid indices in front of it (`get_stats`, `gen_multimg_inds`), the fine-tune and the on-disk
experiment state behind it (`finetune`, `finetune_multimg`, `queries/%d`, `AL_running_times/dt_%d`,
`curr_weights_%d`).  Signatures as in the reference; NumPy on the host like the reference's, patches and training
steps on the device (patch_utils.get_patches* gather, DeviceModel.train_on_batch)."""
import os

import numpy as np

from . import NN, nrrd_io, patch_utils


def _volume(v):
    """A path (NRRD, read like `nrrd.read(path)[0]`) or an array already in memory."""
    if isinstance(v, np.ndarray):
        return v
    return nrrd_io.read(v)[0]


def get_stats(paths):
    """PW_AL.get_stats (PW_AL.py:901-918): mean / std of every modality over the voxels whose mask is not NaN.
    The reference stores them at [i, j*m] and [i, j*m+1] (m = number of modalities; 2*j only for m in {1, 2}): kept."""
    m = len(paths[0]) - 1
    n = len(paths)
    stats = np.zeros((n, 2 * m))
    for i, dat_paths in enumerate(paths):
        mask = _volume(dat_paths[-1])
        for j in range(m):
            img = _volume(dat_paths[j])
            stats[i, j * m] = np.mean(img[~np.isnan(mask)])
            stats[i, j * m + 1] = np.std(img[~np.isnan(mask)])
    return stats


def gen_multimg_inds(dat_paths, grid_spacing):
    """PW_AL.gen_multimg_inds (PW_AL.py:921-975): per subject, the raveled (C-order) indices of the in-plane lattice
    {(x, y): x % spacing == 0, y % spacing == 0} repeated on every slice z - slice by slice, x-major inside a slice, the
    order the reference's meshgrid / ravel produces - with the voxels whose mask is NaN dropped.
    Returns (indices, labels), one list per subject."""
    all_inds, all_labels = [], []
    for sub in dat_paths:
        mask = _volume(sub[-1])
        nx, ny, nz = mask.shape
        xs = np.arange(0, nx, grid_spacing)
        ys = np.arange(0, ny, grid_spacing)
        gx = np.repeat(xs, len(ys))                  # lattice points of one slice, x-major
        gy = np.tile(ys, len(xs))
        plane = (gx * ny + gy) * nz                  # their raveled index at z = 0
        inds = (plane[None, :] + np.arange(nz)[:, None]).reshape(-1)          # slice after slice
        labels = mask[gx, gy, :].T.reshape(-1)
        keep = ~np.isnan(labels)
        all_inds.append(list(inds[keep]))
        all_labels.append(list(labels[keep]))
    return all_inds, all_labels


def load_and_pad(sub_paths, patch_shape):
    """PW_AL.py:737-761 / PW_NN.py:430-444: the m modalities zero-padded by the patch radii, the mask (last) as is."""
    rads = [int((patch_shape[i] - 1) / 2.) for i in range(3)]
    m = len(sub_paths) - 1
    out = []
    for i, p in enumerate(sub_paths):
        img = _volume(p)
        out.append(img if i == m else np.pad(img, ((rads[0], rads[0]), (rads[1], rads[1]), (rads[2], rads[2])), 'constant'))
    return out


def _hot(labels):
    hot = np.zeros((2, len(labels)))
    hot[0, np.asarray(labels) == 0] = 1
    hot[1, np.asarray(labels) == 1] = 1
    return hot


def finetune(model, sess, expr, padded_imgs, mask, train_inds):
    """PW_AL.finetune (PW_AL.py:1030-1088): `epochs` passes over random batches of `b` training voxels of one image;
    channel-index normalisation (:1069-1072), one train_step per batch at keep_prob = model.dropout_rate."""
    n, m = len(train_inds), len(padded_imgs)
    train_inds = np.asarray(train_inds)
    stats = expr.pars['stats']
    for _ in range(expr.pars['epochs']):
        for batch in NN.gen_batch_inds(n, expr.pars['b']):
            patches, labels = patch_utils.get_patches(padded_imgs, train_inds[batch], expr.pars['patch_shape'], True, mask)
            for j in range(m):
                patches[:, :, :, j] = (patches[:, :, :, j] - stats[j][0]) / stats[j][1]
            sess.run(model.train_step, feed_dict={model.x: patches, model.y_: _hot(labels), model.keep_prob: model.dropout_rate})


def finetune_multimg(expr, model, sess, all_padded_imgs, training_inds):
    """PW_AL.finetune_multimg (PW_AL.py:1091-1147): batches drawn over the concatenation of the subjects' training
    voxels, slab-normalised patches (get_patches_multimg), one train_step per batch."""
    s = len(training_inds)
    sizes = [len(training_inds[i]) for i in range(s)]
    n = int(np.sum(sizes))
    for _ in range(expr.pars['epochs']):
        for batch in NN.gen_batch_inds(n, expr.pars['b']):
            local = patch_utils.global2local_inds(batch, sizes)
            img_inds = [np.array(training_inds[j])[local[j]] for j in range(s)]
            b_patches, b_labels = patch_utils.get_patches_multimg(all_padded_imgs, img_inds, expr.pars['patch_shape'], expr.train_stats)
            b_patches = np.concatenate([b_patches[j] for j in range(s) if len(img_inds[j]) > 0], axis=0)
            b_labels = np.concatenate([b_labels[j] for j in range(s) if len(img_inds[j]) > 0])
            sess.run(model.train_step, feed_dict={model.x: b_patches, model.y_: _hot(b_labels), model.keep_prob: model.dropout_rate})


# ------------------------------------------------------------------------------------------ experiment state on disk
class LoopState(object):
    """The per-method directory of Experiment_MultiImg.run_method (PW_AL.py:690-898): `queries/<iter>` = rows
    [voxel index, subject index] (np.savetxt fmt '%d', :862-884), `AL_running_times/dt_<iter>` (:866-885) and
    `curr_weights_<iter>` after every fine-tune (:896-898; .npz twin of the HDF5 file, h5py is absent).  Resume =
    count the files in queries/ (:724-735)."""

    def __init__(self, root):
        self.root = root
        for d in ('queries', 'AL_running_times'):
            os.makedirs(os.path.join(root, d), exist_ok=True)

    def iters_done(self):
        return len(os.listdir(os.path.join(self.root, 'queries')))

    def save_round(self, it, Q_mat, dt):
        np.savetxt(os.path.join(self.root, 'queries', '%d' % it), np.asarray(Q_mat).reshape(-1, 2), fmt='%d')
        np.savetxt(os.path.join(self.root, 'AL_running_times', 'dt_%d' % it), [dt])

    def load_queries(self):
        """All queries so far as an int array [nq, 2] (np.loadtxt like :727-732), in iteration order."""
        rows = []
        for it in range(self.iters_done()):
            q = np.int64(np.loadtxt(os.path.join(self.root, 'queries', '%d' % it), ndmin=2))
            rows.append(q.reshape(-1, 2))
        return np.concatenate(rows) if rows else np.zeros((0, 2), np.int64)

    def weights_path(self, it):
        return os.path.join(self.root, 'curr_weights_%d.npz' % it)
