"""Patch gather / index maps of the scoring path (reference: patch_utils.py).

`get_patches`, `get_patches_multimg` run on the device (alq_gather_normalize); `global2local_inds`
is host index arithmetic like the reference's."""
import ctypes as C

import numpy as np

from . import device
from ._lib import check


def patch_radii(patch_shape):
    # int((d-1)/2.): patch_utils.py:1119-1121
    return [int((d - 1) / 2.) for d in patch_shape]


class DeviceVolumes(object):
    """m zero-padded modalities resident in HBM (uploaded once per batch_eval / query call)."""

    def __init__(self, sess, padded_imgs):
        torch = sess.torch
        self.sess = sess
        arrs = [np.asarray(v) for v in padded_imgs]
        shp = arrs[0].shape
        if len(shp) != 3 or any(a.shape != shp for a in arrs):
            raise ValueError('modalities must be 3-D volumes of one shape')
        self.is_f64 = not all(a.dtype == np.float32 for a in arrs)
        dt = torch.float64 if self.is_f64 else torch.float32
        self.tensors = [sess.to_device(a, dt) for a in arrs]
        self.pad_dims = tuple(int(v) for v in shp)
        self.m = len(arrs)

    def gather(self, inds, patch_shape, stats=None, quirk=2, out_f64=False):
        """-> device tensor [n, d1, d2, m*d3]; quirk: 0 slab stats, 1 channel-index stats, 2 none."""
        sess, torch = self.sess, self.sess.torch
        inds = np.ascontiguousarray(np.asarray(inds, dtype=np.int64))
        n = int(inds.shape[0])
        d1, d2, d3 = [int(v) for v in patch_shape]
        out = sess.empty((n, d1, d2, self.m * d3), torch.float64 if out_f64 else torch.float32)
        if n == 0:
            return out
        r = patch_radii(patch_shape)
        orig = [self.pad_dims[a] - 2 * r[a] for a in range(3)]
        if inds.min() < 0 or inds.max() >= orig[0] * orig[1] * orig[2]:
            raise IndexError('voxel index outside the un-padded volume %r' % (orig,))
        sess.bind_stream()
        d_inds = sess.to_device(inds, torch.int64)
        ptrs = (C.c_void_p * self.m)(*[t.data_ptr() for t in self.tensors])
        pd = (C.c_int64 * 3)(*self.pad_dims)
        ps = (C.c_int32 * 3)(d1, d2, d3)
        st = None
        if quirk != 2:
            flat = np.asarray(stats, dtype=np.float64).reshape(-1)[:2 * self.m]
            st = (C.c_double * (2 * self.m))(*flat)
        check(sess.lib.alq_gather_normalize(sess.ctx, ptrs, self.m, 1 if self.is_f64 else 0, pd,
                                            C.c_void_p(d_inds.data_ptr()), n, ps, st, quirk,
                                            1 if out_f64 else 0, C.c_void_p(out.data_ptr())))
        return out


def get_patches(imgs, inds, patch_shape, padded=True, mask=None):
    """patch_utils.get_patches (patch_utils.py:1087-1173): float64 [b, d1, d2, m*d3] (+ labels)."""
    sess = device.default_session()
    r = patch_radii(patch_shape)
    vols = list(imgs)
    if not padded:
        vols = [np.pad(np.asarray(v), [(r[0], r[0]), (r[1], r[1]), (r[2], r[2])], 'constant') for v in vols]
    dv = DeviceVolumes(sess, vols)
    patches = dv.gather(inds, patch_shape, None, 2, out_f64=True).cpu().numpy()
    if mask is not None:
        orig = tuple(dv.pad_dims[a] - 2 * r[a] for a in range(3))
        return patches, np.asarray(mask)[np.unravel_index(np.asarray(inds), orig)]
    return patches


def get_patches_multimg(all_padded_imgs, img_inds, patch_shape, stats):
    """patch_utils.get_patches_multimg (patch_utils.py:1175-1212): per subject, slab-normalised."""
    sess = device.default_session()
    m = len(all_padded_imgs[0]) - 1
    s = len(img_inds)
    r = patch_radii(patch_shape)
    b_patches = [[] for _ in range(s)]
    b_labels = [[] for _ in range(s)]
    stats = np.asarray(stats, dtype=np.float64)
    for j in range(s):
        if len(img_inds[j]) == 0:
            continue
        dv = DeviceVolumes(sess, all_padded_imgs[j][:m])
        b_patches[j] = dv.gather(img_inds[j], patch_shape, stats[j, :2 * m], 0, out_f64=True).cpu().numpy()
        orig = tuple(dv.pad_dims[a] - 2 * r[a] for a in range(3))
        b_labels[j] = np.asarray(all_padded_imgs[j][m])[np.unravel_index(np.asarray(img_inds[j]), orig)]
    return b_patches, b_labels


def global2local_inds(batch_inds, set_sizes):
    """patch_utils.global2local_inds (patch_utils.py:829-866): positions in the concatenation of
    ordered sets -> per-set local positions, input order kept inside each set."""
    batch_inds = np.asarray(batch_inds)
    sizes = np.asarray(set_sizes, dtype=np.int64)
    ends = np.cumsum(sizes)
    owner = np.searchsorted(ends, batch_inds, side='right')
    return [batch_inds[owner == k] - (ends[k] - sizes[k]) for k in range(len(sizes))]
