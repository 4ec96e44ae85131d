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
    `curr_weights_<iter>` after every fine-tune (:896-898; HDF5 like the reference's when h5py is importable, else the .npz twin).  Resume =
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
        """`curr_weights_<it>.h5` like the reference (PW_AL.py:896-898) when h5py is importable - an existing file of either
        kind wins, so a directory started with one format is resumed in it - else the .npz twin."""
        from . import weights_io
        h5 = os.path.join(self.root, 'curr_weights_%d.h5' % it)
        nz = os.path.join(self.root, 'curr_weights_%d.npz' % it)
        if os.path.exists(nz) or (not weights_io.have_h5py() and not os.path.exists(h5)):
            return nz
        return h5


def _plain(obj, where='pars'):
    """Parameter values as plain Python for the YAML files: numpy scalars -> float / int / bool, arrays -> nested lists,
    OrderedDict -> dict, tuples stay tuples (`patch_shape`).  Anything else fails HERE, on the writing rank at write time, with
    the key that holds it - not on the reading ranks after the barrier."""
    if isinstance(obj, np.generic):        # (first: np.float64 is also a float)
        return obj.item()
    if obj is None or isinstance(obj, (bool, int, float, str)):
        return obj
    if isinstance(obj, np.ndarray):
        if obj.dtype.hasobject:
            raise TypeError('%s: an object array cannot go into a parameter file' % where)
        return obj.tolist()
    if isinstance(obj, dict):
        return {_plain(k, where): _plain(v, '%s[%r]' % (where, k)) for k, v in obj.items()}
    if isinstance(obj, tuple):
        return tuple(_plain(v, where) for v in obj)
    if isinstance(obj, list):
        return [_plain(v, where) for v in obj]
    raise TypeError('%s: a %s cannot go into a parameter file (plain numbers, strings, lists, tuples, dicts and numpy '
                    'scalars / arrays can)' % (where, type(obj).__name__))


def _yaml_dump(obj, f):
    """The writer that matches _yaml_load: yaml.safe_dump of _plain(obj) plus the tuple tag yaml.dump gives `patch_shape` in
    the reference's files (PW_AL.py:91-110) - what this writes, every rank can read back."""
    import yaml

    class Dumper(yaml.SafeDumper):
        pass
    Dumper.add_representer(tuple, lambda d, data: d.represent_sequence('tag:yaml.org,2002:python/tuple', data))
    yaml.dump(_plain(obj), f, Dumper=Dumper)


def _yaml_load(f):
    """yaml.load of the reference's parameter files (PW_AL.py:91-113) without arbitrary object construction: a SafeLoader that
    additionally understands what yaml.dump puts into such files - tuples (`patch_shape`), numpy scalars and arrays
    (`pars['stats']`, np.float64 learning rates: decoded from their bytes, nothing is called), OrderedDict.  A file in a shared
    experiment directory cannot run code when it is loaded; any other python tag raises a ValueError that names it."""
    import yaml

    class Loader(yaml.SafeLoader):
        pass

    def parts(ld, node):
        if isinstance(node, yaml.SequenceNode):
            return ld.construct_sequence(node, deep=True), None
        m = ld.construct_mapping(node, deep=True)
        return list(m.get('args', [])), m.get('state')

    def apply(ld, suffix, node):
        name = suffix.replace('numpy._core.', 'numpy.core.')
        if name == 'numpy.dtype':
            args, state = parts(ld, node)
            dt = np.dtype(str(args[0]))
            if dt.hasobject:
                raise yaml.constructor.ConstructorError(None, None, 'object dtype in a parameter file', node.start_mark)
            if state is not None and len(state) > 1 and state[1] in ('<', '>', '=', '|'):
                dt = dt.newbyteorder(state[1])
            return dt
        if name == 'numpy.core.multiarray.scalar':
            args, _ = parts(ld, node)
            return np.frombuffer(bytes(args[1]), dtype=args[0])[0].item()
        if name == 'numpy.core.multiarray._reconstruct':
            _, state = parts(ld, node)
            _ver, shape, dt, fortran, raw = state
            return np.frombuffer(bytes(raw), dtype=dt).reshape(tuple(shape), order='F' if fortran else 'C').copy()
        if name == 'collections.OrderedDict':
            args, _ = parts(ld, node)
            return {k: v for k, v in (args[0] if args else [])}
        raise yaml.constructor.ConstructorError(None, None, 'python tag %r is not one a parameter file may hold' % suffix, node.start_mark)

    def pyname(ld, suffix, node):
        if suffix == 'numpy.ndarray':
            return np.ndarray
        raise yaml.constructor.ConstructorError(None, None, 'python name %r is not one a parameter file may hold' % suffix, node.start_mark)

    Loader.add_constructor('tag:yaml.org,2002:python/tuple', lambda ld, node: tuple(ld.construct_sequence(node, deep=True)))
    Loader.add_multi_constructor('tag:yaml.org,2002:python/object/apply:', apply)
    Loader.add_multi_constructor('tag:yaml.org,2002:python/name:', pyname)
    try:
        return yaml.load(f, Loader=Loader)
    except yaml.constructor.ConstructorError as e:
        raise ValueError('%s: not a parameter file this build reads (plain YAML + tuples + numpy scalars / arrays): %s'
                         % (getattr(f, 'name', 'parameter file'), e))


def _rank0_then_all(fn):
    """Runs fn() on rank 0 only; a failure there is raised on EVERY rank (the others would otherwise wait in the barrier for a
    writer that is gone, or run on into the next collective alone)."""
    from . import pool_shard
    rank, ws = pool_shard.world()
    err = None
    if rank == 0:
        try:
            fn()
        except Exception as e:          # noqa: BLE001 - re-raised below, on every rank
            err = e
    failed = pool_shard.max_over_ranks(1.0 if err is not None else 0.0) > 0.0
    if err is not None:
        raise err
    if failed:
        raise RuntimeError('rank 0 failed while writing the experiment files (see its traceback)')


# ------------------------------------------------------------------------------------------ the experiment object
class Experiment(object):
    """PW_AL.Experiment's constructor and parameter file (PW_AL.py:29-110): the experiment's root directory and
    `parameters.txt` (YAML).  The single-image `run_method` (:278-498) is not on the scored path's multi-image caller and
    is not mirrored."""

    def __init__(self, root_dir, pars={}):
        from . import pool_shard
        self.root_dir = root_dir
        self.nclass = 2
        # one process per GPU: rank 0 creates the directory and writes the files, the others wait and read them back (a peer
        # that raced the writer would load a half-written parameters.txt)
        rank, ws = pool_shard.world()

        def write():
            os.makedirs(root_dir, exist_ok=True)
            if len(pars) > 0:
                if os.path.exists(os.path.join(root_dir, 'parameters.txt')):
                    print("Some parameters already exist")
                else:
                    self.save_parameters(pars)
        _rank0_then_all(write)
        pool_shard.barrier()
        if rank != 0 and len(pars) > 0 and os.path.exists(os.path.join(root_dir, 'parameters.txt')):
            self.load_parameters()

    def save_parameters(self, pars):
        plain = _plain(pars)          # (an unsupported value fails before the file is touched)
        with open(os.path.join(self.root_dir, 'parameters.txt'), 'w') as f:
            _yaml_dump(plain, f)
        # every rank - the writer included - holds what the FILE holds (arrays as lists): rank 0 keeping the caller's
        # numpy arrays while the others load plain lists made `pars['stats']`-style values differ in type across ranks
        self.load_parameters()

    def load_parameters(self):
        with open(os.path.join(self.root_dir, 'parameters.txt'), 'r') as f:
            self.pars = _yaml_load(f)


class Experiment_MultiImg(Experiment):
    """PW_AL.Experiment_MultiImg (PW_AL.py:586-898): active learning over several subjects, each a list of modality
    volumes + a mask (NaN = voxel to ignore).  `train_paths.txt`, `train_stats.txt` and, per method, `queries/<iter>`,
    `AL_running_times/dt_<iter>`, `curr_weights_<iter>` as the reference writes them (weights: HDF5 with h5py, else .npz).

    `run_method` is the reference's loop (:690-898): grid indices -> resume from queries/ -> load + pad -> model ->
    perform_assign_ops(init_weights_path) -> [query_multimg -> pool -> training bookkeeping -> files -> finetune_multimg
    -> weights] until max_queries.  One process per GPU under torch.distributed: every rank holds the volumes and the
    model, the query's device work is split by contiguous blocks of the pool (PW_NNAL.bin_uncertainty_filter_multimg,
    query_multimg), everything else is the same deterministic code on the same bits on every rank; rank 0 writes the
    files."""

    def __init__(self, root_dir, pars={}, train_paths={}, test_paths={}):
        from . import pool_shard
        Experiment.__init__(self, root_dir, pars)
        if not hasattr(self, 'pars'):
            self.load_parameters()
        rank, ws = pool_shard.world()
        tr_file = os.path.join(self.root_dir, 'train_paths.txt')
        st_file = os.path.join(self.root_dir, 'train_stats.txt')
        def write():         # rank 0 is the writer; its peers read the finished files behind the barrier
            if not os.path.exists(tr_file):
                plain = _plain(train_paths, 'train_paths')
                with open(tr_file, 'w') as f:
                    _yaml_dump(plain, f)
            if not os.path.exists(st_file):
                with open(tr_file, 'r') as f:
                    np.savetxt(st_file, get_stats(_yaml_load(f)))
        _rank0_then_all(write)
        pool_shard.barrier()
        with open(tr_file, 'r') as f:
            self.train_paths = _yaml_load(f)
        self.train_stats = np.loadtxt(st_file)
        if self.train_stats.ndim == 1:                           # one subject: savetxt dropped the dimension (:626-631)
            self.train_stats = np.expand_dims(self.train_stats, axis=0)
        self.model_factory = None     # callable(expr, input_shape, sess) -> model for nets other than the reference's 'PW'

    def add_method(self, method_name):
        from . import pool_shard
        method_path = os.path.join(self.root_dir, method_name)
        if pool_shard.world()[0] == 0:
            for d in (method_path, os.path.join(method_path, 'queries'), os.path.join(method_path, 'AL_running_times')):
                os.makedirs(d, exist_ok=True)
        pool_shard.barrier()

    def _create_model(self, sess):
        m = len(self.train_paths[0]) - 1
        patch_shape = tuple(self.pars['patch_shape'][:2]) + (m * self.pars['patch_shape'][2],)
        if self.model_factory is not None:
            return self.model_factory(self, patch_shape, sess)
        return NN.create_model(self.pars['model_name'], self.pars['dropout_rate'], self.nclass, self.pars['learning_rate'],
                               self.pars['grad_layers'], self.pars['train_layers'], self.pars['optimizer_name'], patch_shape,
                               sess=sess)

    def run_method(self, method_name, max_queries, sess=None):
        import time
        from . import PW_NNAL, device, pool_shard
        rank, _ = pool_shard.world()
        method_path = os.path.join(self.root_dir, method_name)
        state = LoopState(method_path)
        # pool indices (:696-707)
        if 'pool_paths' in self.pars:
            pool_inds = [[] for _ in range(len(self.train_paths))]
            for i in self.pars['pool_paths']:
                pool_inds[i] = gen_multimg_inds([self.train_paths[i]], self.pars['grid_spacing'])[0][0]
        else:
            pool_inds, _ = gen_multimg_inds(self.train_paths, self.pars['grid_spacing'])
        pool_inds = [[int(v) for v in p] for p in pool_inds]
        # initial training indices (:709-720), then the queries already on disk (:721-735; files in iteration order)
        init_training_inds = [[] for _ in range(len(self.train_paths))]
        init_train_path = os.path.join(self.root_dir, 'init_train_inds.txt')
        if os.path.exists(init_train_path):
            init_inds = np.int32(np.loadtxt(init_train_path, ndmin=2))
            for ind in np.unique(init_inds[:, 1]):
                init_training_inds[ind] += init_inds[init_inds[:, 1] == ind, 0].tolist()
        training_inds = [[] for _ in range(len(self.train_paths))]
        iters = state.iters_done()
        for it in range(iters):
            Qs = np.int64(np.loadtxt(os.path.join(method_path, 'queries', '%d' % it), ndmin=2))
            for ind in np.unique(Qs[:, 1]):
                I = Qs[Qs[:, 1] == ind, 0]
                training_inds[ind] += I.tolist()
                for v in I:
                    pool_inds[ind].remove(int(v))
        pool_shard.barrier()                                   # every rank has read the state before rank 0 adds to it
        # volumes (:737-761)
        all_padded_imgs = [load_and_pad(sub, self.pars['patch_shape']) for sub in self.train_paths]
        # model (:763-798)
        sess = sess or device.default_session()
        model = self._create_model(sess)
        model.add_assign_ops()
        init = self.pars['init_weights_path']
        if iters > 0 and os.path.exists(state.weights_path(iters)):
            init = state.weights_path(iters)                   # resume: the weights the last complete iteration left
        model.perform_assign_ops(init, sess)
        nqueries = 0
        log = []
        while nqueries < max_queries:
            self.labeled_paths = self.train_paths               # (:818-821; the core-set bootstrap from a private data set, :806-816, is not mirrored)
            labeled_inds = training_inds
            self.labeled_stats = self.train_stats
            t1 = time.time()
            Q_inds = PW_NNAL.query_multimg(self, model, sess, all_padded_imgs, pool_inds, labeled_inds, method_name)
            dt = time.time() - t1
            nQ = int(np.sum([len(q) for q in Q_inds]))
            if nQ == 0:
                break                                           # an exhausted pool would spin forever in the reference
            nqueries += nQ
            Q_mat = np.zeros((nQ, 2))
            cnt = 0
            for ind in range(len(Q_inds)):
                q = np.asarray(Q_inds[ind], dtype=np.int64)
                if len(q) > 0:
                    vox = np.array(pool_inds[ind])[q]
                    Q_mat[cnt:cnt + len(q), 0] = vox
                    Q_mat[cnt:cnt + len(q), 1] = ind
                    cnt += len(q)
                    training_inds[ind] += list(vox)
                    for i in -np.sort(-q):                      # from the back, so positions stay valid (:880-882)
                        pool_inds[ind].pop(int(i))
            if rank == 0:
                state.save_round(iters, Q_mat, dt)
            iters += 1
            finetune_multimg(self, model, sess, all_padded_imgs, training_inds)
            if rank == 0:
                model.save_weights(state.weights_path(iters))
            pool_shard.barrier()
            log.append(dict(Q_mat=Q_mat.astype(np.int64), seconds=dt, pool_left=int(np.sum([len(p) for p in pool_inds]))))
        self.model = model
        return log
