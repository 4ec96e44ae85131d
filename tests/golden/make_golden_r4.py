#!/usr/bin/env python3
"""Round-4 golden, produced by running the REFERENCE's own Python in the build container:

    python tests/golden/make_golden_r4.py   ->  tests/golden/r4_run_method.npz

`PW_AL.Experiment_MultiImg.__init__`, `add_method` and `run_method` (PW_AL.py:586-637, :679-688, :690-898) executed from
/root/reference, unmodified, on a synthetic two-subject experiment: the grid indices (`gen_multimg_inds`), the volume
statistics (`get_stats`), the resume scan of `queries/`, the pool -> training bookkeeping between query and fine-tune
(:857-884) and the `queries/<iter>` files it writes (np.savetxt '%d').

Stand-ins bound for that run (nothing the bookkeeping depends on):
  * `nrrd.read(path)` -> an in-memory table of arrays (the subjects are synthetic; pynrrd is absent);
  * `tf.reset_default_graph / tf.Session / tf.global_variables_initializer` -> no-ops, `NN.create_model` -> an object whose
    `add_assign_ops / perform_assign_ops / save_weights` do nothing (TensorFlow and h5py are absent; the bookkeeping never
    looks at the model);
  * `PW_NNAL.query_multimg` -> a seeded picker: per subject `k_s` distinct positions into the CURRENT pool list (what the real
    query returns: positions into `pool_inds[i]`, PW_NNAL.py:626-629), recorded with the pool it saw;
  * `finetune_multimg` -> a recorder of the training index lists it is handed;
  * `yaml.load(f)` -> `yaml.load(f, Loader=UnsafeLoader)`: the reference calls the one-argument form PyYAML >= 6 removed
    (PW_AL.py:113); same loader, same result.
Two calls: a fresh run of 3 iterations, then a second `run_method` on the same directory (the resume path: the query files
are read back and the pool is rebuilt, :724-735) for 2 more.  The reference reads the query files back in `os.listdir` order, which the file system chooses: the order this
run saw is recorded (`resume_listdir_order`) so that the resumed training lists can be compared block by block.  The file
holds data only."""
import copy
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden  # noqa: E402

SHAPES = [(11, 9, 5), (8, 12, 4)]
PARS = dict(grid_spacing=2, patch_shape=(5, 5, 3), model_name='PW', dropout_rate=1., learning_rate=1e-3, grad_layers=[],
            train_layers=[], optimizer_name='SGD', init_weights_path='init', k=4, B=10, lambda_=0., ntb=40, b=4, epochs=1)
PICKS = [3, 2]          # queries per subject and iteration


def subjects(seed=4200):
    rs = np.random.RandomState(seed)
    table, paths = {}, []
    for s_, shp in enumerate(SHAPES):
        sub = []
        for j in range(2):
            p = '/synthetic/sub%d_mod%d.nrrd' % (s_, j)
            table[p] = rs.randn(*shp) * (1. + j) + 0.3 * s_
            sub.append(p)
        mask = rs.randint(0, 2, size=shp).astype(np.float64)
        mask[rs.rand(*shp) < 0.15] = np.nan
        p = '/synthetic/sub%d_mask.nrrd' % s_
        table[p] = mask
        sub.append(p)
        paths.append(sub)
    return table, paths


def picker(seed):
    """The stand-in query: seeded positions into the current pool lists, the same stream for the reference run and the test."""
    rs = np.random.RandomState(seed)

    def pick(pool_inds):
        out = []
        for i, pool in enumerate(pool_inds):
            k = min(PICKS[i], len(pool))
            out.append(np.sort(rs.permutation(len(pool))[:k]).astype(np.int64) if k else np.zeros(0, np.int64))
        return out
    return pick


def main():
    make_golden.import_reference()
    sys.path.insert(0, make_golden.REF)
    import PW_AL
    table, paths = subjects()
    PW_AL.nrrd.read = lambda p: (table[p], None)
    import yaml
    _yload = yaml.load
    PW_AL.yaml.load = lambda f, Loader=None: _yload(f, Loader=Loader or yaml.UnsafeLoader)

    class _Graph(object):
        def finalize(self):
            pass

    class _Sess(object):
        graph = _Graph()

        def run(self, *a, **k):
            return None

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    class _TF(object):
        @staticmethod
        def reset_default_graph():
            pass

        @staticmethod
        def Session():
            return _Sess()

        @staticmethod
        def global_variables_initializer():
            return None

    class _Model(object):
        def add_assign_ops(self):
            pass

        def perform_assign_ops(self, path, sess):
            pass

        def save_weights(self, path):
            pass
    PW_AL.tf = _TF
    PW_AL.NN.create_model = lambda *a, **k: _Model()
    rec = {'pools_seen': [], 'train_seen': []}
    pick = picker(77)

    def fake_query(expr, model, sess, all_padded_imgs, pool_inds, labeled_inds, method_name):
        rec['pools_seen'].append(copy.deepcopy(pool_inds))
        return pick(pool_inds)

    def fake_finetune(expr, model, sess, all_padded_imgs, training_inds):
        rec['train_seen'].append(copy.deepcopy(training_inds))
    PW_AL.PW_NNAL.query_multimg = fake_query
    PW_AL.finetune_multimg = fake_finetune
    root = tempfile.mkdtemp(prefix='r4_run_method_')
    expr = PW_AL.Experiment_MultiImg(root, PARS, paths)
    expr.add_method('fi')
    per_iter = sum(PICKS)
    expr.run_method('fi', 3 * per_iter)           # iterations 0, 1, 2
    listing = [int(f) for f in os.listdir(os.path.join(root, 'fi', 'queries'))]   # the order the resume scan will see (:726-728)
    expr2 = PW_AL.Experiment_MultiImg(root)       # a new object on the same directory: parameters / paths / stats read back
    expr2.run_method('fi', 2 * per_iter)          # resume: iterations 3, 4
    out = dict(resume_listdir_order=np.array(listing), shapes=np.array(SHAPES), picks=np.array(PICKS), pick_seed=np.array(77), subject_seed=np.array(4200),
               train_stats=np.asarray(expr.train_stats), train_stats_reloaded=np.asarray(expr2.train_stats))
    for k, v in PARS.items():
        if isinstance(v, (int, float, tuple)):
            out['par_' + k] = np.array(v)
    qdir = os.path.join(root, 'fi', 'queries')
    assert sorted(os.listdir(qdir), key=int) == ['0', '1', '2', '3', '4']
    for it in range(5):
        out['queries_%d' % it] = np.int64(np.loadtxt(os.path.join(qdir, '%d' % it), ndmin=2))
        with open(os.path.join(qdir, '%d' % it)) as f:
            out['queries_text_%d' % it] = np.frombuffer(f.read().encode(), dtype=np.uint8)
        for s_ in range(2):
            out['pool_seen_%d_%d' % (it, s_)] = np.asarray(rec['pools_seen'][it][s_], dtype=np.int64)
            out['train_seen_%d_%d' % (it, s_)] = np.asarray(rec['train_seen'][it][s_], dtype=np.int64)
    assert sorted(os.listdir(os.path.join(root, 'fi', 'AL_running_times'))) == ['dt_%d' % i for i in range(5)]
    np.savez_compressed(os.path.join(HERE, 'r4_run_method.npz'), **out)
    print('r4_run_method: pool sizes seen', [[len(p) for p in ps] for ps in rec['pools_seen']],
          'training sizes', [[len(t) for t in ts] for ts in rec['train_seen']])


if __name__ == '__main__':
    main()
