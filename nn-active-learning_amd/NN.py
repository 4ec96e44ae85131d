"""`NN.CNN`-schema models on the device (reference: NN.py:56-188 class CNN, :1217-1245
create_model, :1319-1359 create_PW1)."""
from collections import OrderedDict

import numpy as np

from .device import DeviceModel, default_session


def gen_batch_inds(data_size, batch_size):
    """NN.gen_batch_inds (NN.py:1529-1555): one random permutation (global NumPy stream) cut into batches of
    `batch_size`, the remainder as a last shorter batch."""
    quot, rem = np.divmod(data_size, batch_size)
    rand_perm = np.random.permutation(data_size).tolist()
    batches = [rand_perm[i * batch_size:(i + 1) * batch_size] for i in range(quot)]
    if rem > 0:
        batches += [rand_perm[-rem:]]
    return batches


def _imread(path):
    """The image at `path` as float64 [H, W, C].  The reference decodes with OpenCV (`cv2.imread`, NN.py:1493; its
    `import cv2` is commented out, NN.py:9); OpenCV is not in this image, so image files are NumPy arrays on disk
    (`.npy`).  Tests may rebind this name."""
    if not str(path).endswith('.npy'):
        raise NotImplementedError('image decoding needs OpenCV (absent): store images as .npy arrays, got %r' % (path,))
    return np.float64(np.load(path))


def load_winds(inds, imgs_path_file, target_shape, mean=None, labels_file=None):
    """NN.load_winds (NN.py:1479-1527): the images whose paths stand on lines inds[i] + 1 of `imgs_path_file` as one
    float64 array [n, H, W, C], `mean` subtracted when truthy, + their labels from `labels_file`.  `cv2.resize` is not
    available: an image must already have `target_shape`."""
    import linecache
    inds = np.asarray(inds)
    imgs, labels = [], []
    for i in inds:
        path = linecache.getline(imgs_path_file, int(i) + 1).splitlines()[0]
        img = _imread(path)
        if tuple(img.shape[:2]) != tuple(target_shape):
            raise NotImplementedError('resizing %r -> %r needs OpenCV (absent)' % (img.shape[:2], tuple(target_shape)))
        if mean:
            img = img - mean
        imgs.append(img)
        if labels_file:
            labels.append(int(linecache.getline(labels_file, int(i) + 1).splitlines()[0]))
    return np.stack(imgs), labels


class CNN(DeviceModel):
    """NN.CNN(x, layer_dict, name, feature_layer, dropout, probes): same layer-dict schema
    ``{name: [depth,'conv',[kh,kw]] | [depth,'fc'] | [[window,stride],'pool']}``; `x` is replaced
    by the placeholder SHAPE (H, W, C)."""

    def __init__(self, in_shape, layer_dict, name, feature_layer=None, dropout=None, probes=(),
                 sess=None, max_batch=256):
        super(CNN, self).__init__(sess or default_session(), layer_dict, in_shape, (), feature_layer,
                                  dropout, max_batch, name)
        self.probes = list(probes)

    def extract_features(self, inds, expr, session):
        """NN.CNN.extract_features (NN.py:522-554): feature_layer [d, n] of the images `inds`, in random batches of
        expr.pars['batch_size'] (one batch of everything, and no RNG draw, when batch_size > n)."""
        inds = np.asarray(inds)
        n = len(inds)
        features = np.zeros((self.feature_dim, n))
        batch_size = expr.pars['batch_size']
        batches = [np.arange(n).tolist()] if batch_size > n else gen_batch_inds(n, batch_size)
        for inner in batches:
            X, _ = load_winds(inds[inner], expr.imgs_path_file, expr.pars['target_shape'], expr.pars['mean'])
            features[:, inner] = session.run(self.feature_layer, feed_dict={self.x: X, self.keep_prob: 1.})
        return features


def pw1_layer_dict(nclass):
    """The layer dict of create_PW1 (NN.py:1328-1336)."""
    return OrderedDict([
        ('conv1', [24, 'conv', [5, 5]]), ('conv2', [32, 'conv', [5, 5]]), ('max1', [[2, 2], 'pool']),
        ('conv3', [48, 'conv', [3, 3]]), ('conv4', [96, 'conv', [3, 3]]), ('max2', [[2, 2], 'pool']),
        ('fc1', [4096, 'fc']), ('fc2', [4096, 'fc']), ('fc3', [nclass, 'fc'])])


def create_PW1(nclass, dropout_rate, learning_rate, optimizer_name, patch_shape, sess=None,
               max_batch=256):
    """NN.create_PW1 (NN.py:1319-1359): the patch-wise net of PW_AL; feature layer = fc2
    (index len-2, :1346); dropout on layers 6-8 is identity at keep_prob = 1."""
    d = pw1_layer_dict(nclass)
    model = CNN(tuple(patch_shape), d, 'PatchWise', len(d) - 2, [[6, 7, 8], dropout_rate], [5], sess, max_batch)
    model.get_optimizer(learning_rate, [], optimizer_name)      # NN.py:1354
    model.get_gradients()                                        # NN.py:1357
    return model


def create_model(model_name, dropout_rate, nclass, learning_rate, grad_layers=[], train_layers=[],
                 optimizer_name='SGD', patch_shape=None, sess=None, max_batch=256):
    """NN.create_model (NN.py:1217-1245), 'PW' only."""
    if model_name != 'PW':
        raise NotImplementedError("model %r: only the patch-wise 'PW' net is on the scored path" % model_name)
    # like the reference's 'PW' branch (NN.py:1238-1243), grad_layers / train_layers are not forwarded: all layers
    return create_PW1(nclass, dropout_rate, learning_rate, optimizer_name, patch_shape, sess, max_batch)
