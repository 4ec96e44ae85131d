"""The one function of the reference's model_utils.py that reuses the scoring kernels."""
import numpy as np


def diagonal_Fisher(model, sess, batch_dat):
    """model_utils.diagonal_Fisher (model_utils.py:294-330): per parameter, the mean over the samples of the squared
    gradient of the sample's loss (= the squared gradient of log posteriors[label]); `batch_dat` = (x [N, ...],
    one-hot labels [c, N]).  Returns the arrays in variable order and TF shapes, like the reference's list.
    The reference runs one sess.run per sample; here the per-sample gradients of a device pass are squared and
    accumulated on the device (alq_param_grads + alq_sq_accum)."""
    x, y = batch_dat
    y = np.asarray(y)
    labels = np.where(y.sum(0) > 0, y.argmax(0), -1)
    if (labels < 0).any():
        raise ValueError('every sample needs a label (one-hot column)')
    return model.diagonal_fisher(np.asarray(x), labels)
