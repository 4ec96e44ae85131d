"""Oracle restatement of the reference's training step.  TEST INFRASTRUCTURE (see oracle/__init__.py).

`model.train_step` = tf.train.GradientDescentOptimizer(lr).minimize(loss) or AdamOptimizer(lr).minimize(loss)
(NN.py:591-615) on loss = mean softmax cross-entropy (NN.py:583-588), optionally restricted to `train_layers`.
TensorFlow is absent: the update rules are restated from the TF-1.x documentation of the two optimizers
(GradientDescent: theta -= lr * g;  Adam, defaults beta1 = .9, beta2 = .999, epsilon = 1e-8:
lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t);  m = beta1 m + (1 - beta1) g;  v = beta2 v + (1 - beta2) g^2;
theta -= lr_t * m / (sqrt(v) + epsilon)) - parity unpinned by the reference, like every TF op."""
import numpy as np
import torch


class OracleOptimizer(object):
    def __init__(self, model, learning_rate, train_layers=(), optimizer_name='SGD'):
        self.model = model
        self.lr = float(learning_rate)
        self.name = optimizer_name
        self.train_layers = list(train_layers)
        self.t = 0
        self.m = None
        self.v = None

    def step(self, x, y_onehot, drop=None):
        """One sess.run(model.train_step, {x, y_, keep_prob}); returns the loss before the step."""
        mdl = self.model
        loss, grads = mdl.loss_and_grads(x, y_onehot, drop)
        names = list(mdl.params.keys())
        self.t += 1
        if self.m is None:
            self.m = [np.zeros_like(g) for g in grads]
            self.v = [np.zeros_like(g) for g in grads]
        dt = grads[0].dtype
        for k, g in enumerate(grads):
            if self.train_layers and names[k // 2] not in self.train_layers:
                continue
            p = mdl.params[names[k // 2]][k % 2]
            th = p.detach().numpy().copy()
            if self.name == 'SGD':
                th = th - dt.type(self.lr) * g
            else:
                b1, b2, eps = dt.type(0.9), dt.type(0.999), dt.type(1e-8)
                lr_t = dt.type(self.lr * np.sqrt(1.0 - 0.999 ** self.t) / (1.0 - 0.9 ** self.t))
                self.m[k] = b1 * self.m[k] + (dt.type(1) - b1) * g
                self.v[k] = b2 * self.v[k] + (dt.type(1) - b2) * g * g
                th = th - lr_t * self.m[k] / (np.sqrt(self.v[k]) + eps)
            with torch.no_grad():
                p.copy_(torch.as_tensor(th))
        return loss
