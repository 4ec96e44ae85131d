"""`NN_extended.CNN`-schema models on the device (reference: NN_extended.py:20-295): layer dict
``{name: [type, specs, op_order]}`` with conv / conv_transpose / pool / fc in 2-D or 3-D, 'con'
skip connections, fc head (the form `get_gradients` is defined for, NN_extended.py:1025)."""
from .device import DeviceModel, default_session


class CNN(DeviceModel):
    """NN_extended.CNN(x, layer_dict, name, skips, feature_layer, dropout, probes, **kwargs) with
    `x` replaced by the placeholder SHAPE ((H,W,C) or (D,H,W,C))."""

    def __init__(self, in_shape, layer_dict, name, skips=[], feature_layer=None, dropout=None,
                 probes=[[], []], sess=None, max_batch=256, **kwargs):
        for key in kwargs:
            if key not in ('activation',) or kwargs[key] != 'ReLU':
                raise NotImplementedError('hyper-parameter %r belongs to training, outside the scored path' % key)
        super(CNN, self).__init__(sess or default_session(), layer_dict, in_shape, skips, feature_layer,
                                  dropout, max_batch, name)
