"""MI355X-native Fisher / entropy query scoring with the signatures of jsourati/nn-active-learning.

Module names mirror the reference's (`PW_NNAL`, `PW_NN`, `NNAL_tools`, `patch_utils`, `NN`,
`NN_extended`); `device` holds the session/model objects, `pool_shard` the multi-GPU pool
sharding.  The directory name is not an importable identifier: import through the `nnal_amd`
shim at the repository root."""
from . import _lib  # noqa: F401
from . import device, patch_utils, NNAL_tools, PW_NN, PW_NNAL, NN, NN_extended, pool_shard, model_utils, nrrd_io, PW_AL, NNAL  # noqa: F401

__all__ = ['device', 'patch_utils', 'NNAL_tools', 'PW_NN', 'PW_NNAL', 'NN', 'NN_extended', 'pool_shard', 'model_utils', 'nrrd_io', 'PW_AL', 'NNAL']
