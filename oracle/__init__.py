"""CPU oracle for the Fisher / entropy query-scoring path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (numpy + torch-CPU, fp32 with an fp64 switch)
of the reference's algorithm for the path named in BASELINE.json:north_star.
Each function cites the reference file:line it follows.  Nothing in the product
package (`nn-active-learning_amd/`) may import it: only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg do, and only as
the checker / the timed CPU baseline.

How parity is pinned
--------------------
* The reference holds NO golden vectors, tests or fixtures.
* Its host-side NumPy layers (`shrink_gradient`, `get_patches`,
  `get_patches_multimg`, `global2local_inds`, `binary_uncertainty_filter`,
  `compute_entropy`, `uncertainty_filtering`, `sample_query_dstr`) and its
  orchestration (`gen_A_matrices`, `batch_eval`, `bin_uncertainty_filter_multimg`,
  `CNN_query('entropy')`) were executed VERBATIM from /root/reference in the build
  container by `tests/golden/make_golden.py`; the outputs are committed under
  `tests/golden/*.npz` and `tests/test_oracle_golden.py` checks this oracle
  against them.
* The TensorFlow-1.x graph arithmetic itself (conv / pool / matmul / softmax /
  tf.gradients) is a third-party dependency that is absent from the image
  (tensorflow: version unpinned by the reference, API implies <= 1.15).  Its
  semantics are restated here from the documented op definitions; that layer is
  "parity unpinned" by the reference and pinned here only by (i) an independent
  loop-level numpy restatement (`oracle.tfops.naive_*`) and (ii) fp64
  finite-difference checks of the gradients (`tests/test_oracle_selfcheck.py`).
"""

from . import netspec, tfops, model, alpath, train  # noqa: F401
