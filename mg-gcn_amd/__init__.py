"""mg-gcn_amd -- MI355X-native engine for the MG-GCN hot path.

The directory name is not a Python identifier; load it with
``__graft_entry__.load_package()`` (registers it as ``mg_gcn_amd``).

Layout:
  csrc/        HIP kernels + the C ABI of include/mggcn.h  -> lib/libmggcn_hip.so
  host/        C++17 header layer with the reference's class names + the mg_gcn CLI
  _lib.py      ctypes binding (fails loudly when the engine is missing; no CPU path)
  matrix.py    context / csr_matrix / dn_matrix        (reference src/matrix.hpp)
  ops.py       matmul / get_matmul_buffer / kernels     (reference src/cuda_utils.hpp)
  gcn.py       sparse_linear / linear / gcn_layer / gcn (reference src/gcn.hpp)
  dist.py      1D row partition, one process per GPU    (reference src/dist_matrix.hpp, gcn.hpp dist_*)
  datasets.py  on-disk format + synthetic generators    (reference test/data/prep.py)
"""
from . import _lib, datasets                                   # noqa: F401
from ._lib import engine_error                                  # noqa: F401
from . import matrix                                            # noqa: F401
from .matrix import context, csr_matrix, dn_matrix, host_scalars, matrix_error  # noqa: F401
from . import ops                                               # noqa: F401
from .ops import get_matmul_buffer, matmul                      # noqa: F401
from .gcn import (gcn, gcn_layer, linear, softmax, softmax_cross_entropy_loss,  # noqa: F401
                  sparse_linear)

from . import dist                                              # noqa: F401
from .dist import (dist_context, dist_gcn, dist_gcn_layer, dist_row_csr_matrix,  # noqa: F401
                   dist_row_dn_matrix, dist_row_linear, dist_sparse_linear, repl_dn_matrix)

__all__ = ["context", "csr_matrix", "dn_matrix", "matrix_error", "engine_error", "ops", "matmul",
           "get_matmul_buffer", "sparse_linear", "linear", "gcn_layer", "softmax",
           "softmax_cross_entropy_loss", "gcn", "datasets"]
