"""rails_amd -- MI355X (gfx950) back end of the RAILS inner loop.

The product is librails_hip.so (hand-written HIP kernels behind the C ABI of include/rails_hip.h)
plus the header-only C++ wrappers in rails_amd/include/rails/ that drop into the reference's
templated Solver<Matrix, MultiVector, DenseMatrix>.  This Python package is the thin host-side
mirror used by the tests and bench.py.  There is no CPU fallback anywhere in this package.
"""
from ._lib import LIB_PATH, RailsError, load  # noqa: F401
from .solver import Solver  # noqa: F401
from .wrappers import (Context, HipMultiVectorWrapper, HipOperatorWrapper, lanczos_vectors,  # noqa: F401
                       resid_lanczos)
