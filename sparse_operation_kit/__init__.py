"""`import sparse_operation_kit as sok` -- the package name of the reference's SparseOperationKit
(R/sparse_operation_kit/sparse_operation_kit/__init__.py), served by `hugectr_amd.sok`: the same
lookup-op surface (`init`, `Variable`, `DynamicVariable`, `lookup_sparse`, `OptimizerWrapper`,
`dump` / `load`, ...) on PyTorch tensors (the reference's is on TensorFlow tensors, which this
stack does not carry)."""
from hugectr_amd import sok as _impl
from hugectr_amd.sok import *  # noqa: F401,F403

for _n in dir(_impl):
    if not _n.startswith("_"):
        globals()[_n] = getattr(_impl, _n)
