"""Minimal stand-in for `mpi4py` -- ONLY for boxes without it (put `hugectr_amd/compat` on
PYTHONPATH).  The reference's scripts do `from mpi4py import MPI` because its multi-node launch is
MPI; most read nothing but `MPI.COMM_WORLD.Get_size() / Get_rank()` as the NODE count / node index
(R/test/embedding_collection_test/dgx_a100_one_hot.py:161-165).  Here nodes = 1: one node, the GPUs
of which are driven by one process per GPU under torch.distributed."""
import os as _os


class _Comm:
    def Get_size(self):
        return int(_os.environ.get("HCTR_NUM_NODES", "1"))

    def Get_rank(self):
        return int(_os.environ.get("HCTR_NODE_RANK", "0"))

    def Barrier(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            dist.barrier()

    barrier = Barrier


class _MPI:
    COMM_WORLD = _Comm()


MPI = _MPI()
