import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # the HIP library is a build artefact (git-ignored): build it when a fresh checkout has none
    lib = os.path.join(ROOT, "hugectr_amd", "libhugectr_amd.so")
    if not os.path.exists(lib):
        import __graft_entry__
        __graft_entry__.build()


    # HCTR_EMU=1: the `-m gpu` tests on a machine without a GPU, the kernels' own source run by the
    # host interpreter of tests/emu (a logic pre-flight; see tests/emu/fakecuda.py)
    if os.environ.get("HCTR_EMU") == "1":
        sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
        import fakecuda
        fakecuda.install(os.environ.get("HCTR_EMU_VARIANT"))
        site = os.path.join(ROOT, "tests", "emu", "site")  # (worker processes: sitecustomize)
        os.environ["PYTHONPATH"] = site + os.pathsep + os.environ.get("PYTHONPATH", "")


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def _node_seed(nodeid):
    """32-bit seed of a test: a hash of its node id, mixed with HCTR_TEST_SEED (default 0) so the
    whole suite can be replayed under other seeds (`HCTR_TEST_SEED=3 pytest -m gpu`)."""
    import zlib
    base = int(os.environ.get("HCTR_TEST_SEED", "0"))
    return (zlib.crc32(nodeid.encode()) ^ (base * 0x9E3779B1)) & 0x7FFFFFFF


@pytest.fixture(autouse=True)
def _seed_every_test(request):
    """No test may depend on the process's RNG state: python / numpy / torch (CPU and every GPU)
    global generators are seeded per test from the test's node id, so a test draws the same
    numbers whether it runs alone, after 29 others, or on a box it has never seen."""
    import random
    import numpy as np
    seed = _node_seed(request.node.nodeid)
    random.seed(seed)
    np.random.seed(seed)
    try:
        import torch
        torch.manual_seed(seed)  # (also seeds every CUDA device's default generator)
    except Exception:
        pass
    yield
    # ... and no test may leave its handles to be destroyed at a random point of a LATER test: what a
    # test created (tables, hash indices, caches: their __del__ frees device memory and unmaps
    # address ranges) goes when the test ends.  One `-m gpu -x` run in sixty died of a segmentation
    # fault inside a dynamic table's destroy that the garbage collector had started from the call of
    # the NEXT test (profiles/r6_det_destroy_segfault.txt; not reproduced in 44 looped runs).
    import gc
    gc.collect()
    if "gpu" in request.node.keywords:
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.synchronize()
        except Exception:
            pass


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle
