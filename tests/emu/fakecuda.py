"""TEST INFRASTRUCTURE ONLY.  `HCTR_EMU=1 python -m pytest tests -m gpu ...` on a machine WITHOUT a
GPU: the `-m gpu` parity tests run unchanged, with

  * the C ABI served by the host interpreter's build of the kernels' own source
    (tests/emu/_build/libhctr_emu.so, see hipemu.h) instead of libhugectr_amd.so,
  * torch "cuda" tensors living in host memory (a TorchFunctionMode rewrites the device of every
    factory call / .to() / .cuda(); streams and events are inert objects).

It is a pre-flight for kernel and host LOGIC before GPU minutes are spent; it proves nothing about
the hardware, and the product never imports it (tests/conftest.py does, and only when HCTR_EMU=1).
Everything is monkeypatching from the outside: hugectr_amd/ has no switch for it."""
import contextlib
import ctypes
import os
import sys

import numpy as np


def _is_cuda_dev(d):
    import torch
    if isinstance(d, torch.device):
        return d.type == "cuda"
    return isinstance(d, str) and d.startswith("cuda")


class _Stream:
    cuda_stream = 0

    def __init__(self, *a, **k):
        pass

    def wait_event(self, e):
        pass

    def wait_stream(self, s):
        pass

    def synchronize(self):
        pass

    def record_event(self, e=None):
        return e or _Event()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class _Event:
    def __init__(self, *a, **k):
        pass

    def record(self, s=None):
        pass

    def wait(self, s=None):
        pass

    def synchronize(self):
        pass

    def query(self):
        return True

    def elapsed_time(self, other):
        return 1e-3


_installed = None


def install(variant=None):
    global _installed
    if _installed is not None:
        return _installed
    _installed = _install(variant)
    return _installed


def _install(variant=None):
    import torch
    from torch.overrides import TorchFunctionMode

    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    import emu
    elib = emu.load(variant, os.path.basename(os.path.normpath(variant))) if variant else emu.load()
    emu.bind(elib)

    # ---- the C ABI ------------------------------------------------------------------------------
    from hugectr_amd import _lib
    real = _lib.lib
    _lib.lib = elib
    _lib.stream_ptr = lambda: ctypes.c_void_p(0)
    import hugectr_amd  # noqa: F401
    import importlib
    for name in ("embedding", "layers", "dense", "dynamic_table", "cache", "unique_exchange",
                 "embedding_collection", "sok", "hugectr", "parallel", "data", "sharding"):
        try:
            importlib.import_module("hugectr_amd." + name)
        except Exception:  # (a module the tests do not need here)
            pass
    for mname, mod in list(sys.modules.items()):
        if mname.startswith("hugectr_amd") and mod is not None:
            if mod.__dict__.get("lib") is real:
                mod.lib = elib
            if hasattr(mod, "stream_ptr"):
                mod.stream_ptr = _lib.stream_ptr

    # ---- views of "device" memory by address ------------------------------------------------------
    def view_bytes(addr, nbytes):
        buf = (ctypes.c_char * nbytes).from_address(int(addr))
        return torch.from_numpy(np.frombuffer(buf, dtype=np.uint8))

    from hugectr_amd import embedding as _e

    def _view(self, addr, shape, dtype):
        n = 1
        for s in shape:
            n *= s
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        return view_bytes(addr, nbytes).view(dtype).view(*shape)

    _e.SparseEmbeddingHash._view = _view
    try:
        from hugectr_amd import sok as _s

        def _view_f32(addr, shape):
            n = 1
            for d in shape:
                n *= d
            return view_bytes(addr, n * 4).view(torch.float32).view(*shape)

        _s._view_f32 = _view_f32
    except Exception:
        pass

    # ---- torch.cuda -----------------------------------------------------------------------------
    tc = torch.cuda
    tc.is_available = lambda: True
    tc.device_count = lambda: 1
    tc.current_device = lambda: 0
    tc.set_device = lambda d: None
    tc.synchronize = lambda d=None: None
    tc.empty_cache = lambda: None
    tc.mem_get_info = lambda d=None: (64 << 30, 64 << 30)
    tc.get_device_name = lambda d=None: "host interpreter (tests/emu)"
    tc.current_stream = lambda d=None: _Stream()
    tc.default_stream = lambda d=None: _Stream()
    tc.Stream = _Stream
    tc.Event = _Event
    tc.stream = lambda s: contextlib.nullcontext()
    tc.device = lambda d: contextlib.nullcontext()
    tc.is_current_stream_capturing = lambda: False
    tc.graphs.is_current_stream_capturing = lambda: False
    os.environ.setdefault("HCTR_HIP_GRAPH", "0")  # (no capture here: the eager schedule runs)
    # (the interpreter keeps at most 128 workgroups alive at once: the index stage's cooperative
    #  finish kernel must not ask for more)
    os.environ.setdefault("HCTR_HT_FINISH_BLOCKS", "128")
    tc.manual_seed = lambda s: None
    tc.manual_seed_all = lambda s: None
    torch.Tensor.is_cuda = property(lambda self: True)
    _Gen = torch.Generator

    class _HostGenerator(_Gen):
        def __new__(cls, device=None):
            return _Gen.__new__(cls, device="cpu")

        def __init__(self, device=None):
            pass

    torch.Generator = _HostGenerator

    class FakeCuda(TorchFunctionMode):
        def __torch_function__(self, func, types, args=(), kwargs=None):
            kwargs = dict(kwargs or {})
            to_cuda = False
            if "device" in kwargs and kwargs["device"] is not None and _is_cuda_dev(kwargs["device"]):
                kwargs["device"] = "cpu"
                to_cuda = True
            if any(_is_cuda_dev(a) for a in args):
                args = tuple("cpu" if _is_cuda_dev(a) else a for a in args)
                to_cuda = True
            if func is torch.Tensor.cuda:
                return args[0].clone()  # (an upload is a copy: the source may change afterwards)
            if func is torch.Tensor.cpu:
                return args[0].clone()
            if func is torch.Tensor.pin_memory or func is torch.Tensor.is_pinned:
                return args[0] if func is torch.Tensor.pin_memory else True
            if func is torch.Tensor.to and to_cuda:
                kwargs.pop("non_blocking", None)
                return func(*args, **kwargs).clone()
            kwargs.pop("pin_memory", None) if "pin_memory" in kwargs else None
            return func(*args, **kwargs)

    mode = FakeCuda()
    mode.__enter__()
    return elib
