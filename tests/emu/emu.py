"""TEST INFRASTRUCTURE ONLY: builds and loads the host interpreter's build of the kernels' source
(tests/emu/Makefile -> _build/libhctr_emu[_TAG].so) and hands out ctypes views.  "Device" memory
is host memory here: numpy arrays go in by pointer.  Never imported by hugectr_amd/."""
import ctypes
import os
import shutil
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CXX = "/opt/rocm/lib/llvm/bin/clang++"


def available():
    return os.path.exists(CXX) and shutil.which("make") is not None


def build(csrc=None, tag=None):
    args = ["make", "-C", HERE, "-j", str(min(8, os.cpu_count() or 1))]
    if csrc:
        args += [f"CSRC={os.path.abspath(csrc)}", f"TAG={tag}"]
    r = subprocess.run(args, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("host-interpreter build failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    return os.path.join(HERE, "_build", f"libhctr_emu{'_' + tag if tag else ''}.so")


_libs = {}


def load(csrc=None, tag=None):
    # (the interpreter keeps at most 128 workgroups alive at once -- HIPEMU_MAX_WORKERS: the index
    #  stage's cooperative finish kernel must not ask for more)
    os.environ.setdefault("HCTR_HT_FINISH_BLOCKS", "128")
    key = tag or ""
    if key not in _libs:
        lib = ctypes.CDLL(build(csrc, tag))
        lib.hctr_last_error.restype = ctypes.c_char_p
        lib.hctr_ht_table_size.restype = ctypes.c_size_t
        lib.hctr_radix_sort_temp_bytes.restype = ctypes.c_size_t
        lib.hctr_radix_sort_temp_bytes.argtypes = [ctypes.c_size_t]
        _libs[key] = lib
    return _libs[key]


def load_under_test():
    """the product's kernel source, or -- HCTR_EMU_VARIANT=<dir> -- the variant laid over it (a
    kernel variant goes through the same tests before it replaces anything)"""
    var = os.environ.get("HCTR_EMU_VARIANT")
    lib = load(var, os.path.basename(os.path.normpath(var))) if var else load()
    bind(lib)
    return lib


def ptr(a):
    return ctypes.c_void_p(a.ctypes.data) if a is not None else None


def check(lib, rc):
    if rc != 0:
        raise RuntimeError(lib.hctr_last_error().decode())


def stats(lib):
    out = (ctypes.c_uint64 * 4)()
    lib.hipemu_stats(out)
    return {"launches": out[0], "blocks": out[1], "divergent_collectives": out[2],
            "shfl_from_inactive": out[3]}


class HashTable:
    """hctr_ht_* (include/hugectr_amd.h) on host memory"""

    def __init__(self, lib, capacity, key_type):
        self.lib = lib
        self.h = ctypes.c_void_p()
        check(lib, lib.hctr_ht_create(ctypes.c_size_t(capacity), key_type, ctypes.byref(self.h)))

    def __del__(self):
        if self.h:
            self.lib.hctr_ht_destroy(self.h)
            self.h = None

    def get_insert(self, keys):
        out = np.empty(keys.size, dtype=np.uint64)
        check(self.lib, self.lib.hctr_ht_get_insert(self.h, ptr(keys), ctypes.c_size_t(keys.size),
                                                    None, ptr(out), None))
        return out

    def get_mark(self, keys):
        out = np.empty(keys.size, dtype=np.uint64)
        check(self.lib, self.lib.hctr_ht_get_mark(self.h, ptr(keys), ctypes.c_size_t(keys.size),
                                                  None, ptr(out), None))
        return out

    def size(self):
        n = ctypes.c_size_t()
        check(self.lib, self.lib.hctr_ht_size(self.h, None, ctypes.byref(n)))
        return n.value

    def value_head(self):
        n = ctypes.c_size_t()
        check(self.lib, self.lib.hctr_ht_value_head(self.h, None, ctypes.byref(n)))
        return n.value

    def dump(self):
        n = self.lib.hctr_ht_table_size(self.h)
        k = np.empty(n, dtype=np.int64)
        v = np.empty(n, dtype=np.uint64)
        c = ctypes.c_size_t()
        check(self.lib, self.lib.hctr_ht_dump(self.h, ptr(k), ptr(v), ctypes.byref(c), None))
        return k[:c.value], v[:c.value]


def radix_sort_pairs(lib, keys, vals, end_bit):
    n = keys.size
    tb = lib.hctr_radix_sort_temp_bytes(n)
    temp = np.empty(tb // 4 + 64, dtype=np.uint32)
    ko = np.empty(n, dtype=np.uint32)
    vo = np.empty(n, dtype=np.uint32)
    check(lib, lib.hctr_radix_sort_pairs_u32(ptr(temp), ctypes.c_size_t(temp.nbytes), ptr(keys),
                                             ptr(ko), ptr(vals), ptr(vo), ctypes.c_size_t(n),
                                             end_bit, None))
    return ko, vo


def bind(lib):
    """argument / result types of every C-ABI entry point, taken from the product's binding table
    (hugectr_amd/_lib.py: declarations only -- nothing of the product runs here)"""
    from hugectr_amd import _lib
    for name, (res, args) in _lib._SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return _lib


class Embedding:
    """hctr_emb_* on host memory (numpy in, numpy views out)"""

    def __init__(self, lib, embedding_type, batch, vocab, D, max_feature_num, slot_num, combiner,
                 opt, key_dtype=np.int64, out_dtype=0, slot_size_array=None, rank=0, world=1,
                 seed=0):
        _lib = bind(lib)
        self.lib, self._lib = lib, _lib
        p = _lib.EmbeddingParams()
        p.embedding_type = embedding_type
        p.key_type = _lib.KEY_I64 if key_dtype == np.int64 else _lib.KEY_U32
        p.out_dtype = out_dtype
        p.train_batch_size = batch
        p.evaluate_batch_size = batch
        p.max_vocabulary_size_per_gpu = vocab
        p.embedding_vec_size = D
        p.max_feature_num = max_feature_num
        p.slot_num = slot_num
        p.combiner = combiner
        if slot_size_array is not None:
            self._ss = (ctypes.c_size_t * slot_num)(*[int(x) for x in slot_size_array])
            p.slot_size_array = ctypes.cast(self._ss, ctypes.POINTER(ctypes.c_size_t))
        for k, v in opt.items():
            setattr(p, k, v)
        p.rank, p.world, p.seed = rank, world, seed
        self.batch, self.D, self.slot_num, self.out_dtype = batch, D, slot_num, out_dtype
        self.h = ctypes.c_void_p()
        check(lib, lib.hctr_emb_create(ctypes.byref(p), ctypes.byref(self.h)))
        self.vocab = int(lib.hctr_emb_get_max_vocabulary_size(self.h))
        self.slots_on_rank = int(lib.hctr_emb_slots_on_rank(self.h))
        check(lib, lib.hctr_emb_init_params(self.h, None))

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.hctr_emb_destroy(self.h)
            self.h = None

    def _view(self, addr, shape, dtype):
        n = int(np.prod(shape))
        buf = (ctypes.c_char * (n * np.dtype(dtype).itemsize)).from_address(addr)
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    def table(self):
        return self._view(self.lib.hctr_emb_table_ptr(self.h), (self.vocab, self.D), np.float32)

    def opt_state(self, k):
        a = self.lib.hctr_emb_opt_state_ptr(self.h, k)
        if not a:
            return None
        if self.out_dtype == 1:  # fp16 embeddings keep their optimizer state in fp16
            return self._view(a, (self.vocab, self.D), np.float16).astype(np.float32)
        return self._view(a, (self.vocab, self.D), np.float32)

    def value_index(self, nnz):
        return self._view(self.lib.hctr_emb_value_index_ptr(self.h), (nnz,), np.uint64)

    def forward(self, is_train, row_offset, keys):
        odt = {0: np.float32, 1: np.float16, 2: np.uint16}[self.out_dtype]
        out = np.empty((self.batch, self.slots_on_rank, self.D), dtype=odt)
        check(self.lib, self.lib.hctr_emb_forward(self.h, 1 if is_train else 0, ptr(row_offset),
                                                  ptr(keys), keys.size, ptr(out), None))
        return out

    def backward(self, top_grad):
        self._g = np.ascontiguousarray(top_grad)
        check(self.lib, self.lib.hctr_emb_backward(self.h, ptr(self._g), None))

    def update_params(self):
        check(self.lib, self.lib.hctr_emb_update_params(self.h, None))

    def check_overflow(self):
        check(self.lib, self.lib.hctr_emb_check_overflow(self.h, None))
