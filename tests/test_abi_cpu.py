"""CPU: the C-ABI library loads without a GPU and exports every symbol include/hugectr_amd.h
declares (no compute calls here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "hugectr_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(hctr_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    from hugectr_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared_functions()
    assert len(declared) >= 40
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, f"declared in hugectr_amd.h but not exported: {missing}"


def test_python_binding_covers_the_header():
    from hugectr_amd import _lib
    declared = set(_declared_functions())
    bound = set(_lib.EXPORTED_SYMBOLS)
    assert declared == bound, f"header/binding mismatch: {declared ^ bound}"


def test_error_reporting_without_gpu():
    """argument validation happens before any HIP call: errors come back as codes + message,
    never as exceptions across the ABI"""
    from hugectr_amd import _lib
    rc = _lib.lib.hctr_emb_create(None, None)
    assert rc == -1 and "null" in _lib.last_error()
    rc = _lib.lib.hctr_forward_pool(10, 0, 0, None, 0, None, None, None, 0, None)
    assert rc == -1 and "vec_size" in _lib.last_error()
    rc = _lib.lib.hctr_forward_pool(10, 16, 2, None, 0, None, None, None, 0, None)
    assert rc == -1 and "combiner" in _lib.last_error()
    assert _lib.lib.hctr_version() >= 100


def test_product_package_never_imports_the_oracle():
    """the oracle is test infrastructure: nothing under hugectr_amd/ may reference it"""
    pkg = os.path.join(ROOT, "hugectr_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "pyoracle" not in txt and "hctr_oracle" not in txt, f


def test_new_entry_points_validate_before_touching_the_device():
    """dynamic table / unique exchange / loss glue: bad arguments come back as error codes with a
    message, without a GPU"""
    from hugectr_amd import _lib
    L = _lib.lib
    assert L.hctr_det_create(0, None, b"", 0, 1, 0, None) == -1 and _lib.last_error()
    assert L.hctr_det_lookup(None, None, None, 4, None, None, 0, None) == -1
    assert "null handle" in _lib.last_error()
    assert L.hctr_det_update(None, None, None, None, 0, None, None, 0, None, None, None) == -1
    assert L.hctr_uniq_create(0, None) == -1
    assert L.hctr_uniq_plan(None, 8, 4, 2, 2, 4, 0, 2, None, 100, None, None, None, None) == -1
    assert L.hctr_uniq_expand(8, 2, None, None, None, None, 128, 2, None, None, None, None, None) == -1
    assert L.hctr_updater_reduce_presorted(None, 1, 1, None, None, None, None, 0, 1, None,
                                           None) == -1
    assert L.hctr_relu_bwd_bias(4, 12, None, None, None, None, None, 2, None) == -1
    assert "multiple of 8" in _lib.last_error()
    assert L.hctr_bce_loss(0, None, None, 1.0, None, None, None, 0, None) == -1
    assert L.hctr_forward_pool_weighted(4, 0, 0, None, None, None, None, None, None) == -1
    assert L.hctr_ebc_routed_keys_to_indices(8, 2, 3, None, None, None, None, None) == -1
    assert L.hctr_logit_head(8, 6, None, None, None, None, 1.0, None, None, None, None, None, 2,
                             None) == -1
    assert "multiple of 4" in _lib.last_error()
    assert L.hctr_forward_pool_ptrs(4, 16, 2, None, None, None, 0, None) == -1
    assert "combiner" in _lib.last_error()
    assert L.hctr_det_lookup_rows(None, None, 0, None, None, 0, 1, None, None, None, None) == -1
    import ctypes
    n = ctypes.c_size_t(7)
    assert L.hctr_ebc_local_reduce(None, 1, 1, None, None, 10, None, None, 0, ctypes.byref(n), None,
                                   None, None, None) == -1
    assert L.hctr_emb_index(None, 1, None, None, 0, None) == -1
    assert L.hctr_emb_update_rows(None, 1, None, None, None, 0, None) == -1
