"""hugectr_amd.sok_format against the reference's own pure-Python format functions
(save_meta_file / load_meta_file / write_file_head / read_file_head of
R/sparse_operation_kit/sparse_operation_kit/dump_load.py), extracted from the source file and run
in a namespace without TensorFlow.  Both directions: the reference reads what we write, we read what
the reference writes, and the bytes are identical.  Skipped where the reference is not mounted."""
import ast
import os
from enum import Enum

import numpy as np
import pytest

REF = "/root/reference/sparse_operation_kit/sparse_operation_kit/dump_load.py"
WANTED = {"SOK_var_info", "MetaVarType", "FileType", "get_meta_info_offset", "save_meta_file",
          "convert_bytes_to_int_list", "convert_bytes_to_string_list", "load_meta_file",
          "write_file_head", "read_file_head"}


def _reference_namespace():
    src = open(REF).read()
    tree = ast.parse(src)
    ns = {"os": os, "Enum": Enum, "np": np, "global_gpu_id": lambda: 0}
    for node in tree.body:
        seg = ast.get_source_segment(src, node)
        if isinstance(node, ast.Assign) and all(isinstance(t, ast.Name) for t in node.targets):
            name = node.targets[0].id
            if name.endswith("_length") or name in ("MetaVarOffsetDict", "save_buffer_size_bytes"):
                exec(seg, ns)
        elif isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in WANTED:
            exec(seg, ns)
    assert WANTED <= set(ns)
    return ns


pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="reference checkout not mounted")


def _infos(fmt):
    return [fmt.VarInfo("user_id_table", "Adam", 1, 5, 123456789012, 128),
            fmt.VarInfo("item_category", "SGD", 3, 5, 17, 16),
            fmt.VarInfo("t", "", 0, 6, 0, 1)]


def test_meta_file_both_directions(tmp_path):
    from hugectr_amd import sok_format as fmt
    ref = _reference_namespace()
    ours, theirs = tmp_path / "ours", tmp_path / "theirs"
    ours.mkdir()
    theirs.mkdir()
    infos = _infos(fmt)
    fmt.save_meta_file(str(ours), infos)
    rinfos = []
    for i in infos:
        r = ref["SOK_var_info"]()
        r.opt_name, r.key_type, r.emb_type = i.opt_name, i.key_type, i.emb_type
        r.emb_num, r.emb_length, r.emb_name = i.emb_num, i.emb_length, i.emb_name
        rinfos.append(r)
    ref["save_meta_file"](str(theirs), rinfos)
    assert open(ours / "meta_info", "rb").read() == open(theirs / "meta_info", "rb").read()
    got = fmt.load_meta_file(str(theirs))
    assert [got[i.emb_name] for i in infos] == infos
    # the reference's reader on our file: `[SOK_var_info()] * n` (dump_load.py:280) makes all n
    # records ONE object, so it returns a single entry holding the last table -- enough to see that
    # every field of our file sits where the reference looks for it
    r = ref["load_meta_file"](str(ours))
    last = infos[-1]
    assert list(r) == [last.emb_name]
    rec = r[last.emb_name]
    assert (rec.key_type, rec.emb_type, rec.emb_length, rec.emb_num, rec.opt_name) == (
        last.key_type, last.emb_type, last.emb_length, last.emb_num, last.opt_name)
    one = tmp_path / "one"
    one.mkdir()
    fmt.save_meta_file(str(one), infos[:1])
    rec = ref["load_meta_file"](str(one))[infos[0].emb_name]
    assert (rec.key_type, rec.emb_type, rec.emb_length, rec.emb_num, rec.opt_name) == (
        1, 5, 128, 123456789012, "Adam")
    assert ref["get_meta_info_offset"](ref["MetaVarType"].OptName, 3) == 4 + (20 + 256) * 3


@pytest.mark.parametrize("ftype,var,dtype", [(0, "", np.int64), (1, "", np.float32),
                                             (2, "accumulator", np.float32), (0, "", np.uint64)])
def test_array_file_heads_both_directions(tmp_path, ftype, var, dtype):
    from hugectr_amd import sok_format as fmt
    ref = _reference_namespace()
    assert fmt.FILE_HEAD_LENGTH == ref["file_head_length"] == 296
    arr = (np.arange(40).reshape(10, 4) * 3).astype(dtype)
    info = ref["SOK_var_info"]()
    info.emb_name = "some/table:0"
    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    fmt.write_array_file(a, info.emb_name, ftype, var, arr)
    ref["write_file_head"](b, info, ftype, var, fmt.DTYPE_INDEX[np.dtype(dtype)])
    with open(b, "ba+") as f:                       # as save_table_to_filesystem_* appends
        arr.tofile(f)
    assert open(a, "rb").read() == open(b, "rb").read()
    name, ft, vn, idx = ref["read_file_head"](a)
    assert (name, ft.value, vn, idx) == (info.emb_name, ftype, var, fmt.DTYPE_INDEX[np.dtype(dtype)])
    assert fmt.read_file_head(b) == (info.emb_name, ftype, var, idx)
    assert (fmt.read_array_file(b).reshape(10, 4) == arr).all()


def test_rounds_and_name_rules():
    from hugectr_amd import sok_format as fmt
    assert fmt.file_table_name("emb/user:0") == "emb_user_0"
    assert fmt.rows_per_round(1000, 128, 4) == (1, 1000)
    rounds, per = fmt.rows_per_round(1_000_000, 128, 4)   # 512 MB -> 64 MiB rounds
    assert per == (64 << 20) // 512 and rounds == -(-1_000_000 // per)
