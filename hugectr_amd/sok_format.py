"""On-disk format of SparseOperationKit's dump / load
(R/sparse_operation_kit/sparse_operation_kit/dump_load.py:46-362): a directory with
  meta_info                      table count + per-table key type, vector type, vector length, row
                                 count, name (256 bytes) and optimizer name (32 bytes); integers
                                 big-endian, strings right-justified with blanks
  <table>-key, <table>-weight    296-byte head (table name[256] | file type u32 | variable name[32] |
  <table>-<Optimizer>-<slot>     dtype index u32, big-endian) followed by the raw array (native
                                 order, as numpy.tofile writes it)
Pure Python + numpy (no torch, no device): the byte layout is pinned against the reference's own
reader / writer functions in tests/test_sok_format_cpu.py."""
from __future__ import annotations

import os
import string
from dataclasses import dataclass
from typing import Dict, List

import numpy as np

INTEGER_LENGTH, LONG_LONG_LENGTH = 4, 8
OPT_NAME_MAX, OPT_VAR_NAME_MAX, TABLE_NAME_MAX = 32, 32, 256
FILE_HEAD_LENGTH = TABLE_NAME_MAX + INTEGER_LENGTH + OPT_VAR_NAME_MAX + INTEGER_LENGTH  # 296
SAVE_BUFFER_BYTES = 1024 * 1024 * 64
FILE_KEY, FILE_EMB, FILE_OPT_STATE = 0, 1, 2
# data_type_convert (dump_load.py:58-123): index <-> numpy dtype
DTYPE_INDEX = {np.dtype(np.int32): 0, np.dtype(np.int64): 1, np.dtype(np.uint32): 2,
               np.dtype(np.uint64): 3, np.dtype(np.float16): 4, np.dtype(np.float32): 5,
               np.dtype(np.float64): 6}
INDEX_DTYPE = {v: k for k, v in DTYPE_INDEX.items()}


@dataclass
class VarInfo:
    emb_name: str = ""
    opt_name: str = ""
    key_type: int = 0
    emb_type: int = 0
    emb_num: int = 0
    emb_length: int = 0


def file_table_name(name: str) -> str:
    """punctuation in a variable name becomes '_' in file names (dump_load.py:469-471)"""
    for ch in string.punctuation:
        name = name.replace(ch, "_")
    return name


def _be(v: int, n: int) -> bytes:
    return int(v).to_bytes(n, "big", signed=False)


def save_meta_file(path: str, infos: List[VarInfo]):
    t = len(infos)
    with open(os.path.join(path, "meta_info"), "wb") as f:
        f.write(_be(t, INTEGER_LENGTH))
        for i in infos:
            f.write(_be(i.key_type, INTEGER_LENGTH))
        for i in infos:
            f.write(_be(i.emb_type, INTEGER_LENGTH))
        for i in infos:
            f.write(_be(i.emb_length, INTEGER_LENGTH))
        for i in infos:
            f.write(_be(i.emb_num, LONG_LONG_LENGTH))
        f.write("".join(i.emb_name.rjust(TABLE_NAME_MAX, " ") for i in infos).encode())
        f.write("".join(i.opt_name.rjust(OPT_NAME_MAX, " ") for i in infos).encode())


def load_meta_file(path: str) -> Dict[str, VarInfo]:
    p = os.path.join(path, "meta_info")
    if not os.path.exists(p):
        raise Exception(f"can't find meta_info data from path = {path} ,please ensure the integrity "
                        "of weight file")
    b = open(p, "rb").read()
    t = int.from_bytes(b[:INTEGER_LENGTH], "big")
    off = INTEGER_LENGTH

    def ints(width):
        nonlocal off
        out = [int.from_bytes(b[off + k * width:off + (k + 1) * width], "big") for k in range(t)]
        off += t * width
        return out

    def strs(width):
        nonlocal off
        out = [b[off + k * width:off + (k + 1) * width].decode().strip() for k in range(t)]
        off += t * width
        return out

    key_type, emb_type, emb_length = ints(INTEGER_LENGTH), ints(INTEGER_LENGTH), ints(INTEGER_LENGTH)
    emb_num = ints(LONG_LONG_LENGTH)
    names, opts = strs(TABLE_NAME_MAX), strs(OPT_NAME_MAX)
    return {names[k]: VarInfo(names[k], opts[k], key_type[k], emb_type[k], emb_num[k], emb_length[k])
            for k in range(t)}


def write_file_head(path: str, table_name: str, file_type: int, var_name: str, dtype_index: int):
    with open(path, "wb") as f:
        f.write(table_name.rjust(TABLE_NAME_MAX, " ").encode())
        f.write(_be(file_type, INTEGER_LENGTH))
        f.write(var_name.rjust(OPT_VAR_NAME_MAX, " ").encode())
        f.write(_be(dtype_index, INTEGER_LENGTH))


def read_file_head(path: str):
    """-> (table name, file type, variable name, dtype index)"""
    b = open(path, "rb").read(FILE_HEAD_LENGTH)
    name = b[:TABLE_NAME_MAX].decode().strip()
    ftype = int.from_bytes(b[TABLE_NAME_MAX:TABLE_NAME_MAX + INTEGER_LENGTH], "big")
    o = TABLE_NAME_MAX + INTEGER_LENGTH
    var = b[o:o + OPT_VAR_NAME_MAX].decode().strip()
    idx = int.from_bytes(b[o + OPT_VAR_NAME_MAX:o + OPT_VAR_NAME_MAX + INTEGER_LENGTH], "big")
    return name, ftype, var, idx


def write_array_file(path: str, table_name: str, file_type: int, var_name: str, arr: np.ndarray):
    arr = np.ascontiguousarray(arr)
    write_file_head(path, table_name, file_type, var_name, DTYPE_INDEX[arr.dtype])
    with open(path, "ba+") as f:
        arr.tofile(f)


def read_array_file(path: str) -> np.ndarray:
    _, _, _, idx = read_file_head(path)
    with open(path, "rb") as f:
        f.seek(FILE_HEAD_LENGTH, os.SEEK_SET)
        return np.fromfile(f, dtype=INDEX_DTYPE[idx])


def rows_per_round(num_rows: int, row_elems: int, elem_bytes: int):
    """get_save_rounds (dump_load.py:395-411): arrays above 64 MiB are written in rounds; with
    several GPUs every round holds that slice of every rank, rank after rank"""
    total = num_rows * row_elems * elem_bytes
    if total <= SAVE_BUFFER_BYTES:
        return 1, num_rows
    per = int(np.floor(SAVE_BUFFER_BYTES / (row_elems * elem_bytes)))
    return int(np.ceil(num_rows / per)), per


def check_weight_files(key_path: str, weight_path: str, state_paths: List[str]):
    """check_weight_file_valid (dump_load.py:773-844): -> (ok, message, key count, vector length)"""
    for p, what in ((key_path, "key"), (weight_path, "weight")):
        if not os.path.exists(p):
            return False, f"{what} file {p} is not exist", 0, 0
    ksz = INDEX_DTYPE[read_file_head(key_path)[3]].itemsize
    esz = INDEX_DTYPE[read_file_head(weight_path)[3]].itemsize
    kbytes = os.stat(key_path).st_size - FILE_HEAD_LENGTH
    wbytes = os.stat(weight_path).st_size - FILE_HEAD_LENGTH
    if kbytes % ksz or wbytes % esz:
        return False, "file length is not divisible by the element size", 0, 0
    n = kbytes // ksz
    if n == 0 or (wbytes // esz) % n:
        return False, "weight file length is not divisible by the key count", 0, 0
    ev = wbytes // esz // n
    for sp in state_paths:
        if not os.path.exists(sp):
            return False, f"optimizer state file {sp} is not exist", 0, 0
        ssz = INDEX_DTYPE[read_file_head(sp)[3]].itemsize
        sbytes = os.stat(sp).st_size - FILE_HEAD_LENGTH
        if sbytes % ssz or (sbytes // ssz) % n or sbytes // esz // n != ev:
            return False, f"optimizer state file {sp} does not match the weight file", 0, 0
    return True, "", n, ev
