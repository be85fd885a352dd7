"""TEST INFRASTRUCTURE ONLY.  `kernel<T...><<<grid, block, shmem, stream>>>(args);` is CUDA syntax,
not C++: to compile a reference .cu file as plain C++ for the host interpreter (tests/emu), every
launch statement is rewritten to `REFEMU_LAUNCH((kernel<T...>), (grid, block, shmem, stream), args);`
(macro in ref_shims/cuda/cuda_runtime_api.h), and a declaration of dynamically sized shared memory
`extern __shared__ T name[];` (an unsized extern array cannot be expressed on the host) to
`T* name = (T*)hipemu::dyn_shared();`.  Nothing else in the text changes.  The output goes to
oracle/_ref/gen/ (generated at build time from the checkout, never committed).

    python ref_launch_rewrite.py <in.cu> <out.cpp>
"""
import re
import sys


def _match_back(s, i, open_c, close_c):
    """s[i] == close_c: index of the matching open_c"""
    depth = 0
    while i >= 0:
        if s[i] == close_c:
            depth += 1
        elif s[i] == open_c:
            depth -= 1
            if depth == 0:
                return i
        i -= 1
    raise ValueError("unbalanced")


def _match_fwd(s, i, open_c, close_c):
    depth = 0
    while i < len(s):
        if s[i] == open_c:
            depth += 1
        elif s[i] == close_c:
            depth -= 1
            if depth == 0:
                return i
        i += 1
    raise ValueError("unbalanced")


def rewrite(src: str) -> str:
    out, pos = [], 0
    while True:
        k = src.find("<<<", pos)
        if k < 0:
            out.append(src[pos:])
            return "".join(out)
        # kernel expression: identifier [<template arguments>] right before the chevrons
        j = k - 1
        while src[j].isspace():
            j -= 1
        if src[j] == ">":
            j = _match_back(src, j, "<", ">") - 1
            while src[j].isspace():
                j -= 1
        end_name = j
        while src[j].isalnum() or src[j] in "_:":
            j -= 1
        start = j + 1
        assert start <= end_name, "no kernel name before <<<"
        kernel = src[start:k].strip()
        e = src.index(">>>", k)
        cfg = src[k + 3:e]
        a0 = e + 3
        while src[a0].isspace():
            a0 += 1
        assert src[a0] == "(", "no argument list after >>>"
        a1 = _match_fwd(src, a0, "(", ")")
        args = src[a0 + 1:a1]
        out.append(src[pos:start])
        out.append(f"REFEMU_LAUNCH(({kernel}), ({cfg}), {args})")
        pos = a1 + 1


if __name__ == "__main__":
    text = open(sys.argv[1]).read()
    res = rewrite(text)
    res = re.sub(r"extern\s+__shared__\s+([A-Za-z_][\w ]*?)\s+(\w+)\[\];",
                 r"\1* \2 = (\1*)hipemu::dyn_shared();", res)
    assert "<<<" not in res
    open(sys.argv[2], "w").write(res)
