"""The grouped weight-gradient GEMM's host side without a GPU: trx_gemm_tn_grouped_block_bytes / _plan (include/trx_nn.h) --
which problems the path takes, and the plan it writes: every tile of every problem exactly once, the problems dealt whole to
the eight XCD lists in balance, the long tiles first inside a list (textreact_amd/csrc/gemm_tn.hip: struct GHeader)."""
import ctypes
import os
import struct

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "textreact_amd", "csrc", "libtrxnn.so")


class Problem(ctypes.Structure):
    _fields_ = [("A", ctypes.c_void_p), ("B", ctypes.c_void_p), ("C", ctypes.c_void_p), ("colsum", ctypes.c_void_p),
                ("M", ctypes.c_int), ("N", ctypes.c_int), ("K", ctypes.c_int), ("lda", ctypes.c_int), ("ldb", ctypes.c_int), ("ldc", ctypes.c_int)]


def _lib():
    if not os.path.exists(SO):
        pytest.skip("libtrxnn.so is not built")
    L = ctypes.CDLL(SO)
    L.trx_gemm_tn_grouped_block_bytes.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.trx_gemm_tn_grouped_block_bytes.restype = ctypes.c_int64
    L.trx_gemm_tn_grouped_plan.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64]
    return L


def _problems(shapes):
    arr = (Problem * len(shapes))()
    for i, (q, (M, N, K)) in enumerate(zip(arr, shapes)):
        q.A, q.B, q.C, q.colsum = 0x10000000 + i * 0x1000000, 0x20000000 + i * 0x1000000, 0x30000000 + i * 0x1000000, 0x40000000 + i * 0x100
        q.M, q.N, q.K, q.lda, q.ldb, q.ldc = M, N, K, N, K, K
    return arr


STEP = [(16384, 2304, 768), (16384, 768, 768), (16384, 3072, 768), (16384, 768, 3072)] * 12 + \
       [(5120, 2304, 768), (5120, 768, 768), (5120, 768, 768), (16384, 1536, 768), (5120, 768, 768), (5120, 3072, 768), (5120, 768, 3072)] * 6 + \
       [(5120, 600, 768)]          # the vocabulary projection: N % 256 != 0


def test_the_plan_of_a_training_step():
    L = _lib()
    arr = _problems(STEP)
    n = len(STEP)
    nbytes = L.trx_gemm_tn_grouped_block_bytes(ctypes.addressof(arr), n)
    tiles = [((N + 255) // 256) * (K // 256) for (M, N, K) in STEP]
    assert nbytes == 32 + 8 * 4 + 12 * 4 + n * 64 + sum(tiles) * 4
    buf = ctypes.create_string_buffer(nbytes)
    assert L.trx_gemm_tn_grouped_plan(ctypes.addressof(arr), n, buf, nbytes) == 0
    assert L.trx_gemm_tn_grouped_plan(ctypes.addressof(arr), n, buf, nbytes - 1) != 0      # a block that is too small is refused
    magic, nprob, nitems, off_c, off_b, off_p, off_i, total = struct.unpack_from("<Iiiiiiii", buf.raw, 0)
    assert magic == 0x54524E47 and nprob == n and nitems == sum(tiles) and total == nbytes
    assert struct.unpack_from("<8i", buf.raw, off_c) == (0,) * 8                             # the cursors start at zero
    begin = struct.unpack_from("<9i", buf.raw, off_b)
    assert begin[0] == 0 and begin[8] == nitems and all(a <= b for a, b in zip(begin, begin[1:]))
    items = struct.unpack_from("<%dI" % nitems, buf.raw, off_i)
    seen = sorted((it >> 12, it & 4095) for it in items)
    assert seen == sorted((i, t) for i in range(n) for t in range(tiles[i]))                 # every tile exactly once
    steps = [(M + 63) // 64 for (M, N, K) in STEP]
    load = []
    for x in range(8):
        mine = items[begin[x]:begin[x + 1]]
        probs = []
        for it in mine:                                          # a problem's tiles are one run of its XCD's list
            if not probs or probs[-1] != it >> 12:
                probs.append(it >> 12)
        assert len(probs) == len(set(probs))
        assert [steps[i] for i in probs] == sorted((steps[i] for i in probs), reverse=True)  # the long tiles first
        load.append(sum(steps[it >> 12] for it in mine))
    assert all(set(items[begin[x]:begin[x + 1]]).isdisjoint(items[begin[y]:begin[y + 1]]) for x in range(8) for y in range(x))
    assert max(load) <= 1.02 * sum(load) / 8, load                                           # whole problems, and still in balance
    # the problem table as the kernel reads it: pointers, sizes, tile counts
    A, B, C, cs, M, N, K, lda, ldb, ldc, tn, tk = struct.unpack_from("<QQQQiiiiiiii", buf.raw, off_p + 64 * (n - 1))
    assert (A, M, N, K, tn, tk) == (arr[n - 1].A, 5120, 600, 768, 3, 3)


@pytest.mark.parametrize("bad", [dict(N=604), dict(K=700), dict(lda=2303), dict(lda=2300 + 5), dict(ldc=767), dict(M=0), dict(A=0), dict(C=0x30000008),
                                 dict(M=1 << 20, lda=4096)])
def test_problems_the_grouped_path_does_not_take(bad):
    L = _lib()
    arr = _problems([(16384, 2304, 768), (5120, 768, 768)])
    assert L.trx_gemm_tn_grouped_block_bytes(ctypes.addressof(arr), 2) > 0
    for k, v in bad.items():
        setattr(arr[1], k, v)
    assert L.trx_gemm_tn_grouped_block_bytes(ctypes.addressof(arr), 2) == -1
    assert L.trx_gemm_tn_grouped_block_bytes(ctypes.addressof(arr), 0) == -1
