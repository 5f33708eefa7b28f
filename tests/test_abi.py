"""The C ABI without a GPU: the library builds, loads, exports every symbol include/trx_knn.h
declares, and fails loudly (never falls back to a CPU path) when no device is present."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "trx_knn.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(trx_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_list_the_same_symbols():
    from textreact_amd import _lib
    assert _declared() == sorted(_lib.SYMBOLS)


def test_library_loads_and_exports_every_symbol():
    from textreact_amd import _lib
    L = _lib.lib()
    for sym in _declared():
        assert hasattr(L, sym), sym
    assert L.trx_version().startswith(b"trxknn 0.2")
    assert L.trx_search_stats_size() == ctypes.sizeof(_lib.SearchStats) == 56


def test_every_header_entry_cites_the_reference_call_it_replaces():
    src = open(os.path.join(ROOT, "include", "trx_knn.h")).read()
    assert "retrieve/retrieve_faiss.py:65" in src and "retrieve_faiss.py:66" in src and "retrieve_faiss.py:70-71" in src


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from textreact_amd import _lib
    import textreact_amd.faiss_compat as faiss
    with pytest.raises(_lib.TrxError) as e:
        faiss.IndexFlatIP(8)
    assert "no HIP device" in str(e.value) or "HIP" in str(e.value)


def test_argument_errors_do_not_need_a_device():
    from textreact_amd import _lib
    L = _lib.lib()
    out = ctypes.c_void_p()
    assert L.trx_index_create(0, 0, 0, ctypes.byref(out)) == -1        # d <= 0
    assert b"d must be" in L.trx_last_error()
    assert L.trx_index_create(8, 7, 0, ctypes.byref(out)) == -1        # unknown metric
    assert L.trx_merge_topk_device(0, 99, 1, 1, None, None, None, None, None) == -1
    assert L.trx_index_ntotal(None) == -1


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "textreact_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "libtrxoracle" not in text and not re.search(r"\bnn_ref\b", text.replace("oracle/nn_ref.py", "")), f
                # one implementation of the predictor ops: no backend switch, no PyTorch statement of them in the product
                assert not re.search(r"kernel_backend|backend\s*==\s*[\"']torch", text), f


def test_predictor_ops_have_no_cpu_implementation():
    import torch
    from textreact_amd.predictor import ops
    q = torch.zeros(1, 4, 1, 64)
    x = torch.zeros(4, 64)
    for call in (lambda: ops.attention(q, q, q), lambda: ops.add_layernorm(x, x, torch.ones(64), torch.zeros(64), 1e-5),
                 lambda: ops.attention_qkv(torch.zeros(1, 4, 3, 1, 64)), lambda: ops.require_device("cpu")):
        with pytest.raises(ops.TrxNNError):
            call()
