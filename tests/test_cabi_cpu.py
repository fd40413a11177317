"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/star_flash_attn.h declares, validates its arguments, and the product never routes through
the oracle.  No compute calls (there is no GPU here)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from starflashattention_amd import _lib

HEADER = os.path.join(ROOT, "include", "star_flash_attn.h")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.load()


def test_header_symbols_all_exported(lib):
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = set(re.findall(r"\b(sfa_[a-z0-9_]+)\s*\(", text))
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), f"{name} not exported by {_lib.LIB_PATH}"


def test_struct_layouts_match_header():
    # the ctypes mirrors must match the C structs field for field
    text = open(HEADER).read()
    for cname, cls in (("sfa_decode_args", _lib.DecodeArgs), ("sfa_prefill_args", _lib.PrefillArgs)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), text, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = [re.search(r"(\w+)(\[\d+\])?\s*$", d.strip()).group(1) for d in body.split(";") if d.strip()]
        assert names == [f[0] for f in cls._fields_], cname
    assert ctypes.sizeof(_lib.DecodeArgs) == 10 * 8 + 12 * 4 + 8 + 8
    assert ctypes.sizeof(_lib.PrefillArgs) == 5 * 8 + 6 * 4 + 12 * 8 + 3 * 4 + 4 - 0


def test_argument_validation_without_gpu(lib):
    assert lib.sfa_abi_version() == 1
    a = _lib.DecodeArgs()
    assert lib.sfa_decode(ctypes.byref(a), None) == -1          # null pointers
    assert b"non-NULL" in lib.sfa_last_error()
    for f in ("qkv", "o", "seq_len", "k_cache_table", "v_cache_table"):
        setattr(a, f, 0x1000)
    a.batch_size, a.num_heads, a.memory_max_len, a.num_layer = 1, 2, 16, 1
    a.head_dim = 96
    assert lib.sfa_decode(ctypes.byref(a), None) == -4          # unsupported head_dim
    a.head_dim = 128
    a.rotary_embedding_dim = 129
    assert lib.sfa_decode(ctypes.byref(a), None) == -2
    a.rotary_embedding_dim = 128
    a.dtype = 7
    assert lib.sfa_decode(ctypes.byref(a), None) == -3
    a.dtype = 0
    a.idx_layer = 1
    assert lib.sfa_decode(ctypes.byref(a), None) == -2          # idx_layer >= num_layer
    a.idx_layer = 0
    assert lib.sfa_decode(ctypes.byref(a), None) == -1          # workspace NULL
    assert b"workspace" in lib.sfa_last_error()
    a.workspace, a.workspace_bytes = 0x2000, 16
    assert lib.sfa_decode(ctypes.byref(a), None) == -5          # workspace too small
    p = _lib.PrefillArgs()
    assert lib.sfa_prefill_fwd(ctypes.byref(p), None) == -1
    assert lib.sfa_status_string(-7) == b"seq_len out of range"


def test_auto_splits_and_workspace(lib):
    # BASELINE config 4 fills the machine without splitting; the reference harness shape needs splits
    assert lib.sfa_decode_auto_splits(256, 32, 128, 8192) == 1
    s = lib.sfa_decode_auto_splits(2, 32, 128, 8192)
    assert s > 1
    assert lib.sfa_decode_workspace_bytes(256, 32, 128, 8192, 1) == 256
    assert lib.sfa_decode_workspace_bytes(2, 32, 128, 8192, 4) == 256 + 2 * 32 * 4 * 128 * 4 + 2 * 32 * 4 * 8
    assert lib.sfa_decode_workspace_bytes(2, 32, 128, 8192, 0) == lib.sfa_decode_workspace_bytes(2, 32, 128, 8192, s)


def test_product_never_imports_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    offenders = []
    for base in ("starflashattention_amd", "src", "include", "examples"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".cpp", ".hip", ".h", ".cc")):
                    txt = open(os.path.join(dp, f), errors="replace").read()
                    if re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M) or "oracle/" in txt:
                        offenders.append(os.path.join(dp, f))
    assert not offenders, offenders


def test_ops_fail_loudly_without_library(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libStarFlashAttention.so")
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()
