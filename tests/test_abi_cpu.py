"""`not gpu`: the C-ABI shared library loads here (no GPU) and exports every symbol include/vmv.h declares; the
ctypes argument blocks have the C layout; argument validation (which runs before any launch) reports VMV_E* codes."""
import ctypes as C
import os
import re

import pytest
import torch

from videomv_amd import _lib as L
from videomv_amd import ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "vmv.h")).read()
    declared = set(re.findall(r"\b(vmv_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "header parse failed"
    lib = L.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert declared == set(L.SYMBOLS), declared ^ set(L.SYMBOLS)
    assert lib.vmv_abi_version() == 5


def test_struct_layouts_match_c():
    lib = L.load()
    for which, st in ((L.OP_GEMM, L.GemmParams), (L.OP_GN_STATS, L.GroupNormParams), (L.OP_GN_APPLY, L.GroupNormParams),
                      (L.OP_LAYERNORM, L.LayerNormParams), (L.OP_ATTENTION, L.AttnParams), (L.OP_SOFTMAX, L.SoftmaxParams),
                      (100, L.DdimParams),
                      (101, L.GemmSeg), (102, L.SeqMap)):
        assert lib.vmv_sizeof(which) == C.sizeof(st), st.__name__


def test_argument_validation_needs_no_gpu():
    lib = L.load()
    a = torch.zeros(64, 64, dtype=L.elem())
    p = ops.gemm_params(64, 64, ops.linear_segs([(a, 64, 64)]), a, a, 64)
    p.N = 63
    assert lib.vmv_gemm(C.byref(p), None) == -1
    p = ops.gemm_params(64, 64, ops.linear_segs([(a, 64, 64)]), None, a, 64)
    assert lib.vmv_gemm(C.byref(p), None) == -3
    p = ops.gemm_params(64, 64, ops.linear_segs([(a.data_ptr() + 2, 64, 64)]), a, a, 64)
    assert lib.vmv_gemm(C.byref(p), None) == -2
    assert lib.vmv_gemm(None, None) == -3
    assert b"VMV_EALIGN" in lib.vmv_error_string(-2)
    ln = ops.ln_params(a, 64, a, 64, None, None, 64, 64)
    assert lib.vmv_layernorm(C.byref(ln), None) == -3
    ln = ops.ln_params(a, 64, a, 64, torch.zeros(64), torch.zeros(64), 64, 4096)
    assert lib.vmv_layernorm(C.byref(ln), None) == -4


def test_plan_records_and_sizes():
    lib = L.load()
    plan = lib.vmv_plan_create()
    a = torch.zeros(64, 64, dtype=L.elem())
    p = ops.gemm_params(64, 64, ops.linear_segs([(a, 64, 64)]), a, a, 64)
    assert lib.vmv_plan_add(plan, L.OP_GEMM, C.byref(p), C.sizeof(p)) == 0
    assert lib.vmv_plan_add(plan, L.OP_GEMM, C.byref(p), C.sizeof(p) - 8) == -1     # wrong block size
    assert lib.vmv_plan_add(plan, 99, C.byref(p), C.sizeof(p)) == -1
    assert lib.vmv_plan_size(plan) == 1
    lib.vmv_plan_destroy(plan)


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "lib_path", lambda: "/nonexistent/libvmv_hip_f16.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        L.load()
