"""CPU: the C-ABI library loads and exports every symbol include/dq_hip.h declares; the ctypes table binds all of them
with matching arity (no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(REPO, "include", "dq_hip.h")


def declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(dq_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        out[name] = 0 if args in ("", "void") else args.count(",") + 1
    return out


def test_header_symbols_exported_and_bound():
    from dquartic import _native as N

    decl = declared()
    assert len(decl) >= 16, decl
    lib = N.lib()
    for name, nargs in decl.items():
        assert hasattr(lib, name), f"{name} declared in dq_hip.h but not exported by libdq_hip.so"
        assert name in N.PROTOTYPES, f"{name} has no ctypes prototype"
        assert len(N.PROTOTYPES[name][1]) == nargs, f"{name}: header has {nargs} parameters, binding {len(N.PROTOTYPES[name][1])}"
    assert set(N.PROTOTYPES) == set(decl), set(N.PROTOTYPES) ^ set(decl)
    assert lib.dq_abi_version() == N.ABI_VERSION == 10  # DQ_ABI_VERSION in include/dq_hip.h


def test_plan_layout_without_a_gpu():
    from dquartic import _native as N

    lib = N.lib()
    mults = (ctypes.c_int * 7)(1, 2, 2, 3, 3, 4, 4)
    plan = lib.dq_plan_create(4, 7, mults, 64, 1000)
    assert plan
    assert lib.dq_plan_num_params(plan) == 395 and lib.dq_plan_param_floats(plan) == 128847
    ws_inf = lib.dq_unet_workspace_bytes(plan, 32, 400, 0)
    ws_trn = lib.dq_unet_workspace_bytes(plan, 32, 400, 1)
    assert 0 < ws_inf and ws_trn == 2 * ws_inf
    lib.dq_plan_destroy(plan)
    # unsupported configurations fail loudly
    bad = (ctypes.c_int * 2)(1, 8)
    assert not lib.dq_plan_create(4, 2, bad, 64, 1000) and b"16" in lib.dq_last_error()
    assert not lib.dq_plan_create(4, 7, mults, 65, 1000) and b"divisible" in lib.dq_last_error()


def test_missing_library_raises(monkeypatch):
    from dquartic import _native as N

    monkeypatch.setattr(N, "_lib", None)
    monkeypatch.setattr(N, "LIB_PATH", "/nonexistent/libdq_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        N.lib()
