"""The C-ABI library loads without a GPU and exports exactly what include/salnmf.h declares."""

import os
import re

import pytest

from conftest import ROOT
from salamander_amd import _lib


def _header_functions():
    text = open(os.path.join(ROOT, "include", "salnmf.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(salnmf_[a-z_A-Z0-9]+)\s*\(", text))


def test_library_exists():
    assert os.path.exists(_lib.LIB_PATH), "build the extension first: python -c 'import __graft_entry__ as g; g.build()'"


def test_every_declared_symbol_is_exported_and_bound():
    declared = _header_functions()
    assert declared, "no functions parsed from the header"
    lib = _lib.load()  # getattr on each bound symbol happens inside load()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in salnmf.h but not exported"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))


def test_version_and_error_channel():
    lib = _lib.load()
    assert lib.salnmf_version() >= 100
    assert isinstance(_lib.last_error(), str)


def test_no_compute_without_device():
    """Without a HIP device the engine refuses loudly instead of falling back to the CPU."""
    lib = _lib.load()
    if lib.salnmf_device_count() > 0:
        pytest.skip("a device is present")
    from salamander_amd import Engine, EngineUnavailable

    with pytest.raises(EngineUnavailable):
        Engine(16, 96, 2)


def test_closed_engine_raises_instead_of_passing_null():
    """Every method of a closed Engine raises (the C ABI also rejects NULL handles, see the GPU suite)."""
    from salamander_amd import Engine

    e = Engine.__new__(Engine)
    e._lib, e._handle, e.N, e.V, e.K = _lib.load(), None, 16, 96, 2
    import numpy as np

    for call in (lambda: e.upload_W(np.ones((2, 96))), lambda: e.download_W(), lambda: e.kl_step(1), lambda: e.sync()):
        with pytest.raises(RuntimeError, match="closed"):
            call()
    e.close()  # idempotent


def test_null_handle_is_an_error_code_not_a_crash():
    lib = _lib.load()
    import ctypes

    buf = (ctypes.c_double * 4)()
    for fn, args in (
        (lib.salnmf_upload_W, (None, buf)),
        (lib.salnmf_download_W, (None, buf)),
        (lib.salnmf_upload_X, (None, buf, 0)),
        (lib.salnmf_kl_step, (None, 1, 0)),
        (lib.salnmf_sync, (None,)),
    ):
        assert fn(*args) != 0
        assert _lib.last_error()


@pytest.mark.parametrize("order", ["salamander_first", "torch_first"])
def test_one_hip_runtime_and_one_rccl_whatever_the_import_order(order):
    """Round 1's hazard: loaded before torch, the engine bound the system RCCL and `import torch` brought a second
    copy.  RCCL is now bound at first use (dlopen by SONAME, RTLD_NOLOAD first), so both sides share one copy; the
    child process must also exit cleanly (no abort in a library destructor)."""
    import subprocess
    import sys

    code = """
import ctypes, json, sys
sys.path.insert(0, %r)
if %r == "torch_first":
    import torch
from salamander_amd import _lib
lib = _lib.load()
import torch
buf = ctypes.create_string_buffer(_lib.UNIQUE_ID_BYTES)
rc = lib.salnmf_comm_unique_id(buf)   # first RCCL use: binds the library
print(json.dumps({"rc": rc, "err": _lib.last_error() if rc else "", "libs": _lib.mapped_runtime_libraries()}))
""" % (ROOT, order)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    import json

    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["rc"] == 0, rec
    assert len(rec["libs"]["libamdhip64"]) == 1, rec
    assert len(rec["libs"]["librccl"]) == 1, rec
    assert "torch" in rec["libs"]["librccl"][0] and "torch" in rec["libs"]["libamdhip64"][0], rec


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under salamander_amd/ may reference it."""
    pkg = os.path.join(ROOT, "salamander_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(base, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"


def test_oracle_is_used_only_where_allowed():
    """Outside tests/: only bench.py's cpu_baseline leg and __graft_entry__.smoke() may import the oracle."""
    import ast

    tools = os.path.join(ROOT, "tools")
    for f in os.listdir(tools):
        if f.endswith(".py"):
            src = open(os.path.join(tools, f)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"tools/{f} imports the oracle"
    for name, allowed in (("bench.py", {"cpu_baseline"}), ("__graft_entry__.py", {"smoke"})):
        tree = ast.parse(open(os.path.join(ROOT, name)).read())
        for node in ast.walk(tree):
            if isinstance(node, ast.FunctionDef):
                for sub in ast.walk(node):
                    if isinstance(sub, (ast.Import, ast.ImportFrom)):
                        mods = [a.name for a in sub.names] if isinstance(sub, ast.Import) else [sub.module or ""]
                        if any(m.split(".")[0] == "oracle" for m in mods):
                            assert node.name in allowed, f"{name}:{node.name} imports the oracle"
        for node in tree.body:  # no module-level import of the oracle
            if isinstance(node, (ast.Import, ast.ImportFrom)):
                mods = [a.name for a in node.names] if isinstance(node, ast.Import) else [node.module or ""]
                assert not any(m.split(".")[0] == "oracle" for m in mods), f"{name} imports the oracle at module level"
