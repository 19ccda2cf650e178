"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports exactly the
entry points include/blockcg_hip.h declares, and refuses to run without a gfx950 device (no CPU
fallback).  No compute call is made here."""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    import blockcg_amd
    if not os.path.exists(blockcg_amd.LIB_PATH):
        blockcg_amd.build()
    return blockcg_amd.load()


def _declared(header="blockcg_hip.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bcg_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(lib):
    names = _declared()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/blockcg_hip.h but not exported"


def test_bindings_cover_the_header():
    from blockcg_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()


def test_rccl_library_exports_its_header(lib):
    """libblockcg_rccl.so (native RCCL bcg_comm, include/blockcg_rccl.h): builds, loads without a GPU and exports every
    declared entry point; no transfer is attempted here."""
    from blockcg_amd import rccl
    rl = rccl.load()
    names = [n for n in _declared("blockcg_rccl.h") if n != "bcg_comm"]
    assert len(names) == 12 and sorted(rccl.SIGNATURES) == sorted(names)
    for n in names:
        assert hasattr(rl, n), n
    hdr = open(os.path.join(ROOT, "include", "blockcg_rccl.h")).read()
    assert int(re.search(r"BCG_RCCL_UNIQUE_ID_BYTES (\d+)", hdr).group(1)) == rccl.UNIQUE_ID_BYTES


def test_header_is_plain_c(tmp_path):
    """The boundary is a C ABI: the header must compile as C99 with no C++ (or torch) types in any signature."""
    src = tmp_path / "use_header.c"
    src.write_text('#include "blockcg_hip.h"\n#include "blockcg_rccl.h"\n'
                   "int probe(void) { bcg_context* c = 0; size_t n = 0; bcg_comm k; (void)k;\n"
                   "  return bcg_sbcgrq_device_bytes(c, 16, 4, 1, &n) + bcg_capacity_mode(c, 0); }\n")
    out = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                          "-c", str(src), "-o", str(tmp_path / "use_header.o")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr


def test_library_contains_gfx950_code_object():
    import blockcg_amd
    data = open(blockcg_amd.LIB_PATH, "rb").read()
    assert b"amdgcn-amd-amdhsa--gfx950" in data
    assert b"gfx942" not in data and b"sm_" not in data  # one target, no dual paths


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from blockcg_amd import BlockCGError, Context
    with pytest.raises(BlockCGError) as e:
        Context([16])
    assert e.value.code == 4  # BCG_ERR_NO_DEVICE


def test_product_does_not_import_oracle():
    # the oracle is test infrastructure; nothing under blockcg_amd/ may reference it
    for dp, _, fns in os.walk(os.path.join(ROOT, "blockcg_amd")):
        if "_build" in dp:
            continue
        for fn in fns:
            if fn.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dp, fn)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", txt, flags=re.M), fn
                assert not re.search(r'#include\s+"[^"]*oracle/', txt), fn
