"""CPU: the C-ABI library builds, loads and exports every symbol include/pintron_gpu.h declares
(no compute calls: there is no GPU here), and refuses to run without a device."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    import pintron_amd.capi as capi
    return capi


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "pintron_gpu.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(pgpu_[a-z_0-9]+)\s*\(", hdr)))


def test_header_symbols_exported(lib):
    L = lib.lib()
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(L, s), "libpintron_gpu.so does not export %s" % s
    assert sorted(lib.EXPORTS) == syms


def test_struct_layouts(lib):
    assert C.sizeof(lib.DpJob) == 48
    assert C.sizeof(lib.DpResult) == 48
    assert C.sizeof(lib.GroupInfo) == 96      # + t0_ms: the launch on the device time line
    assert lib.lib().pgpu_abi_version() == 1


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(lib.PgpuError) as e:
        lib.Context(0)
    assert e.value.code == lib.PGPU_EDEVICE
