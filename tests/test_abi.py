"""The C-ABI library loads and exports every symbol include/*.h declares (no compute calls: no GPU needed)."""
import ctypes
import os
import re

import pytest

from libyafaray_amd import interface

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"typedef struct \w+\s*\{.*?\}\s*\w+;", "", txt, flags=re.S)   # drop struct bodies (callback members)
    return sorted(set(re.findall(r"\b(yaf(?:aray|gpu)_\w+)\s*\(", txt)))


@pytest.mark.parametrize("header,listed", [("yafaray_c_api.h", interface.C_API_SYMBOLS), ("yafgpu.h", interface.GPU_ABI_SYMBOLS)])
def test_every_declared_symbol_is_exported(header, listed):
    lib = ctypes.CDLL(interface.lib_path())
    names = declared_functions(header)
    assert len(names) > 10
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/{header} but not exported by libyafaray_gpu.so"
    assert set(listed) == set(names), f"python symbol list out of sync with include/{header}: {set(listed) ^ set(names)}"


def test_version_and_interface_lifecycle():
    yi = interface.Interface(strict=False)
    assert "yafgpu" in yi.getVersion()
    assert yi.getLastError() == ""
    yi.close()
