import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _ensure_native():
    """The C-ABI library and the oracle are build products (git-ignored): build them when missing, exactly as
    __graft_entry__.build() does (hipcc cross-compiles gfx950 without a GPU; gcc for the oracle)."""
    import subprocess
    lib = os.path.join(ROOT, "libyafaray_amd", "libyafaray_gpu.so")
    if not os.path.exists(lib):
        subprocess.run(["bash", os.path.join(ROOT, "libyafaray_amd", "csrc", "build.sh")], check=True, timeout=1800)
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"], check=True, timeout=600)


def pytest_sessionstart(session):
    _ensure_native()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "container: needs /root/reference (build container only)")
    config.addinivalue_line("markers", "timeout: per-test time limit (pytest-timeout); a no-op where the plugin is absent")


def pytest_collection_modifyitems(config, items):
    have_ref = os.path.isdir("/root/reference")
    for item in items:
        if "container" in item.keywords and not have_ref:
            item.add_marker(pytest.mark.skip(reason="/root/reference not present"))
