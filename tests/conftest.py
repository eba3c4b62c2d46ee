import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU-marked tests are skipped (not failed) when no GPU is visible, so `pytest tests/`
    without -m still passes in the CPU-only build container."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pytest_report_header(config):
    """First lines of every pytest log: how many GPUs this process sees (the two-device test of test_gpu_misc.py only runs with >= 2)
    and which native libraries the package will load -- evidence that travels with the log copied to profiles/."""
    try:
        import torch
        n = torch.cuda.device_count() if torch.cuda.is_available() else 0
        names = [torch.cuda.get_device_name(i) for i in range(n)]
    except Exception as ex:  # pragma: no cover
        n, names = 0, [repr(ex)]
    pkg = os.path.join(ROOT, "flash_cosine_sim_attention_amd")
    libs = [f for f in sorted(os.listdir(pkg)) if f.endswith(".so")]
    return ["fcsa: torch.cuda.device_count() = %d %s; HIP_VISIBLE_DEVICES=%r" % (n, names, os.environ.get("HIP_VISIBLE_DEVICES")),
            "fcsa: native libraries in the package: %s" % libs]
