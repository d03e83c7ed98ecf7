import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# tests/test_gpu_tp.py emulates a whole tensor-parallel group on ONE device: up to 8 ranks = 16 streams whose kernels wait
# for each other.  ROCm multiplexes a process's streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues; two ranks
# sharing a queue would serialise behind each other's spinning collectives.  (Must be set before the HIP runtime starts;
# irrelevant on a real node where every rank has its own device.)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def load_pplhip():
    """imports ppl.llm.serving_amd/pplhip.py (the directory name is not an importable identifier)."""
    import importlib.util
    name = "pplhip_binding"
    if name in sys.modules:
        return sys.modules[name]
    path = os.path.join(ROOT, "ppl.llm.serving_amd", "pplhip.py")
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def pplhip():
    return load_pplhip()
