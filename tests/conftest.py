import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def load_package():
    """import dealii-asm_amd/ (the directory name carries a hyphen) as dealii_asm_amd"""
    if "dealii_asm_amd" in sys.modules:
        return sys.modules["dealii_asm_amd"]
    spec = importlib.util.spec_from_file_location(
        "dealii_asm_amd", os.path.join(ROOT, "dealii-asm_amd", "__init__.py")
    )
    mod = importlib.util.module_from_spec(spec)
    sys.modules["dealii_asm_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return load_package()
