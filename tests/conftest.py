import importlib.util
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def load_package():
    """import the hyphen-named package directory as `formula_vad_amd`"""
    if "formula_vad_amd" in sys.modules:
        return sys.modules["formula_vad_amd"]
    pkg_dir = os.path.join(ROOT, "formula-vad_amd")
    spec = importlib.util.spec_from_file_location(
        "formula_vad_amd", os.path.join(pkg_dir, "__init__.py"),
        submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["formula_vad_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return load_package()


@pytest.fixture(scope="session")
def fv(pkg):
    return pkg.binding


@pytest.fixture(scope="session")
def weights7(fv):
    """the library's synthetic NSNet2-shaped weights, seed 7 (BASELINE config 3)"""
    return fv.synth_weights(7)


@pytest.fixture(scope="session")
def gpu_ctx(fv):
    """one context for the whole GPU session; fails loudly if the extension or device is missing"""
    ctx = fv.Context(0)
    ctx.load_synth(7)
    yield ctx
    ctx.close()


def rel_err(a, b, floor=0.0):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b) / np.maximum(np.abs(b), floor)
