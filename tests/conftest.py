import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_artifacts():
    """A clean checkout has no binaries: build the HIP library (hipcc cross-compiles gfx950 without a GPU) and the
    oracle once per session, exactly as __graft_entry__.build() does."""
    import importlib
    pt = importlib.import_module("path-tracing_amd")
    if not os.path.exists(pt.LIB_PATH) or not os.path.exists(os.path.join(ROOT, "path-tracing_amd", "bin", "pt_render")):
        pt.build()
    import oracle_lib
    oracle_lib.build()


@pytest.fixture(scope="session")
def models_dir():
    return os.path.join(ROOT, "models") + "/"


@pytest.fixture(scope="session")
def oracle_scene(models_dir):
    import oracle_lib as O
    return O.Scene.load(models_dir, "Tor.obj")
