import importlib
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dic():
    return importlib.import_module("diffusion-image-captioning_amd")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    return z, meta


@pytest.fixture(autouse=True)
def _shipped_options():
    """Every test starts on the shipped configuration (options.py: nothing differs from the defaults unless DIC_OPTIONS / a legacy variable was set for
    the whole run) and whatever a test switches is switched back -- a benchmark line and an assertion can then only differ in configuration visibly."""
    import copy
    import importlib
    opts = importlib.import_module("diffusion-image-captioning_amd.options")
    if not os.environ.get("DIC_OPTIONS") and not any(v in os.environ for v in opts.LEGACY_ENV):
        assert opts.OPT.non_default() == {}, opts.OPT.non_default()
    keep = copy.copy(opts.OPT.__dict__)
    yield
    opts.OPT.__dict__.update(keep)
