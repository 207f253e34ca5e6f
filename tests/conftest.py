import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)

    return load


@pytest.fixture(scope="session")
def golden_json():
    import json

    def load(name):
        with open(os.path.join(GOLDEN, name + ".json")) as f:
            return json.load(f)

    return load


@pytest.fixture(autouse=True)
def _isolated_train_dir(tmp_path, monkeypatch):
    """every test gets its own default --train_dir: Runner.run() writes checkpoints on stop and restart_behavior
    defaults to "resume" (as in the reference), so a shared directory would leak weights between tests"""
    from sample_factory_amd.cfg import arguments
    flags = [(("train_dir", str, str(tmp_path / "train_dir")) if f[0] == "train_dir" else f) for f in arguments.FLAGS]
    monkeypatch.setattr(arguments, "FLAGS", flags)
