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


# ---- the reference's own example scripts, staged unmodified by `make -C oracle ref` (test infrastructure like the oracle)
STUBS = os.path.join(ROOT, "tests", "stubs")
EX_ZIP = os.path.join(ROOT, "oracle", "_ref", "sf_examples_ref.zip")


def staged_scripts(tmp_path):
    """extract the staged archive (built where /root/reference exists; it travels to the GPU box) and return the
    directory to put on sys.path / PYTHONPATH"""
    import zipfile
    if not os.path.isfile(EX_ZIP):
        pytest.skip("oracle/_ref/sf_examples_ref.zip not staged (make -C oracle ref needs /root/reference)")
    with zipfile.ZipFile(EX_ZIP) as z:
        z.extractall(tmp_path)
    return str(tmp_path)


def gymnasium_is_real() -> bool:
    try:
        import gymnasium
    except ImportError:
        return False
    return "test-stub" not in getattr(gymnasium, "__version__", "")


@pytest.fixture
def ref_scripts(tmp_path, monkeypatch):
    """`import sf_examples.<script>` resolves to the reference's unmodified files, `import sample_factory` to this engine,
    `import gymnasium` to the real package or, where it is not installed, to tests/stubs/gymnasium"""
    d = staged_scripts(tmp_path / "ref_scripts")
    import sample_factory  # noqa: F401  (installs the alias finder)
    if not gymnasium_is_real():
        monkeypatch.syspath_prepend(STUBS)
    monkeypatch.syspath_prepend(d)
    from sample_factory.algo.utils.context import reset_global_context
    yield d
    reset_global_context()
    for m in [m for m in sys.modules if m == "sf_examples" or m.startswith("sf_examples.")]:
        del sys.modules[m]
