import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def model_dir(tmp_path_factory):
    return str(tmp_path_factory.mktemp("models"))


def _model(model_dir, name, seed=1, **kw):
    from speaksense_amd import ggml_io
    path = os.path.join(model_dir, f"{name}-s{seed}.bin")
    if not os.path.exists(path):
        ggml_io.write_model(path, name, seed=seed, **kw)
    return path


@pytest.fixture(scope="session")
def toy_en_path(model_dir):
    return _model(model_dir, "toy.en")


@pytest.fixture(scope="session")
def toy_ml_path(model_dir):
    return _model(model_dir, "toy")


@pytest.fixture(scope="session")
def toy256_path(model_dir):
    return _model(model_dir, "toy256")


@pytest.fixture(scope="session")
def tiny_en_path(model_dir):
    return _model(model_dir, "tiny.en")


@pytest.fixture(scope="session")
def base_en_path(model_dir):
    return _model(model_dir, "base.en")


@pytest.fixture(scope="session")
def wide2_path(model_dir):
    return _model(model_dir, "wide2")


def report(msg: str):
    """Parity numbers worth keeping (margins, flip counts, stage errors): printed and, on the GPU box, appended to gpurun_out/parity_report.txt."""
    print(msg)
    d = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_report.txt"), "a") as f:
            f.write(msg + "\n")
    except OSError:
        pass
