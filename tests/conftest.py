import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def kat():
    with open(os.path.join(ROOT, "tests", "golden", "cornell_first_hit_kat.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.lib()  # builds liboracle.so with g++ when missing
    return O


@pytest.fixture(scope="session")
def rt():
    import ray_tracing_amd
    ray_tracing_amd.lib()
    return ray_tracing_amd


@pytest.fixture(scope="session")
def earth(rt):
    e = rt.load_earthmap()
    assert e is not None and e.shape == (512, 1024, 4)
    return e


def observe(test, **values):
    """Append what a tolerance test saw (close-pixel share, worst difference, ...) to gpurun_out/observed_parity.jsonl:
    the thresholds in the tests are these observations minus a stated margin, and the file shows how much is left."""
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "observed_parity.jsonl"), "a") as f:
            f.write(json.dumps({"test": test, **{k: (float(v) if isinstance(v, (float, np.floating)) else v) for k, v in values.items()}}) + "\n")
    except OSError:
        pass


def kat_rays(kat):
    o = kat["camera"]["origin"]
    return np.array([o + r["dir"] + [0.5] for r in kat["rays"]], dtype=np.float32)
