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
    config.addinivalue_line("markers", "variants: needs the A/B build of the library (earlier kernel generations); "
                                       "opt-in: tools/gpu_ci.sh variants")


_gpu_ok = None


def _have_gpu():
    global _gpu_ok
    if _gpu_ok is None:
        import torch
        _gpu_ok = bool(torch.cuda.is_available())
    return _gpu_ok


def pytest_runtest_setup(item):
    # A gpu-marked test that gets selected on a box without a GPU FAILS: it is never skipped, so a
    # `-m gpu` run can not go green without the hardware (the CPU suite deselects them with -m "not gpu").
    if item.get_closest_marker("gpu") is not None and not _have_gpu():
        pytest.fail("this test needs an MI355X (torch.cuda.is_available() is False); "
                    "deselect GPU tests with -m 'not gpu'", pytrace=False)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def decode_golden():
    return load_golden("decode_llama_ref.npz")


@pytest.fixture(scope="session")
def prefill_golden():
    return load_golden("prefill_sdpa_cpu.npz")


@pytest.fixture(scope="session")
def ones_kat():
    return load_golden("decode_ones_kat.npz")


def bf16bits_to_f32(bits):
    return (np.asarray(bits, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)
