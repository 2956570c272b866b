import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "perf: wall-clock statements (kernel durations, throughput ratios); NOT part of "
                            "the parity suite -- run with -m perf.  The parity tests assert the CHOICE a timing was a "
                            "proxy for (fuelmi_map_last_esdf_family, evaluation counts)")


def _has_gpu():
    try:
        import fuel_amd
        return fuel_amd.lib().fuelmi_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu_available():
    return _has_gpu()


def pytest_collection_modifyitems(config, items):
    # gpu tests fail loudly (not skip) when selected with -m gpu on a box without a device or
    # without the built extension: a silent skip would hide "native code not loaded".
    pass
