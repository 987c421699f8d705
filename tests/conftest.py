import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")
    config.addinivalue_line("markers", "reference: needs /root/reference mounted (build container only)")
    config.addinivalue_line("markers", "slow: full-size CPU oracle case (tens of seconds)")


def pytest_collection_modifyitems(config, items):
    have_ref = os.path.isfile("/root/reference/Dynam3D_VLN/vlnce_baselines/models/feature_fields.py")
    skip_ref = pytest.mark.skip(reason="/root/reference not mounted")
    try:
        import pytest_timeout  # noqa: F401
        have_timeout = True
    except ImportError:
        have_timeout = False
    for item in items:
        if "reference" in item.keywords and not have_ref:
            item.add_marker(skip_ref)
        if have_timeout and "gpu" in item.keywords and item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(900))          # a hung GPU test fails after 15 min instead of eating the box's time limit
