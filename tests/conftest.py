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
    for item in items:
        if "reference" in item.keywords and not have_ref:
            item.add_marker(skip_ref)
