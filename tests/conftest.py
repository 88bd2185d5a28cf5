import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(HERE, "golden", "golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def hip():
    """The HIP engine binding; GPU tests fail (not skip) when the library or device is missing."""
    from ropebwt2_amd import build_all, load_hip_lib
    build_all()
    lib = load_hip_lib()
    assert lib.rb2_hip_device_count() > 0, "no HIP device visible: -m gpu tests need an MI355X"
    import ropebwt2_amd
    return ropebwt2_amd
