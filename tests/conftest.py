import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib

    return oracle_lib.Oracle()


@pytest.fixture(scope="session")
def ctx():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import wgpu_morton_code_radix_sort_test_amd as mrs

    return mrs.Context(0)


@pytest.fixture()
def aids(ctx):
    """Setter for the MRS_ATTR_TEST_* attributes of the shared context: `aids(which, value)`.  A plan
    copies them when it is created; everything set is put back to 0 when the test ends."""
    touched = []

    def set_(which, value):
        ctx.set_attribute(which, value)
        touched.append(which)

    yield set_
    for which in touched:
        ctx.set_attribute(which, 0)
