import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def O():
    import oracle_lib
    oracle_lib.oracle()
    return oracle_lib


@pytest.fixture(scope="session")
def gpu_ctx():
    import pintron_amd.capi as capi
    ctx = capi.Context(0)          # raises (test fails) when the HIP library or the GPU is missing
    yield ctx
    ctx.close()
