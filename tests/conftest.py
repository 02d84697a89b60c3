import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# MUNDY_ORACLE_PATH: directory holding an alternative build of the `oracle` package (e.g. an ASan/UBSan build of the
# CPU restatement, see tests/run_oracle_sanitizers.sh); it then shadows the in-tree one
if os.environ.get("MUNDY_ORACLE_PATH"):
    sys.path.insert(0, os.environ["MUNDY_ORACLE_PATH"])


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle as o  # test infrastructure: the CPU restatement of the reference path
    o.build()
    return o
