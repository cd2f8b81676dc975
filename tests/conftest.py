import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # every synthetic ResNet-50-U-Net carries upstream's one_side_pad Lambda: the lowering warns by design (test_keras_config_fixture
    # checks the warning and the strict mode explicitly)
    config.addinivalue_line("filterwarnings", "ignore:layer .*Lambda after ZeroPadding2D:UserWarning")


@pytest.fixture(scope="session")
def repo_root():
    return ROOT
