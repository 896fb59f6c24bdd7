import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'timeout: per-test limit (pytest-timeout, when installed)')


def pytest_collection_modifyitems(config, items):
    # a GPU test that stops making progress should fail by itself after 20 minutes (the slowest one takes ~1) instead of sitting
    # on the box until whoever launched the suite gives up; needs pytest-timeout (in this image), a no-op marker otherwise
    for item in items:
        if 'gpu' in item.keywords and not any(m.name == 'timeout' for m in item.iter_markers()):
            item.add_marker(pytest.mark.timeout(1200, method='thread'))


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')
