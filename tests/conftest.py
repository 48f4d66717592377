import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


# The library reads its dispatch switches (csrc/policy.h) from the environment ONCE.  The tests flip SEGFAC_* variables in the running
# process (monkeypatch.setenv, os.environ[...] = ...): every such change -- and monkeypatch's undo -- is followed by a re-read, so a
# test sees the switch it has just set and the next test sees the defaults again.
class _PolicyFollowsEnviron(type(os.environ)):
    def __setitem__(self, key, value):
        super().__setitem__(key, value)
        _policy_sync(key)

    def __delitem__(self, key):
        super().__delitem__(key)
        _policy_sync(key)


def _policy_sync(key):
    if isinstance(key, str) and key.startswith('SEGFAC_'):
        hip = sys.modules.get('segmentation_factory_amd.hip')
        if hip is not None and hip._lib is not None:
            hip.policy_reload()


os.environ.__class__ = _PolicyFollowsEnviron


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def _gpu_ok():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # -m gpu on a box without a GPU must fail loudly rather than skip: only auto-skip when the
    # user did not ask for gpu tests explicitly.
    if 'gpu' in (config.getoption('-m') or ''):
        return
    if _gpu_ok():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for it in items:
        if 'gpu' in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
