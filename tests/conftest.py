import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref/libsgbm_ref.so (the real reference build)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU checker (oracle/): built on demand; never imported by the product."""
    from oracle import pyoracle as po
    po.build(ref=None if not po.have_ref() else False)
    return po


@pytest.fixture(scope="session", autouse=True)
def _no_broker_left_behind():
    """Tests whose Pool workers reach the GPU start a broker on demand (s2p_amd/broker.py); it would leave by itself after two idle
    minutes -- ask it to leave when the session ends, so nothing of this run keeps the device."""
    yield
    try:
        import glob
        from s2p_amd import broker
        for path in glob.glob(os.path.join(broker.broker_dir(), "gpu*.sock")):
            broker.shutdown(int(os.path.basename(path)[3:-5]))
    except Exception:
        pass
