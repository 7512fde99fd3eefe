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
def _no_broker_left_behind(tmp_path_factory):
    """Tests whose Pool workers reach the GPU start a broker on demand (s2p_amd/broker.py); it would leave by itself after two idle
    minutes.  The session gets a broker directory OF ITS OWN (S2P_HIP_BROKER_DIR, inherited by every worker and broker the tests start)
    and asks the brokers found THERE to leave when it ends: a test run never talks to, or stops, the brokers of a live job of the same
    user on the box."""
    import shutil
    import tempfile
    old = os.environ.get("S2P_HIP_BROKER_DIR")
    d = tempfile.mkdtemp(prefix="s2p_broker_test_")           # short path: a Unix socket's name is limited to ~100 bytes
    os.environ["S2P_HIP_BROKER_DIR"] = d
    # ... and its own accounting of the device process fence (csrc/api.hip: acquire_device_slot): the slots of a live job do not count
    # against the tests, the test processes do not take a live job's, and the session allows itself 16 processes per device -- the
    # pytest process, the session's broker and the direct-mode Pools of 6 sit side by side here (the fence test sets its own values)
    old_slot = {k: os.environ.get(k) for k in ("S2P_HIP_SLOT_DIR", "S2P_HIP_MAX_PROCS_PER_DEVICE")}
    os.environ["S2P_HIP_SLOT_DIR"] = d
    os.environ.setdefault("S2P_HIP_MAX_PROCS_PER_DEVICE", "16")
    yield
    try:
        from s2p_amd import broker
        broker.shutdown_all()
    except Exception:
        pass
    finally:
        for k, v in old_slot.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        if old is None:
            os.environ.pop("S2P_HIP_BROKER_DIR", None)
        else:
            os.environ["S2P_HIP_BROKER_DIR"] = old
        shutil.rmtree(d, ignore_errors=True)
