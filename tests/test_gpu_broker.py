"""GPU side of the broker path (s2p_amd/broker.py): the host-batch entry it calls and page-locking of foreign memory."""
import ctypes
import mmap
import os

import numpy as np
import pytest

from helpers import same, synth_pair
from s2p_amd import _lib

pytestmark = pytest.mark.gpu


def _pairs(n, h, w, amp=9.0):
    return [synth_pair(700 + k, h, w, lambda x, y: 5 + amp * np.sin(x / 41.) * np.cos(y / 31.)) for k in range(n)]


@pytest.mark.parametrize("rec,n,h,w,dmin,dmax", [(2, 3, 150, 200, -20, 27), (1, 5, 96, 160, -8, 39), (0, 2, 96, 160, -8, 39), (2, 8, 256, 256, -32, 31)])
def test_host_batch_equals_single_host_calls(rec, n, h, w, dmin, dmax):
    p = _lib.default_census_params(recursion=rec, median=1)
    pairs = _pairs(n, h, w)
    singles = [_lib.census_sgm(a, b, dmin, dmax, params=p) for a, b in pairs]
    disp = [np.full((h, w), 7, np.float32) for _ in range(n)]
    conf = [np.full((h, w), 7, np.float32) for _ in range(n)]
    mask = [np.full((h, w), 7, np.uint8) for _ in range(n)]
    conf_addr = [c.ctypes.data for c in conf]
    conf_addr[-1] = 0                                         # a slot that does not want its confidence image
    _lib.census_sgm_host_batch(_lib.context(), [a.ctypes.data for a, _ in pairs], [b.ctypes.data for _, b in pairs], w, h, dmin, dmax, p,
                               [d.ctypes.data for d in disp], conf_addr, [m.ctypes.data for m in mask])
    for k in range(n):
        assert same(singles[k]["disp"], disp[k]) and np.array_equal(singles[k]["mask"], mask[k]), k
        if k < n - 1:
            assert same(singles[k]["conf"], conf[k]), k
    assert np.all(conf[-1] == 7)


def test_multi_scale_parameters_run_through_the_batch_entry_one_by_one():
    p = _lib.default_census_params(recursion=1, median=0, scales=6, remove_small_cc=25)
    pairs = _pairs(2, 300, 400)
    singles = [_lib.census_sgm(a, b, -30, 33, params=p) for a, b in pairs]
    disp = [np.empty((300, 400), np.float32) for _ in range(2)]
    _lib.census_sgm_host_batch(_lib.context(), [a.ctypes.data for a, _ in pairs], [b.ctypes.data for _, b in pairs], 400, 300, -30, 33, p,
                               [d.ctypes.data for d in disp], [0, 0], [0, 0])
    for k in range(2):
        assert same(singles[k]["disp"], disp[k])


def test_a_memfd_mapping_can_be_page_locked_and_used_as_io_buffer():
    """What the broker does with a worker's arena: map the descriptor, hipHostRegister it, run a call on addresses inside it."""
    h, w = 128, 192
    size = 8 << 20
    fd = os.memfd_create("s2p_test_arena")
    os.ftruncate(fd, size)
    mm = mmap.mmap(fd, size)
    arr = np.frombuffer(mm, np.uint8)
    _lib.host_register(arr.ctypes.data, size)                 # raises HipError if the driver refuses this kind of memory
    try:
        (a, b), = _pairs(1, h, w)
        n4 = h * w * 4
        v = lambda o, dt, cnt: arr[o:o + cnt * np.dtype(dt).itemsize].view(dt).reshape(h, w)
        v(0, np.float32, h * w)[:] = a
        v(n4, np.float32, h * w)[:] = b
        p = _lib.default_census_params(recursion=2, median=1)
        base = arr.ctypes.data
        _lib.census_sgm_host_batch(_lib.context(), [base], [base + n4], w, h, -16, 15, p, [base + 2 * n4], [base + 3 * n4], [base + 4 * n4])
        ref = _lib.census_sgm(a, b, -16, 15, params=p)
        assert same(ref["disp"], v(2 * n4, np.float32, h * w)) and same(ref["conf"], v(3 * n4, np.float32, h * w))
        assert np.array_equal(ref["mask"], v(4 * n4, np.uint8, h * w))
    finally:
        _lib.host_unregister(arr.ctypes.data)
        del arr
        mm.close()
        os.close(fd)
