"""tools/pcie_probe.py -- host <-> device copy rates for the buffer kinds and sizes the broker moves (round 6): a memfd mapping page-locked
with hipHostRegister (what an arena is) against hipHostMalloc memory, 1 / 2.25 / 4 / 8 / 9 MB per copy, one stream and three streams side
by side, each direction alone and both at once."""
import ctypes
import mmap
import os
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch

hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
hip.hipHostRegister.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint]
hip.hipHostMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
H2D, D2H = 1, 2
torch.cuda.init()
dev = torch.device("cuda", 0)
TOTAL = 256 << 20
dbuf = [torch.empty(TOTAL, dtype=torch.uint8, device=dev) for _ in range(3)]
streams = [torch.cuda.Stream() for _ in range(3)]


def host_registered(n):
    fd = os.memfd_create("probe")
    os.ftruncate(fd, n)
    mm = mmap.mmap(fd, n)
    a = np.frombuffer(mm, dtype=np.uint8)
    a[:] = 1
    assert hip.hipHostRegister(a.ctypes.data, n, 1) == 0          # hipHostRegisterPortable
    return a, a.ctypes.data


def host_malloc(n):
    p = ctypes.c_void_p()
    assert hip.hipHostMalloc(ctypes.byref(p), n, 1) == 0           # hipHostMallocPortable
    ctypes.memset(p, 1, n)
    return p, p.value


def run(hptrs, size, direction, nstreams, both=False):
    ncopies = TOTAL // size
    torch.cuda.synchronize()
    t = time.perf_counter()
    for k in range(nstreams):
        st = streams[k].cuda_stream
        for i in range(ncopies):
            off = i * size
            d = dbuf[k].data_ptr() + off
            h = hptrs[k] + off
            if both:
                dirn = H2D if (i & 1) == 0 else D2H
            else:
                dirn = direction
            if dirn == H2D:
                hip.hipMemcpyAsync(d, h, size, H2D, st)
            else:
                hip.hipMemcpyAsync(h, d, size, D2H, st)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    return TOTAL * nstreams / dt / 1e9, dt / (ncopies * nstreams) * 1e6


for kind, maker in (("memfd + hipHostRegister (an arena)", host_registered), ("hipHostMalloc", host_malloc)):
    keep = [maker(TOTAL) for _ in range(3)]
    hptrs = [k[1] for k in keep]
    print(kind)
    for size in (1 << 20, 2359296, 4 << 20, 8 << 20, 9437184):
        row = []
        for label, direction, ns, both in (("H2D x1", H2D, 1, False), ("D2H x1", D2H, 1, False), ("H2D x3", H2D, 3, False), ("D2H x3", D2H, 3, False), ("both x3", 0, 3, True)):
            run(hptrs, size, direction or H2D, ns, both)
            gbs, us = run(hptrs, size, direction or H2D, ns, both)
            row.append("%s %.1f GB/s (%.0f us/copy)" % (label, gbs, us))
        print("  %5.2f MB per copy: %s" % (size / 2 ** 20, " | ".join(row)))
