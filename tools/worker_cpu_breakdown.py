"""tools/worker_cpu_breakdown.py [size] -- where a Pool worker's CPU goes per tile (host only, no GPU): the file-level pieces of one
compute_disparity_map call through the broker, each timed in wall and in process CPU (user + system, encoder threads included) over 40
repetitions on files in /dev/shm.  Round 6: the Pool model is bounded by the box's CPU budget below the headline tile size."""
import os
import sys
import tempfile
import time

sys.path.insert(0, ".")
import numpy as np
from s2p_amd import io as rio

size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rng = np.random.default_rng(0)
d = tempfile.mkdtemp(prefix="wcb_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
im = rng.random((size, size), np.float32)
disp = rng.random((size, size), np.float32)
mask = np.ones((size, size), np.uint8)
blob = rng.random((size // 16, size // 16)) < 0.08
mask[np.kron(blob, np.ones((16, 16), bool))] = 0
mask[rng.random((size, size)) < 0.01] = 0
p1, p2 = os.path.join(d, "a.tif"), os.path.join(d, "b.tif")
rio.write_image(p1, im)
rio.write_image(p2, im)
arena = np.empty((2, size, size), np.float32)


def timed(name, fn, n=40):
    for _ in range(n):                                     # warm: pages touched, pool threads started, allocator settled
        fn()
    w0, c0 = time.perf_counter(), time.process_time()
    for _ in range(n):
        fn()
    w, c = (time.perf_counter() - w0) / n * 1e3, (time.process_time() - c0) / n * 1e3
    print("  %-58s wall %6.3f ms   CPU %6.3f ms" % (name, w, c))
    return c


def alloc_into(k):
    return lambda shape, dtype=np.float32: arena[k].reshape(-1)[:int(np.prod(shape))].reshape(shape)


print("%d x %d tile, files in %s" % (size, size, d))
tot = 0.0
def read_both():                                         # as broker.match does: image 1 on a pool thread, image 0 here, both straight into the arena
    f = rio._pool().submit(rio.read_image, p2, np.float32, alloc_into(1))
    rio.read_image(p1, np.float32, alloc_into(0))
    f.result()


tot += timed("read two float32 TIFFs straight into the arena (readinto)", read_both)
timed("  (for comparison: read_images into fresh arrays)", lambda: rio.read_images([p1, p2]) and None)
od, oc, om = os.path.join(d, "d.tif"), os.path.join(d, "c.tif"), os.path.join(d, "m.png")


def write_all():
    rio.write_images([(od, disp), (oc, disp), (om, mask)])


tot += timed("write disp + confidence TIFFs and the mask PNG (write_images)", write_all)
timed("  of which: one float32 TIFF", lambda: rio.write_image(od, disp))
timed("  of which: the mask PNG (rows + deflate level 1 + crc)", lambda: rio.write_image(om, mask))
import zlib
rows = np.zeros((size, size + 1), np.uint8)
rows[:, 1:] = mask
buf = memoryview(rows).cast("B")
timed("     zlib.compress(level 1) of the PNG rows, one thread", lambda: zlib.compress(buf, 1))
timed("     zlib.compress(level 0 = stored)", lambda: zlib.compress(buf, 0))
timed("     crc32 + adler32 of the rows", lambda: (zlib.crc32(buf), zlib.adler32(buf)))
tot += timed("unlink the three outputs + rewrite (unlink alone = this - write)", lambda: (write_all(), [os.unlink(p) for p in (od, oc, om)]))
print("  (sum of read + write + unlink-and-rewrite: %.2f ms CPU)" % tot)
for p in (p1, p2):
    os.unlink(p)
os.rmdir(d)
