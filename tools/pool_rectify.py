"""tools/pool_rectify.py [workers tiles] -- (GPU box) the rectification step under the reference's Pool model: forked workers x
s2p_amd.common.image_apply_homography (s2p/rectification.py:366-380 calls it twice per tile) on a uint16 1200 x 1200 window of a larger
image -> 1088 x 1024 rectified tile, files in /dev/shm, through the GPU broker (@broker.remote on _lib.warp)."""
import multiprocessing as mp, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1536


def task(a):
    i, src, d, w, h = a
    from s2p_amd import common
    H = np.array([[0.98, 0.04, -60.0 - (i % 7)], [-0.04, 0.98, 30.0 + (i % 5)], [0.0, 0.0, 1.0]])
    so, sys.stdout = sys.stdout, open(os.devnull, "w")
    t0 = time.monotonic()
    try:
        out = os.path.join(d, "r_%d.tif" % i)
        common.image_apply_homography(out, src, H, w, h)
        os.unlink(out)
    finally:
        sys.stdout.close(); sys.stdout = so
    return os.getpid(), t0, time.monotonic()


if __name__ == "__main__":
    from s2p_amd import io as rio, broker
    d = tempfile.mkdtemp(prefix="s2p_rect_", dir="/dev/shm")
    rng = np.random.default_rng(0)
    img = (rng.random((2400, 2400)) * 1000).astype(np.uint16)
    src = os.path.join(d, "img.tif")
    rio.write_image(src, img)
    for mode in ("1", "0"):
        os.environ["S2P_HIP_BROKER"] = mode
        t0 = time.monotonic()
        with mp.get_context("fork").Pool(P) as pool:
            res = pool.map(task, [(i, src, d, 1088, 1024) for i in range(N)], chunksize=1)
        el = time.monotonic() - t0
        first = {}
        for pid, a, b in res:
            first[pid] = min(first.get(pid, 1e18), b)
        warm = max(first.values())
        tail = max(a for _, a, _ in res)
        st = [r for r in res if warm < r[2] <= tail]
        print("%s: %d workers, %d calls: fork -> join %.0f calls/s, steady %.0f calls/s, %.2f ms per call in the worker"
              % ("broker" if mode == "1" else "direct", P, N, N / el, len(st) / max(tail - warm, 1e-9), 1e3 * np.mean([b - a for _, a, b in st])), flush=True)
        if mode == "1":
            try:
                print("   broker:", {k: v for k, v in broker.stats(0).items() if k in ("fn_calls", "requests", "errors")})
                broker._clients[(os.getpid(), 0)].close()
            except Exception as e:
                print("   (no stats: %r)" % e)
            broker.shutdown(0)
    import shutil; shutil.rmtree(d, ignore_errors=True)
