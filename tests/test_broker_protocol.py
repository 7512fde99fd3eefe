"""The GPU broker's host logic without a GPU (s2p_amd/broker.py): framing, descriptor passing, arenas, queueing, batching by
compatibility, replies to the right worker, workers that die mid-request, arena growth, idle exit.  The lanes run a numpy
stand-in instead of libs2p_hip (`Server(backend=...)` exists for exactly this test; the command line only offers the HIP
backend).  The GPU side of the same paths is tests/test_gpu_broker.py."""
import multiprocessing as mp
import os
import threading
import time

import numpy as np
import pytest

from s2p_amd import _lib, broker


class FakeBackend:
    """disp = im1 - im2, conf = number of requests in the call, mask = im1 > 0; sleeps so that requests pile up."""

    def __init__(self, delay=0.01, fail_on=None):
        self.delay, self.fail_on, self.pins = delay, fail_on, 0
        self.groups = []
        self.entered = 0                                          # calls that have reached a lane (counted before the delay)

    def start(self, device, nlanes):
        return 3                                                  # "three devices visible"

    def pin(self, addr, size):
        self.pins += 1
        return True

    def unpin(self, addr):
        self.pins -= 1

    def run(self, lane, grp, tmo, cap=1):
        self.entered += 1
        time.sleep(self.delay)
        assert len(grp) <= cap
        m = grp[0].msg
        if self.fail_on is not None and m["dmin"] == self.fail_on:
            raise _lib.HipError(_lib.EMPTY_RANGE, "empty range")
        self.groups.append([(r.msg["w"], r.msg["h"], r.msg["dmin"], r.msg["dmax"]) for r in grp])
        for r in grp:
            assert r.msg["params"] == m["params"] and r.msg["op"] == m["op"]
            v = lambda k, dt: r.arena.plane(r.msg["off"][k], (r.msg["h"], r.msg["w"]), dt)
            v("disp", np.float32)[:] = v("im1", np.float32) - v("im2", np.float32)
            v("mask", np.uint8)[:] = v("im1", np.float32) > 0
            if m["op"] == "census":
                v("conf", np.float32)[:] = len(grp)
        if getattr(self, "late_groups", False) and len(grp) > 1:       # the library's way: the stream is drained, THEN the call says TIMEOUT
            raise _lib.HipError(_lib.TIMEOUT, "deadline exceeded (the enqueued kernels are left to drain)")


@pytest.fixture
def server(tmp_path, monkeypatch):
    monkeypatch.setenv("S2P_HIP_BROKER_DIR", str(tmp_path))
    monkeypatch.delenv("S2P_HIP_DEVICE", raising=False)
    monkeypatch.delenv("LOCAL_RANK", raising=False)
    be = FakeBackend()
    srv = broker.Server(0, lanes=2, max_batch=4, idle_s=30.0, max_wait_ms=2.0, backend=be)     # (the idle test shortens it: a loaded box must not lose the broker between two steps of a test)
    th = threading.Thread(target=srv.serve, daemon=True)
    th.start()
    for _ in range(500):
        if os.path.exists(broker.sock_path(0)):
            break
        time.sleep(0.01)
    yield srv, be, th
    broker.shutdown(0)
    th.join(timeout=10)
    broker._clients.clear()
    broker._ndev.clear()


def _pair(seed, h, w):
    rng = np.random.default_rng(seed)
    return rng.random((h, w), np.float32) - 0.3, rng.random((h, w), np.float32)


def _call(seed, h=40, w=56, dmin=-8, dmax=7, kind="census", device=0):
    a, b = _pair(seed, h, w)

    def read_one(i, alloc):
        dst = alloc((h * w,), np.float32).reshape(h, w)
        dst[:] = (a, b)[i]
        return dst
    p = _lib.CensusParams(census_win=5, P1=8, P2=32, nb_dir=8, recursion=2, scales=1, subpix=1) if kind == "census" else \
        _lib.SgbmParams(win=3, P1=8, P2=32, lr=1)
    r = broker.match(kind, p, read_one, w, h, dmin, dmax, 30.0, device=device)
    ok = np.array_equal(r["disp"], a - b) and np.array_equal(r["mask"], (a > 0).astype(np.uint8))
    return ok, int(r["conf"][0, 0]) if kind == "census" else 1, os.getpid()


def _worker(seed):
    return _call(seed)


def test_requests_of_forked_workers_are_answered_and_batched(server):
    srv, be, _ = server
    ctx = mp.get_context("fork")
    with ctx.Pool(6) as pool:
        res = pool.map(_worker, range(48))
    assert all(ok for ok, _, _ in res)
    assert len({pid for _, _, pid in res}) >= 3
    assert max(n for _, n, _ in res) > 1                          # requests that waited together went through one call
    assert max(n for _, n, _ in res) <= 4                         # ... of at most max_batch
    for _ in range(200):                                          # the workers are gone: their arenas wait, mapped and page-locked, for the next Pool
        if len(srv.spare) == 6:
            break
        time.sleep(0.02)
    assert len(srv.spare) == 6 and be.pins == 6 and not srv.limbo
    st = srv.stat
    assert st["requests"] == 48 and st["calls"] < 48 and st["errors"] == 0 and st["recycled"] == 0
    with ctx.Pool(4) as pool:                                     # the next step's Pool: four of the six arenas serve again, nothing is page-locked anew
        res = pool.map(_worker, range(100, 132))
    assert all(ok for ok, _, _ in res)
    assert st["recycled"] == 4 and st["attached"] == 10 and be.pins == 6
    for _ in range(200):
        if len(srv.spare) == 6:
            break
        time.sleep(0.02)
    assert len(srv.spare) == 6


def test_an_arena_of_a_living_process_is_never_handed_out_again(server):
    """A process that hangs up (a timeout closes the connection, the process goes on) may still map its arena: it is released after
    the grace period, not recycled; the arena a client replaces by a larger one is released at once."""
    srv, be, _ = server
    srv.limbo_grace = 0.3
    c = broker.Client(0)
    c.reserve(1 << 20)
    c.reserve(8 << 20)                                           # replaced: the first one goes
    for _ in range(100):
        if be.pins == 1:
            break
        time.sleep(0.02)
    assert be.pins == 1 and not srv.spare and not srv.limbo
    c.sock.close()                                               # hung up, but this process lives
    time.sleep(0.1)
    assert len(srv.limbo) == 1 and not srv.spare
    for _ in range(200):
        if be.pins == 0:
            break
        time.sleep(0.02)
    assert be.pins == 0 and not srv.spare and not srv.limbo
    c2 = broker.Client(0)
    c2.reserve(8 << 20)
    assert not c2.recycled and srv.stat["recycled"] == 0
    c2.sock.close()


def test_only_compatible_requests_share_a_call(server):
    srv, be, _ = server
    out = {}

    def run(k, **kw):
        out[k] = _call_thread_safe(k, **kw)
    # threads of one process need their own connections: use distinct "devices" of the client cache
    ths = [threading.Thread(target=run, args=(k,), kwargs=dict(dmin=-8 - (k % 2))) for k in range(8)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert all(v[0] for v in out.values())
    assert all(v[1] <= 4 for v in out.values())                   # two keys x 4 requests: never 8 in one call


def _call_thread_safe(k, **kw):
    c = broker.Client(0)                                          # a private connection (the cache is per process and device)
    a, b = _pair(k, 24, 32)
    npx = 24 * 32
    a4 = broker._round_up(npx * 4, 4096)
    off = {"im1": 0, "im2": a4, "disp": 2 * a4, "conf": 3 * a4, "mask": 4 * a4}
    c.reserve(5 * a4)
    c.view(0, (24, 32), np.float32)[:] = a
    c.view(a4, (24, 32), np.float32)[:] = b
    p = _lib.CensusParams(recursion=2, scales=1)
    r = c.request({"op": "census", "w": 32, "h": 24, "dmin": kw.get("dmin", -8), "dmax": 7, "params": broker._params_dict(p), "off": off, "timeout": 30.0})
    ok = r["ok"] and np.array_equal(c.view(2 * a4, (24, 32), np.float32), a - b)
    n = r.get("batch", 0)
    c.sock.close()
    return ok, n


def test_errors_travel_back_as_the_library_status(server, monkeypatch):
    srv, be, _ = server
    be.fail_on = -99
    with pytest.raises(_lib.HipError) as e:
        _call(1, dmin=-99)
    assert e.value.code == _lib.EMPTY_RANGE
    assert _call(2)[0]                                            # the connection survives an error reply


def test_bad_requests_are_refused_not_executed(server):
    c = broker.Client(0)
    p = broker._params_dict(_lib.CensusParams())
    r = c.request({"op": "census", "w": 8, "h": 8, "dmin": 0, "dmax": 3, "params": p, "off": {"im1": 0, "im2": 0, "disp": 0, "conf": 0, "mask": 0}, "timeout": 1})
    assert not r["ok"] and "arena" in r["msg"]
    c.reserve(1 << 20)
    r = c.request({"op": "census", "w": 4096, "h": 4096, "dmin": 0, "dmax": 3, "params": p,
                   "off": {"im1": 0, "im2": 0, "disp": 0, "conf": 0, "mask": c.size - 16}, "timeout": 1})
    assert not r["ok"] and "outside" in r["msg"]
    assert not c.request({"op": "nonsense"})["ok"]
    c.sock.close()


def test_what_a_peer_may_not_ask(server, tmp_path, monkeypatch):
    """A message longer than the protocol allows closes the connection (arrays travel through the arena, never in the JSON); arrays
    are numbers only -- an object dtype over shared bytes would be pointers --; and the directory that holds the socket must be
    the user's own, closed to everybody else."""
    import socket
    import struct
    s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    s.connect(broker.sock_path(0))
    s.sendall(struct.pack("<I", broker.MAX_MSG + 1) + b"{")
    s.settimeout(5)
    try:
        assert s.recv(16) == b""                                 # closed, nothing executed
    except ConnectionResetError:
        pass                                                     # (closed with our bytes unread)
    s.close()
    c = broker.Client(0)
    c.reserve(1 << 20)
    r = c.request({"op": "fn", "name": "s2p_amd._lib:erode_mask", "free": 0,
                   "args": {"mask": {"__arr__": 0, "shape": [4, 4], "dtype": "|O"}, "radius": 1}})
    assert not r["ok"] and "dtype" in r["msg"]
    r = c.request({"op": "fn", "name": "os:system", "free": 0, "args": {"command": "true"}})
    assert not r["ok"] and "unknown function" in r["msg"]         # the registry of @broker.remote functions is the whitelist
    c.sock.close()
    open_dir = tmp_path / "open"
    open_dir.mkdir(mode=0o755)
    os.chmod(str(open_dir), 0o755)
    monkeypatch.setenv("S2P_HIP_BROKER_DIR", str(open_dir))
    with pytest.raises(broker.BrokerError):
        broker.sock_path(0)
    monkeypatch.setenv("S2P_HIP_BROKER_DIR", str(tmp_path))      # (the fixture's teardown talks to the server again)


def test_the_arena_grows_and_sgbm_requests_are_not_batched(server):
    srv, be, _ = server
    assert _call(3, h=64, w=64)[0]
    size0 = broker.client(0).size
    assert _call(4, h=2100, w=2100)[0]                            # 75 MB of planes: a new, larger arena replaces the first
    assert broker.client(0).size > size0
    ok, n, _ = _call(5, kind="sgbm")
    assert ok and n == 1


def test_default_device_is_pid_modulo_what_the_broker_reports(server, tmp_path):
    # the fake backend reports 3 devices; only broker 0 runs here, so ask which device a client WOULD pick
    c0 = broker.Client(0)
    assert c0.hello["ndev"] == 3
    c0.sock.close()


def _worker_any_device(seed):
    a, b = _pair(seed, 40, 56)

    def read_one(i, alloc):
        dst = alloc((40 * 56,), np.float32).reshape(40, 56)
        dst[:] = (a, b)[i]
        return dst
    p = _lib.CensusParams(census_win=5, P1=8, P2=32, nb_dir=8, recursion=2, scales=1, subpix=1)
    r = broker.match("census", p, read_one, 56, 40, -8, 7, 30.0)          # device=None: this worker's pid modulo the device count
    c = [k for k in broker._clients if k[0] == os.getpid() and broker._clients[k].mm is not None]
    return bool(np.array_equal(r["disp"], a - b)), os.getpid(), sorted(k[1] for k in c)


def test_workers_of_one_pool_spread_over_the_brokers_of_a_node(server, monkeypatch):
    """One broker per device (the node-wide Pool bench, `bench.py --workload pool --gpus N`): a worker without S2P_HIP_DEVICE asks the
    broker of device 0 how many devices there are and takes its pid modulo that count -- the rule of _lib.default_device.  Three
    brokers here (the stand-in backend reports three devices): every worker's requests land on the broker of ITS device, all three
    serve, and the answers are right."""
    srv0, be0, _ = server
    others = []
    for dev in (1, 2):
        be = FakeBackend()
        srv = broker.Server(dev, lanes=2, max_batch=4, idle_s=30.0, max_wait_ms=2.0, backend=be)
        th = threading.Thread(target=srv.serve, daemon=True)
        th.start()
        others.append((srv, be, th))
    for _ in range(500):
        if all(os.path.exists(broker.sock_path(d)) for d in (1, 2)):
            break
        time.sleep(0.01)
    try:
        ctx = mp.get_context("fork")
        with ctx.Pool(9) as pool:
            res = pool.map(_worker_any_device, range(200, 272))
        assert all(ok for ok, _, _ in res)
        for ok, pid, devs in res:
            assert devs == [pid % 3] or devs == sorted({0, pid % 3}), (pid, devs)     # (device 0 is also asked for the count; it only holds an arena when it is the worker's own)
        served = [srv0.stat["requests"]] + [s.stat["requests"] for s, _, _ in others]
        assert sum(served) == 72 and all(n > 0 for n in served), served
    finally:
        for dev in (1, 2):
            broker.shutdown(dev)
        for _, _, th in others:
            th.join(timeout=10)


def test_a_worker_that_dies_mid_request_does_not_hurt_the_others(server):
    srv, be, _ = server
    be.delay = 0.2
    ctx = mp.get_context("fork")
    p = ctx.Process(target=_worker, args=(7,))
    p.start()
    time.sleep(0.1)                                               # its request is inside a lane now
    p.kill()
    p.join()
    be.delay = 0.0
    assert _call(8)[0]
    for _ in range(200):                                          # the dead worker's arena: spare once its request left the lane -- or already this process's own
        if not srv.limbo and be.pins == 1 + len(srv.spare):
            break
        time.sleep(0.02)
    assert not srv.limbo and be.pins == 1 + len(srv.spare) and len(srv.spare) <= 1


def test_the_broker_leaves_when_idle(server):
    srv, be, th = server
    assert _call(9)[0]
    srv.idle_s = 0.6                                              # (read at every turn of the accept loop)
    broker.client(0).close()
    th.join(timeout=5)
    assert not th.is_alive()                                      # 0.6 s without a connection
    assert not os.path.exists(broker.sock_path(0))


def test_selection_rule(monkeypatch):
    monkeypatch.setenv("S2P_HIP_BROKER", "1")
    assert broker.wanted()
    monkeypatch.setenv("S2P_HIP_BROKER", "0")
    assert not broker.wanted()
    monkeypatch.delenv("S2P_HIP_BROKER")
    assert not broker.wanted()                                    # pytest's process has no multiprocessing parent
    ctx = mp.get_context("fork")
    with ctx.Pool(1) as pool:
        assert pool.apply(broker.wanted)                          # a Pool worker has


@broker.remote(inplace=("acc",))
def _remote_demo(img, stack, scale, rpc, acc, mode="sum", device=None):
    """What an array-level function of the package looks like to the broker: arrays in, scalars, a struct, arrays out, one in place."""
    acc += 1
    tot = sum(a.astype(np.float64).sum() for a in stack)
    out = img.astype(np.float64) * scale + rpc.delta
    return out, {"total": tot, "mode": mode, "big": np.arange(3 << 20, dtype=np.float32)}


def test_any_registered_function_travels_through_the_arena(server, monkeypatch):
    srv, be, _ = server
    monkeypatch.setenv("S2P_HIP_BROKER", "1")
    monkeypatch.setenv("S2P_HIP_DEVICE", "0")                                # (the stand-in reports 3 devices; only broker 0 runs here)
    img = np.arange(12, dtype=np.float32).reshape(3, 4)[:, ::2]              # non-contiguous on purpose
    stack = [np.full((2, 2), 1.5, np.float32), np.full((3,), 2, np.uint8)]
    rpc = _lib.RpcStruct()
    rpc.delta = 0.25
    acc = np.zeros(5, np.int32)
    out, info = _remote_demo(img, stack, 2.0, rpc, acc, mode="x", device=3)
    assert np.array_equal(out, img.astype(np.float64) * 2.0 + 0.25)
    assert info["total"] == 12.0 and info["mode"] == "x" and np.array_equal(info["big"], np.arange(3 << 20, dtype=np.float32))
    assert np.array_equal(acc, np.ones(5, np.int32))                        # the in-place argument came back
    assert srv.stat["fn_calls"] >= 1
    with pytest.raises(ValueError):
        broker.call("s2p_amd.broker:no_such_function", {})
    with pytest.raises(ValueError):
        broker.call("os:system", {"command": "true"})                        # only registered functions run in the broker
    monkeypatch.setenv("S2P_HIP_BROKER", "0")
    out2, _ = _remote_demo(img, stack, 2.0, rpc, acc)                       # not wanted: the function itself, here
    assert np.array_equal(out2, out) and acc[0] == 2


def test_tiles_of_different_shapes_share_a_call_when_their_depths_are_close(server):
    """Single-scale MGM requests of different sizes and ranges join one group (s2p_hip_census_sgm_host_batch_v) as long as the
    common depth wastes at most a quarter on any of them; a much narrower range does not join."""
    srv, be, _ = server
    be.delay = 0.15
    out = {}
    shapes = [(24, 32, -8, 7), (26, 30, -9, 8), (22, 36, -7, 9), (24, 32, -2, 1)]     # depths 16, 32?, ...: see below
    go = threading.Barrier(3)

    def run(k, key=None):
        h, w, dmin, dmax = shapes[k]
        key = k if key is None else key
        c = broker.Client(0)
        a, b = _pair(k, h, w)
        a4 = broker._round_up(h * w * 4, 4096)
        off = {"im1": 0, "im2": a4, "disp": 2 * a4, "conf": 3 * a4, "mask": 4 * a4}
        c.reserve(5 * a4)
        c.view(0, (h, w), np.float32)[:] = a
        c.view(a4, (h, w), np.float32)[:] = b
        p = _lib.CensusParams(recursion=2, scales=1, P2=32)
        if k < 3:
            go.wait(10)                                           # the three send together, with everything else already done
        r = c.request({"op": "census", "w": w, "h": h, "dmin": dmin, "dmax": dmax, "params": broker._params_dict(p), "off": off, "timeout": 30.0})
        out[key] = (r["ok"] and np.array_equal(c.view(2 * a4, (h, w), np.float32), a - b), r.get("batch"))
        c.sock.close()
    # one narrow-range request per lane, each seen on its lane before the next step (no sleeps to tune: a cold or loaded box once let the
    # three arrive one lane-time apart, each alone): the three then wait together until a lane is free
    blockers = []
    for n in range(2):
        blockers.append(threading.Thread(target=run, args=(3, "blocker%d" % n)))
        blockers[-1].start()
        for _ in range(1000):
            if be.entered > n:
                break
            time.sleep(0.005)
        assert be.entered == n + 1
    ths = [threading.Thread(target=run, args=(k,)) for k in range(3)]
    for t in ths:
        t.start()
    for t in ths + blockers:
        t.join()
    assert all(v[0] for v in out.values()), out
    mixed = [g for g in be.groups if len({(w, h) for w, h, _, _ in g}) > 1]
    assert mixed, be.groups                                   # tiles of different sizes went through one call
    assert all((24, 32, -2, 1) not in [(h_, w_, a_, b_) for w_, h_, a_, b_ in g] or len(g) == 1 for g in be.groups)   # depth 16 vs 32: alone


@broker.remote(out=lambda a: (a["img"].shape, np.float64))
def _remote_out_demo(img, gain, out=None, device=None):
    if out is None:
        out = np.empty(img.shape, np.float64)
    out[...] = img * gain
    return out


def test_a_declared_output_is_written_straight_into_the_arena(server, monkeypatch):
    """`@broker.remote(out=...)`: the worker reserves room for the result, the broker hands the function a view of it as `out=` and
    replies with a reference instead of a copy."""
    srv, be, _ = server
    monkeypatch.setenv("S2P_HIP_BROKER", "1")
    monkeypatch.setenv("S2P_HIP_DEVICE", "0")
    img = np.arange(5000, dtype=np.float32).reshape(50, 100)
    r = _remote_out_demo(img, 3.0)
    assert r.dtype == np.float64 and np.array_equal(r, img * 3.0)
    mine = np.zeros((50, 100))                                               # an explicit out= of the caller's: filled remotely, copied back? no:
    r2 = _remote_out_demo(img, 2.0, out=None)                                # ... only the declared slot mechanism is supported; None = slot
    assert np.array_equal(r2, img * 2.0)


def test_two_broker_processes_per_device_split_the_workers_by_pid(tmp_path, monkeypatch):
    """S2P_HIP_BROKER_PROCS = 2 (round 6): the device has two brokers -- two interpreters -- and a worker talks to shard `pid mod 2`;
    broker.stats() sums the shards, broker.shutdown() stops both.  (Stand-in backends, threads instead of processes: the protocol side.)"""
    monkeypatch.setenv("S2P_HIP_BROKER_DIR", str(tmp_path))
    monkeypatch.setenv("S2P_HIP_BROKER_PROCS", "2")
    monkeypatch.delenv("S2P_HIP_DEVICE", raising=False)
    monkeypatch.delenv("LOCAL_RANK", raising=False)
    assert broker.shards() == 2 and broker.sock_path(0, 0).endswith("gpu0.sock") and broker.sock_path(0, 1).endswith("gpu0.1.sock")
    bes = [FakeBackend(), FakeBackend()]
    srvs = [broker.Server(0, lanes=1, max_batch=4, idle_s=30.0, max_wait_ms=2.0, backend=bes[k], shard=k) for k in range(2)]
    ths = [threading.Thread(target=s.serve, daemon=True) for s in srvs]
    for t in ths:
        t.start()
    for _ in range(500):
        if all(os.path.exists(broker.sock_path(0, k)) for k in range(2)):
            break
        time.sleep(0.01)
    try:
        ctx = mp.get_context("fork")
        with ctx.Pool(6) as pool:
            res = pool.map(_worker, range(36))
        assert all(ok for ok, _, _ in res)
        pids = {pid for _, _, pid in res}
        want = [sum(1 for _, _, pid in res if pid % 2 == k) for k in range(2)]
        assert [s.stat["requests"] for s in srvs] == want and len({p % 2 for p in pids}) == 2, (want, pids)
        st = broker.stats(0)
        assert st["requests"] == 36 and len(st["shards"]) == 2 and st["lanes"] == 2
    finally:
        assert broker.shutdown(0)
        for t in ths:
            t.join(timeout=10)
        broker._clients.clear()
        broker._ndev.clear()
    assert not os.path.exists(broker.sock_path(0, 0)) and not os.path.exists(broker.sock_path(0, 1))


def test_a_late_batched_call_answers_the_members_that_were_on_time(server):
    """a batched call that reports TIMEOUT has drained its stream first, so every member's outputs are complete.  Members whose own
    deadline has not passed get their results (no second run of the group on an overloaded device); only the expired ones hear TIMEOUT."""
    srv, be, _ = server
    be.delay = 0.15
    be.late_groups = True
    out = {}

    def run(k, timeout):
        c = broker.Client(0)
        a, b = _pair(k, 24, 32)
        a4 = broker._round_up(24 * 32 * 4, 4096)
        off = {"im1": 0, "im2": a4, "disp": 2 * a4, "conf": 3 * a4, "mask": 4 * a4}
        c.reserve(5 * a4)
        c.view(0, (24, 32), np.float32)[:] = a
        c.view(a4, (24, 32), np.float32)[:] = b
        p = _lib.CensusParams(recursion=2, scales=1)
        r = c.request({"op": "census", "w": 32, "h": 24, "dmin": -8, "dmax": 7, "params": broker._params_dict(p), "off": off, "timeout": timeout})
        out[k] = (r, bool(np.array_equal(c.view(2 * a4, (24, 32), np.float32), a - b)))
        c.sock.close()
    blocker = threading.Thread(target=run, args=(9, 30.0))       # keeps lanes busy so that the others wait together and form ONE group
    blocker2 = threading.Thread(target=run, args=(8, 30.0))
    blocker.start(); blocker2.start()
    time.sleep(0.02)
    ths = [threading.Thread(target=run, args=(0, 30.0)), threading.Thread(target=run, args=(1, 30.0)), threading.Thread(target=run, args=(2, 0.05))]
    for t in ths:
        t.start()
    for t in ths + [blocker, blocker2]:
        t.join()
    runs_before = len(be.groups)
    assert out[0][0]["ok"] and out[0][1] and out[1][0]["ok"] and out[1][1], out          # on time: their results, although the call said TIMEOUT
    assert not out[2][0]["ok"] and out[2][0]["code"] == _lib.TIMEOUT, out[2]              # its own 50 ms had passed
    grouped = [g for g in be.groups if len(g) > 1]
    assert grouped and all(len(g) >= 2 for g in grouped)
    assert sum(1 for g in be.groups if len(g) == 1) <= 2, be.groups                      # nobody was run a second time, one by one


def test_census_depth_mirrors_the_library_rule():
    """broker.census_depth (grouping rule of the lanes and of tiles.compatible) = census_D of csrc/census_kernels.hip with the MGM recursion:
    the candidates rounded up to 16, and to 64 from 33 candidates on (whole lines per pixel, round 6)."""
    from s2p_amd.broker import census_depth
    assert [census_depth(0, n - 1) for n in (1, 16, 17, 32, 33, 48, 64, 65, 100, 128, 129, 144, 192, 193, 256, 257)] == \
           [16, 16, 32, 32, 64, 64, 64, 128, 128, 128, 192, 192, 192, 256, 256, 320]
    assert census_depth(-45, 50, subpix=2) == 192            # 191 half-pixel candidates
    assert census_depth(-20, 27) == 64 and census_depth(-5, 5) == 16
