"""s2p_amd/broker.py -- one GPU-owning process per device, serving the reference's forked Pool workers.

Why it exists (profiles/r04/pool_direct_sweep.json, cold_start_*.txt).  The reference runs every step as a fresh
`multiprocessing.Pool(nb_workers)` of forked workers, one tile x pair per task, files in and out (s2p/parallel.py:76-110,
s2p/__init__.py:166-196).  If every worker drives the GPU itself, each one pays the HIP runtime's start-up again at every
step -- 0.12 s alone, 0.7 s when 16 start together, 5-8 s when 64 do -- and their kernels then time-slice the device
process by process: 930 tiles/s with 8 workers, 640 with 16, 410 with 32.  MI355X-first means ONE process per GPU.  So the
workers stay what they are for the reference -- file readers and writers -- and hand the tile to this broker:

    worker (s2p_amd.block_matching.compute_disparity_map, unchanged signature)
        reads the two TIFFs straight into its shared arena (a memfd of the broker's, page-locked there; the worker maps it through a
        descriptor it is sent -- an arena a worker of an earlier Pool left is handed out again as it is)
        -> request over a Unix socket: shape, range, parameters, offsets of the five planes in the arena
        <- reply when the results are in the arena; encodes them into the output files
    broker (this module, `python -m s2p_amd.broker --device d`; started on demand by the first worker that finds no socket)
        `lanes` threads, one libs2p_hip context (= HIP stream + workspace) each; a free lane takes every compatible request
        that is waiting (same shape, range, parameters; at most `max_batch`) and runs them through ONE
        s2p_hip_census_sgm_host_batch call -- the batched launch sequence whose aggregation kernel runs at 0.51 of the HBM
        roofline instead of 0.26 for a lone tile -- with the DMAs going straight from / to the workers' pages.

Nothing of the reference's orchestration changes: same Pool, same task function, same files.  The broker survives the Pools of
the successive steps (it leaves after `idle_s` seconds without a client), so the runtime start-up is paid once per GPU and job.
Results are byte-identical to the in-process path (tests/test_gpu_broker.py).  There is no CPU path here either: a broker that
cannot start or a library error surfaces in the worker as the same HipError / TimeoutExpired / CalledProcessError.

Selection: S2P_HIP_BROKER = 1 (always), 0 (never), or unset / "auto": workers of a multiprocessing Pool (any process with a
multiprocessing parent) go through the broker, a plain process drives the GPU itself.
"""
import ctypes
import json
import mmap
import os
import socket
import struct
import subprocess
import sys
import threading
import time

PROTOCOL = 1
MAX_MSG = 16 << 20                                               # bytes of JSON per message (arrays travel through the arena, never in here)
_ALIGN = 4096
HERE = os.path.dirname(os.path.abspath(__file__))


def broker_dir():
    d = os.environ.get("S2P_HIP_BROKER_DIR")
    if not d:
        d = os.path.join(os.environ.get("XDG_RUNTIME_DIR") if os.access(os.environ.get("XDG_RUNTIME_DIR", "/nonexistent"), os.W_OK) else "/tmp",
                         "s2p_hip_broker_%d" % os.getuid())
    os.makedirs(d, mode=0o700, exist_ok=True)
    # the socket, the lock and the log live here, and whoever can connect can have this user's GPU run array functions on memory it
    # hands over: the directory must be this user's own and closed to everybody else (a /tmp name somebody else created first is refused)
    st = os.lstat(d)
    import stat as _stat
    if not _stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise BrokerError("broker directory %s must be a directory of uid %d with mode 0700 (found uid %d, mode %o): set S2P_HIP_BROKER_DIR"
                          % (d, os.getuid(), st.st_uid, st.st_mode & 0o7777))
    return d


def shards():
    """GPU-owning broker processes per device (S2P_HIP_BROKER_PROCS, default 1).  Every request costs the broker ~0.2 ms of
    interpreter time -- framing, the queue, the arena bookkeeping -- under ONE interpreter lock per process, which is what limits tiles of
    a quarter of the headline's size and below (profiles/r05/broker_small_tiles.txt).  With N processes a worker talks to shard
    `pid mod N`: N interpreters, N x `lanes` contexts on the device, each shard batching the requests of its own workers.  The device's
    process fence (csrc/api.hip) admits 8."""
    try:
        return max(1, min(8, int(os.environ.get("S2P_HIP_BROKER_PROCS", "1"))))
    except ValueError:
        return 1


def shard_of(pid=None):
    return (os.getpid() if pid is None else int(pid)) % shards()


def _stem(device, shard=0):
    return "gpu%d" % int(device) if not shard else "gpu%d.%d" % (int(device), int(shard))


def sock_path(device, shard=0):
    return os.path.join(broker_dir(), _stem(device, shard) + ".sock")


def wanted():
    """Does this process hand its tiles to the broker?  (module docstring: S2P_HIP_BROKER)"""
    v = os.environ.get("S2P_HIP_BROKER", "auto").strip().lower()
    if v in ("1", "on", "yes", "true"):
        return True
    if v in ("0", "off", "no", "false"):
        return False
    import multiprocessing as mp
    return mp.parent_process() is not None


# ---- framing: 4-byte little-endian length + JSON; file descriptors ride on the header bytes (SCM_RIGHTS) -----------------------
def send_msg(sock, obj, fds=()):
    b = json.dumps(obj, separators=(",", ":")).encode()
    data = struct.pack("<I", len(b)) + b
    if fds:
        n = socket.send_fds(sock, [data], list(fds))
        if n < len(data):
            sock.sendall(data[n:])
    else:
        sock.sendall(data)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        c = sock.recv(n - len(buf))
        if not c:
            raise EOFError
        buf += c
    return bytes(buf)


def recv_msg(sock, want_fds=False):
    fds = []
    if want_fds:
        hdr, fds, _, _ = socket.recv_fds(sock, 4, 4)
        if not hdr:
            raise EOFError
        if len(hdr) < 4:
            hdr += _recv_exact(sock, 4 - len(hdr))
    else:
        hdr = _recv_exact(sock, 4)
    (n,) = struct.unpack("<I", hdr)
    try:
        if n > MAX_MSG:
            raise ValueError("message of %d bytes: beyond the protocol's %d" % (n, MAX_MSG))
        return json.loads(_recv_exact(sock, n)), fds
    except Exception:
        for fd in fds:                                          # descriptors that rode on a message nobody will handle
            os.close(fd)
        raise


def _round_up(n, a):
    return (n + a - 1) // a * a


# =================================================================================================================================
# client side (runs in the Pool worker; never touches the HIP runtime)
# =================================================================================================================================
class BrokerError(RuntimeError):
    pass


class Client:
    """One connection + one shared arena per (process, device)."""

    def __init__(self, device, shard=None):
        self.device = int(device)
        self.pid = os.getpid()
        self.shard = shard_of(self.pid) if shard is None else int(shard)
        self.sock = None
        self.mm = None
        self.fd = -1
        self.size = 0
        self.pinned = None
        self.recycled = False
        self.hello = None
        self.setup_ms = 0.0                                     # connecting (+ starting the broker) and attaching arenas, so far
        t = time.perf_counter()
        self._connect()
        self.setup_ms += (time.perf_counter() - t) * 1e3

    # -- connection / on-demand start ------------------------------------------------------------------------------------------
    def _try_connect(self):
        s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        try:
            s.connect(sock_path(self.device, self.shard))
        except OSError:
            s.close()
            return None
        return s

    def _connect(self):
        """Connect (starting a broker when none listens) and say hello.  A broker that leaves idle between accepting this connection
        and answering the hello shows as EOF / a reset here: once more, through the starter's lock, on a fresh one."""
        for attempt in (0, 1):
            s = self._try_connect() if attempt == 0 else None
            if s is None:
                s = self._start_and_connect()
            try:
                send_msg(s, {"op": "hello", "protocol": PROTOCOL, "pid": self.pid})
                r, _ = recv_msg(s)
            except (EOFError, OSError) as e:
                s.close()
                if attempt == 0:
                    continue
                raise BrokerError("the GPU broker for device %d closed the connection at hello (%s)" % (self.device, e.__class__.__name__))
            if not r.get("ok"):
                s.close()
                raise BrokerError("broker refused the connection: %s" % r.get("msg"))
            self.sock, self.hello = s, r
            return

    def _start_and_connect(self):
        """No broker listens for this device: start one (a detached process of its own session that outlives this worker and
        its Pool), unless another worker is doing so right now -- a lock file serialises the starters."""
        import fcntl
        stem = _stem(self.device, self.shard)
        lock = open(os.path.join(broker_dir(), stem + ".lock"), "w")
        try:
            fcntl.flock(lock, fcntl.LOCK_EX)
            s = self._try_connect()
            if s is not None:
                return s
            try:
                os.unlink(sock_path(self.device, self.shard))   # a socket file nobody answers on: its broker is gone
            except OSError:
                pass
            log = open(os.path.join(broker_dir(), stem + ".log"), "ab")
            env = dict(os.environ)
            env["PYTHONPATH"] = os.path.dirname(HERE) + os.pathsep + env.get("PYTHONPATH", "")
            env.pop("S2P_HIP_BROKER", None)
            proc = subprocess.Popen([sys.executable, "-m", "s2p_amd.broker", "--device", str(self.device), "--shard", str(self.shard)], stdin=subprocess.DEVNULL,
                                    stdout=log, stderr=log, start_new_session=True, close_fds=True, env=env, cwd=os.path.dirname(HERE))
            log.close()
            deadline = time.monotonic() + float(os.environ.get("S2P_HIP_BROKER_START_TIMEOUT", "120"))
            while time.monotonic() < deadline:
                s = self._try_connect()
                if s is not None:
                    return s
                if proc.poll() is not None:
                    raise BrokerError("the GPU broker for device %d exited with status %s at start-up; see %s"
                                      % (self.device, proc.returncode, os.path.join(broker_dir(), stem + ".log")))
                time.sleep(0.01)
            raise BrokerError("the GPU broker for device %d did not come up within the start timeout" % self.device)
        finally:
            try:
                fcntl.flock(lock, fcntl.LOCK_UN)
            finally:
                lock.close()

    # -- arena -----------------------------------------------------------------------------------------------------------------
    def reserve(self, nbytes):
        """A shared arena of at least nbytes (memfd: no name in any file system, gone with the last process that maps it)."""
        if nbytes <= self.size:
            return
        t = time.perf_counter()
        size = _round_up(max(int(nbytes), 1 << 20), 2 << 20)     # no larger than needed: the broker page-locks it
        # The BROKER owns the arenas and hands this process a descriptor: an arena a worker of an earlier Pool left behind is mapped
        # and page-locked already, so the workers of the next step attach in a fraction of a millisecond instead of queueing up
        # behind each other's hipHostRegister (64 workers: 126 ms median, 385 ms worst, profiles/r04/pool_broker_sweep_final.json)
        send_msg(self.sock, {"op": "arena", "bytes": size})
        r, fds = recv_msg(self.sock, want_fds=True)
        if not r.get("ok") or len(fds) != 1:
            for fd in fds:
                os.close(fd)
            raise BrokerError("broker could not provide an arena: %s" % r.get("msg"))
        fd, size = fds[0], int(r["bytes"])
        mm = mmap.mmap(fd, size)
        old = (self.mm, self.fd)
        self.mm, self.fd, self.size, self.pinned = mm, fd, size, bool(r.get("pinned"))
        self.recycled = bool(r.get("recycled"))
        if old[0] is not None:
            try:
                old[0].close()
            except BufferError:
                pass                                            # a caller still holds a view: the pages go with its last reference
            os.close(old[1])
        self.setup_ms += (time.perf_counter() - t) * 1e3

    def view(self, off, shape, dtype):
        import numpy as np
        dt = np.dtype(dtype)
        n = 1
        for v in shape:
            n *= int(v)
        return np.frombuffer(self.mm, dtype=dt, count=n, offset=off).reshape(shape)

    def request(self, msg, timeout=None):
        self.sock.settimeout(None if timeout is None else float(timeout))
        try:
            send_msg(self.sock, msg)
            r, _ = recv_msg(self.sock)
        except socket.timeout:
            self.close()                                        # the reply may still come: this connection's stream is out of step
            raise
        finally:
            if self.sock is not None:
                self.sock.settimeout(None)
        return r

    def close(self, keep_arena=False):
        """Drop the connection.  The arena's mapping and descriptor go with it (a reconnect leaked one of each) unless
        `keep_arena` (the caller still copies its inputs out of it) or a caller still holds a view of the pages."""
        try:
            if self.sock is not None:
                self.sock.close()
        finally:
            self.sock = None
            _clients.pop((self.pid, self.device), None)
            if not keep_arena:
                self.release_arena()

    def release_arena(self):
        mm, fd = self.mm, self.fd
        self.mm, self.fd, self.size = None, -1, 0
        if mm is not None:
            try:
                mm.close()
            except BufferError:
                pass                                            # a view is alive: the pages go with its last reference
        if fd is not None and fd >= 0:
            try:
                os.close(fd)
            except OSError:
                pass


_clients = {}
_client_lock = threading.Lock()
_ndev = {}


def client(device=None):
    """The connection of this process to the broker of `device` (default: S2P_HIP_DEVICE, else LOCAL_RANK, else pid mod the number
    of devices the broker of device 0 reports -- the rule of _lib.default_device without initialising HIP here)."""
    pid = os.getpid()
    with _client_lock:
        if device is None:
            for k in ("S2P_HIP_DEVICE", "LOCAL_RANK"):
                if k in os.environ:
                    device = int(os.environ[k])
                    break
        if device is None:
            if pid not in _ndev:
                c0 = _clients.get((pid, 0))
                if c0 is None or c0.sock is None:
                    c0 = _clients[(pid, 0)] = Client(0)
                _ndev.clear()
                _ndev[pid] = max(1, int(c0.hello.get("ndev", 1)))
                if pid % _ndev[pid] != 0 and c0.mm is None:
                    # only asked for the count: leave.  (Kept open, every worker of a node-wide Pool -- 64 per device -- held a
                    # connection and a server thread in the broker of device 0: 644 descriptors there with 8 devices,
                    # tools/pool_dryrun.py, profiles/r05/pool_dryrun_8x64.json)
                    c0.close()
                    _clients.pop((pid, 0), None)
            device = pid % _ndev[pid]
        c = _clients.get((pid, int(device)))
        if c is None or c.sock is None:
            c = _clients[(pid, int(device))] = Client(device)
        return c


def _client_or_hip_error(device):
    """client(), with the failures of connecting -- no broker could be started, or it closed the connection at hello twice -- turned into
    the HipError the module's contract promises (block_matching._raise_for maps it like any library error)."""
    from s2p_amd import _lib
    try:
        return client(device)
    except (BrokerError, EOFError, OSError) as e:
        raise _lib.HipError(_lib.RUNTIME_ERROR, "no GPU broker for this worker: %s" % (e,))


def _params_dict(p):
    return {n: getattr(p, n) for n, _ in p._fields_}


_digests = {}
_last_egress = [0.0]


def _params_digest(pd):
    """A short name of a parameter set: the broker groups requests by it (one string compare per request instead of a canonical dump)."""
    t = tuple(sorted(pd.items()))
    d = _digests.get(t)
    if d is None:
        import hashlib
        if len(_digests) > 256:
            _digests.clear()
        d = _digests[t] = hashlib.sha1(json.dumps(pd, sort_keys=True).encode()).hexdigest()[:20]
    return d


def match(kind, params, read_one, w, h, dmin, dmax, timeout, device=None):
    """One matcher call through the broker.  `read_one(i, alloc)` must return input image i (0, 1) as a float32 (h, w) array,
    using `alloc(shape, dtype)` for its memory where it can (the shim reads the TIFFs straight into the arena that way).
    Returns dict(disp, mask[, conf]) as views of the arena -- valid until this process's next broker call."""
    import numpy as np
    from s2p_amd import _lib
    c = _client_or_hip_error(device)
    npx = int(w) * int(h)
    # the five planes back to back at ONE stride, rounded to 256 bytes: the library then moves a tile in two transfers, inputs up and
    # outputs down (csrc/api.hip: common_plane_stride -- a gap of a page would not count as "back to back")
    a4 = _round_up(npx * 4, 256)
    off = {"im1": 0, "im2": a4, "disp": 2 * a4, "conf": 3 * a4, "mask": 4 * a4}
    c.reserve(4 * a4 + _round_up(npx, _ALIGN))
    def fill(i):
        key = ("im1", "im2")[i]
        dst = c.view(off[key], (h, w), np.float32)

        def alloc(shape, dtype=np.float32, _o=off[key]):
            if np.dtype(dtype) != np.float32 or not np.dtype(dtype).isnative or int(np.prod(shape)) != npx:
                return np.empty(shape, dtype)
            return c.view(_o, shape, np.float32)
        arr = read_one(i, alloc)
        if not (isinstance(arr, np.ndarray) and arr.dtype == np.float32 and arr.shape == (h, w) and arr.ctypes.data == dst.ctypes.data):
            np.copyto(dst, np.asarray(arr, np.float32).reshape(h, w))   # a file the reader could not decode in place
    t_read = time.perf_counter()
    if npx >= (1 << 18):                                         # two 4 MB reads from the page cache: side by side (the copies release the GIL)
        from s2p_amd import io as rio
        other = rio._pool().submit(fill, 1)
        fill(0)
        other.result()
    else:
        fill(0)
        fill(1)
    t_read = (time.perf_counter() - t_read) * 1e3
    pd = _params_dict(params)
    msg = {"op": kind, "w": int(w), "h": int(h), "dmin": int(dmin), "dmax": int(dmax), "params": pd, "pk": _params_digest(pd), "off": off,
           "timeout": -1.0 if timeout is None else float(timeout)}
    if os.environ.get("S2P_HIP_BROKER_TRACE"):                   # latency accounting (bench_pool.py --trace): when this request left, and how long
        msg["ts"], msg["pe"] = time.time(), _last_egress[0]     # the previous reply took from the broker's send to this worker's wake-up
    try:
        r = c.request(msg, None if timeout is None or timeout < 0 else float(timeout) + 30.0)
    except socket.timeout:
        raise _lib.HipError(_lib.TIMEOUT, "no reply from the GPU broker within the call's timeout")
    except (EOFError, OSError) as e:
        # the broker went away (it left idle between two steps just as this worker wrote, or it was killed): one more try on a fresh
        # one -- the inputs are still in this worker's arena, which the new broker has to be given again
        mm_old, size_old = c.mm, c.size
        c.close(keep_arena=True)
        try:
            c2 = client(device)
            c2.reserve(size_old)
            c2.mm[:size_old] = mm_old[:size_old]
            r = c2.request(msg, None if timeout is None or timeout < 0 else float(timeout) + 30.0)
        except (EOFError, OSError, BrokerError, socket.timeout):
            raise _lib.HipError(_lib.RUNTIME_ERROR, "the GPU broker went away during the call (%s) and a second one did not answer" % (e.__class__.__name__,))
        finally:
            c.release_arena()
        c = c2
    if "ts" in r:
        _last_egress[0] = (time.time() - float(r["ts"])) * 1e3
    if not r.get("ok"):
        raise _lib.HipError(int(r.get("code", _lib.RUNTIME_ERROR)), "broker: " + str(r.get("msg")))
    out = {"disp": c.view(off["disp"], (h, w), np.float32), "mask": c.view(off["mask"], (h, w), np.uint8)}
    if kind == "census":
        out["conf"] = c.view(off["conf"], (h, w), np.float32)
    out["batch"] = r.get("batch", 1)
    out["read_ms"] = t_read
    out["setup_ms"], c.setup_ms = c.setup_ms, 0.0               # what this call spent connecting / attaching (first call of a worker)
    return out


# ---- any array-level function of the package through the broker ------------------------------------------------------------------
# The other file-level mirrors a Pool worker calls (image_apply_homography of the rectification step, masking.erosion right after the
# matcher, cargarse_basura / merge_n of the tri-stereo tail, the triangulation helpers) are thin: read files, ONE array-level call,
# write files.  `@broker.remote()` on that array-level function forwards the call when this process hands its GPU work to the
# broker (wanted()): arrays travel through the shared arena, ctypes structs as bytes, scalars as JSON; the broker runs the very same
# function on its own context and the results come back through the arena.  One mechanism for all of them, no per-function protocol.
_REMOTE = {}
_serving = [False]                                               # True inside the broker process: run the function itself


class _OutSlot:
    """Stands for an output array of a remote call: room in the arena, nothing to send."""

    def __init__(self, shape, dtype):
        self.shape, self.dtype = tuple(int(v) for v in shape), dtype


def remote(inplace=(), out=None):
    """Decorator: `inplace` names positional-or-keyword array arguments the function modifies in place (copied back).  `out`: for a
    function with an `out=` keyword, a callable (arguments) -> (shape, dtype) of its result: the broker then lets the function write
    straight into the worker's arena instead of allocating a result and copying it there."""
    import functools
    import inspect

    def deco(fn):
        name = fn.__module__ + ":" + fn.__qualname__
        _REMOTE[name] = fn
        sig = inspect.signature(fn)

        @functools.wraps(fn)
        def wrapper(*a, **k):
            if _serving[0] or not wanted():
                return fn(*a, **k)
            bound = sig.bind(*a, **k)
            bound.arguments.pop("device", None)                  # the broker IS the device
            args = dict(bound.arguments)
            if out is not None and args.get("out") is None:
                args["out"] = _OutSlot(*out(args))
            return call(name, args, inplace)
        wrapper.__wrapped_local__ = fn
        return wrapper
    return deco


def _marshal(v, place):
    import numpy as np
    if isinstance(v, (np.ndarray, _OutSlot)):
        return place(v)
    if isinstance(v, ctypes.Structure):
        return {"__struct__": bytes(v).hex()}
    if isinstance(v, (list, tuple)):
        return {"__seq__": [_marshal(x, place) for x in v], "tuple": isinstance(v, tuple)}
    if isinstance(v, dict):
        return {"__map__": {str(k): _marshal(x, place) for k, x in v.items()}}
    if isinstance(v, (np.integer,)):
        return int(v)
    if isinstance(v, (np.floating,)):
        return float(v)
    if isinstance(v, float) and (v != v or v in (float("inf"), float("-inf"))):
        return {"__float__": repr(v)}
    if v is None or isinstance(v, (bool, int, float, str)):
        return v
    raise TypeError("broker: cannot send a %s" % type(v).__name__)


def _plain_dtype(text):
    """The dtype of an array a peer describes: numbers and booleans only (an object or structured dtype over shared bytes would
    turn them into pointers)."""
    import numpy as np
    dt = np.dtype(str(text))
    if dt.kind not in "biufc" or dt.hasobject or dt.fields is not None:
        raise ValueError("arrays of dtype %s do not travel through the broker" % dt)
    return dt


def _unmarshal(v, view):
    if isinstance(v, dict):
        if "__arr__" in v:
            return view(v)
        if "__struct__" in v:
            from s2p_amd import _lib
            return _lib.RpcStruct.from_buffer_copy(bytes.fromhex(v["__struct__"]))
        if "__seq__" in v:
            seq = [_unmarshal(x, view) for x in v["__seq__"]]
            return tuple(seq) if v.get("tuple") else seq
        if "__map__" in v:
            return {k: _unmarshal(x, view) for k, x in v["__map__"].items()}
        if "__float__" in v:
            return float(v["__float__"])
    return v


def _arrays_bytes(v):
    import numpy as np
    if isinstance(v, _OutSlot):
        return _round_up(int(np.prod(v.shape)) * np.dtype(v.dtype).itemsize, _ALIGN)
    if isinstance(v, np.ndarray):
        return _round_up(v.nbytes, _ALIGN)
    if isinstance(v, (list, tuple)):
        return sum(_arrays_bytes(x) for x in v)
    if isinstance(v, dict):
        return sum(_arrays_bytes(x) for x in v.values())
    return 0


def _arrays_bytes_outside(v, arena):
    """Bytes the results still need in the arena (arrays that already lie in it need none)."""
    import numpy as np
    if isinstance(v, np.ndarray):
        inside = v.flags.c_contiguous and arena.base <= v.ctypes.data and v.ctypes.data + v.nbytes <= arena.base + arena.size
        return 0 if inside else _round_up(v.nbytes, _ALIGN)
    if isinstance(v, (list, tuple)):
        return sum(_arrays_bytes_outside(x, arena) for x in v)
    if isinstance(v, dict):
        return sum(_arrays_bytes_outside(x, arena) for x in v.values())
    return 0


def call(name, arguments, inplace=(), device=None):
    """Run the registered function `name` in the broker with `arguments` (dict); returns what it returns (arrays are copies)."""
    import numpy as np
    from s2p_amd import _lib
    c = _client_or_hip_error(device)
    need_in = _arrays_bytes(arguments)
    want = need_in + max(need_in, 16 << 20)                      # room for results of about the inputs' size; the broker says if it needs more
    retried = False
    attempt = 0
    r = None
    while attempt < 4:
        c.reserve(want)
        top = [0]
        placed = {}

        def place(a0):
            if isinstance(a0, _OutSlot):                         # room only
                off = top[0]
                top[0] += _round_up(int(np.prod(a0.shape)) * np.dtype(a0.dtype).itemsize, _ALIGN)
                return {"__arr__": off, "shape": list(a0.shape), "dtype": np.dtype(a0.dtype).str}
            a = np.ascontiguousarray(a0)
            off = top[0]
            top[0] += _round_up(a.nbytes, _ALIGN)
            if a.nbytes:
                c.view(off, a.shape, a.dtype)[...] = a
            placed[id(a0)] = off
            return {"__arr__": off, "shape": list(a.shape), "dtype": a.dtype.str}
        msg = {"op": "fn", "name": name, "args": {k: _marshal(v, place) for k, v in arguments.items()}}
        msg["free"] = top[0]
        try:
            r = c.request(msg, 900.0)
        except (EOFError, OSError) as e:
            c.close()
            if not retried:                                      # the broker left or was killed: once more on a fresh one (not one of the
                retried = True                                   # four attempts: those are for growing the arena)
                try:
                    c = client(device)
                    continue
                except (BrokerError, EOFError, OSError):
                    pass
            raise _lib.HipError(_lib.RUNTIME_ERROR, "the GPU broker went away during %s (%s)" % (name, e.__class__.__name__))
        attempt += 1
        if r.get("ok"):
            break
        if "need" in r and attempt < 4:
            want = int(r["need"])
            continue
        if r.get("exc") in ("ValueError", "NotImplementedError", "TypeError", "AssertionError"):
            raise {"ValueError": ValueError, "NotImplementedError": NotImplementedError, "TypeError": TypeError,
                   "AssertionError": AssertionError}[r["exc"]](r.get("msg"))
        raise _lib.HipError(int(r.get("code", _lib.RUNTIME_ERROR)), "broker: " + str(r.get("msg")))
    if r is None or not r.get("ok"):                             # (every path out of the loop without a result raised above; belt and braces)
        raise _lib.HipError(_lib.RUNTIME_ERROR, "broker: no result for %s" % name)

    def view(d):
        return np.array(c.view(int(d["__arr__"]), tuple(d["shape"]), _plain_dtype(d["dtype"])))     # a copy: the arena is reused by the next call
    for key in inplace:                                          # arguments the function modified where they lay (the arena): back into the caller's arrays
        v = arguments.get(key)
        if isinstance(v, np.ndarray) and id(v) in placed:
            v[...] = c.view(placed[id(v)], v.shape, v.dtype)
    return _unmarshal(r.get("ret"), view)


def _bare_request(device, shard, msg):
    """One request on a connection of its own to a RUNNING shard (never starts one); None when nobody listens."""
    s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    try:
        s.connect(sock_path(device, shard))
    except OSError:
        s.close()
        return None
    try:
        send_msg(s, msg)
        return recv_msg(s)[0]
    except (EOFError, OSError):
        return None
    finally:
        s.close()


def stats(device=0, reset=False):
    """The counters of the device's broker; with several shards (S2P_HIP_BROKER_PROCS) their sums, the per-shard dicts under "shards"."""
    if shards() == 1:
        return client(device).request({"op": "stats", "reset": bool(reset)})
    per = [r for r in (_bare_request(device, k, {"op": "stats", "reset": bool(reset)}) for k in range(shards())) if r]
    if not per:
        return {"ok": False, "msg": "no shard of device %d is running" % device}
    tot = dict(per[0])
    for r in per[1:]:
        for k, v in r.items():
            if isinstance(v, bool) or k in ("lanes", "max_batch", "started", "uptime_s"):
                continue
            if isinstance(v, (int, float)):
                tot[k] = tot.get(k, 0) + v
            elif isinstance(v, dict) and k in ("batch_hist", "run_ms"):
                acc = dict(tot.get(k, {}))
                for kk, vv in v.items():
                    acc[kk] = [a + b for a, b in zip(acc[kk], vv)] if isinstance(vv, list) and kk in acc else (acc.get(kk, 0) + vv if not isinstance(vv, list) else vv)
                tot[k] = acc
            elif isinstance(v, list) and k == "slow_calls":
                tot[k] = (tot.get(k, []) + v)[-64:]
    tot["lanes"] = sum(int(r.get("lanes", 0)) for r in per)
    tot["shards"] = per
    return tot


def shutdown(device=0):
    """Ask the broker of `device` -- every shard of it -- to leave (tests; a job's end does not need it: a broker leaves by itself when idle)."""
    import glob
    found = False
    names = {sock_path(device, 0)} | set(glob.glob(os.path.join(broker_dir(), "gpu%d.*.sock" % int(device))))
    for path in sorted(names):
        base = os.path.basename(path)[:-5].split(".")
        found = _shutdown_one(device, int(base[1]) if len(base) > 1 else 0) or found
    return found


def shutdown_all():
    """Every broker that listens in this process's broker directory (test sessions)."""
    import glob
    import re
    devs = set()
    for path in glob.glob(os.path.join(broker_dir(), "gpu*.sock")):
        m = re.match(r"gpu(\d+)(?:\.\d+)?\.sock$", os.path.basename(path))
        if m:
            devs.add(int(m.group(1)))
    for d in sorted(devs):
        shutdown(d)


def _shutdown_one(device, shard):
    s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    try:
        s.connect(sock_path(device, shard))
    except OSError:
        s.close()
        return False
    try:
        send_msg(s, {"op": "shutdown"})
        try:
            recv_msg(s)
        except (EOFError, OSError):
            pass
    finally:
        s.close()
    for _ in range(500):
        if not os.path.exists(sock_path(device, shard)):
            break
        time.sleep(0.01)
    return True


# =================================================================================================================================
# server side
# =================================================================================================================================
class HipBackend:
    """What the broker's lanes run: libs2p_hip.so through s2p_amd._lib (the only backend the command line offers; the protocol
    tests in tests/test_broker_protocol.py hand Server a numpy stand-in to exercise queueing, arenas and batching without a GPU)."""

    def start(self, device, nlanes):
        from s2p_amd import _lib
        n = _lib.device_count()                                 # the one HIP initialisation of this GPU's job
        if not (0 <= device < n):
            raise SystemExit("s2p_amd.broker: device %d of %d visible" % (device, n))
        self.ctxs, self.sized, self.pcache = [], set(), {}
        self.device, self.fn_pool, self.fn_lock = device, None, threading.Lock()
        for _ in range(nlanes):
            p = ctypes.c_void_p()
            _lib.check(_lib.lib().s2p_hip_ctx_create(device, None, ctypes.byref(p)))
            self.ctxs.append(p)
        return n

    def pin(self, addr, size):
        from s2p_amd import _lib
        try:
            _lib.host_register(addr, size)
            return True
        except _lib.HipError:
            return False                                        # the transfers then stage through the runtime's own buffers

    def unpin(self, addr):
        from s2p_amd import _lib
        _lib.host_unregister(addr)

    def fn_context(self):
        """A context for one array-level call (with-block): borrowed from a small pool so that the workers' calls overlap."""
        from s2p_amd import _lib
        import contextlib
        import queue

        @contextlib.contextmanager
        def borrowed():
            with self.fn_lock:
                if self.fn_pool is None:
                    self.fn_pool = queue.Queue()
                    for _ in range(int(os.environ.get("S2P_HIP_BROKER_FN_CONTEXTS", "3"))):
                        p = ctypes.c_void_p()
                        _lib.check(_lib.lib().s2p_hip_ctx_create(self.device, None, ctypes.byref(p)))
                        self.fn_pool.put(p)
            c = self.fn_pool.get()
            try:
                with _lib.thread_context(c):
                    yield
            finally:
                self.fn_pool.put(c)
        return borrowed()

    def run(self, lane, grp, tmo, cap=1):
        """grp: compatible requests (same op, shape, range, parameters); cap: the most such requests a call may carry.
        Raises _lib.HipError on failure."""
        from s2p_amd import _lib
        ctx, m = self.ctxs[lane], grp[0].msg
        if m["op"] == "census":
            pc = self.pcache                                    # parameter structs by the requests' parameter key
            p = pc.get(grp[0].key[5])
            if p is None:
                if len(pc) > 256:
                    pc.clear()
                p = pc[grp[0].key[5]] = _lib.CensusParams(**{n: (float(v) if n == "lr_tau" else int(v)) for n, v in m["params"].items()})
            if cap > 1 and (lane, grp[0].key) not in self.sized:
                # first tile of this shape on this lane: size the workspace for full batches at once (it only grows, and every
                # growth is a hipFree + hipMalloc of gigabytes that stalls the whole device)
                try:
                    _lib.census_sgm_host_batch_reserve(ctx, cap, m["w"], m["h"], m["dmin"], m["dmax"], p)
                except _lib.HipError:
                    pass                                        # (too large for `cap` tiles at once: the call below sizes for what it gets)
                self.sized.add((lane, grp[0].key))
            ad = lambda key: [r.arena.base + int(r.msg["off"][key]) for r in grp]
            if all(r.key == grp[0].key for r in grp):
                _lib.census_sgm_host_batch(ctx, ad("im1"), ad("im2"), m["w"], m["h"], m["dmin"], m["dmax"], p, ad("disp"), ad("conf"), ad("mask"), tmo)
            else:                                               # tiles of different sizes / ranges: one launch sequence all the same
                _lib.census_sgm_host_batch_v(ctx, ad("im1"), ad("im2"), [r.msg["w"] for r in grp], [r.msg["h"] for r in grp], [r.msg["dmin"] for r in grp],
                                             [r.msg["dmax"] for r in grp], p, ad("disp"), ad("conf"), ad("mask"), tmo)
        else:
            import numpy as np
            p = _lib.SgbmParams(**{n: int(v) for n, v in m["params"].items()})
            for r in grp:
                v = lambda key, dt: r.arena.plane(r.msg["off"][key], (m["h"], m["w"]), dt)
                _lib.sgbm(v("im1", np.float32), v("im2", np.float32), m["dmin"], m["dmax"], params=p, timeout=tmo, want_cost=False, ctx=ctx,
                          out={"disp": v("disp", np.float32), "mask": v("mask", np.uint8)})


def _pid_gone(pid):
    """No such process any more, or only its zombie (exited, not yet reaped by the Pool's parent)."""
    try:
        with open("/proc/%d/stat" % int(pid), "rb") as f:
            st = f.read()
        return st[st.rindex(b")") + 2:st.rindex(b")") + 3] in (b"Z", b"X")
    except (OSError, ValueError):
        return True


class _Arena:
    def __init__(self, fd, size, backend):
        import numpy as np
        self.backend = backend
        self.mm = mmap.mmap(fd, size)
        self.fd = os.dup(fd)                                    # what a worker maps: sent with the answer to its "arena" request, maybe to several workers in turn
        self.size = size
        self.np = np.frombuffer(self.mm, dtype=np.uint8)
        self.base = self.np.ctypes.data
        self.busy = 0                                           # requests of this arena inside a lane
        self.dead = False
        self.pinned = False
        self.pinning = False                                    # the pinner thread is inside hipHostRegister for this arena: requests wait
        self.served = 0                                         # requests answered from this arena

    def plane(self, off, shape, dtype):
        import numpy as np
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        return self.np[int(off):int(off) + n].view(dtype).reshape(shape)

    def release(self):
        if self.pinned:
            self.backend.unpin(self.base)
            self.pinned = False
        self.np = None
        try:
            self.mm.close()
        except BufferError:
            pass
        if self.fd >= 0:
            os.close(self.fd)
            self.fd = -1


_OK_REPLY = {}                                                  # batch size -> the framed bytes of {"ok":true,"batch":n}


def _ok_reply(n):
    b = _OK_REPLY.get(n)
    if b is None:
        body = json.dumps({"ok": True, "batch": n}, separators=(",", ":")).encode()
        b = _OK_REPLY[n] = struct.pack("<I", len(body)) + body
    return b


class _Conn:
    def __init__(self, sock):
        self.sock = sock
        self.arena = None
        self.wlock = threading.Lock()
        self.pid = None
        self.peer_pid = None                                    # SO_PEERCRED: the process that connected
        self.buf = bytearray()                                  # bytes received and not yet parsed (requests are a few hundred bytes: one recv each)

    def recv(self):
        """The next request of this connection (no request carries descriptors: the arenas are the broker's own).  One recv() per
        message in the common case instead of one for the length and one for the body."""
        buf = self.buf
        while True:
            if len(buf) >= 4:
                (n,) = struct.unpack_from("<I", buf, 0)
                if n > MAX_MSG:
                    raise ValueError("message of %d bytes: beyond the protocol's %d" % (n, MAX_MSG))
                if len(buf) >= 4 + n:
                    body = bytes(buf[4:4 + n])
                    del buf[:4 + n]
                    return json.loads(body)
            c = self.sock.recv(65536)
            if not c:
                raise EOFError
            buf += c

    def reply(self, obj, fds=()):
        try:
            with self.wlock:
                send_msg(self.sock, obj, fds)
        except OSError:
            pass                                                # the worker is gone (Pool.terminate): nothing to tell it

    def reply_ok(self, n):
        try:
            with self.wlock:
                self.sock.sendall(_ok_reply(n))
        except OSError:
            pass


def census_depth(dmin, dmax, subpix=1):
    """Depth of the cost / e-volumes the library lays out for a range with the MGM recursion (csrc/census_kernels.hip: census_D): the
    candidates rounded up to 16, from 48 on to 64 (whole lines per pixel).  For the grouping rules of the broker and the tile scheduler."""
    d = ((2 if int(subpix) == 2 else 1) * (int(dmax) - int(dmin)) + 16) // 16 * 16
    return d if d <= 32 else (d + 63) // 64 * 64


class _Req:
    __slots__ = ("conn", "arena", "msg", "key", "t", "depth", "levels", "npx", "tmo")

    def __init__(self, conn, arena, msg, key):
        self.conn, self.arena, self.msg, self.key, self.t = conn, arena, msg, key, time.monotonic()
        # what the lanes' grouping rule needs, once per request instead of once per look at the queue
        pr = msg["params"]
        w, h = int(msg["w"]), int(msg["h"])
        self.npx = w * h
        self.tmo = float(msg.get("timeout", -1.0))
        if msg["op"] == "census":
            self.depth = census_depth(msg["dmin"], msg["dmax"], pr.get("subpix", 1))
            n, sc = 1, int(pr.get("scales", 1))                # census_levels of csrc/census_kernels.hip: multi-scale tiles need the same count
            while n < sc and min((w + 1) // 2, (h + 1) // 2) >= 128:
                n, w, h = n + 1, (w + 1) // 2, (h + 1) // 2
            self.levels = n
        else:
            self.depth, self.levels = 0, 1


class Server:
    def __init__(self, device, lanes=3, max_batch=8, idle_s=120.0, max_wait_ms=3.0, backend=None, shard=0):
        self.backend = backend if backend is not None else HipBackend()
        self.shard = int(shard)
        self.device, self.nlanes, self.max_batch, self.idle_s, self.max_wait = int(device), int(lanes), int(max_batch), float(idle_s), float(max_wait_ms) * 1e-3
        self.busy = 0                                           # lanes inside the library right now
        self.to_pin = []                                        # arenas waiting for the pinner thread
        self.to_free = []                                       # ... and dead ones waiting to be unmapped
        # Arenas outlive their workers: the reference forks a fresh Pool per step, and mapping + page-locking 64 arenas at every step is
        # what the workers of a new Pool wait for (the registrations serialise on the broker's memory-map lock).  An arena whose worker
        # process is GONE (not merely disconnected: a live process may still hold the mapping) waits in `spare` for the next worker that
        # asks for that much room, mapped and page-locked as it is.
        self.recycle = os.environ.get("S2P_HIP_BROKER_RECYCLE", "1") != "0"
        self.spare = []                                         # arenas ready to be handed out again
        self.limbo = []                                         # (arena, pid, since): the connection closed, is the process gone?
        self.spare_max_bytes = int(float(os.environ.get("S2P_HIP_BROKER_SPARE_MB", "8192")) * (1 << 20))
        self.limbo_grace = 2.0                                  # seconds a disconnected but living process keeps its arena out of `spare`
        self.hetero = os.environ.get("S2P_HIP_BROKER_HETERO", "1") != "0"     # tiles of different shapes may share a launch
        self.last_attach = 0.0
        self.cv = threading.Condition()
        self.pending = []
        self.nconn = 0
        self.last_active = time.monotonic()
        self.stop = False
        self.stat = {"requests": 0, "calls": 0, "batch_hist": {}, "errors": 0, "started": time.time(), "attached": 0, "pinned": 0, "recycled": 0, "run_ms": {}, "queue_ms": 0.0, "slow_calls": []}
        self.t0 = time.monotonic()
        self.path = sock_path(self.device, self.shard)

    # -- life cycle ------------------------------------------------------------------------------------------------------------
    def serve(self):
        n = self.ndev = self.backend.start(self.device, self.nlanes)
        lst = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        tmp = self.path + ".%d" % os.getpid()
        try:
            os.unlink(tmp)
        except OSError:
            pass
        lst.bind(tmp)
        os.chmod(tmp, 0o600)
        lst.listen(1024)
        os.rename(tmp, self.path)                               # the socket appears under its name only once it accepts
        ino = os.stat(self.path).st_ino
        lst.settimeout(0.5)
        self.lanes = [threading.Thread(target=self.lane, args=(k,), daemon=True) for k in range(self.nlanes)]
        self.lanes.append(threading.Thread(target=self.pinner, daemon=True))
        for t in self.lanes:
            t.start()
        print("s2p_amd.broker: device %d of %d%s, %d lanes, batches of up to %d, pid %d, socket %s"
              % (self.device, n, " (shard %d of %d)" % (self.shard, shards()) if shards() > 1 else "", self.nlanes, self.max_batch, os.getpid(), self.path), flush=True)
        try:
            while not self.stop:
                try:
                    s, _ = lst.accept()
                except socket.timeout:
                    with self.cv:
                        idle = self.nconn == 0 and not self.pending and time.monotonic() - self.last_active > self.idle_s
                    if idle:
                        break
                    continue
                peer_pid = None
                try:                                           # the directory is 0700 already; the kernel's word on who is calling all the same
                    peer_pid, uid, _gid = struct.unpack("3i", s.getsockopt(socket.SOL_SOCKET, socket.SO_PEERCRED, struct.calcsize("3i")))
                except (OSError, AttributeError):
                    uid = os.getuid()
                if uid != os.getuid():
                    s.close()
                    continue
                with self.cv:
                    self.nconn += 1
                    self.last_active = time.monotonic()
                conn = _Conn(s)
                conn.peer_pid = peer_pid
                threading.Thread(target=self.connection, args=(conn,), daemon=True).start()
        finally:
            try:
                if os.path.exists(self.path) and os.stat(self.path).st_ino == ino:      # not a successor's socket
                    os.unlink(self.path)
            except OSError:
                pass
            lst.close()
            with self.cv:
                self.stop = True
                self.cv.notify_all()
            for t in self.lanes:
                t.join(timeout=30)
            with self.cv:
                left = self.spare + [x[0] for x in self.limbo] + self.to_free
                self.spare, self.limbo, self.to_free = [], [], []
            for x in left:
                x.release()
            print("s2p_amd.broker: leaving after %d requests in %d calls (batch sizes %s)"
                  % (self.stat["requests"], self.stat["calls"], json.dumps(self.stat["batch_hist"], sort_keys=True)), flush=True)

    # -- one thread per worker connection: parse, queue; the lanes answer --------------------------------------------------------
    def connection(self, conn):
        try:
            while True:
                msg = conn.recv()
                op = msg.get("op")
                if op == "hello":
                    conn.pid = msg.get("pid")
                    if msg.get("protocol") != PROTOCOL:
                        conn.reply({"ok": False, "msg": "protocol %s, broker speaks %d" % (msg.get("protocol"), PROTOCOL)})
                    else:
                        conn.reply({"ok": True, "ndev": self.ndev, "device": self.device, "pid": os.getpid(), "lanes": self.nlanes, "max_batch": self.max_batch})
                elif op == "arena":
                    self.provide(conn, msg)
                elif op in ("census", "sgbm"):
                    self.enqueue(conn, msg)
                elif op == "fn":
                    self.run_fn(conn, msg)
                elif op == "stats":
                    with self.cv:
                        st = json.loads(json.dumps(dict(self.stat, pending=len(self.pending), connections=self.nconn, ok=True, lanes=self.nlanes,
                                                        spare_arenas=len(self.spare), spare_mb=round(sum(x.size for x in self.spare) / 2.0 ** 20, 1),
                                                        max_batch=self.max_batch, uptime_s=round(time.monotonic() - self.t0, 3),
                                                        cpu_s=round(time.process_time(), 4))))     # user + system CPU of this broker process so far, all threads
                        if msg.get("reset"):                   # (bench_pool.py reads the counters of one Pool at a time)
                            keep = {"started": self.stat["started"]}
                            self.stat.update({"requests": 0, "calls": 0, "batch_hist": {}, "errors": 0, "attached": 0, "pinned": 0, "recycled": 0, "run_ms": {},
                                              "queue_ms": 0.0, "slow_calls": [], "fn_calls": 0}, **keep)
                            self.stat.pop("trace", None)
                    conn.reply(st)
                elif op == "shutdown":
                    conn.reply({"ok": True})
                    with self.cv:
                        self.stop = True
                        self.cv.notify_all()
                    return
                else:
                    conn.reply({"ok": False, "code": 5, "msg": "unknown op %r" % (op,)})
        except (EOFError, OSError, ValueError):
            pass
        finally:
            with self.cv:
                a, conn.arena = conn.arena, None
                if a is not None:
                    a.dead = True
                    a.owner_pid = conn.peer_pid
                    free_now = a.busy == 0 and not a.pinning    # (the pinner frees what dies under its hands)
                else:
                    free_now = False
                self.nconn -= 1
                self.last_active = time.monotonic()
            if free_now:
                self.free_later(a)
            try:
                conn.sock.close()
            except OSError:
                pass

    def provide(self, conn, msg):
        """An arena of at least msg["bytes"] for this connection, as a descriptor: a spare one (mapped and page-locked by an earlier
        worker's request) when one is large enough without wasting more than half of itself, else a new memfd."""
        try:
            size = int(msg["bytes"])
            if size <= 0 or size > (64 << 30):
                raise ValueError("arena of %d bytes" % size)
        except Exception as e:
            conn.reply({"ok": False, "msg": "%s: %s" % (e.__class__.__name__, e)})
            return
        a = None
        with self.cv:
            fit = [x for x in self.spare if size <= x.size <= 2 * size]
            if fit:
                a = min(fit, key=lambda x: x.size)
                self.spare.remove(a)
                a.dead, a.owner_pid, a.served = False, None, 0
                self.stat["recycled"] += 1
        recycled = a is not None
        if a is None:
            fd = -1
            try:
                fd = os.memfd_create("s2p_hip_arena")
                os.ftruncate(fd, size)
                a = _Arena(fd, size, self.backend)
            except Exception as e:
                conn.reply({"ok": False, "msg": "%s: %s" % (e.__class__.__name__, e)})
                return
            finally:
                if fd >= 0:
                    os.close(fd)
        # S2P_HIP_BROKER_PIN: "eager" (default) page-locks a new arena before the worker gets its answer; "lazy" leaves it to the pinner
        # thread (after the arena's first tile, outside attach bursts); "0" never.  Measured with 64 workers x 3 Pools of 1 536 tiles
        # BEFORE arenas were recycled (profiles/r04/pin_probe.txt): eager 1 320-1 360 tiles/s steady, 1 020-1 100 fork -> join; lazy
        # 1 090-1 120 / 905-950 (the registrations then run during the steady state and the first tiles travel through the runtime's
        # bounce buffers); none 1 150-1 300 / 930-1 020
        mode = os.environ.get("S2P_HIP_BROKER_PIN", "eager")
        with self.cv:
            old, conn.arena = conn.arena, a
            self.stat["attached"] += 1
            free_now = old is not None and old.busy == 0 and not old.pinning
            if old is not None:
                old.dead = True                                 # (its process lives and may still map it: released, never handed out again)
                old.owner_pid = None
            self.last_attach = time.monotonic()
            if mode == "lazy" and not a.pinned and a not in self.to_pin:   # (a spare arena that was never page-locked gets its turn too)
                self.to_pin.append(a)
                self.cv.notify_all()
        if free_now:
            self.free_later(old)
        if not a.pinned and mode not in ("lazy", "0"):
            a.pinned = bool(self.backend.pin(a.base, a.size))
            with self.cv:
                self.stat["pinned"] += int(a.pinned)
        conn.reply({"ok": True, "bytes": a.size, "recycled": recycled, "pinned": a.pinned if (recycled or mode != "lazy") else "soon"}, fds=[a.fd])

    def free_later(self, a):
        """Hand a dead arena to the pinner thread: un-registering and un-mapping take the memory-map lock the attaching workers of
        the next Pool need, so they wait for a quiet moment too."""
        with self.cv:
            if self.recycle and a.fd >= 0 and getattr(a, "owner_pid", None) and not self.stop:
                self.limbo.append((a, a.owner_pid, time.monotonic()))
            else:
                self.to_free.append(a)
            self.cv.notify_all()

    def pinner(self):
        """Page-locks the arenas one after the other, each at a moment when no request of it is inside the library (a transfer from
        pages that are being registered is not something to rely on); until then the arena works as pageable memory."""
        while True:
            with self.cv:
                # registering faults the arena's pages in under the process's memory-map lock, which every attaching worker's mmap
                # needs: during the burst of a starting Pool the registrations would queue the workers up behind each other (measured:
                # 64 attaches took 140-270 ms each).  So an arena is page-locked once it has served a tile and no worker has attached
                # for 50 ms -- a worker that lives for one tile never is.
                def quiet():
                    return time.monotonic() - self.last_attach > 0.05
                def ready(x):
                    return x.dead or (x.busy == 0 and x.served >= 1 and quiet())
                self.settle_limbo()
                while not self.stop and not any(ready(a) for a in self.to_pin) and not (self.to_free and (quiet() or len(self.to_free) > 512)):
                    self.cv.wait(0.05 if (self.to_pin or self.to_free or self.limbo) else 0.5)
                    self.settle_limbo()
                if self.stop:
                    return
                if self.to_free and (quiet() or len(self.to_free) > 512):
                    a = self.to_free.pop()                       # un-map / un-pin what dead workers left, also outside the bursts
                    freeing = True
                else:
                    freeing = False
                    a = next(x for x in self.to_pin if ready(x))
            if freeing:
                a.release()
                continue
            with self.cv:
                self.to_pin.remove(a)
                if a.dead:
                    continue
                a.pinning = True
            ok = False
            try:
                ok = bool(self.backend.pin(a.base, a.size))
            finally:
                with self.cv:
                    a.pinning = False
                    a.pinned = ok
                    self.stat["pinned"] += int(ok)
                    free_now = a.dead and a.busy == 0
                    self.cv.notify_all()
                if free_now:
                    self.free_later(a)

    def settle_limbo(self):
        """(under self.cv) Arenas whose connection closed: once the worker PROCESS is gone the arena is spare -- a process that only
        hung up may still map it, so after `limbo_grace` seconds of it living on the arena is released instead."""
        if not self.limbo:
            return
        now, keep = time.monotonic(), []
        for a, pid, since in self.limbo:
            if _pid_gone(pid):
                held = sum(x.size for x in self.spare)
                if held + a.size <= self.spare_max_bytes and len(self.spare) < 1024:
                    self.spare.append(a)
                else:
                    self.to_free.append(a)
            elif now - since > self.limbo_grace:
                self.to_free.append(a)
            else:
                keep.append((a, pid, since))
        self.limbo = keep

    def run_fn(self, conn, msg):
        """A registered array-level function on this connection's thread (the library serialises calls that share a context; the lanes'
        batches run on their own streams beside it).  Arguments are views of the worker's arena, results are placed behind them."""
        import importlib
        import numpy as np
        a = conn.arena
        try:
            if a is None:
                raise ValueError("no arena attached")
            name = str(msg["name"])
            fn = _REMOTE.get(name)                               # the registry is the whitelist: only @broker.remote functions run here
            if fn is None and name.split(":")[0].split(".")[0] == "s2p_amd":
                importlib.import_module(name.split(":")[0])      # (a module of this package the broker has not imported yet registers on import)
                fn = _REMOTE.get(name)
            if fn is None:
                raise ValueError("unknown function %s" % name)

            def view(d):
                shape, dt, off = tuple(int(v) for v in d["shape"]), _plain_dtype(d["dtype"]), int(d["__arr__"])
                n = int(np.prod(shape)) * dt.itemsize
                if off < 0 or off + n > a.size or any(v < 0 for v in shape):
                    raise ValueError("array outside the arena")
                return a.plane(off, shape, dt)
            with self.cv:
                while a.pinning:
                    self.cv.wait(0.1)
                a.busy += 1
                self.stat["fn_calls"] = self.stat.get("fn_calls", 0) + 1
                self.last_active = time.monotonic()
            try:
                args = {k: _unmarshal(v, view) for k, v in msg["args"].items()}
                fc = getattr(self.backend, "fn_context", None)
                if fc is not None:
                    with fc():
                        ret = fn(**args)
                else:
                    ret = fn(**args)
                top = [_round_up(int(msg.get("free", 0)), _ALIGN)]
                need = top[0] + _arrays_bytes_outside(ret, a)
                if need > a.size:
                    conn.reply({"ok": False, "need": need + (1 << 20)})
                    return

                def place(r):
                    if r.flags.c_contiguous and a.base <= r.ctypes.data and r.ctypes.data + r.nbytes <= a.base + a.size:
                        return {"__arr__": r.ctypes.data - a.base, "shape": list(r.shape), "dtype": r.dtype.str}   # written where it belongs (out=)
                    r = np.ascontiguousarray(r)
                    off = top[0]
                    top[0] += _round_up(r.nbytes, _ALIGN)
                    if r.nbytes:
                        a.plane(off, r.shape, r.dtype)[...] = r
                    return {"__arr__": off, "shape": list(r.shape), "dtype": r.dtype.str}
                conn.reply({"ok": True, "ret": _marshal(ret, place)})
            finally:
                with self.cv:
                    a.busy -= 1
                    free_now = a.dead and a.busy == 0 and not a.pinning
                    self.last_active = time.monotonic()
                    self.cv.notify_all()
                if free_now:
                    self.free_later(a)
        except Exception as e:
            conn.reply({"ok": False, "code": int(getattr(e, "code", 3)), "exc": e.__class__.__name__, "msg": "%s: %s" % (e.__class__.__name__, e)})

    def enqueue(self, conn, msg):
        a = conn.arena
        try:
            if a is None:
                raise ValueError("no arena attached")
            w, h = int(msg["w"]), int(msg["h"])
            npx = w * h
            if w <= 0 or h <= 0:
                raise ValueError("bad size")
            off = msg["off"]
            for k, nb in (("im1", 4 * npx), ("im2", 4 * npx), ("disp", 4 * npx), ("conf", 4 * npx), ("mask", npx)):
                o = int(off[k])
                if o < 0 or o % 4 or o + nb > a.size:
                    raise ValueError("plane %s outside the arena" % k)
            pk = msg.get("pk")                                  # the client's digest of its parameters (match()): one string compare instead of a dump per request
            key = (msg["op"], w, h, int(msg["dmin"]), int(msg["dmax"]), pk if isinstance(pk, str) and pk else json.dumps(msg["params"], sort_keys=True))
            req = _Req(conn, a, msg, key)                       # (parses the fields the lanes group by: a malformed parameter set is a bad request, here)
            trace = (time.time() - float(msg["ts"]), float(msg.get("pe", 0.0))) if "ts" in msg else None     # (S2P_HIP_BROKER_TRACE in the workers)
        except Exception as e:
            conn.reply({"ok": False, "code": 5, "msg": "bad request: %s" % e})
            return
        with self.cv:
            while a.pinning:
                self.cv.wait(0.1)
            a.busy += 1
            self.pending.append(req)
            self.stat["requests"] += 1
            if trace is not None:
                tr = self.stat.setdefault("trace", {"n": 0, "ingress_ms": 0.0, "egress_ms": 0.0, "reply_ms": 0.0})
                tr["n"] += 1
                tr["ingress_ms"] += trace[0] * 1e3
                tr["egress_ms"] += trace[1]
            self.last_active = time.monotonic()
            self.cv.notify()

    # -- a lane: one context; takes every compatible waiting request, one library call, answers -----------------------------------
    def take(self):
        """The next group of compatible requests for a free lane.  Dispatch rule: at once when the device is idle (no lane inside the
        library: latency matters, there is nothing to share a launch with) or when a full batch waits; otherwise the lane lets the
        group grow -- the device is busy anyway -- until it is full or its oldest request has waited `max_wait_ms`."""
        with self.cv:
            while True:
                if not self.pending:
                    if self.stop:
                        return None
                    self.cv.wait(0.5)
                    continue
                first = self.pending[0]
                pr = first.msg["params"]
                # what one batched launch sequence covers: the MGM modes (multi-scale ones with P2 <= 115: census_batches in
                # csrc/census_kernels.hip); anything else would run one after the other inside the call
                P2 = int(pr.get("P2", 32))
                cap = self.max_batch if (first.key[0] == "census" and int(pr.get("recursion", 0)) >= 1 and
                                         (int(pr.get("scales", 1)) <= 1 or P2 <= 115)) else 1
                bpc = 17 if int(pr.get("nb_dir", 8)) > 8 else 9  # bytes per candidate and tile: the cost volume + 8 (16 directions: 16) e-volumes
                if cap > 1:                                      # ... and what a lane's workspace should hold: 24 GB per lane
                    cap = max(1, min(cap, int(24e9 // (bpc * max(1, first.npx * first.depth)))))
                if cap > 1 and P2 <= 115 and self.hetero:
                    # single-scale tiles of OTHER sizes and ranges join the group (s2p_hip_census_sgm_host_batch_v: one aggregation launch
                    # with per-tile geometry) when the volumes' common depth -- the widest range's -- wastes little on them: at least three
                    # quarters of it are their own candidates; and at most 16 tiles, 24 GB of volumes
                    lim, pkey, l0 = min(cap, 16), first.key[5], first.levels
                    # (requests of the first one's own depth come first: the volumes' depth is rounded up to 64 candidates, so a queue of a
                    # real job's tiles holds a handful of depths and a group of one depth pads nothing)
                    grp, cand, dlo, dhi, taken = [], 0, first.depth, first.depth, set()
                    for same_depth in (True, False):
                        for r in self.pending:
                            if len(grp) >= lim:
                                break
                            if id(r) in taken or r.key[0] != first.key[0] or r.key[5] != pkey or r.levels != l0:
                                continue
                            if same_depth and r.depth != first.depth:
                                continue
                            lo, hi = min(dlo, r.depth), max(dhi, r.depth)
                            if lo * 4 < hi * 3 or (cand + r.npx) * hi * bpc > 24e9:
                                continue
                            grp.append(r)
                            taken.add(id(r))
                            cand += r.npx
                            dlo, dhi = lo, hi
                else:
                    k0 = first.key
                    grp = []
                    for r in self.pending:
                        if r.key == k0:
                            grp.append(r)
                            if len(grp) >= cap:
                                break
                age = time.monotonic() - first.t
                # how long a short group may wait for company: not at all on an idle device, a quarter of max_wait with one lane busy of
                # three, all of it once every other lane is busy (measured: with 16 workers the full wait left lanes idle -- 906 tiles/s
                # against 1 045 for tiles that cannot batch at all --, without any wait 64 workers got 980 instead of 1 370)
                limit = self.max_wait * min(1.0, (self.busy / max(1.0, self.nlanes - 1.0)) ** 2)
                if len(grp) >= cap or self.busy == 0 or age >= limit or self.stop:
                    ids = {id(r) for r in grp}
                    self.pending = [r for r in self.pending if id(r) not in ids]
                    self.busy += 1
                    if self.pending:
                        self.cv.notify()
                    return grp, cap
                self.cv.wait(max(limit - age, 1e-4))              # woken by an arrival, a lane that finished, or the age limit

    def lane(self, k):
        while True:
            got = self.take()
            if got is None:
                return
            grp, cap = got
            err = None
            t_take = time.monotonic()

            def remaining(members, now):                        # the call's deadline: the tightest of its members' (-1 = none)
                ts = [r.tmo - (now - r.t) for r in members if r.tmo >= 0]
                return max(0.001, min(ts)) if ts else -1.0

            def failure(e):                                     # HipError carries the library's status; anything else is a bug here that
                return {"ok": False, "code": int(getattr(e, "code", 3)), "msg": "%s: %s" % (e.__class__.__name__, e)}   # must not take the job down
            replies = None
            try:
                self.backend.run(k, grp, remaining(grp, t_take), cap)
            except Exception as e:
                err = failure(e)
                # One member's fault must not fail its companions: tiles of unrelated workers share a call, and a refused
                # parameter, a tile too large for the hand-off ring or one request's short time-out would fail up to 15 valid tiles that
                # succeed as single calls.  So a group that failed for anything but a device fault (a HIP error poisons the context:
                # everybody hears about it) runs again member by member, each under its own deadline, and every request gets ITS answer.
                if len(grp) > 1 and err["code"] == 2:
                    # TIMEOUT of the batched call: the library drains the stream before it reports it (csrc/api.hip: wait_stream_raw), so
                    # every member's outputs are complete and valid.  Members whose own deadline has not passed get their results; only
                    # the expired ones hear TIMEOUT.  (Until round 6 the group ran again member by member: twice the GPU work at the very
                    # moment the device was overloaded, and members that were on time made late -- ADVICE r05.)
                    replies = []
                    now = time.monotonic()
                    for r in grp:
                        t_own = float(r.msg.get("timeout", -1.0))
                        if t_own >= 0 and t_own - (now - r.t) <= 0:
                            replies.append({"ok": False, "code": 2, "msg": "HipError: the request's time-out passed while its group was being served"})
                        else:
                            replies.append({"ok": True, "batch": len(grp)})
                    err = None if all(x.get("ok") for x in replies) else err
                elif len(grp) > 1 and err["code"] != 3:
                    replies = []
                    for r in grp:
                        now = time.monotonic()
                        t_own = float(r.msg.get("timeout", -1.0))
                        if t_own >= 0 and t_own - (now - r.t) <= 0:
                            replies.append({"ok": False, "code": 2, "msg": "HipError: the request's time-out passed while its group was being served"})
                            continue
                        try:
                            self.backend.run(k, [r], remaining([r], now), 1)
                            replies.append({"ok": True, "batch": 1})
                        except Exception as e1:
                            replies.append(failure(e1))
                    err = None if all(x.get("ok") for x in replies) else err
            t_done = time.monotonic()
            if replies is None and err is None:
                for r in grp:
                    if "ts" in r.msg:
                        r.conn.reply({"ok": True, "batch": len(grp), "ts": time.time()})
                    else:
                        r.conn.reply_ok(len(grp))
                if "trace" in self.stat:
                    self.stat["trace"]["reply_ms"] += (time.monotonic() - t_done) * 1e3
            else:
                for i, r in enumerate(grp):
                    r.conn.reply(replies[i] if replies is not None else err)
            dead = []
            with self.cv:
                rm = self.stat["run_ms"].setdefault(str(len(grp)), [0, 0.0])          # per batch size: calls, total ms inside the library
                rm[0] += 1
                rm[1] = round(rm[1] + (t_done - t_take) * 1e3, 3)
                if (t_done - t_take) > 0.05:
                    self.stat["slow_calls"].append([round(t_take - self.t0, 3), k, len(grp), round((t_done - t_take) * 1e3, 1)])
                    del self.stat["slow_calls"][:-64]
                self.stat["queue_ms"] = round(self.stat["queue_ms"] + sum(t_take - r.t for r in grp) * 1e3, 3)
                self.stat["calls"] += 1
                self.stat["errors"] += int(err is not None)
                h = self.stat["batch_hist"]
                h[str(len(grp))] = h.get(str(len(grp)), 0) + 1
                self.busy -= 1
                self.cv.notify_all()                            # "the device is idle" may hold now
                for r in grp:
                    r.arena.served += 1
                    r.arena.busy -= 1
                    if r.arena.dead and r.arena.busy == 0 and not r.arena.pinning and r.arena not in dead:
                        dead.append(r.arena)
                self.last_active = time.monotonic()
            for a in dead:
                self.free_later(a)


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="s2p_amd GPU broker: one per device; Pool workers connect through s2p_amd.block_matching")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--shard", type=int, default=0, help="which of the device's S2P_HIP_BROKER_PROCS broker processes this one is (workers talk to shard pid mod N)")
    ap.add_argument("--lanes", type=int, default=int(os.environ.get("S2P_HIP_BROKER_LANES", "0")),
                    help="library contexts (HIP streams) taking batches side by side; default: 3 for the device, shared out over its shards (at least 1 each)")
    ap.add_argument("--max-batch", type=int, default=int(os.environ.get("S2P_HIP_BROKER_BATCH", "8")), help="tiles per library call at most")
    ap.add_argument("--idle", type=float, default=float(os.environ.get("S2P_HIP_BROKER_IDLE", "120")), help="leave after this many seconds without a client")
    ap.add_argument("--max-wait-ms", type=float, default=float(os.environ.get("S2P_HIP_BROKER_WAIT_MS", "3")),
                    help="while the device is busy a group may wait this long for more compatible requests before it is dispatched short")
    ap.add_argument("--stats", action="store_true", help="print the counters of the running broker of --device and exit")
    ap.add_argument("--stop", action="store_true", help="ask the running broker of --device to leave and exit")
    a = ap.parse_args(argv)
    if a.stats or a.stop:
        if a.stats:
            got = [(k, _bare_request(a.device, k, {"op": "stats"})) for k in range(shards())]
            if not any(r for _, r in got):
                print("no broker is listening on %s" % sock_path(a.device))
                return 1
            for k, r in got:
                if r:
                    print(json.dumps(dict(r, shard=k), indent=1, sort_keys=True))
        if a.stop:
            print("stopped" if shutdown(a.device) else "nothing to stop")
        return 0
    if a.lanes <= 0:
        a.lanes = max(1, -(-3 // shards()))
    os.environ["S2P_HIP_DEVICE"] = str(a.device)                # what _lib.default_device() answers in this process
    from s2p_amd import broker as canonical                     # (under `python -m` this file is __main__: the registry of remote
    canonical._serving[0] = True                                #  functions lives in the imported module, so serve from there)
    canonical.Server(a.device, a.lanes, a.max_batch, a.idle, a.max_wait_ms, shard=a.shard).serve()


if __name__ == "__main__":
    sys.exit(main())
