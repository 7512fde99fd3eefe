"""RCCL on the box ("RCCL has never been initialised").  The 8-GPU run is the driver's; what a 1-GPU box can show is that
every torch.distributed call bench.py and s2p_amd.tiles make at N > 1 -- init_process_group("nccl", device_id=...), barrier, the
MAX all-reduce of the timing, the gather of a resident tile, gather_mosaic's owner count and padded gather of device buffers -- runs
through RCCL on this torch / ROCm build, in a group of one rank.  (The same calls with 2, 3 and 8 ranks run on gloo in
tests/test_tiles_dist.py.)"""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %r)
assert torch.cuda.is_available()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)          # bench.py: main()
assert dist.get_backend() == "nccl"
dist.barrier()
t = torch.tensor([3.25], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)                                       # the max-over-ranks clock
assert float(t.item()) == 3.25
payload = torch.arange(1024 * 1024, dtype=torch.float32, device=dev).reshape(1024, 1024)
out = [torch.empty_like(payload)]
dist.gather(payload, out, dst=0)                                               # bench.py: the resident-tile gather
torch.cuda.synchronize()
assert torch.equal(out[0], payload)
from s2p_amd import tiles as T
rng = np.random.default_rng(5)
layout = [(y, x, 100, 120) for y in range(0, 400, 100) for x in range(0, 480, 120)]
mine = {i: rng.random((100, 120), dtype=np.float32) for i in range(len(layout))}
m = T.gather_mosaic(mine, layout, (400, 480), dst=0, device=dev, dynamic=True, collectives=True)   # owner count (all-reduce) + padded gather, device buffers
for i, (y, x, h, w) in enumerate(layout):
    assert np.array_equal(m[y:y + h, x:x + w], mine[i])
dist.barrier()
dist.destroy_process_group()
print("rccl ok", torch.cuda.nccl.version() if hasattr(torch.cuda, "nccl") else "")
"""


def test_the_collectives_of_the_multi_gpu_path_run_on_rccl():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", SCRIPT % ROOT], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
