"""Data-parallel invariance on the device: two ranks (sharing the visible GPU over gloo when there is only one),
averaged flat-bucket gradients == single-process two-chunk emulation with per-chunk BatchNorm statistics."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_two_rank_averaged_gradients_match_two_chunk_emulation():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(29700 + os.getpid() % 200), os.path.join(HERE, "ddp_worker.py")]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=110)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["ok"] and out["world"] == 2 and out["rank0_bn_buffers_equal_single_process"], out


def test_flat_bucket_through_rccl_on_one_rank():
    """The RCCL backend itself (not gloo): initialise, all-reduce the flat gradient bucket in place, device-pinned
    barrier, clean shutdown - in a process of its own so that the group does not outlive the test."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    res = subprocess.run([sys.executable, os.path.join(HERE, "rccl_worker.py")], env=env, capture_output=True, text=True,
                         timeout=110)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["ok"] and out["backend"] == "nccl", out
