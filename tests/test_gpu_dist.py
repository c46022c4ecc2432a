"""The multi-GPU path on a one-GPU box: the RCCL collectives in a process group of one rank, and
the real driver under two ranks that share the GPU (gloo through host memory)."""
import os
import socket
import subprocess
import sys

import pytest

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run(cmd, timeout=240):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stdout[-3000:] + "\n" + p.stderr[-3000:]
    return p.stdout


def test_rccl_collectives_in_a_one_rank_group(gpu):
    """all_to_all_single / all_reduce / all_gather_into_tensor on device tensors through RCCL: a sum
    over one rank must give the Hessian back bit for bit (pack -> exchange -> unpack -> mirror)."""
    out = _run([sys.executable, os.path.join(ROOT, "tests", "_rccl_one_rank.py"), str(_free_port())])
    assert "rccl one-rank ok" in out


def test_driver_under_two_ranks(gpu):
    """quantize_llama on a tiny random-init Llama under torch.distributed with two ranks (sample
    shards, summed Hessians, row-sharded column loop, gathered results): both ranks end with the same
    quantised model, and it is the single-process result up to the summation order of H."""
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_driver_two_ranks.py")])
    assert out.count("driver rank ok") == 2
