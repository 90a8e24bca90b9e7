"""GPU: the data-parallel step (packed embedding all-gather, global losses, SUM gradient
reduction) on 2 ranks equals the oracle's single-process global-batch computation with
per-shard BatchNorm.  The ranks share the one visible GPU and talk over gloo; the RCCL path
differs only in the transport of the same two collectives (textreid_amd/parallel.py)."""

import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_dp_two_ranks_match_global_batch_oracle():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   TRID_DIST_BACKEND="gloo", OMP_NUM_THREADS="8")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_gpu_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    assert "DP_OK" in outs[0] and "DP_REPLICAS_IDENTICAL" in outs[0] and "DP_RETRIEVAL_OK" in outs[0], outs[0][-3000:]
    print([ln for ln in outs[0].splitlines() if ln.startswith("DP_ERRS")])
