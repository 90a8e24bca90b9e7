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


def _run_ranks(world, worker="dp_gpu_worker.py", need=("DP_OK", "DP_REPLICAS_IDENTICAL", "DP_RETRIEVAL_OK"), timeout=600, **extra_env):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="8", HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", worker)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    try:
        outs = [p.communicate(timeout=timeout)[0] for p in procs]
    finally:
        for p in procs:  # (exactly the processes started here)
            if p.poll() is None:
                p.kill()
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    assert all(tag in outs[0] for tag in need), outs[0][-3000:]
    print([ln for ln in outs[0].splitlines() if ln.startswith(("DP_ERRS", "DP_TRANSPORT", "DPFULL_"))])
    return outs[0]


@pytest.mark.parametrize("fc", ["0", "1"], ids=["FC=False", "FC=True"])
def test_dp_two_ranks_match_global_batch_oracle(fc):
    """(FC=True: MODEL.MOCO.FC projection heads - the packed gather then carries six embedding blocks.)"""
    _run_ranks(2, TRID_DIST_BACKEND="gloo", TRID_TEST_FC=fc)


def test_dp_do_train_broadcasts_and_keeps_replicas():
    """engine.trainer.do_train with two ranks whose models start from DIFFERENT seeds: the initial broadcast of parameters and
    buffers (train_net.py:50-56) makes them replicas, three optimizer steps on sharded batches keep them bit-identical, the
    replica digest check runs on every step."""
    out = _run_ranks(2, TRID_DIST_BACKEND="gloo", TRID_DP_TRAINER="1")
    assert "DP_TRAINER_REPLICAS_IDENTICAL" in out


def test_dp_collectives_execute_on_rccl():
    """The SAME step with backend `nccl` (= RCCL on ROCm) in a ONE-rank group with TRID_DP_FORCE=1: the packed
    `all_gather_into_tensor` inside the forward, the asynchronous SUM all-reduce of staged flats issued from inside
    the image encoder's backward (`GradReducer.stage`, joined with `work.wait()` there), the bucketed post-backward
    all-reduce and the sharded-retrieval merge all execute on RCCL's streams on this box's one GPU; results are
    checked against the oracle exactly as in the two-rank gloo run (a one-rank SUM is the identity, so any stream-order
    or staged-buffer-lifetime bug shows up as a wrong gradient)."""
    out = _run_ranks(1, TRID_DIST_BACKEND="nccl", TRID_DP_FORCE="1")
    line = [ln for ln in out.splitlines() if ln.startswith("DP_TRANSPORT")][0]
    assert "backend=nccl" in line
    staged = int(line.split("staged_bytes=")[1].split()[0])
    post = int(line.split("post_bytes=")[1].split()[0])
    assert staged > 0 and post > 0, line  # both reducer entry points really ran their collectives


def test_dp_step_recorded_with_its_rccl_collectives():
    """The data-parallel step on the fast host path: recorded once, replayed in SEGMENTS around its collectives (the packed
    all-gather, the in-backward staged all-reduces, the bucketed ones are cut points of the recording; RCCL runs each through its
    own launch path on the stream the marker was recorded on).  The `nccl` one-rank group with TRID_DP_FORCE=1 drives every
    collective; five steps (two eager warm-ups, the recording, replays) equal the eager data-parallel steps bit for bit: losses,
    every parameter, queue, BatchNorm buffer."""
    out = _run_ranks(1, need=("DP_OK", "DP_REPLICAS_IDENTICAL", "DP_RETRIEVAL_OK", "DP_CAPTURED_OK"), TRID_DIST_BACKEND="nccl",
                     TRID_DP_FORCE="1", TRID_DP_CAPTURED="1")
    assert "DP_CAPTURED_OK backend=nccl" in out


def test_dp_step_segmented_replay_two_ranks():
    """... and with TWO ranks (sharing the one GPU, gloo transport): the same recording - a host-staged transport is replayable
    because the collectives are cut points, not nodes - bit-identical to the eager two-rank steps on every rank."""
    out = _run_ranks(2, need=("DP_OK", "DP_REPLICAS_IDENTICAL", "DP_RETRIEVAL_OK", "DP_CAPTURED_OK"), TRID_DIST_BACKEND="gloo", TRID_DP_CAPTURED="1")
    assert "DP_CAPTURED_OK backend=gloo world=2" in out


def test_dp_do_train_replays_in_segments_and_keeps_replicas():
    """engine.trainer.do_train's DEFAULT under data parallelism: two eager steps, then the recorded step replayed in segments -
    two ranks from different seeds end as bit-identical replicas, the digest check passes on every step."""
    out = _run_ranks(2, TRID_DIST_BACKEND="gloo", TRID_DP_TRAINER="1", TRID_DP_TRAINER_CAPTURE="1")
    assert "DP_TRAINER_REPLICAS_IDENTICAL" in out
    assert "DP_TRAINER_LOG train step captured" in out and "segments of stream launches around" in out, out[-2000:]


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL with more than one rank needs two GPUs (none on the one-GPU build pool)")
def test_dp_step_recorded_two_ranks_on_rccl():
    """... and with one rank per GPU when the box has two."""
    out = _run_ranks(2, need=("DP_OK", "DP_REPLICAS_IDENTICAL", "DP_RETRIEVAL_OK", "DP_CAPTURED_OK"), TRID_DIST_BACKEND="nccl", TRID_DP_CAPTURED="1")
    assert "DP_CAPTURED_OK backend=nccl world=2" in out


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL with more than one rank needs two GPUs (none on the one-GPU build pool)")
@pytest.mark.parametrize("fc", ["0", "1"], ids=["FC=False", "FC=True"])
def test_dp_two_ranks_on_rccl(fc):
    """The two-rank case above with ONE RANK PER GPU over `nccl` (= RCCL over xGMI; reference `train_net.py:148-154`
    derives the world from WORLD_SIZE and calls init_process_group("nccl")): armed for the first box that shows two
    devices - the packed all-gather, the in-backward staged all-reduce and the bucketed one then run between GPUs."""
    out = _run_ranks(2, TRID_DIST_BACKEND="nccl", TRID_TEST_FC=fc)
    assert "backend=nccl world=2" in out


def _full(world, arch, K, **env):
    # one rank per GPU over RCCL when the box has them; else the ranks share the GPU(s) and talk over gloo
    backend = "nccl" if torch.cuda.device_count() >= world else "gloo"
    out = _run_ranks(world, worker="dp_full_worker.py", need=("DPFULL_REPLICAS_IDENTICAL", "DPFULL_OK"), timeout=1500,
                     TRID_DIST_BACKEND=backend, TRID_DP_ARCH=arch, TRID_DP_K=str(K), TRID_DP_BLOCAL="128", **env)
    assert "world=%d backend=%s" % (world, backend) in out and "B_global=%d K=%d" % (128 * world, K) in out
    return out


def test_config2_bs512_four_ranks_full_size():
    """BASELINE configs[2] in its N-rank form: CLIP-RN50 + BiGRU, 4 ranks x 128 = global batch 512 through the packed
    embedding all-gather, MoCo queue 8192, fp32-class.  See tests/dp_full_worker.py for what is compared (replicas
    bit-identical; global losses vs the CPU oracle on the gathered [512,256] blocks; reduced gradients vs a
    single-process evaluation of the same global batch with per-shard BatchNorm)."""
    _full(4, "m_resnet50", 8192)


def test_config3_bs1024_eight_ranks_bf16_full_size():
    """BASELINE configs[3] in its N-rank form: CLIP-RN101 + BiGRU, 8 ranks x 128 = global batch 1024, MoCo queue
    65536 (the 1024-row batch takes the queue kernel's flag pre-pass against the 65536 slots), bf16 mode
    (TRID_CONV_PRECISION=1)."""
    out = _full(8, "m_resnet101", 65536, TRID_CONV_PRECISION="1", TRID_DP_SEED="43")
    assert "conv_precision=1" in out
