"""CPU: the C-ABI library loads and exports every symbol the header declares, the
product path refuses CPU tensors (no fallback), host logic (config, captions,
LR schedule, DP gather/reduce over gloo with world_size 2)."""

import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_header_symbols():
    import textreid_amd.lib as L

    assert os.path.exists(L.LIB_PATH), "build with `python -c 'import __graft_entry__ as g; g.build()'`"
    lib = L.load()
    assert lib.trid_version() >= 100 and lib.trid_arch() == b"gfx950"
    assert len(L.EXPORTS) >= 40
    for name in L.EXPORTS:
        assert hasattr(lib, name), name
    # the header is the single source of truth: nothing exported under trid_* is undeclared
    out = subprocess.run(["nm", "-D", "--defined-only", L.LIB_PATH], capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T trid_" in ln}
    assert exported == set(L.EXPORTS), exported ^ set(L.EXPORTS)


def test_argument_errors_are_reported_not_crashing():
    import textreid_amd.lib as L

    L.load()
    with pytest.raises(RuntimeError, match="trid_sum_f32"):
        L.call("trid_sum_f32", None, None, 0, 1.0, 0, None)
    assert "bad arguments" in L.last_error()


def test_product_path_refuses_cpu_tensors():
    from textreid_amd import losses
    from textreid_amd.backbones.m_resnet import ModifiedResNet

    m = ModifiedResNet([1, 1, 1, 1], 64, 4, 1, (96, 32), 16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(2, 3, 96, 32))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        losses.global_align_loss(torch.zeros(4, 8), torch.zeros(4, 8), torch.zeros(4, dtype=torch.long))


def test_product_does_not_import_oracle():
    code = "import sys; import textreid_amd.model, textreid_amd.evaluation, textreid_amd.solver, textreid_amd.engine.trainer; assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules), 'product imports oracle'"
    subprocess.check_call([sys.executable, "-c", code], cwd=ROOT)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "textreid_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_state_dict_names_match_reference_layout():
    """Parameter / buffer names and shapes equal the oracle's table, which
    make_golden.py asserts equal to the reference modules' state_dict."""
    import types

    import oracle.head as OH
    import oracle.visual as OV
    from textreid_amd.backbones.gru import GRU
    from textreid_amd.backbones.m_resnet import ModifiedResNet
    from textreid_amd.embeddings.moco_head.head import MoCoHead

    spec = OV.TINY
    vis = ModifiedResNet(list(spec.layers), spec.output_dim, spec.heads, spec.last_stride, (spec.height, spec.in_width), spec.width)
    txt = GRU(64, 64, 64, 1, 0.0, True, "clip_vit", "./", vocab_dict=torch.zeros(10, 64))
    ns = types.SimpleNamespace
    cfg = ns(MODEL=ns(EMBEDDING=ns(FEATURE_SIZE=32, EPSILON=0.1), MOCO=ns(K=32, M=0.9, FC=False), NUM_CLASSES=53))
    head = MoCoHead(cfg, vis, txt)
    got = {k: tuple(v.shape) for k, v in head.state_dict().items()}
    want = {k: tuple(s) for k, s in OH.state_shapes(spec, 32, 32, 53, 64, 64).items()}
    assert got == want
    assert head.t_queue.t().is_contiguous()  # queue storage is row-major [K,C]
    assert head.v_encoder_q.conv2.weight.is_contiguous(memory_format=torch.channels_last)


def test_config_merges_reference_style_yaml(tmp_path):
    from textreid_amd.config import get_cfg_defaults

    y = tmp_path / "c.yaml"
    y.write_text("MODEL:\n  VISUAL_MODEL: 'm_resnet50'\n  MOCO:\n    K: 2048\n    FC: False\nSOLVER:\n  STEPS: (40, 70)\n  BASE_LR: 0.0001\nINPUT:\n  HEIGHT: 384\n")
    cfg = get_cfg_defaults()
    cfg.merge_from_file(str(y))
    cfg.merge_from_list(["MODEL.MOCO.K", "8192", "ROOT", "/data"])
    cfg.freeze()
    assert cfg.MODEL.MOCO.K == 8192 and cfg.SOLVER.STEPS == (40, 70) and cfg.ROOT == "/data" and cfg.MODEL.MOCO.M == 0.999
    with pytest.raises(AttributeError):
        cfg.MODEL.MOCO.K = 1
    with pytest.raises(KeyError):
        get_cfg_defaults().merge_from_list(["MODEL.NOPE", "1"])


def test_caption_containers():
    from textreid_amd.caption import Caption, CaptionBatch

    caps = []
    for i, n in enumerate([3, 5, 2]):
        c = Caption([list(range(1, n + 1))], max_length=8)
        c.add_field("id", torch.tensor(10 + i))
        caps.append(c)
    assert caps[0].text.shape == (1, 8) and int(caps[1].length) == 5
    cb = CaptionBatch.from_list(caps)
    assert cb.tokens.shape == (3, 8) and cb.lengths.tolist() == [3, 5, 2] and cb.ids.tolist() == [10, 11, 12] and cb.max_len == 5
    assert cb.tokens[0].tolist() == [1, 2, 3, 0, 0, 0, 0, 0]
    assert CaptionBatch.from_list(cb) is cb


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


DP_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from textreid_amd.parallel import gather_embeddings, GradReducer, world_size, rank
import oracle.losses as OL
dist.init_process_group("gloo", init_method="env://")
W, r = world_size(), rank()
torch.manual_seed(0)
B, C, NC = 4, 8, 11
# identical "global" data on every rank; each rank owns rows [r*B, (r+1)*B)
Vg, Tg = torch.randn(W * B, C), torch.randn(W * B, C)
ids_g = torch.tensor([0, 0, 1, 2, 3, 3, 4, 5][: W * B])
proj = torch.randn(C, NC, requires_grad=True)
w = torch.randn(C, C, requires_grad=True)          # a pre-gather parameter (shared "encoder")
def local_embed(x):
    return x @ w
v_loc = local_embed(Vg[r * B:(r + 1) * B]); t_loc = local_embed(Tg[r * B:(r + 1) * B])
vk = torch.nn.functional.normalize(v_loc.detach(), dim=1); tk = torch.nn.functional.normalize(t_loc.detach(), dim=1)
v, t, vkg, tkg, ids = gather_embeddings(v_loc, t_loc, vk, tk, ids_g[r * B:(r + 1) * B])
assert torch.equal(ids, ids_g)
loss = OL.instance_loss(proj, v, t, ids, 0.1) + OL.global_align_loss(v, t, ids)
loss.backward()
red = GradReducer(bucket_mb=1)
red.reduce([w]); red.wait()                           # SUM over ranks for pre-gather params only
# single-process oracle on the global batch
w2 = w.detach().clone().requires_grad_(True); p2 = proj.detach().clone().requires_grad_(True)
l2 = OL.instance_loss(p2, Vg @ w2, Tg @ w2, ids_g, 0.1) + OL.global_align_loss(Vg @ w2, Tg @ w2, ids_g)
l2.backward()
assert torch.allclose(loss, l2, rtol=1e-5), (loss, l2)
assert torch.allclose(w.grad, w2.grad, rtol=1e-4, atol=1e-6), (w.grad - w2.grad).abs().max()
assert torch.allclose(proj.grad, p2.grad, rtol=1e-4, atol=1e-6)      # post-gather: identical, unreduced
print("rank", r, "ok")
dist.destroy_process_group()
"""


def test_dp_gather_and_reduce_gloo_world2(tmp_path):
    """W-rank DP == the single-process global-batch computation."""
    script = tmp_path / "dp_worker.py"
    script.write_text(DP_WORKER % ROOT)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert "rank %d ok" % r in o


STAGE_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from textreid_amd.parallel import gather_embeddings, GradReducer, world_size, rank
dist.init_process_group("gloo", init_method="env://")
W, r = world_size(), rank()
torch.manual_seed(100 + r)
# in-backward staging (ModifiedResNet.grad_sync): gradients arrive in channels_last (3x3 filters), contiguous and
# strided layouts; finish_stages() must hand back the SUM in each gradient's ORIGINAL layout
ps = [torch.nn.Parameter(torch.zeros(6, 4, 3, 3).contiguous(memory_format=torch.channels_last)), torch.nn.Parameter(torch.zeros(7)),
      torch.nn.Parameter(torch.zeros(5, 3))]
gs = [torch.randn(6, 4, 3, 3).contiguous(memory_format=torch.channels_last), torch.randn(7), torch.randn(3, 5).t()]
red = GradReducer(bucket_mb=1)
red.stage(ps[:2], gs[:2]); red.stage(ps[2:], gs[2:])
out = red.finish_stages()
ref = []
for g in gs:
    parts = [torch.empty_like(g.contiguous()) for _ in range(W)]
    dist.all_gather(parts, g.contiguous())
    ref.append(sum(parts))
for p, g, want in zip(ps, gs, ref):
    got = out[id(p)]
    assert got.shape == g.shape and torch.allclose(got, want, atol=1e-6), (got.shape, g.shape)
assert out[id(ps[0])].is_contiguous(memory_format=torch.channels_last)
# staged parameters are skipped by the post-backward reduce; the rest is reduced there
q = torch.nn.Parameter(torch.zeros(9)); q.grad = torch.full((9,), float(r + 1))
for p, g in zip(ps, gs): p.grad = g
red.reduce(ps + [q]); red.wait()
assert torch.allclose(q.grad, torch.full((9,), float(sum(range(1, W + 1)))))
assert torch.equal(ps[1].grad, gs[1])                                   # untouched: it was reduced in the stage
st = red.stats()
assert st["allreduce_bytes_per_step"] == 4 * (6 * 4 * 9 + 7 + 15 + 9) and 0.9 < st["staged_fraction"] < 1.0
# ids beyond 2^24 survive the packed gather bit for bit (they travel as two 32-bit lanes, not as fp32 values)
B, C = 3, 4
ids = torch.tensor([2 ** 40 + 7 * r + 1, -5, 16777217 + r])
e = torch.randn(B, C)
v, t, vk, tk, gid = gather_embeddings(e, e + 1, e + 2, e + 3, ids)
want = torch.cat([torch.tensor([2 ** 40 + 7 * k + 1, -5, 16777217 + k]) for k in range(W)])
assert torch.equal(gid, want), (gid, want)
print("rank", r, "ok")
dist.destroy_process_group()
"""


def test_dp_stage_finish_layouts_and_id_lanes_gloo_world2(tmp_path):
    """GradReducer's in-backward staging path with channels-last / strided gradients, the post-backward bucket path
    skipping staged parameters, the per-step accounting bench.py prints, and exact 64-bit ids through the packed
    embedding all-gather - two gloo ranks on CPU."""
    script = tmp_path / "stage_worker.py"
    script.write_text(STAGE_WORKER % ROOT)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert "rank %d ok" % r in o


PRED_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from textreid_amd.engine.inference import _gather_predictions
dist.init_process_group("gloo", init_method="env://")
W, r = dist.get_world_size(), dist.get_rank()
C = 6
# uneven shards (rank 1 holds more samples), indices beyond 2^24 (an fp32 VALUE could not carry them)
mine = {0: [3, 16777217, 40], 1: [1, 2 ** 33 + 5, 7, 9]}[r]
g = lambda k, s: torch.arange(C, dtype=torch.float32) * (s + 1) + float(k %% 1000)
pred = {k: [g(k, 0), g(k, 1)] for k in mine}
out = _gather_predictions(pred)
if r == 0:
    want = [3, 16777217, 40, 1, 2 ** 33 + 5, 7, 9]
    assert sorted(out) == sorted(want), sorted(out)
    for k in want:
        assert torch.equal(out[k][0], g(k, 0)) and torch.equal(out[k][1], g(k, 1)), k
else:
    assert out is None
# a rank with no samples at all still takes part
out = _gather_predictions(pred if r == 0 else {}, device="cpu")  # (the empty rank allocates on ITS device, not on rank 0's)
if r == 0:
    assert sorted(out) == sorted(mine)
    assert all(v.device.type == "cpu" and t.device.type == "cpu" for v, t in out.values())
print("rank", r, "ok")
dist.destroy_process_group()
"""


def test_inference_prediction_gather_gloo_world2(tmp_path):
    """engine.inference._gather_predictions (the cross-rank accumulation of `lib/engine/inference.py:28-45`): uneven
    shards, 64-bit dataset indices, an empty rank - two gloo ranks on CPU."""
    script = tmp_path / "pred_worker.py"
    script.write_text(PRED_WORKER % ROOT)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert "rank %d ok" % r in o


REPLICA_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from textreid_amd.parallel import broadcast_module_state, check_replicas, replica_digest
dist.init_process_group("gloo", init_method="env://")
W, r = dist.get_world_size(), dist.get_rank()
torch.manual_seed(100 + r)  # DIFFERENT seeds: the ranks draw different weights, statistics and queues
class M(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = torch.nn.Conv2d(3, 8, 3)
        self.bn = torch.nn.BatchNorm2d(8)
        self.register_buffer("v_queue", torch.nn.functional.normalize(torch.rand(16, 64), dim=0))
        self.register_buffer("id_queue", torch.randint(0, 1000, (1, 64)))
        self.register_buffer("queue_ptr", torch.tensor([8 * r], dtype=torch.long))
m = M()
m.bn.running_mean.add_(float(r))
before = [t.detach().clone() for t in list(m.parameters()) + list(m.buffers())]
watch = [("queue_ptr", m.queue_ptr), ("id_queue", m.id_queue), ("parameter conv.weight", m.conv.weight)]
try:
    check_replicas(watch, "before the broadcast")
    raise SystemExit("ranks drawn from different seeds passed the replica check")
except RuntimeError as e:
    assert "diverged" in str(e) and "id_queue" in str(e), str(e)
nbytes = broadcast_module_state(m)
assert nbytes == sum(t.numel() * t.element_size() for t in before)
state = [t.detach().clone() for t in list(m.parameters()) + list(m.buffers())]
ref = [None] * W
dist.all_gather_object(ref, [t.numpy().tobytes() for t in state])
assert ref[0] == ref[1]                      # bit-identical replicas ...
if r == 0:
    assert all(torch.equal(a, b) for a, b in zip(before, state))   # ... equal to what rank 0 had
check_replicas(watch, "after the broadcast")
# a replica that drifts in ONE element of the id queue is caught on every rank
if r == 1:
    m.id_queue[0, 5] += 1
try:
    check_replicas(watch, "after a drift")
    raise SystemExit("a drifted replica passed the check")
except RuntimeError as e:
    assert "id_queue" in str(e) and "conv.weight" not in str(e), str(e)
# digests: position-weighted wrap-around sums of the words (weight of word i: i * c + 1 with c odd), exact for int64 and float tensors
d = replica_digest([torch.tensor([2 ** 62, 2 ** 62, 2 ** 62, 2 ** 62]), torch.tensor([1.0, -1.0]), torch.tensor([-1.0, 1.0]), torch.tensor([3, 5, 7]), torch.tensor([5, 3, 7])])
c = -7046029254386353131
def signed(v):
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v
assert int(d[0]) == signed(sum((2 ** 62) * (i * c + 1) for i in range(4)))
assert int(d[1]) == signed(0x3F800000 + (0xBF800000 - (1 << 32)) * (c + 1))
assert int(d[1]) != int(d[2]) and int(d[3]) != int(d[4])   # a swap of two unequal entries is seen (a plain sum would not see it)
print("rank", r, "ok")
dist.destroy_process_group()
"""


def test_replicas_broadcast_and_digest_gloo_world2(tmp_path):
    """parallel.broadcast_module_state / check_replicas (engine.trainer.do_train under data parallelism; DDP's broadcast at wrap,
    train_net.py:50-56): two gloo ranks that start from DIFFERENT seeds end bit-identical to rank 0, the digest check passes
    then, fails before, and fails - naming the tensor - after one rank drifts in one element."""
    script = tmp_path / "replica_worker.py"
    script.write_text(REPLICA_WORKER % ROOT)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert "rank %d ok" % r in o


def test_p16_flow_size_guard():
    """The P16 kernels address operands with 31-bit offsets: per-GPU batches whose largest activation reaches 2 GB must not take
    the P16 flow (about 640 images of 384 x 128) - a pure shape rule, checked on meta tensors."""
    from textreid_amd.backbones.m_resnet import p16_fits

    assert p16_fits(torch.empty(128, 3, 384, 128, device="meta")) and p16_fits(torch.empty(640, 3, 384, 128, device="meta"))
    assert not p16_fits(torch.empty(683, 3, 384, 128, device="meta"))  # 683 x 192 x 64 x 64 x 4 B = 2 GB + 1 MB
    assert p16_fits(torch.empty(2000, 3, 96, 32, device="meta"))


def test_lr_schedule_values():
    from textreid_amd.solver import LRSchedulerWithWarmup

    p = [torch.nn.Parameter(torch.zeros(1))]
    opt = torch.optim.SGD(p, lr=1.0)
    sch = LRSchedulerWithWarmup(opt, milestones=(4, 7), gamma=0.1, mode="step", warmup_factor=0.1, warmup_epochs=2, total_epochs=10)
    lrs = []
    for _ in range(9):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sch.step()
    assert np.allclose(lrs, [0.1, 0.55, 1.0, 1.0, 0.1, 0.1, 0.1, 0.01, 0.01])


def test_clip_checkpoint_ingestion_matches_reference(golden_dir):
    """state_filter / resize_pos_embed (host-side checkpoint ingestion, m_resnet.py:220-243)
    against vectors captured from the reference functions."""
    import oracle.fill as OF
    from textreid_amd.backbones.m_resnet import resize_pos_embed, state_filter

    g = np.load(os.path.join(golden_dir, "ingest.npz"))
    pe = torch.from_numpy(g["pos_in"])
    assert np.allclose(resize_pos_embed(pe, (24, 8)).numpy(), g["pos_a"], atol=1e-6)
    assert np.allclose(resize_pos_embed(pe, (6, 2)).numpy(), g["pos_b"], atol=1e-6)
    sd = {"visual.conv1.weight": OF.randn("ingest:w", (4, 3, 3, 3), 13), "visual.attnpool.positional_embedding": pe,
          "token_embedding.weight": OF.randn("ingest:t", (5, 4), 13)}
    flt = state_filter(sd, (6, 2))
    assert sorted(flt.keys()) == list(g["filter_keys"])
    assert np.allclose(flt["attnpool.positional_embedding"].numpy(), g["filter_pos"], atol=1e-6)


def test_suffix_aligner_matches_reference_mapping(golden_dir):
    """Checkpointer's loading rule (lib/utils/checkpoint.py:90-148): `module.` stripped, every model key takes the
    loaded key that is its longest suffix, unmatched keys keep their value - against the mapping the reference's
    own align_and_update_state_dicts produced for the same key lists."""
    from textreid_amd.checkpoint import match_by_longest_suffix, strip_prefix_if_present

    g = np.load(os.path.join(golden_dir, "ingest.npz"))
    model_keys, loaded_keys = list(g["align_model_keys"]), list(g["align_loaded_keys"])
    loaded = strip_prefix_if_present({k: 100.0 + i for i, k in enumerate(loaded_keys)})
    m = match_by_longest_suffix(sorted(model_keys), sorted(loaded))
    got = [loaded[m[k]] if k in m else float(i) for i, k in enumerate(model_keys)]
    assert got == list(g["align_values"])


def test_reference_best_pth_and_torchscript_clip_load(tmp_path):
    """SURVEY 8 f5 end to end, on stand-ins built here (no CLIP weights / dataset in the image):
    (a) a `best.pth`-shaped file - {"model": state saved from a DistributedDataParallel-wrapped model (`module.`
        prefix), "epoch": ...} - loads through the suffix aligner and reproduces every tensor;
    (b) a TorchScript archive exposing `visual.*` tensors with a 7x7 positional grid (what `RN50.pt` is) loads through
        torch.jit.load + state_filter into an encoder with a 6x2 final grid: positional embedding resized, the rest
        copied, text-tower tensors ignored."""
    from textreid_amd.backbones.m_resnet import ModifiedResNet, resize_pos_embed
    from textreid_amd.checkpoint import load_checkpoint_file, load_clip_visual

    torch.manual_seed(3)
    mk = lambda: ModifiedResNet([1, 1, 1, 1], 64, 4, 1, (96, 32), 16)
    src, dst = mk(), mk()
    for p in src.parameters():
        p.data.normal_()
    ck = {"model": {"module." + k: v.clone() for k, v in src.state_dict().items()}, "epoch": 7, "iteration": 123}
    torch.save(ck, tmp_path / "best.pth")
    rest = load_checkpoint_file(dst, str(tmp_path / "best.pth"))
    assert rest == {"epoch": 7, "iteration": 123}
    for (k, a), (_, b) in zip(src.state_dict().items(), dst.state_dict().items()):
        assert torch.equal(a, b), k

    class Tower(torch.nn.Module):  # stand-in for CLIP's scripted model: visual.* parameters + unrelated text tensors
        def __init__(self, sd):
            super().__init__()
            self.visual = torch.nn.Module()
            for k, v in sd.items():
                mod = self.visual
                parts = k.split(".")
                for q in parts[:-1]:
                    if not hasattr(mod, q):
                        setattr(mod, q, torch.nn.Module())
                    mod = getattr(mod, q)
                mod.register_buffer(parts[-1], v.clone())
            self.token_embedding = torch.nn.Embedding(5, 4)

        def forward(self, x):
            return x

    sd = {k: v for k, v in src.state_dict().items() if "num_batches_tracked" not in k}
    sd["attnpool.positional_embedding"] = torch.randn(50, 512)  # CLIP's 7x7 grid + class token
    torch.jit.script(Tower(sd)).save(str(tmp_path / "RN_stub.pt"))
    dst2 = mk()
    missing, unexpected = load_clip_visual(dst2, str(tmp_path / "RN_stub.pt"))
    assert not [k for k in missing if "num_batches_tracked" not in k]
    assert all(k.startswith("token_embedding") for k in unexpected)
    got = dst2.state_dict()
    assert torch.allclose(got["attnpool.positional_embedding"], resize_pos_embed(sd["attnpool.positional_embedding"], (6, 2)))
    for k, v in sd.items():
        if k != "attnpool.positional_embedding":
            assert torch.equal(got[k], v), k


def test_image_resize_restatement_is_pillow_bit_exact():
    """SURVEY 8 f4 oracle pin: Resize on a PIL image is Pillow's antialiased two-pass fixed-point BILINEAR.  The
    oracle's restatement of Resample.c and the product's vectorised weight tables reproduce Pillow (the library the
    reference calls through torchvision) bit for bit on up-scaling, down-scaling and mixed cases."""
    from PIL import Image

    import oracle.transforms as OT
    from textreid_amd.transforms import resample_tables

    rs = np.random.RandomState(0)
    for (h, w) in [(97, 53), (384, 128), (500, 200), (60, 30), (777, 333), (384, 53), (201, 128)]:
        a = (rs.rand(h, w, 3) * 255).astype(np.uint8)
        ref = np.asarray(Image.fromarray(a).resize((128, 384), Image.BILINEAR))
        assert np.array_equal(OT.resize_bilinear_u8(a, 384, 128), ref), (h, w)
    for n_in, n_out in [(53, 128), (500, 384), (333, 128), (128, 128), (3, 128), (1000, 384)]:
        b0, k0 = OT.resample_coeffs(n_in, n_out)
        b1, k1 = resample_tables(n_in, n_out)
        assert np.array_equal(b0, b1) and np.array_equal(k0, k1), (n_in, n_out)


def test_wgrad_split_rule():
    """ops._wgrad_splits: one >= 90 %-full round of resident workgroups when the tile count allows it, else two; counts
    >= 8 are multiples of 8 (every XCD owns whole splits); at least 512 reduction rows per split."""
    from textreid_amd.ops import _wgrad_splits

    B = 128
    cases = {  # (tiles, pixels, slots) -> splits, the RN50 layer shapes at B=128 (profiles/r03i_wgrad_split_sweep.txt)
        (9, B * 48 * 16, 512): 56, (9, B * 96 * 32, 512): 56, (36, B * 24 * 8, 512): 24, (144, B * 12 * 4, 512): 7,
        (5, B * 96 * 32, 768): 152, (16, B * 24 * 8, 512): 32, (64, B * 12 * 4, 512): 8, (2, B * 96 * 32, 512): 256,
    }
    for (tiles, K, slots), want in cases.items():
        assert _wgrad_splits(tiles, K, slots=slots) == want, (tiles, K, slots)
    for tiles in (1, 3, 7, 20, 100, 500):
        for K in (300, 4096, 100000, 2000000):
            s = _wgrad_splits(tiles, K)
            assert s >= 1 and (s < 8 or s % 8 == 0) and s <= max(1, K // 512) and s * tiles <= 2 * 512 + tiles


def test_oracle_bf16_mode_rounding_points():
    """oracle.visual.bf16_conv (the definition of configs[3]'s bf16 mode): conv outputs, block outputs and the gradients
    w.r.t. them are bf16-representable; filters' gradients are NOT rounded; outside the context nothing is rounded."""
    import oracle.visual as OV

    torch.manual_seed(0)
    x = torch.randn(2, 8, 6, 4, requires_grad=True)
    w = torch.randn(8, 8, 3, 3, requires_grad=True)
    rep = lambda t: torch.equal(t.to(torch.bfloat16).to(t.dtype), t)
    with OV.bf16_conv():
        y = OV._conv(x, w, padding=1)
        assert rep(y.detach())
        (y * torch.randn_like(y)).sum().backward()
    assert rep(x.grad) and not rep(w.grad)
    x.grad = None
    y = OV._conv(x, w, padding=1)
    (y * torch.randn_like(y)).sum().backward()
    assert not rep(y.detach()) and not rep(x.grad)


def test_bench_gpus_n_never_prints_a_smaller_jobs_line():
    """`bench.py --gpus N` with fewer than N visible GPUs (here: none) refuses with a non-zero exit code and no JSON
    line - never a silent N=1 measurement; likewise a launcher whose WORLD_SIZE contradicts --gpus."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "--gpus 2 requested" in r.stderr and "{" not in r.stdout
    env.update(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "{" not in r.stdout


def test_caption_bucket_selection(monkeypatch):
    """engine.graph.BucketedTrainStep.bucket_of: the smallest recurrence bound that holds the batch's longest caption, never wider
    than the token tensor (lib/data/build.py:26 pads to 105; lib/models/backbones/gru.py:66-82 packs to the batch maximum)."""
    from textreid_amd.caption import CaptionBatch
    from textreid_amd.engine.graph import BucketedTrainStep

    monkeypatch.delenv("TRID_CAPTION_BUCKETS", raising=False)
    bs = BucketedTrainStep(None, None)
    assert bs.buckets == [32, 48, 64, 105]

    def cb(width, longest):
        lengths = torch.full((4,), longest, dtype=torch.long)
        return CaptionBatch(torch.zeros(4, width, dtype=torch.long), lengths, None, max_len=longest)

    assert [bs.bucket_of(cb(105, n)) for n in (1, 32, 33, 48, 49, 64, 65, 105)] == [32, 32, 48, 48, 64, 64, 105, 105]
    assert bs.bucket_of(cb(50, 49)) == 50  # the bucket is clipped to the tensor's width
    assert bs.bucket_of(cb(120, 110)) == 120  # longer than every bucket: the width itself
    monkeypatch.setenv("TRID_CAPTION_BUCKETS", "16, 80")
    assert BucketedTrainStep(None, None).buckets == [16, 80]
    monkeypatch.delenv("TRID_CAPTION_BUCKETS")
    with pytest.raises(ValueError):
        BucketedTrainStep(None, None, buckets=())
    with pytest.raises(ValueError):
        BucketedTrainStep(None, None, buckets=(0, 32))
    assert BucketedTrainStep(None, None).recorded == {}
