#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the IMPORTED reference.

Runs only in the build container (needs /root/reference, read-only).  The
reference is imported with three in-process shims, no reference file modified
(SURVEY.md section 8 c1):
  (i)   torch.Tensor.cuda -> identity   (gru.py:34, head.py:154, losses.py:36,215)
  (ii)  lib.models.backbones.gru.load_vocab_dict -> synthetic table from fill()
  (iii) visual model built with modified_resnet50/101(..., pretrained_path=None)
Weights are NOT stored: both the reference modules here and the HIP path on the
GPU box fill every tensor from oracle.fill(name, shape, seed).  Fixtures hold
inputs that cannot be regenerated cheaply plus expected outputs (data only).

Every fixture is also checked here against the oracle restatement
(oracle/*.py); the script fails if they disagree beyond 2e-5 relative.

Usage:  python tests/golden/make_golden.py
"""

import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

torch.Tensor.cuda = lambda self, *a, **k: self  # shim (i)
torch.set_num_threads(8)

import oracle.fill as OF  # noqa: E402
from oracle.fill import digest, digest_err, grad_floor, strided_sample  # noqa: E402
import oracle.head as OH  # noqa: E402
import oracle.text as OT  # noqa: E402
import oracle.visual as OV  # noqa: E402
import oracle.evaluation as OE  # noqa: E402
import oracle.losses as OL  # noqa: E402

import lib.models.backbones.gru as ref_gru  # noqa: E402
import lib.models.backbones.m_resnet as ref_mr  # noqa: E402
import lib.models.losses as ref_losses  # noqa: E402
from lib.models.embeddings.moco_head.head import MoCoHead  # noqa: E402
from lib.utils.caption import Caption  # noqa: E402

spec_eval = importlib.util.spec_from_file_location(
    "ref_evaluation", os.path.join(REF, "lib/data/metrics/evaluation.py")
)
sys.modules.setdefault("lib.utils.logger", types.SimpleNamespace(table_log=lambda *a, **k: ""))
ref_eval = importlib.util.module_from_spec(spec_eval)
spec_eval.loader.exec_module(ref_eval)


WARM = 25


def f32(d):
    return {k: (np.asarray(v).astype(np.float32) if np.asarray(v).dtype == np.float64 and k.startswith('truth:') else v) for k, v in d.items()}


def rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def check(name, a, b, tol=2e-5):
    r = rel(a, b)
    print("  %-40s rel %.2e" % (name, r))
    assert r < tol, (name, r)


def ref_visual(spec):
    return ref_mr.ModifiedResNet(
        layers=list(spec.layers),
        output_dim=spec.output_dim,
        heads=spec.heads,
        last_stride=spec.last_stride,
        input_resolution=(spec.height, spec.in_width),
        width=spec.width,
    )


def load_filled(module, seed, prefix=""):
    sd = module.state_dict()
    module.load_state_dict(OF.fill_state(sd, seed, prefix))
    return module


def oracle_state(shapes, seed, prefix="", grad_names=None):
    st = {}
    for k, s in shapes.items():
        if k.endswith("num_batches_tracked"):
            st[k] = torch.zeros((), dtype=torch.int64)
        else:
            st[k] = OF.fill(prefix + k, s, seed)
    return st


def make_captions(tokens, lengths, ids=None):
    caps = []
    for i in range(tokens.shape[0]):
        n = int(lengths[i])
        c = Caption([tokens[i, :n].tolist()], max_length=tokens.shape[1])
        if ids is not None:
            c.add_field("id", ids[i].clone())
        caps.append(c)
    return caps


def synth_tokens(name, B, lens, vocab, Lpad, seed):
    tok = OF.randint(name, 1, vocab, (B, Lpad), seed)
    for i, n in enumerate(lens):
        tok[i, n:] = 0
    return tok, torch.tensor(lens, dtype=torch.int64)


# --------------------------------------------------------------------------
def gen_visual(tag, spec, B, seed, grads=()):
    print("[visual %s]" % tag)
    m = load_filled(ref_visual(spec), seed)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {
        k: tuple(s) for k, s in OV.state_shapes(spec).items()
    }, "oracle state_shapes != reference state_dict"
    x = OF.randn("img:" + tag, (B, 3, spec.height, spec.in_width), seed)
    m.train()
    y = m(x)
    w_out = OF.randn("gout:" + tag, tuple(y.shape), seed)
    (y * w_out).sum().backward()
    out = {"out_train": y.detach().numpy(), "spec": np.array(list(spec.layers) + [spec.width, spec.heads, spec.output_dim, spec.height, spec.in_width, B, seed])}
    sd_after = m.state_dict()
    out["bn1_running_mean"] = sd_after["bn1.running_mean"].numpy().copy()
    out["bn1_running_var"] = sd_after["bn1.running_var"].numpy().copy()
    last = [k for k in sd_after if k.endswith("bn3.running_var")][-1]
    out["last_running_var"] = sd_after[last].numpy().copy()
    named = dict(m.named_parameters())
    for g in grads:
        out["grad:" + g] = named[g].grad.numpy().copy()
    m.eval()
    with torch.no_grad():
        out["out_eval"] = m(x).numpy()

    # warm running statistics: WARM more train-mode forwards, then eval (well-conditioned eval check)
    m.train()
    with torch.no_grad():
        for _ in range(WARM):
            m(x)
        m.eval()
        out["out_eval_warm"] = m(x).numpy()

    # oracle check (fp32) + fp64 oracle = "truth" used by the noise-aware GPU criterion
    for dt in (torch.float32, torch.float64):
        st = oracle_state(OV.state_shapes(spec), seed)
        for k in st:
            if st[k].dtype.is_floating_point:
                st[k] = st[k].to(dt)
            if OV.is_param(k):
                st[k].requires_grad_(True)
        xx = x.to(dt)
        yo = OV.visual_forward(st, xx, spec, True)
        (yo * w_out.to(dt)).sum().backward()
        with torch.no_grad():
            ye = OV.visual_forward(st, xx, spec, False)
            for _ in range(WARM):
                OV.visual_forward(st, xx, spec, True)
            yw = OV.visual_forward(st, xx, spec, False)
        if dt == torch.float32:
            check("out_train", yo, y)
            for g in grads:
                check("grad " + g, st[g].grad, named[g].grad, 1e-4)
            check("bn1.running_mean (after warm)", st["bn1.running_mean"], m.state_dict()["bn1.running_mean"])
            check("out_eval", ye, torch.from_numpy(out["out_eval"]), 1e-4)
            check("out_eval_warm", yw, torch.from_numpy(out["out_eval_warm"]), 1e-4)
        else:
            out["truth:out_train"] = yo.detach().numpy()
            out["truth:out_eval"] = ye.numpy()
            out["truth:out_eval_warm"] = yw.numpy()
            for g in grads:
                out["truth:grad:" + g] = st[g].grad.numpy()
            print("  noise(ref fp32 vs fp64): out_train %.1e out_eval %.1e warm %.1e" % (rel(y, yo), rel(torch.from_numpy(out["out_eval"]), ye), rel(torch.from_numpy(out["out_eval_warm"]), yw)))
    np.savez_compressed(os.path.join(HERE, "visual_%s.npz" % tag), **f32(out))


# conv filters whose gradient is stored as a strided sample (in addition to the digest of EVERY gradient):
# stem, first / middle / last residual layers, 1x1 and 3x3, plain and downsample paths
FULL_CONV_GRADS = ("conv1.weight", "conv3.weight", "layer1.0.conv2.weight", "layer1.0.downsample.0.weight",
                   "layer2.0.conv2.weight", "layer2.3.conv1.weight", "layer3.0.conv3.weight", "layer3.2.conv2.weight",
                   "layer4.0.downsample.0.weight", "layer4.2.conv2.weight", "layer4.2.conv3.weight")


def relu_margin(spec, B, tag, seed):
    """Smallest |ReLU input| of the fp32 oracle's train-mode forward for this (seed, batch)."""
    st = {k: (torch.zeros((), dtype=torch.int64) if k.endswith("num_batches_tracked") else OF.fill(k, s, seed, style="margin"))
          for k, s in OV.state_shapes(spec).items()}
    taps = {}
    with torch.no_grad():
        OV.visual_forward(st, OF.randn("img:" + tag, (B, 3, spec.height, spec.in_width), seed), spec, True, taps)
    return taps["relu_min"]


def pick_seed(spec, B, tag, start):
    for seed in range(start, start + 200):
        mn = relu_margin(spec, B, tag, seed)
        print("  seed %d: smallest |ReLU input| %.2e" % (seed, mn), flush=True)
        if mn >= OF.RELU_MIN:
            return seed, mn
    raise RuntimeError("no seed with a ReLU margin >= %g" % OF.RELU_MIN)


def gen_visual_full(tag, spec, B, seed):
    """Full-size encoder fixture under the ``margin`` fill style (oracle/fill.py): the reference's fp32
    result is within 1e-4 of an fp64 evaluation for EVERY stored quantity (asserted below), so the GPU
    test holds a flat 1e-3 on all of them.  Stored: train output, a digest of every parameter gradient,
    strided samples of 11 filter gradients, a digest of every stage's activation (G2 of SURVEY 8 c2),
    a digest of every BatchNorm running statistic, eval output cold and after WARM more train passes."""
    print("[visual %s, margin style, B=%d]" % (tag, B))
    seed, rmin = pick_seed(spec, B, tag, seed)
    m = ref_visual(spec)
    m.load_state_dict(OF.fill_state(m.state_dict(), seed, style="margin"))
    x = OF.randn("img:" + tag, (B, 3, spec.height, spec.in_width), seed)
    acts = {}
    hooks = [m.avgpool.register_forward_hook(lambda mod, i, o: acts.__setitem__("stem", o.detach()))]
    for li in range(1, 5):
        for bi, blk in enumerate(getattr(m, "layer%d" % li)):
            hooks.append(blk.register_forward_hook(lambda mod, i, o, k="layer%d.%d" % (li, bi): acts.__setitem__(k, o.detach())))
    m.train()
    y = m(x)
    for h in hooks:
        h.remove()
    w_out = OF.randn("gout:" + tag, tuple(y.shape), seed)
    (y * w_out).sum().backward()
    out = {"out_train": y.detach().numpy(), "spec": np.array(list(spec.layers) + [spec.width, spec.heads, spec.output_dim, spec.height, spec.in_width, B, seed])}
    named = dict(m.named_parameters())
    for k, p in named.items():
        out["gdig:" + k] = digest("grad:" + k, p.grad)
    for k in FULL_CONV_GRADS:
        out["grad:" + k] = strided_sample(named[k].grad).numpy()
    for k, a in acts.items():
        out["adig:" + k] = digest("act:" + k, a)
    for k, v in m.state_dict().items():
        if k.endswith(("running_mean", "running_var")):
            out["rdig:" + k] = digest("run:" + k, v)
    m.eval()
    with torch.no_grad():
        out["out_eval"] = m(x).numpy()
    m.train()
    with torch.no_grad():
        for _ in range(WARM):
            m(x)
        m.eval()
        out["out_eval_warm"] = m(x).numpy()

    worst = 0.0
    for dt in (torch.float32, torch.float64):
        st = {}
        for k, s in OV.state_shapes(spec).items():
            st[k] = torch.zeros((), dtype=torch.int64) if k.endswith("num_batches_tracked") else OF.fill(k, s, seed, style="margin").to(dt)
            if OV.is_param(k) and st[k].dtype.is_floating_point:
                st[k].requires_grad_(True)
        xx = x.to(dt)
        taps = {}
        yo = OV.visual_forward(st, xx, spec, True, taps)
        (yo * w_out.to(dt)).sum().backward()
        run1 = {k: v.detach().clone() for k, v in st.items() if k.endswith(("running_mean", "running_var"))}
        with torch.no_grad():
            ye = OV.visual_forward(st, xx, spec, False)
            for _ in range(WARM):
                OV.visual_forward(st, xx, spec, True)
            yw = OV.visual_forward(st, xx, spec, False)
        errs = {"out_train": rel(yo, y), "out_eval": rel(ye, torch.from_numpy(out["out_eval"])), "out_eval_warm": rel(yw, torch.from_numpy(out["out_eval_warm"]))}
        gfl = grad_floor([v for k, v in out.items() if k.startswith("gdig:")])
        for k, p in named.items():
            if k == "attnpool.k_proj.bias":  # analytically zero (softmax shift invariance): rounding residue on both sides
                continue
            errs["grad:" + k] = float((st[k].grad.double() - p.grad.double()).abs().max() / max(float(p.grad.abs().max()), gfl))
        for k in acts:
            errs["act:" + k] = rel(taps[k], acts[k])
        for k, v in run1.items():
            errs["run:" + k] = digest_err(digest("run:" + k, v), out["rdig:" + k])
        top = sorted(errs.items(), key=lambda kv: -kv[1])[:4]
        if dt == torch.float32:
            print("  oracle fp32 vs reference: worst", [(k, "%.1e" % v) for k, v in top])
            assert top[0][1] < 3e-4, top
        else:
            print("  reference fp32 vs fp64 truth (conditioning): worst", [(k, "%.1e" % v) for k, v in top])
            assert top[0][1] < 5e-4, ("fixture is not well-conditioned: choose another seed", top)
            worst = top[0][1]
            out["truth:out_train"] = yo.detach().numpy()
    out["conditioning"] = np.array(worst)
    out["relu_min"] = np.array(rmin)
    np.savez_compressed(os.path.join(HERE, "visual_%s.npz" % tag), **f32(out))


def gen_text(seed=3):
    print("[text]")
    hidden, embed, vocab, Lpad = 512, 512, 300, 105
    table = OF.randn("vocab_table", (vocab, embed), seed, 0.5)
    ref_gru.load_vocab_dict = lambda root, onehot: table.numpy()  # shim (ii)
    g = ref_gru.GRU(hidden, embed, embed, 1, 0.0, True, "clip_vit", "./")
    load_filled(g, seed)
    lens = [64, 9, 105, 33, 64, 1]
    tok, ln = synth_tokens("tok:text", len(lens), lens, vocab, Lpad, seed)
    y = g(make_captions(tok, ln))
    w_out = OF.randn("gout:text", tuple(y.shape), seed)
    (y * w_out).sum().backward()
    out = {"tokens": tok.numpy(), "lengths": ln.numpy(), "out": y.detach().numpy(), "seed": np.array(seed), "vocab": np.array(vocab)}
    named = dict(g.named_parameters())
    for k, p in named.items():
        out["grad:" + k] = p.grad[::7, ::5].numpy().copy()  # strided sample keeps the file small
    # a second batch whose max length is < 105 (zero-pad-in-max quirk: only up to batch max)
    lens2 = [12, 40, 7, 40]
    tok2, ln2 = synth_tokens("tok:text2", len(lens2), lens2, vocab, Lpad, seed)
    with torch.no_grad():
        y2 = g(make_captions(tok2, ln2))
    out.update(tokens2=tok2.numpy(), lengths2=ln2.numpy(), out2=y2.numpy())

    st = oracle_state(OT.state_shapes(hidden, embed), seed)
    for k in st:
        st[k].requires_grad_(True)
    yo = OT.text_forward(st, table, tok, ln)
    check("out", yo, y)
    (yo * w_out).sum().backward()
    for k, p in named.items():
        check("grad " + k, st[k].grad, p.grad, 1e-4)
    with torch.no_grad():
        check("out2", OT.text_forward(st, table, tok2, ln2), y2)
    np.savez_compressed(os.path.join(HERE, "text.npz"), **out)


def gen_text_embed(seed=4):
    """The text encoder's non-default input forms (gru.py:22-31,59-60): a trainable nn.Embedding(padding_idx=0) (`use_onehot ==
    "yes"`) and nn.Linear(vocab_size, embed_size) over the frozen table's rows (vocab_size != embed_size)."""
    print("[text_embed]")
    hidden, embed, vocab, Lpad, vdim = 64, 96, 120, 40, 48
    lens = [40, 9, 33, 1, 17, 40, 25]
    tok, ln = synth_tokens("tok:textemb", len(lens), lens, vocab, Lpad, seed)
    tok[2, 5] = tok[2, 11] = tok[4, 3] = tok[0, 0]  # repeated tokens across captions: their embedding rows accumulate
    out = {"tokens": tok.numpy(), "lengths": ln.numpy(), "seed": np.array(seed), "dims": np.array([hidden, embed, vocab, vdim])}
    # form 1: nn.Embedding
    g1 = ref_gru.GRU(hidden, vocab, embed, 1, 0.0, True, "yes", "./")
    load_filled(g1, seed, "emb1.")
    y1 = g1(make_captions(tok, ln))
    w_out = OF.randn("gout:textemb", tuple(y1.shape), seed)
    (y1 * w_out).sum().backward()
    out["out_embedding"] = y1.detach().numpy()
    for k, p in g1.named_parameters():
        out["grad_embedding:" + k] = p.grad.numpy().copy()
    st = {k: OF.fill("emb1." + k, tuple(v.shape), seed).requires_grad_(True) for k, v in g1.state_dict().items()}
    yo = OT.text_forward(st, None, tok, ln)
    check("embedding form: out", yo, y1)
    (yo * w_out).sum().backward()
    for k, p in g1.named_parameters():
        check("embedding form: grad " + k, st[k].grad, p.grad, 1e-4)
    assert float(g1.embed.weight.grad[0].abs().max()) == 0.0  # padding_idx
    # form 2: nn.Linear over the frozen rows
    table = OF.randn("vocab_table_lin", (vocab, vdim), seed, 0.5)
    ref_gru.load_vocab_dict = lambda root, onehot: table.numpy()  # shim (ii)
    g2 = ref_gru.GRU(hidden, vdim, embed, 1, 0.0, True, "clip_vit", "./")
    load_filled(g2, seed, "emb2.")
    y2 = g2(make_captions(tok, ln))
    (y2 * w_out).sum().backward()
    out["out_linear"] = y2.detach().numpy()
    for k, p in g2.named_parameters():
        out["grad_linear:" + k] = p.grad.numpy().copy()
    st2 = {k: OF.fill("emb2." + k, tuple(v.shape), seed).requires_grad_(True) for k, v in g2.state_dict().items()}
    yo2 = OT.text_forward(st2, table, tok, ln)
    check("linear form: out", yo2, y2)
    (yo2 * w_out).sum().backward()
    for k, p in g2.named_parameters():
        check("linear form: grad " + k, st2[k].grad, p.grad, 1e-4)
    np.savez_compressed(os.path.join(HERE, "text_embed.npz"), **out)


def ns(**kw):
    return types.SimpleNamespace(**kw)


HEAD_LR, HEAD_MOMENTUM, HEAD_WD = 0.02, 0.9, 4e-5


def head_groups(named):
    """Parameter groups of the reference's make_optimizer (lib/solver/build.py:6-18): one group per
    tensor, bias lr x2 and no weight decay."""
    groups = []
    for k, p in named:
        if not p.requires_grad:
            continue
        lr, wd = (2 * HEAD_LR, 0.0) if "bias" in k else (HEAD_LR, HEAD_WD)
        groups.append({"params": [p], "lr": lr, "weight_decay": wd})
    return groups


def head_step_inputs(s, spec, B, vocab, Lpad, seed):
    x = OF.randn("img:head%d" % s, (B, 3, spec.height, spec.in_width), seed)
    lens = [int(v) for v in OF.randint("len:head%d" % s, 3, 30, (B,), seed)]
    tok, ln = synth_tokens("tok:head%d" % s, B, lens, vocab, Lpad, seed)
    # ids: duplicates inside the batch, and from step 1 on, hits in the queue
    ids = torch.tensor([10 * s + (i // 2) for i in range(B)], dtype=torch.int64)
    if s > 0:
        ids[0] = 10 * (s - 1)  # equals an id already enqueued -> filtered column
    return x, tok, ln, ids


def gen_head(seed=5, steps=3, fc=False, fname="head.npz"):
    """Tiny visual encoder + small BiGRU + MoCo head, 3 optimiser steps with the reference's
    make_optimizer rule (SOLVER.OPTIMIZER "SGD", momentum 0.9, bias lr x2 / wd 0; lib/solver/build.py:6-23).
    SGD, not Adam: Adam's first steps are lr*sign(g), which turns rounding noise on near-zero gradients
    into full-size parameter differences; the fused Adam kernel has its own test against torch.optim.Adam.
    ``margin`` fill style (oracle/fill.py) so that every stored quantity of the fp32 reference is within
    3e-4 of an fp64 evaluation (asserted) and the GPU test can hold a flat 1e-3."""
    print("[head]")
    spec = OV.TINY
    hidden, embed, vocab, Lpad = 64, 64, 200, 105
    C, K, NC, B = 32, 32, 53, 8
    table = OF.randn("vocab_table_head", (vocab, embed), seed, 0.5)
    ref_gru.load_vocab_dict = lambda root, onehot: table.numpy()
    vis = ref_visual(spec)
    txt = ref_gru.GRU(hidden, embed, embed, 1, 0.0, True, "clip_vit", "./")
    cfg = ns(MODEL=ns(EMBEDDING=ns(FEATURE_SIZE=C, EPSILON=0.1), MOCO=ns(K=K, M=0.9, FC=fc), NUM_CLASSES=NC))
    head = MoCoHead(cfg, vis, txt)
    sd = head.state_dict()
    shapes = OH.state_shapes(spec, K, C, NC, hidden, embed, fc=fc)
    assert {k: tuple(v.shape) for k, v in sd.items()} == {k: tuple(s) for k, s in shapes.items()}, "head state mismatch"
    filled = OF.fill_state(sd, seed, "head.", style="margin")
    st0 = {k: v.clone() for k, v in filled.items()}
    OH.init_queues(st0, seed)
    for k in ("t_queue", "v_queue", "id_queue", "queue_ptr"):
        filled[k] = st0[k].clone()
    head.load_state_dict(filled)
    head.train()
    opt = torch.optim.SGD(head_groups(head.named_parameters()), lr=HEAD_LR, momentum=HEAD_MOMENTUM)

    out = {"dims": np.array([hidden, embed, vocab, Lpad, C, K, NC, B, seed, steps]), "m": np.array(0.9),
           "sgd": np.array([HEAD_LR, HEAD_MOMENTUM, HEAD_WD])}
    for s in range(steps):
        x, tok, ln, ids = head_step_inputs(s, spec, B, vocab, Lpad, seed)
        ld = head(x, make_captions(tok, ln, ids))
        opt.zero_grad()
        sum(ld.values()).backward()
        if s == 0:
            for k, p in head.named_parameters():
                if p.grad is not None:
                    out["gdig0:" + k] = digest("grad0:" + k, p.grad)
            for k in ("v_embed_layer.weight", "loss_evaluator.projection", "t_encoder_q.gru.weight_hh_l0", "v_encoder_q.conv1.weight",
                      "v_encoder_q.layer2.0.conv2.weight", "v_encoder_q.attnpool.q_proj.weight"):
                out["grad0:" + k] = dict(head.named_parameters())[k].grad.numpy().copy()
        opt.step()
        for k in ld:
            out["loss%d:%s" % (s, k)] = ld[k].detach().numpy()
        out["images%d" % s], out["tokens%d" % s], out["lengths%d" % s], out["ids%d" % s] = x.numpy(), tok.numpy(), ln.numpy(), ids.numpy()
    sd2 = head.state_dict()
    for k, v in sd2.items():  # the WHOLE state after 3 steps: parameters, key encoders, BN statistics, queues
        if v.dtype.is_floating_point:
            out["fdig:" + k] = digest("final:" + k, v)
    for k in ("v_queue", "t_queue", "id_queue", "queue_ptr", "v_encoder_k.conv1.weight", "t_encoder_k.gru.weight_ih_l0", "v_encoder_k.bn1.running_mean", "v_embed_layer.weight"):
        out["final:" + k] = sd2[k].numpy()
    head.eval()  # eval path (head.py:178-183)
    with torch.no_grad():
        ev = head(x, make_captions(tok, ln, ids))
    out["eval_v"], out["eval_t"] = ev[0].numpy(), ev[1].numpy()

    # the same trajectory on the oracle: fp32 must reproduce the reference, fp64 measures the conditioning
    for dt in (torch.float32, torch.float64):
        res = head_oracle(out, filled, spec, table, dt)
        errs = {}
        gfl = grad_floor([v for k, v in out.items() if k.startswith("gdig0:")])
        for k, v in res.items():
            ref = out[k]
            if k.startswith("gdig0:"):
                errs[k] = digest_err(v, ref, gfl * (100.0 if k.endswith("attnpool.k_proj.bias") else 1.0))
            elif k.startswith("fdig:"):
                errs[k] = digest_err(v, ref)
            elif k.startswith("grad0:"):
                errs[k] = float(np.abs(np.asarray(v, dtype=np.float64) - ref).max() / max(np.abs(ref).max(), gfl))
            else:
                errs[k] = rel(torch.as_tensor(v), torch.as_tensor(ref))
        top = sorted(errs.items(), key=lambda kv: -kv[1])[:4]
        if dt == torch.float32:
            print("  oracle fp32 vs reference: worst", [(k, "%.1e" % v) for k, v in top])
            assert top[0][1] < 3e-4, top
        else:
            print("  reference fp32 vs fp64 truth (conditioning): worst", [(k, "%.1e" % v) for k, v in top])
            assert top[0][1] < 5e-4, ("fixture is not well-conditioned", top)
            out["conditioning"] = np.array(top[0][1])
    out["fc"] = np.array(int(fc))
    np.savez_compressed(os.path.join(HERE, fname), **f32(out))


def head_oracle(out, filled, spec, table, dt):
    """The head fixture's trajectory on the oracle in precision ``dt``; returns the same keys."""
    hidden, embed, vocab, Lpad, C, K, NC, B, seed, steps = (int(v) for v in out["dims"])
    st = {k: (v.clone().to(dt) if v.dtype.is_floating_point else v.clone()) for k, v in filled.items()}
    tr = OH.trainable_names(st)
    for k in tr:
        st[k].requires_grad_(True)
    opt = torch.optim.SGD(head_groups([(k, st[k]) for k in tr]), lr=HEAD_LR, momentum=HEAD_MOMENTUM)
    res = {}
    for s in range(steps):
        x, tok, ln, ids = (torch.from_numpy(out["%s%d" % (k, s)]) for k in ("images", "tokens", "lengths", "ids"))
        taps = {}
        ld = OH.train_forward(st, spec, table.to(dt), x.to(dt), tok, ln, ids, m=0.9, epsilon=0.1, taps=taps)
        assert taps["visual_q"]["relu_min"] >= OF.RELU_MIN, ("head fixture: ReLU margin too small, change the seed", s, taps["visual_q"]["relu_min"])
        opt.zero_grad()
        sum(ld.values()).backward()
        if s == 0:
            for k in out:
                if k.startswith("gdig0:"):
                    res[k] = digest("grad0:" + k[6:], st[k[6:]].grad)
                elif k.startswith("grad0:"):
                    res[k] = st[k[6:]].grad.numpy().copy()
        opt.step()
        for k in ld:
            res["loss%d:%s" % (s, k)] = ld[k].detach().numpy()
    for k in out:
        if k.startswith("fdig:"):
            res[k] = digest("final:" + k[5:], st[k[5:]])
        elif k.startswith("final:") and st[k[6:]].dtype.is_floating_point:
            res[k] = st[k[6:]].detach().numpy().copy()
    ev = OH.eval_forward(st, spec, table.to(dt), x.to(dt), tok, ln)
    res["eval_v"], res["eval_t"] = ev[0].numpy(), ev[1].numpy()
    return res


def gen_losses(seed=11):
    print("[losses]")
    B, C, NC, K = 16, 32, 101, 48
    v = OF.randn("l:v", (B, C), seed)
    t = OF.randn("l:t", (B, C), seed)
    proj = OF.randn("l:p", (C, NC), seed, 0.3)
    lab = OF.randint("l:lab", 0, NC, (B,), seed)
    lab[1] = lab[0]
    out = {"v": v.numpy(), "t": t.numpy(), "proj": proj.numpy(), "labels": lab.numpy()}
    out["instance"] = ref_losses.instance_loss(proj, v, t, lab, epsilon=0.1).numpy()
    out["instance_eps0"] = ref_losses.instance_loss(proj, v, t, lab, epsilon=0.0).numpy()
    # the non-default forms of the signature (losses.py:42-54) and its epsilon quirk (any epsilon > 0 smooths with 0.1: losses.py:56,18)
    out["instance_s28_norm"] = ref_losses.instance_loss(proj, v, t, lab, scale=28, norm=True, epsilon=0.1).numpy()
    out["instance_s5"] = ref_losses.instance_loss(proj, v, t, lab, scale=5, norm=False, epsilon=0.0).numpy()
    out["instance_eps03"] = ref_losses.instance_loss(proj, v, t, lab, epsilon=0.3).numpy()
    out["global_align"] = ref_losses.global_align_loss(v, t, lab).numpy()
    vp, tp = OF.randn("l:vp", (B, 1), seed), OF.randn("l:tp", (B, 1), seed)
    vn, tn = OF.randn("l:vn", (B, K), seed), OF.randn("l:tn", (B, K), seed)
    out.update(v_pos=vp.numpy(), t_pos=tp.numpy(), v_neg=vn.numpy(), t_neg=tn.numpy())
    out["infonce"] = ref_losses.infonce_loss(vp, vn, tp, tn, 0.07).numpy()
    check("instance", OL.instance_loss(proj, v, t, lab, 0.1), torch.from_numpy(out["instance"]))
    check("instance eps0", OL.instance_loss(proj, v, t, lab, 0.0), torch.from_numpy(out["instance_eps0"]))
    check("instance s28 norm", OL.instance_loss(proj, v, t, lab, 0.1, scale=28, norm=True), torch.from_numpy(out["instance_s28_norm"]))
    check("instance s5", OL.instance_loss(proj, v, t, lab, 0.0, scale=5), torch.from_numpy(out["instance_s5"]))
    check("instance eps 0.3", OL.instance_loss(proj, v, t, lab, 0.3), torch.from_numpy(out["instance_eps03"]))
    check("global_align", OL.global_align_loss(v, t, lab), torch.from_numpy(out["global_align"]))
    check("infonce", OL.infonce_loss(vp, vn, tp, tn, 0.07), torch.from_numpy(out["infonce"]))
    np.savez_compressed(os.path.join(HERE, "losses.npz"), **out)


def gen_rank(seed=7):
    print("[rank]")
    out = {}
    for tag, (Q, G, P) in {"a": (50, 30, 12), "b": (200, 1000, 40)}.items():
        sim = OF.randn("rank:sim" + tag, (Q, G), seed)
        qp = OF.randint("rank:q" + tag, 0, P, (Q,), seed)
        gp = OF.randint("rank:g" + tag, 0, P, (G,), seed)
        if tag == "a":
            qp[3] = P + 5  # a query with no relevant gallery item -> AP NaN
        topk = torch.tensor([1, 5, 10])
        cmc, mAP, idx = ref_eval.rank(sim, qp, gp, topk, get_mAP=True)
        cmc2, idx2 = ref_eval.rank(sim, qp, gp, topk, get_mAP=False)
        out.update({"sim" + tag: sim.numpy(), "q" + tag: qp.numpy(), "g" + tag: gp.numpy(), "cmc" + tag: cmc.numpy(), "mAP" + tag: mAP.numpy(), "top10" + tag: idx2.numpy(), "cmc_topk" + tag: cmc2.numpy()})
        ocmc, omap, oidx = OE.rank(sim, qp, gp, (1, 5, 10), True)
        ocmc2, oidx2 = OE.rank(sim, qp, gp, (1, 5, 10), False)
        assert torch.equal(oidx, idx) and torch.equal(oidx2, idx2)
        assert torch.allclose(ocmc, cmc) and torch.allclose(ocmc2, cmc2)
        assert torch.allclose(omap, mAP, equal_nan=True), (omap, mAP)
        print("  rank %s ok (mAP %s)" % (tag, float(mAP)))
    # similarity (evaluation.py:117-120)
    te, ie = OF.randn("rank:te", (40, 32), seed), OF.randn("rank:ie", (25, 32), seed)
    import torch.nn.functional as F

    s_ref = torch.matmul(F.normalize(te, p=2, dim=1), F.normalize(ie, p=2, dim=1).t())
    check("similarity", OE.similarity(te, ie), s_ref)
    out.update(te=te.numpy(), ie=ie.numpy(), sim_ti=s_ref.numpy())
    # k-reciprocal re-rank (evaluation.py:40-65,122-124) and the re-ranked metrics (:144-163)
    tn, im = F.normalize(te, p=2, dim=1), F.normalize(ie, p=2, dim=1)
    rtn = ref_eval.k_reciprocal(im, tn)
    rvn = ref_eval.k_reciprocal(tn, im)
    check("k_reciprocal rtn", OE.k_reciprocal(im, tn), rtn, 1e-12)
    check("k_reciprocal rvn", OE.k_reciprocal(tn, im), rvn, 1e-12)
    tp = OF.randint("rank:tp", 0, 9, (40,), seed)
    ip = OF.randint("rank:ip", 0, 9, (25,), seed)
    topk = torch.tensor([1, 5, 10])
    re_t2i_cmc, re_t2i_map, re_idx = ref_eval.rank(rvn + s_ref, tp, ip, topk, get_mAP=True)
    re_i2t_cmc, re_i2t_map, _ = ref_eval.rank(rtn + s_ref.t(), ip, tp, topk, get_mAP=True)
    out.update(rtn=rtn.numpy(), rvn=rvn.numpy(), tp=tp.numpy(), ip=ip.numpy(), re_t2i_cmc=re_t2i_cmc.numpy(),
               re_t2i_map=re_t2i_map.numpy(), re_i2t_cmc=re_i2t_cmc.numpy(), re_i2t_map=re_i2t_map.numpy(),
               re_t2i_idx=re_idx.numpy())
    np.savez_compressed(os.path.join(HERE, "rank.npz"), **out)


def gen_ingest(seed=13):
    """resize_pos_embed / state_filter (m_resnet.py:220-243): CLIP 7x7 grid -> 24x8 / 6x2."""
    print("[ingest]")
    out = {}
    pe = OF.randn("ingest:pos", (50, 64), seed)
    for tag, gs in (("a", (24, 8)), ("b", (6, 2))):
        out["pos_" + tag] = ref_mr.resize_pos_embed(pe, gs).numpy()
    sd = {"visual.conv1.weight": OF.randn("ingest:w", (4, 3, 3, 3), seed), "visual.attnpool.positional_embedding": pe,
          "token_embedding.weight": OF.randn("ingest:t", (5, 4), seed)}
    flt = ref_mr.state_filter(sd, (6, 2))
    out["filter_keys"] = np.array(sorted(flt.keys()))
    out["filter_pos"] = flt["attnpool.positional_embedding"].numpy()
    out["pos_in"] = pe.numpy()
    # suffix-matching aligner of the Checkpointer (lib/utils/checkpoint.py:90-148): which loaded key each model key takes
    import lib.utils.checkpoint as ref_ckpt

    model_keys = ["embed_model.v_encoder_q.conv1.weight", "embed_model.v_encoder_q.layer1.0.conv1.weight", "visual_model.conv1.weight",
                  "embed_model.v_encoder_k.layer1.0.bn1.running_mean", "embed_model.t_queue", "embed_model.loss_evaluator.projection",
                  "embed_model.v_embed_layer.bias", "textual_model.gru.weight_hh_l0"]
    loaded_keys = ["module.conv1.weight", "module.v_encoder_q.conv1.weight", "module.layer1.0.conv1.weight", "module.bn1.running_mean",
                   "module.embed_model.t_queue", "module.projection", "module.unrelated.bias", "module.gru.weight_hh_l0"]
    ms = {k: torch.full((1,), float(i)) for i, k in enumerate(model_keys)}
    ls = ref_ckpt.strip_prefix_if_present({k: torch.full((1,), 100.0 + i) for i, k in enumerate(loaded_keys)}, prefix="module.")
    ref_ckpt.align_and_update_state_dicts(ms, ls)
    out["align_model_keys"] = np.array(model_keys)
    out["align_loaded_keys"] = np.array(loaded_keys)
    out["align_values"] = np.array([float(ms[k]) for k in model_keys])  # >= 100: taken from loaded key (value - 100)
    np.savez_compressed(os.path.join(HERE, "ingest.npz"), **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["losses", "rank", "text", "text_embed", "tiny", "head", "head_fc", "rn50", "rn101", "ingest"]
    if "ingest" in which:
        gen_ingest()
    if "losses" in which:
        gen_losses()
    if "rank" in which:
        gen_rank()
    if "text" in which:
        gen_text()
    if "text_embed" in which:
        gen_text_embed()
    if "tiny" in which:
        gen_visual("tiny", OV.TINY, 4, 1, grads=("conv1.weight", "bn1.weight", "conv2.weight", "layer1.0.conv2.weight", "layer2.0.downsample.0.weight", "layer3.0.conv2.weight", "layer4.0.bn3.bias", "attnpool.k_proj.weight", "attnpool.q_proj.bias", "attnpool.positional_embedding", "attnpool.c_proj.weight"))
    if "head" in which:
        gen_head()
    if "head_fc" in which:  # MODEL.MOCO.FC = True (the default of lib/config/defaults.py:56): projection-head branch
        gen_head(seed=6, steps=2, fc=True, fname="head_fc.npz")
    if "rn50" in which:
        gen_visual_full("rn50", OV.RN50, 8, 2)
    if "rn101" in which:
        gen_visual_full("rn101", OV.RN101, 8, 2)
    print("done")
