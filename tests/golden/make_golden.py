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


def ns(**kw):
    return types.SimpleNamespace(**kw)


def gen_head(seed=5, steps=3):
    """Tiny visual encoder + small BiGRU + MoCo head, 3 optimiser steps (Adam via
    the reference's make_optimizer rule: bias lr x2, wd 0)."""
    print("[head]")
    spec = OV.TINY
    hidden, embed, vocab, Lpad = 64, 64, 200, 105
    C, K, NC, B = 32, 32, 53, 8
    table = OF.randn("vocab_table_head", (vocab, embed), seed, 0.5)
    ref_gru.load_vocab_dict = lambda root, onehot: table.numpy()
    vis = ref_visual(spec)
    txt = ref_gru.GRU(hidden, embed, embed, 1, 0.0, True, "clip_vit", "./")
    cfg = ns(MODEL=ns(EMBEDDING=ns(FEATURE_SIZE=C, EPSILON=0.1), MOCO=ns(K=K, M=0.9, FC=False), NUM_CLASSES=NC))
    head = MoCoHead(cfg, vis, txt)
    sd = head.state_dict()
    shapes = OH.state_shapes(spec, K, C, NC, hidden, embed)
    assert {k: tuple(v.shape) for k, v in sd.items()} == {k: tuple(s) for k, s in shapes.items()}, "head state mismatch"
    filled = OF.fill_state(sd, seed, "head.")
    st = {k: v.clone() for k, v in filled.items()}
    OH.init_queues(st, seed)
    for k in ("t_queue", "v_queue", "id_queue", "queue_ptr"):
        filled[k] = st[k].clone()
    head.load_state_dict(filled)
    head.train()

    # reference optimiser rule (lib/solver/build.py:6-25)
    groups = []
    for k, p in head.named_parameters():
        if not p.requires_grad:
            continue
        lr, wd = 1e-3, 4e-5
        if "bias" in k:
            lr, wd = 2e-3, 0.0
        groups.append({"params": [p], "lr": lr, "weight_decay": wd})
    opt = torch.optim.Adam(groups, lr=1e-3, betas=(0.9, 0.999), eps=1e-8)

    tr = OH.trainable_names(st)
    for k in tr:
        st[k].requires_grad_(True)
    ogroups = []
    for k in tr:
        lr, wd = (2e-3, 0.0) if "bias" in k else (1e-3, 4e-5)
        ogroups.append({"params": [st[k]], "lr": lr, "weight_decay": wd})
    oopt = torch.optim.Adam(ogroups, lr=1e-3, betas=(0.9, 0.999), eps=1e-8)

    out = {"dims": np.array([hidden, embed, vocab, Lpad, C, K, NC, B, seed, steps]), "m": np.array(0.9)}
    for s in range(steps):
        x = OF.randn("img:head%d" % s, (B, 3, spec.height, spec.in_width), seed)
        lens = [int(v) for v in OF.randint("len:head%d" % s, 3, 30, (B,), seed)]
        tok, ln = synth_tokens("tok:head%d" % s, B, lens, vocab, Lpad, seed)
        # ids: duplicates inside the batch, and from step 1 on, hits in the queue
        ids = torch.tensor([10 * s + (i // 2) for i in range(B)], dtype=torch.int64)
        if s > 0:
            ids[0] = 10 * (s - 1)  # equals an id already enqueued -> filtered column
        caps = make_captions(tok, ln, ids)
        ld = head(x, caps)
        loss = sum(ld.values())
        opt.zero_grad()
        loss.backward()
        if s == 0:
            g0 = {k: p.grad.clone() for k, p in head.named_parameters() if p.grad is not None}
        opt.step()

        old = OH.train_forward(st, spec, table, x, tok, ln, ids, m=0.9, epsilon=0.1)
        oloss = sum(old.values())
        oopt.zero_grad()
        oloss.backward()
        if s == 0:
            for k in ("v_embed_layer.weight", "loss_evaluator.projection", "t_encoder_q.gru.weight_hh_l0", "v_encoder_q.conv1.weight", "v_encoder_q.attnpool.q_proj.weight"):
                check("step0 grad " + k, st[k].grad, g0[k], 2e-4)
                out["grad0:" + k] = g0[k].numpy()
        oopt.step()
        for k in ld:
            check("step%d %s" % (s, k), old[k], ld[k], 1e-4)
            out["loss%d:%s" % (s, k)] = ld[k].detach().numpy()
        out["images%d" % s] = x.numpy()
        out["tokens%d" % s] = tok.numpy()
        out["lengths%d" % s] = ln.numpy()
        out["ids%d" % s] = ids.numpy()
    sd2 = head.state_dict()
    for k in ("v_queue", "t_queue", "id_queue", "queue_ptr", "v_encoder_k.conv1.weight", "t_encoder_k.gru.weight_ih_l0", "v_encoder_k.bn1.running_mean", "v_embed_layer.weight"):
        check("final " + k, st[k].float(), sd2[k].float(), 2e-4)
        out["final:" + k] = sd2[k].numpy()
    # eval path (head.py:178-183)
    head.eval()
    with torch.no_grad():
        ev = head(x, caps)
    eo = OH.eval_forward(st, spec, table, x, tok, ln)
    check("eval v", eo[0], ev[0], 2e-4)
    check("eval t", eo[1], ev[1], 2e-4)
    out["eval_v"], out["eval_t"] = ev[0].numpy(), ev[1].numpy()

    # fp64 oracle trajectory = "truth" for the noise-aware GPU criterion
    t64 = head_truth(out, filled, spec, table, (hidden, embed, vocab, Lpad, C, K, NC, B, seed, steps), 0.9)
    for k, v in t64.items():
        out["truth:" + k] = v
        if k in out:
            print("  noise(ref fp32 vs fp64) %-40s %.1e" % (k, rel(torch.from_numpy(np.asarray(out[k])).double(), torch.from_numpy(v))))
    np.savez_compressed(os.path.join(HERE, "head.npz"), **f32(out))


def head_truth(out, filled, spec, table, dims, m):
    hidden, embed, vocab, Lpad, C, K, NC, B, seed, steps = dims
    dt = torch.float64
    st = {k: (v.clone().to(dt) if v.dtype.is_floating_point else v.clone()) for k, v in filled.items()}
    tr = OH.trainable_names(st)
    groups = []
    for k in tr:
        st[k].requires_grad_(True)
        lr, wd = (2e-3, 0.0) if "bias" in k else (1e-3, 4e-5)
        groups.append({"params": [st[k]], "lr": lr, "weight_decay": wd})
    opt = torch.optim.Adam(groups, lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
    res = {}
    for s in range(steps):
        x, tok, ln, ids = (torch.from_numpy(out["%s%d" % (k, s)]) for k in ("images", "tokens", "lengths", "ids"))
        ld = OH.train_forward(st, spec, table.to(dt), x.to(dt), tok, ln, ids, m=m, epsilon=0.1)
        opt.zero_grad()
        sum(ld.values()).backward()
        if s == 0:
            for k in out:
                if k.startswith("grad0:"):
                    res[k] = st[k[6:]].grad.numpy().copy()
        opt.step()
        for k in ld:
            res["loss%d:%s" % (s, k)] = ld[k].detach().numpy()
    for k in out:
        if k.startswith("final:") and st[k[6:]].dtype.is_floating_point:
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
    out["global_align"] = ref_losses.global_align_loss(v, t, lab).numpy()
    vp, tp = OF.randn("l:vp", (B, 1), seed), OF.randn("l:tp", (B, 1), seed)
    vn, tn = OF.randn("l:vn", (B, K), seed), OF.randn("l:tn", (B, K), seed)
    out.update(v_pos=vp.numpy(), t_pos=tp.numpy(), v_neg=vn.numpy(), t_neg=tn.numpy())
    out["infonce"] = ref_losses.infonce_loss(vp, vn, tp, tn, 0.07).numpy()
    check("instance", OL.instance_loss(proj, v, t, lab, 0.1), torch.from_numpy(out["instance"]))
    check("instance eps0", OL.instance_loss(proj, v, t, lab, 0.0), torch.from_numpy(out["instance_eps0"]))
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
    np.savez_compressed(os.path.join(HERE, "ingest.npz"), **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["losses", "rank", "text", "tiny", "head", "rn50", "rn101", "ingest"]
    if "ingest" in which:
        gen_ingest()
    if "losses" in which:
        gen_losses()
    if "rank" in which:
        gen_rank()
    if "text" in which:
        gen_text()
    if "tiny" in which:
        gen_visual("tiny", OV.TINY, 4, 1, grads=("conv1.weight", "bn1.weight", "conv2.weight", "layer1.0.conv2.weight", "layer2.0.downsample.0.weight", "layer3.0.conv2.weight", "layer4.0.bn3.bias", "attnpool.k_proj.weight", "attnpool.q_proj.bias", "attnpool.positional_embedding", "attnpool.c_proj.weight"))
    if "head" in which:
        gen_head()
    if "rn50" in which:
        gen_visual("rn50", OV.RN50, 2, 2, grads=("bn1.weight", "layer4.2.bn3.bias", "attnpool.c_proj.bias"))
    if "rn101" in which:
        gen_visual("rn101", OV.RN101, 2, 2, grads=("bn1.weight", "attnpool.c_proj.bias"))
    print("done")
