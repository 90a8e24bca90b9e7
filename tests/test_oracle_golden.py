"""CPU: the oracle restatement reproduces the golden vectors captured from the
imported reference (tests/golden/make_golden.py).  No GPU, no /root/reference."""

import os

import numpy as np
import pytest
import torch

import oracle.evaluation as OE
import oracle.fill as OF
import oracle.head as OH
import oracle.losses as OL
import oracle.text as OT
import oracle.visual as OV


def rel(a, b):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def filled(shapes, seed, prefix=""):
    st = {}
    for k, s in shapes.items():
        st[k] = torch.zeros((), dtype=torch.int64) if k.endswith("num_batches_tracked") else OF.fill(prefix + k, s, seed)
    return st


def test_losses(golden_dir):
    g = load(golden_dir, "losses.npz")
    v, t, p, lab = (torch.from_numpy(g[k]) for k in ("v", "t", "proj", "labels"))
    assert rel(OL.instance_loss(p, v, t, lab, 0.1), g["instance"]) < 1e-6
    assert rel(OL.instance_loss(p, v, t, lab, 0.0), g["instance_eps0"]) < 1e-6
    assert rel(OL.instance_loss(p, v, t, lab, 0.1, scale=28, norm=True), g["instance_s28_norm"]) < 1e-6
    assert rel(OL.instance_loss(p, v, t, lab, 0.0, scale=5), g["instance_s5"]) < 1e-6
    assert rel(OL.instance_loss(p, v, t, lab, 0.3), g["instance_eps03"]) < 1e-6  # (the reference smooths with 0.1 for any epsilon > 0)
    assert rel(OL.global_align_loss(v, t, lab), g["global_align"]) < 1e-6
    args = [torch.from_numpy(g[k]) for k in ("v_pos", "v_neg", "t_pos", "t_neg")]
    assert rel(OL.infonce_loss(*args, 0.07), g["infonce"]) < 1e-6


@pytest.mark.parametrize("tag", ["a", "b"])
def test_rank(golden_dir, tag):
    g = load(golden_dir, "rank.npz")
    sim, q, gp = (torch.from_numpy(g[k + tag]) for k in ("sim", "q", "g"))
    cmc, mAP, idx = OE.rank(sim, q, gp, (1, 5, 10), True)
    cmc2, idx2 = OE.rank(sim, q, gp, (1, 5, 10), False)
    assert np.array_equal(idx2.numpy(), g["top10" + tag])
    assert np.allclose(cmc.numpy(), g["cmc" + tag]) and np.allclose(cmc2.numpy(), g["cmc_topk" + tag])
    assert np.allclose(float(mAP), float(g["mAP" + tag]), equal_nan=True)
    if tag == "a":
        assert np.isnan(float(mAP))  # no-relevant query -> NaN, as the reference


def test_similarity(golden_dir):
    g = load(golden_dir, "rank.npz")
    assert rel(OE.similarity(torch.from_numpy(g["te"]), torch.from_numpy(g["ie"])), g["sim_ti"]) < 1e-6


def test_text(golden_dir):
    g = load(golden_dir, "text.npz")
    seed, vocab = int(g["seed"]), int(g["vocab"])
    table = OF.randn("vocab_table", (vocab, 512), seed, 0.5)
    st = filled(OT.state_shapes(512, 512), seed)
    for k in st:
        st[k].requires_grad_(True)
    y = OT.text_forward(st, table, torch.from_numpy(g["tokens"]), torch.from_numpy(g["lengths"]))
    assert rel(y, g["out"]) < 1e-5
    (y * OF.randn("gout:text", tuple(y.shape), seed)).sum().backward()
    for k in st:
        assert rel(st[k].grad[::7, ::5], g["grad:" + k]) < 1e-4
    with torch.no_grad():
        y2 = OT.text_forward(st, table, torch.from_numpy(g["tokens2"]), torch.from_numpy(g["lengths2"]))
    assert rel(y2, g["out2"]) < 1e-5


def test_text_input_forms(golden_dir):
    """The oracle's non-default text input forms (gru.py:22-31,59-60) against vectors captured from the reference module: a
    trainable nn.Embedding(padding_idx=0) and nn.Linear over the frozen table's rows - outputs and every gradient."""
    g = load(golden_dir, "text_embed.npz")
    seed = int(g["seed"])
    hidden, embed, vocab, vdim = (int(v) for v in g["dims"])
    tok, ln = torch.from_numpy(g["tokens"]), torch.from_numpy(g["lengths"])
    w_out = OF.randn("gout:textemb", (tok.shape[0], 2 * hidden), seed)
    shapes = dict(OT.state_shapes(hidden, embed))
    for form, pre, table, extra in (("embedding", "emb1.", None, {"embed.weight": (vocab, embed)}),
                                    ("linear", "emb2.", OF.randn("vocab_table_lin", (vocab, vdim), seed, 0.5), {"embed.weight": (embed, vdim), "embed.bias": (embed,)})):
        st = {k: OF.fill(pre + k, s, seed).requires_grad_(True) for k, s in dict(shapes, **extra).items()}
        y = OT.text_forward(st, table, tok, ln)
        assert rel(y, g["out_" + form]) < 1e-5
        (y * w_out).sum().backward()
        for k in st:
            assert rel(st[k].grad, g["grad_%s:%s" % (form, k)]) < 1e-4, (form, k)
    assert float(np.abs(g["grad_embedding:embed.weight"][0]).max()) == 0.0  # padding_idx = 0


def test_text_zero_pad_enters_max():
    """gru.py:63 quirk: a caption shorter than the batch max gets max(.,0)."""
    st = filled(OT.state_shapes(16, 16), 0)
    table = OF.randn("tb", (20, 16), 0)
    tok = OF.randint("tk", 1, 20, (2, 12), 0)
    alone = OT.text_forward(st, table, tok[:1], torch.tensor([5]))
    both = OT.text_forward(st, table, tok, torch.tensor([5, 9]))
    assert torch.allclose(both[0], alone[0].clamp(min=0), atol=1e-6)


@pytest.mark.parametrize("tag,spec", [("rn50", OV.RN50), ("rn101", OV.RN101)])
def test_visual_full_size(golden_dir, tag, spec):
    """Full-size encoders (B=8, 384x128): the oracle reproduces every quantity the fixture pins - train
    output, all parameter gradients (digests + strided filter-gradient samples), every stage's
    activation digest, every BatchNorm running statistic - to 3e-4 (measured <= 1.4e-4; the largest are
    residual-branch BatchNorm bias gradients, which are near-zero by shift invariance)."""
    from fixture_check import assert_within, visual_full_errors

    g = load(golden_dir, "visual_%s.npz" % tag)
    B, seed = int(g["spec"][-2]), int(g["spec"][-1])
    st = {k: (torch.zeros((), dtype=torch.int64) if k.endswith("num_batches_tracked") else OF.fill(k, s, seed, style="margin"))
          for k, s in OV.state_shapes(spec).items()}
    for k in st:
        if OV.is_param(k):
            st[k].requires_grad_(True)
    x = OF.randn("img:" + tag, (B, 3, spec.height, spec.in_width), seed)
    taps = {}
    y = OV.visual_forward(st, x, spec, True, taps)
    (y * OF.randn("gout:" + tag, tuple(y.shape), seed)).sum().backward()
    errs = visual_full_errors(g, y, lambda k: st[k].grad, taps, st)
    with torch.no_grad():
        errs["out_eval"] = rel(OV.visual_forward(st, x, spec, False), g["out_eval"])
    assert float(g["conditioning"]) < 5e-4  # reference fp32 vs fp64, measured when the fixture was made
    assert_within(errs, 3e-4)


@pytest.mark.parametrize("tag,spec", [("tiny", OV.TINY)])
def test_visual(golden_dir, tag, spec):
    g = load(golden_dir, "visual_%s.npz" % tag)
    B, seed = int(g["spec"][-2]), int(g["spec"][-1])
    st = filled(OV.state_shapes(spec), seed)
    for k in st:
        if OV.is_param(k):
            st[k].requires_grad_(True)
    x = OF.randn("img:" + tag, (B, 3, spec.height, spec.in_width), seed)
    y = OV.visual_forward(st, x, spec, True)
    assert rel(y, g["out_train"]) < 2e-5
    (y * OF.randn("gout:" + tag, tuple(y.shape), seed)).sum().backward()
    for k in g.files:
        if k.startswith("grad:"):
            assert rel(st[k[5:]].grad, g[k]) < 1e-4, k
    assert rel(st["bn1.running_mean"], g["bn1_running_mean"]) < 1e-6
    assert rel(st["bn1.running_var"], g["bn1_running_var"]) < 1e-6
    with torch.no_grad():
        assert rel(OV.visual_forward(st, x, spec, False), g["out_eval"]) < 2e-5


def head_setup(g):
    hidden, embed, vocab, Lpad, C, K, NC, B, seed, steps = (int(v) for v in g["dims"])
    spec = OV.TINY
    table = OF.randn("vocab_table_head", (vocab, embed), seed, 0.5)
    shapes = OH.state_shapes(spec, K, C, NC, hidden, embed, fc=bool(int(g["fc"])) if "fc" in g.files else False)
    st = {}
    for k, s in shapes.items():
        if k.endswith("num_batches_tracked"):
            st[k] = torch.zeros((), dtype=torch.int64)
        elif k in ("id_queue", "queue_ptr"):
            st[k] = torch.zeros(s, dtype=torch.int64)
        else:
            st[k] = OF.fill("head." + k, s, seed, style="margin")
    OH.init_queues(st, seed)
    return st, spec, table, (hidden, embed, vocab, Lpad, C, K, NC, B, seed, steps)


def sgd_groups(named, lr, wd):
    """lib/solver/build.py:6-18: one group per tensor, bias lr x2 / no weight decay."""
    return [{"params": [p], "lr": 2 * lr if "bias" in k else lr, "weight_decay": 0.0 if "bias" in k else wd} for k, p in named]


@pytest.mark.parametrize("fname", ["head.npz", "head_fc.npz"])
def test_head_three_steps(golden_dir, fname):
    """SGD steps of the whole MoCo head on the oracle against the reference-captured trajectory: losses, every
    step-0 gradient, the entire final state, eval embeddings.  head_fc.npz: MODEL.MOCO.FC = True (projection heads
    with momentum copies, head.py:32-49,86-94,117-144)."""
    from fixture_check import assert_within, head_errors

    g = load(golden_dir, fname)
    st, spec, table, dims = head_setup(g)
    steps = dims[-1]
    lr, mom, wd = (float(v) for v in g["sgd"])
    tr = OH.trainable_names(st)
    for k in tr:
        st[k].requires_grad_(True)
    opt = torch.optim.SGD(sgd_groups([(k, st[k]) for k in tr], lr, wd), lr=lr, momentum=mom)
    losses, g0 = {}, {}
    for s in range(steps):
        x, tok, ln, ids = (torch.from_numpy(g["%s%d" % (k, s)]) for k in ("images", "tokens", "lengths", "ids"))
        ld = OH.train_forward(st, spec, table, x, tok, ln, ids, m=float(g["m"]), epsilon=0.1)
        opt.zero_grad()
        sum(ld.values()).backward()
        if s == 0:
            g0 = {k: st[k].grad.clone() for k in tr}
        opt.step()
        for k in ld:
            losses["loss%d:%s" % (s, k)] = ld[k].detach()
    ev = OH.eval_forward(st, spec, table, x, tok, ln)
    assert float(g["conditioning"]) < 5e-4
    assert_within(head_errors(g, losses, lambda k: g0[k], st, ev), 3e-4)


def test_negative_filter_is_batch_wide():
    idq = torch.tensor([[5, 7, -1, 9, 7]])
    neg = OH.negative_columns(idq, torch.tensor([7, 3]))
    assert neg.tolist() == [0, 2, 3]
