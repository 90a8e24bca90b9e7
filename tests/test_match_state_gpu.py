"""GPU parity: retrieval match / rank metric vs the reference-captured goldens
(bit-exact indices), fused similarity+top-k, MoCo state kernels (EMA, enqueue),
fused Adam vs torch.optim.Adam, and size-independent properties at full size."""

import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import oracle.evaluation as OE  # noqa: E402
import oracle.fill as OF  # noqa: E402
import oracle.head as OH  # noqa: E402
import oracle.visual as OV  # noqa: E402


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda")


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


@pytest.mark.parametrize("tag", ["a", "b"])
def test_rank_matches_reference_golden(gpu, golden_dir, tag):
    from textreid_amd.evaluation import rank

    g = load(golden_dir, "rank.npz")
    sim, q, gp = (torch.from_numpy(g[k + tag]).to(gpu) for k in ("sim", "q", "g"))
    cmc, mAP, idx = rank(sim, q, gp, (1, 5, 10), get_mAP=True)
    cmc2, idx2 = rank(sim, q, gp, (1, 5, 10), get_mAP=False)
    assert np.array_equal(idx2.cpu().numpy(), g["top10" + tag]), "top-10 indices"  # index work: bit exact
    ref_full = torch.argsort(torch.from_numpy(g["sim" + tag]), dim=1, descending=True).numpy()
    mine = idx.cpu().numpy()
    simn = g["sim" + tag]
    # exact ties have no defined order in torch.argsort: positions may differ only where the values tie
    assert np.array_equal(np.take_along_axis(simn, mine, 1), np.take_along_axis(simn, ref_full, 1)), "sorted values"
    assert np.array_equal(np.sort(mine, 1), np.sort(ref_full, 1)), "permutation"
    assert int((mine != ref_full).sum()) <= 4
    assert np.allclose(cmc.cpu().numpy(), g["cmc" + tag], atol=1e-4), (cmc, g["cmc" + tag])
    assert np.allclose(cmc2.cpu().numpy(), g["cmc_topk" + tag], atol=1e-4), (cmc2, g["cmc_topk" + tag])
    assert np.allclose(float(mAP), float(g["mAP" + tag]), rtol=1e-5, equal_nan=True), (float(mAP), float(g["mAP" + tag]))


def test_similarity_matches_reference_golden(gpu, golden_dir):
    from textreid_amd.evaluation import similarity

    g = load(golden_dir, "rank.npz")
    s = similarity(torch.from_numpy(g["te"]).to(gpu), torch.from_numpy(g["ie"]).to(gpu))
    assert np.allclose(s.cpu().numpy(), g["sim_ti"], atol=2e-6)


@pytest.mark.parametrize("Q,G,k", [(37, 1000, 10), (128, 20000, 10), (5, 64, 1), (64, 8192 * 2 + 77, 16)])
def test_fused_similarity_topk(gpu, Q, G, k):
    from textreid_amd.evaluation import similarity_topk

    te, ie = OF.randn("tk:q%d" % Q, (Q, 256), 1), OF.randn("tk:g%d" % G, (G, 256), 1)
    vals, idx = similarity_topk(te.to(gpu), ie.to(gpu), k)
    sim = OE.similarity(te, ie)
    rv, ri = torch.topk(sim, k, dim=1)
    assert torch.equal(idx.cpu(), ri)
    assert torch.allclose(vals.cpu(), rv, atol=2e-6)


@pytest.mark.parametrize("G,k", [(10, 10), (63, 5), (257, 10), (4099, 16), (20000, 10)])
def test_topk_rows_adversarial_orders_and_ties(gpu, G, k):
    """Row top-k kernel on the orders a threshold filter is weakest on: ascending (every element
    is admitted), descending, constant (pure ties -> lowest indices), few distinct values, -inf."""
    from textreid_amd.evaluation import call, _p, stream

    gen = torch.Generator().manual_seed(G)
    rows = [
        torch.arange(G, dtype=torch.float32),                      # ascending
        -torch.arange(G, dtype=torch.float32),                     # descending
        torch.full((G,), 0.25),                                    # constant
        torch.randint(0, 3, (G,), generator=gen).float(),          # heavy ties
        torch.randn(G, generator=gen),
        torch.where(torch.rand(G, generator=gen) < 0.5, torch.tensor(-float("inf")), torch.randn(G, generator=gen)),
    ]
    sim = torch.stack(rows).to(gpu)
    Q = sim.shape[0]
    vals = torch.empty(Q, k, device=gpu)
    idx = torch.empty(Q, k, dtype=torch.int64, device=gpu)
    call("trid_topk_rows_f32", _p(sim), G, Q, G, k, _p(vals), _p(idx), stream())
    # expected: stable descending sort = (value desc, index asc)
    order = torch.sort(sim.cpu(), dim=1, descending=True, stable=True)
    assert torch.equal(vals.cpu(), order.values[:, :k])
    finite = torch.isfinite(order.values[:, :k])
    assert torch.equal(idx.cpu()[finite], order.indices[:, :k][finite])


@pytest.mark.parametrize("Q,G,k", [(1, 100, 10), (5, 8192, 10), (5, 8193, 3), (70, 8192 + 64, 10), (130, 8192 * 3 + 5, 16), (64, 8192 + 63, 1)])
def test_fused_similarity_topk_shape_edges(gpu, Q, G, k):
    """Chunk boundaries of the retrieval path: exactly one panel, one column past it, a filtered tail narrower
    than a GEMM tile, fewer queries than a tile (falls back to panel passes), k = 1 and k = 16."""
    from textreid_amd.evaluation import similarity_topk

    te, ie = OF.randn("tke:q%d" % Q, (Q, 64), 2), OF.randn("tke:g%d" % G, (G, 64), 2)
    vals, idx = similarity_topk(te.to(gpu), ie.to(gpu), k)
    rv, ri = torch.topk(OE.similarity(te, ie), k, dim=1)
    assert torch.equal(idx.cpu(), ri)
    assert torch.allclose(vals.cpu(), rv, atol=2e-6)


@pytest.mark.parametrize("Q,G,k", [(1, 8193, 3), (70, 8192 + 64 + 5, 10), (130, 8192 * 3 + 5, 16), (300, 8192 + 40000, 10), (33, 8192 + 63, 1)])
def test_fused_similarity_topk_presplit_operands(gpu, Q, G, k):
    """C = 256 embeddings: queries and gallery are split into their fp16 planes once and the admission-filter pass runs on the
    streaming kernel with the queries resident in registers (trid_sim_topk_p16) - query counts that are not multiples of the
    32-column wave panel (zero padding rows must never produce candidates), a ragged last gallery tile, k = 1 / 16; against the
    dense product, and against the on-the-fly-split path it replaces (same arithmetic, another summation order: last-bit
    differences, 6e-8 measured - the indices agree wherever two neighbours are further apart than that)."""
    import textreid_amd.evaluation as E

    te, ie = OF.randn("tkp:q%d" % Q, (Q, 256), 4), OF.randn("tkp:g%d" % G, (G, 256), 4)
    assert E.USE_SIM_P16
    vals, idx = E.similarity_topk(te.to(gpu), ie.to(gpu), k)
    rv, ri = torch.topk(OE.similarity(te, ie), k, dim=1)
    assert torch.allclose(vals.cpu(), rv, atol=2e-6)
    clear = torch.ones_like(ri, dtype=torch.bool)  # (neighbours closer than the arithmetic's last bits may swap)
    clear[:, :-1] &= (rv[:, :-1] - rv[:, 1:]) > 1e-6
    clear[:, 1:] &= (rv[:, :-1] - rv[:, 1:]) > 1e-6
    assert torch.equal(idx.cpu()[clear], ri[clear]) and float(clear.float().mean()) > 0.9
    try:
        E.USE_SIM_P16 = False
        v0, i0 = E.similarity_topk(te.to(gpu), ie.to(gpu), k)
    finally:
        E.USE_SIM_P16 = True
    assert torch.allclose(v0, vals, atol=3e-7, rtol=0)
    gap_ok = torch.ones_like(idx, dtype=torch.bool)
    gap_ok[:, :-1] &= (vals[:, :-1] - vals[:, 1:]) > 3e-7
    gap_ok[:, 1:] &= (vals[:, :-1] - vals[:, 1:]) > 3e-7
    assert torch.equal(i0[gap_ok], idx[gap_ok])


def test_fused_similarity_topk_presplit_overflow_and_negative_thresholds(gpu):
    """The pre-split path on the orders a threshold filter is weakest on: a gallery ascending along its index (every later row
    beats the first chunk's thresholds: the lists overflow, the gated dense passes must answer), and queries whose best
    similarities are all NEGATIVE (the zero rows a ragged tile reads must not be admitted as 0 >= threshold)."""
    from textreid_amd.evaluation import similarity_topk

    G, Q, C = 8192 * 2 + 1234, 96, 256
    gen = torch.Generator().manual_seed(3)
    gal = torch.zeros(G, C)
    gal[:, 0] = torch.arange(G, dtype=torch.float32) * 1e-4
    gal[:, 1] = torch.randn(G, generator=gen)
    gal[:, 2] = -torch.rand(G, generator=gen) - 0.5               # strictly negative
    q = torch.zeros(Q, C)
    q[:32, 0] = 1.0            # ascending with the index: overflow
    q[32:64, 1] = 1.0          # random order
    q[64:, 2] = 1.0            # every similarity negative
    vals, idx = similarity_topk(q.to(gpu), gal.to(gpu), 10, normalize=False)
    order = torch.sort(q @ gal.t(), dim=1, descending=True, stable=True)
    assert torch.equal(idx.cpu(), order.indices[:, :10])
    assert torch.allclose(vals.cpu(), order.values[:, :10], rtol=1e-6, atol=0)


@pytest.mark.parametrize("device_gated", [False, True])
def test_fused_similarity_topk_presplit_segments_and_late_overflow(gpu, device_gated):
    """A gallery long enough for the filter pass to run in SEGMENTS (8192 | 57344 | the rest, each merged before the next
    starts): random queries (thresholds tighten from segment to segment), and queries whose similarity ascends with the
    index only in the LAST segment - its lists overflow after the earlier segments were merged, so the fall-back must
    restart from the top-k of the first chunk (no row twice).  Both fall-back forms: host-read flag, device-gated."""
    import textreid_amd.evaluation as E

    G, Q, C = 8192 * 8 + 40000, 70, 256
    gen = torch.Generator().manual_seed(5)
    gal = torch.randn(G, C, generator=gen) * 0.05
    gal[:, 0] = torch.randn(G, generator=gen)
    gal[8192 * 8:, 0] = 4.0 + torch.arange(G - 8192 * 8, dtype=torch.float32) * 1e-4   # ascending: 40000 admissions per list
    q = torch.randn(Q, C, generator=gen) * 0.05
    q[:32, 0] = 1.0
    q[32:, 0] = 0.0            # these never see the ascending column: ordinary lists
    try:
        E.DEVICE_GATED_FALLBACK = device_gated
        vals, idx = E.similarity_topk(q.to(gpu), gal.to(gpu), 10, normalize=False)
    finally:
        E.DEVICE_GATED_FALLBACK = False
    order = torch.sort(q.double() @ gal.double().t(), dim=1, descending=True, stable=True)
    assert all(len(set(r)) == 10 for r in idx.cpu().tolist())
    assert torch.allclose(vals.cpu().double(), order.values[:, :10], rtol=2e-6, atol=2e-6)
    clear = torch.ones(Q, 10, dtype=torch.bool)
    gap = order.values[:, :10] - order.values[:, 1:11]
    clear &= gap > 1e-5
    clear[:, 1:] &= gap[:, :-1] > 1e-5
    assert torch.equal(idx.cpu()[clear], order.indices[:, :10][clear]) and float(clear.float().mean()) > 0.8


def test_rank_full_argsort_beyond_the_lds_sort(gpu):
    """rank(get_mAP=True) on a gallery wider than the in-LDS bitonic sort (G > 16384: ICFG-PEDES i2t has 19 848
    captions): packed keys + segmented radix sort (argsort_large.hip).  Index-exact against a stable descending sort,
    with heavy ties (quantised similarities, +-0) - ties resolve to the lower column, like the in-LDS kernel."""
    from textreid_amd.evaluation import rank

    Q, G = 37, 20011
    gen = torch.Generator().manual_seed(11)
    sim = torch.randn(Q, G, generator=gen)
    sim[:20] = (sim[:20] * 4).round() / 4          # many exact ties
    sim[5, ::3] = 0.0
    sim[5, 1::3] = -0.0                             # signed zeros compare equal
    q_pids = torch.randint(0, 50, (Q,), generator=gen)
    g_pids = torch.randint(0, 50, (G,), generator=gen)
    cmc, mAP, idx = rank(sim.to(gpu), q_pids, g_pids, topk=[1, 5, 10], get_mAP=True)
    ref = torch.sort(sim.double(), dim=1, descending=True, stable=True).indices
    assert torch.equal(idx.cpu(), ref)
    # metrics against the reference formulas (evaluation.py:20-36) on the same ranking
    matches = (g_pids[ref] == q_pids[:, None]).float()
    first = matches.argmax(dim=1)
    for t, k in enumerate([1, 5, 10]):
        assert abs(float(cmc[t]) - 100.0 * float((first < k).float().mean())) < 1e-3
    prec = matches.cumsum(1) / torch.arange(1, G + 1)[None, :]
    ap = (prec * matches).sum(1) / matches.sum(1)
    assert abs(float(mAP) - 100.0 * float(ap.mean())) < 1e-2


def test_fused_similarity_topk_overflow_fallback(gpu):
    """Gallery ordered so that every later column beats the thresholds set by the first chunk: the
    admission-filter candidate lists overflow and the gated dense passes must produce the exact answer.
    Mixed with rows that do not overflow (random) and heavy ties."""
    from textreid_amd.evaluation import similarity_topk

    G, Q, C = 8192 * 2 + 1234, 96, 64
    gen = torch.Generator().manual_seed(3)
    gal = torch.zeros(G, C)
    gal[:, 0] = torch.arange(G, dtype=torch.float32) * 1e-4      # ascending along the gallery
    gal[:, 1] = torch.randn(G, generator=gen)
    gal[:, 2] = torch.randint(0, 2, (G,), generator=gen).float()  # two-valued -> ties
    q = torch.zeros(Q, C)
    q[:32, 0] = 1.0            # rows 0-31: similarity ascending with the index (worst case)
    q[32:64, 1] = 1.0          # rows 32-63: random order
    q[64:, 2] = 1.0            # rows 64-95: ties everywhere -> lowest indices win
    vals, idx = similarity_topk(q.to(gpu), gal.to(gpu), 10, normalize=False)
    sim = q @ gal.t()
    order = torch.sort(sim, dim=1, descending=True, stable=True)
    # indices exact (ties -> lowest index); values carry the fp16 two-plane split of the operands (2^-22 relative each)
    assert torch.equal(idx.cpu(), order.indices[:, :10])
    assert torch.allclose(vals.cpu(), order.values[:, :10], rtol=1e-6, atol=0)


def test_topk_full_size_properties(gpu):
    """Config-5-shaped shard at reduced Q: top-k is sorted, indices valid and
    unique, and every returned value equals the recomputed dot product."""
    from textreid_amd.evaluation import similarity_topk

    Q, G = 256, 125000  # one of 8 gallery shards of the 1e6 gallery
    gen = torch.Generator(device="cpu").manual_seed(7)
    q = torch.nn.functional.normalize(torch.randn(Q, 256, generator=gen), dim=1).to(gpu)
    gal = torch.nn.functional.normalize(torch.randn(G, 256, generator=gen), dim=1).to(gpu)
    vals, idx = similarity_topk(q, gal, 10, normalize=False)
    assert bool((vals[:, :-1] >= vals[:, 1:]).all())
    assert int(idx.min()) >= 0 and int(idx.max()) < G
    assert all(len(set(r.tolist())) == 10 for r in idx.cpu())
    re = (q[:, None, :] * gal[idx]).sum(-1)
    assert torch.allclose(re, vals, atol=2e-6)
    kth = vals[:, -1:]
    full = q[:8] @ gal.t()  # spot check of 8 rows against the dense product (torch only as checker)
    assert bool(((full > kth[:8] + 1e-6).sum(1) <= 9).all())


def test_config4_full_gallery_one_gpu(gpu):
    """configs[4] at full size on ONE GPU: Q = 1e4 text queries x G = 1e6 gallery images (1 GB of fp32
    embeddings), fused similarity + top-10.  Index-exact against a dense CPU product for 64 spot rows
    (near-ties, |gap| < 2e-6, may swap: then the VALUES must agree); size-independent properties for
    all 1e4 rows: sorted, indices valid and distinct, every value equals the recomputed dot product."""
    from textreid_amd.evaluation import similarity_topk

    Q, G, C, k = 10000, 1000000, 256, 10
    gen = torch.Generator(device="cpu").manual_seed(7)
    qc = torch.nn.functional.normalize(torch.randn(Q, C, generator=gen), dim=1)
    gc = torch.nn.functional.normalize(torch.randn(G, C, generator=gen), dim=1)
    q, gal = qc.to(gpu), gc.to(gpu)
    vals, idx = similarity_topk(q, gal, k, normalize=False)
    assert bool((vals[:, :-1] >= vals[:, 1:]).all())
    assert int(idx.min()) >= 0 and int(idx.max()) < G
    srt = torch.sort(idx, dim=1).values
    assert bool((srt[:, 1:] != srt[:, :-1]).all())  # distinct per row
    re = torch.empty_like(vals)
    for r0 in range(0, Q, 1000):  # recomputed dot products, chunked (torch only as checker)
        re[r0:r0 + 1000] = (q[r0:r0 + 1000, None, :] * gal[idx[r0:r0 + 1000]]).sum(-1)
    assert torch.allclose(re, vals, atol=2e-6)
    rows = torch.randperm(Q, generator=gen)[:64]
    dense = qc[rows] @ gc.t()  # CPU fp32, 64 x 1e6
    rv, ri = torch.topk(dense, k, dim=1)
    gv, gi = vals[rows.to(gpu)].cpu(), idx[rows.to(gpu)].cpu()
    assert torch.allclose(gv, rv, atol=2e-6)
    diff = gi != ri
    if bool(diff.any()):  # only near-ties may differ in order
        assert bool(((gv - rv).abs()[diff] < 2e-6).all())
        assert bool((dense.gather(1, gi)[diff] - rv[diff]).abs().max() < 2e-6)
    assert float(diff.float().mean()) < 0.01


def test_fused_adam_matches_torch_adam(gpu):
    from textreid_amd.solver import FusedAdam

    torch.manual_seed(0)
    shapes = [(64, 32, 3, 3), (128,), (1000, 17), (5,), (70001,)]
    ps1 = [torch.nn.Parameter(torch.randn(s, device=gpu)) for s in shapes]
    ps1[0].data = ps1[0].data.contiguous(memory_format=torch.channels_last)
    ps2 = [torch.nn.Parameter(p.detach().clone()) for p in ps1]
    groups = lambda ps: [{"params": [p], "lr": 1e-2 * (1 + i % 2), "weight_decay": 0.0 if i % 2 else 4e-2} for i, p in enumerate(ps)]
    o1 = FusedAdam(groups(ps1), lr=1e-2, betas=(0.9, 0.999), eps=1e-8)
    o2 = torch.optim.Adam(groups(ps2), lr=1e-2, betas=(0.9, 0.999), eps=1e-8)
    for it in range(5):
        for i, (a, b) in enumerate(zip(ps1, ps2)):
            if i == 3 and it < 2:  # this parameter starts receiving gradients two steps late: its OWN step count
                a.grad = b.grad = None  # (and bias correction) lags the others, as in torch.optim.Adam
                continue
            gr = torch.randn_like(b)
            a.grad = gr.clone()
            b.grad = gr.clone()
        o1.step()
        o2.step()
        if it == 2:
            for grp in o1.param_groups + o2.param_groups:
                grp["lr"] *= 0.1  # LR scheduler changes per-group lr
    for a, b in zip(ps1, ps2):
        d = (a.detach() - b.detach()).abs()
        assert torch.allclose(a.detach(), b.detach(), rtol=1e-5, atol=1e-6), (float(d.max()), float((d / (b.detach().abs() + 1e-12)).max()))
    assert [int(o1.state[p]["step"]) for p in ps1] == [5, 5, 5, 3, 5] == [int(o2.state[p]["step"]) for p in ps2]


def test_fused_adam_resume_from_state_dict(gpu):
    """Resume flow of the reference (train_net.py:69-72: build the optimiser, then Checkpointer.resume ->
    optimizer.load_state_dict): after a state-dict round trip into a FRESH FusedAdam the bias-correction step
    count and the moment tensors are the saved ones - the continued trajectory equals torch.optim.Adam's."""
    from textreid_amd.solver import FusedAdam

    torch.manual_seed(1)
    shapes = [(33, 17), (64,), (8, 4, 3, 3)]
    mk = lambda: [torch.nn.Parameter(torch.randn(s, device=gpu)) for s in shapes]
    ref_p = mk()
    fus_p = [torch.nn.Parameter(p.detach().clone()) for p in ref_p]
    ref = torch.optim.Adam(ref_p, lr=1e-2)
    fus = FusedAdam(fus_p, lr=1e-2)
    grads = [[torch.randn(s, device=gpu) for s in shapes] for _ in range(6)]

    def run(opt, ps, its):
        for it in its:
            for p, g in zip(ps, grads[it]):
                p.grad = g.clone()
            opt.step()

    run(ref, ref_p, range(3))
    run(fus, fus_p, range(3))
    saved = fus.state_dict()
    fus2_p = [torch.nn.Parameter(p.detach().clone()) for p in fus_p]
    fus2 = FusedAdam(fus2_p, lr=1e-2)
    run(fus2, fus2_p, [0])          # a step BEFORE loading: the cached pointer tables must not survive the load
    for p, q in zip(fus2_p, fus_p):
        p.data.copy_(q.data)
    fus2.load_state_dict(saved)
    run(ref, ref_p, range(3, 6))
    run(fus2, fus2_p, range(3, 6))
    for a, b in zip(fus2_p, ref_p):
        assert torch.allclose(a.detach(), b.detach(), rtol=1e-5, atol=1e-6)
    assert all(int(st["step"]) == 6 for st in fus2.state.values())


def test_enqueue_with_a_misaligned_pointer_wraps(gpu):
    """A queue_ptr that is not a multiple of the batch (checkpoint written with another batch / world size) must
    wrap around the ring, never write past the end of the queues (the reference's slice assignment raises there)."""
    from textreid_amd import ops

    K, C, B = 64, 32, 16
    vq, tq = torch.zeros(K + B, C, device=gpu), torch.zeros(K + B, C, device=gpu)  # B guard rows behind the queue
    idq = -torch.ones(K + B, dtype=torch.int64, device=gpu)
    ptr = torch.tensor([K - 8], dtype=torch.int64, device=gpu)
    vk, tk = torch.randn(B, C, device=gpu), torch.randn(B, C, device=gpu)
    ids = torch.arange(B, device=gpu) + 500
    ops.call("trid_enqueue_f32", ops._p(vq), ops._p(tq), ops._p(idq), ops._p(ptr), ops._p(vk), ops._p(tk), ops._p(ids), K, C, B, ops.stream())
    assert int(ptr) == 8
    assert torch.equal(vq[K - 8 : K], vk[:8]) and torch.equal(vq[:8], vk[8:]) and torch.equal(tq[:8], tk[8:])
    assert torch.equal(idq[K - 8 : K], ids[:8]) and torch.equal(idq[:8], ids[8:])
    assert float(vq[K:].abs().max()) == 0.0 and bool((idq[K:] == -1).all())  # nothing behind the ring was touched


def test_ema_and_enqueue(gpu):
    import types

    from textreid_amd.backbones.gru import GRU
    from textreid_amd.backbones.m_resnet import ModifiedResNet
    from textreid_amd.embeddings.moco_head.head import MoCoHead

    ns = types.SimpleNamespace
    vis = ModifiedResNet([1, 1, 1, 1], 64, 4, 1, (96, 32), 16)
    txt = GRU(64, 64, 64, 1, 0.0, True, "clip_vit", "./", vocab_dict=torch.zeros(10, 64))
    cfg = ns(MODEL=ns(EMBEDDING=ns(FEATURE_SIZE=32, EPSILON=0.1), MOCO=ns(K=64, M=0.9, FC=False), NUM_CLASSES=53))
    head = MoCoHead(cfg, vis, txt).to(gpu)
    with torch.no_grad():
        for p in head.v_encoder_q.parameters():
            p.add_(torch.randn_like(p) * 0.1)
    q = [p.detach().clone() for p in list(head.v_encoder_q.parameters()) + list(head.t_encoder_q.parameters())]
    k0 = [p.detach().clone() for p in list(head.v_encoder_k.parameters()) + list(head.t_encoder_k.parameters())]
    rm0 = head.v_encoder_k.bn1.running_mean.clone()
    head._momentum_update_key_encoder()
    k1 = list(head.v_encoder_k.parameters()) + list(head.t_encoder_k.parameters())
    for a, b, c in zip(k1, k0, q):
        assert torch.equal(a, b * 0.9 + c * (1.0 - 0.9))  # same association as head.py:79,83 -> bit exact
    assert torch.equal(head.v_encoder_k.bn1.running_mean, rm0)  # buffers untouched
    # enqueue: ring buffer at device-resident pointer, reference layout [C,K] preserved in state_dict
    vq0, tq0 = head.v_queue.clone(), head.t_queue.clone()
    for step in range(5):  # K=64, B=16: wraps after 4 pushes
        vk, tk = torch.randn(16, 32, device=gpu), torch.randn(16, 32, device=gpu)
        ids = torch.arange(16, device=gpu) + 100 * step
        ptr = int(head.queue_ptr)
        head._dequeue_and_enqueue(vk, tk, ids)
        vq0[:, ptr : ptr + 16] = vk.t()
        tq0[:, ptr : ptr + 16] = tk.t()
        assert int(head.queue_ptr) == (ptr + 16) % 64
        assert torch.equal(head.id_queue[0, ptr : ptr + 16], ids)
    assert torch.equal(head.state_dict()["v_queue"], vq0) and torch.equal(head.state_dict()["t_queue"], tq0)
    assert head.state_dict()["v_queue"].shape == (32, 64)
    with pytest.raises(AssertionError):
        head._dequeue_and_enqueue(torch.randn(24, 32, device=gpu), torch.randn(24, 32, device=gpu), torch.arange(24, device=gpu))


def test_queue_wraparound_and_all_positive_ids(gpu):
    """K = 2B: the ring pointer wraps on the second step and the third overwrites the first slab; every sample
    of a step shares ONE id (all pairs positive) and later steps hit those ids in the queue (columns masked
    out of the InfoNCE negatives).  Losses and queue state against the oracle for 3 steps."""
    from textreid_amd.backbones.gru import GRU
    from textreid_amd.backbones.m_resnet import ModifiedResNet
    from textreid_amd.caption import CaptionBatch
    from textreid_amd.embeddings.moco_head.head import MoCoHead
    import types

    spec, hidden, embed, vocab, C, B, NC, seed = OV.TINY, 64, 64, 200, 32, 4, 53, 77
    K = 2 * B
    ns = types.SimpleNamespace
    table = OF.randn("vocab_table_wrap", (vocab, embed), seed, 0.5)
    vis = ModifiedResNet(list(spec.layers), spec.output_dim, spec.heads, spec.last_stride, (spec.height, spec.in_width), spec.width)
    txt = GRU(hidden, embed, embed, 1, 0.0, True, "clip_vit", "./", vocab_dict=table)
    cfg = ns(MODEL=ns(EMBEDDING=ns(FEATURE_SIZE=C, EPSILON=0.1), MOCO=ns(K=K, M=0.9, FC=False), NUM_CLASSES=NC))
    head = MoCoHead(cfg, vis, txt)
    filled = OF.fill_state(head.state_dict(), seed, "wrap.")
    st = {k: v.clone() for k, v in filled.items()}
    OH.init_queues(st, seed)
    for k in ("t_queue", "v_queue", "id_queue", "queue_ptr"):
        filled[k] = st[k].clone()
    head.load_state_dict(filled)
    head.to(gpu).train()
    for step, id_value in enumerate((7, 7, 9)):  # step 1 finds step 0's ids in the queue; step 2 wraps
        x = OF.randn("img:wrap%d" % step, (B, 3, spec.height, spec.in_width), seed)
        tok = OF.randint("tok:wrap%d" % step, 1, vocab, (B, 105), seed)
        ln = OF.randint("len:wrap%d" % step, 2, 50, (B,), seed)
        for i, n in enumerate(ln.tolist()):
            tok[i, n:] = 0
        ids = torch.full((B,), id_value, dtype=torch.int64)
        ld = head(x.to(gpu), CaptionBatch(tok.to(gpu), ln.to(gpu), ids.to(gpu)))
        with torch.no_grad():
            old = OH.train_forward(st, spec, table, x, tok, ln, ids, m=0.9, epsilon=0.1)
        for k in old:
            assert rel(ld[k], old[k]) < 1e-3, (step, k, float(ld[k]), float(old[k]))
        sd = head.state_dict()
        assert int(sd["queue_ptr"]) == int(st["queue_ptr"]) == ((step + 1) * B) % K
        assert torch.equal(sd["id_queue"].cpu(), st["id_queue"])
        assert rel(sd["v_queue"], st["v_queue"]) < 1e-3 and rel(sd["t_queue"], st["t_queue"]) < 1e-3


def test_train_step_is_deterministic(gpu):
    """Two identical runs give bit-identical losses (no atomics anywhere on the path)."""
    import bench
    from textreid_amd.caption import CaptionBatch
    from textreid_amd.config import moco_cfg
    from textreid_amd.model import build_model
    from textreid_amd.solver import make_optimizer

    outs = []
    for _ in range(2):
        torch.manual_seed(0)
        cfg = moco_cfg("m_resnet50", K=64)
        model = build_model(cfg, vocab_dict=torch.randn(1000, 512) * 0.02).to(gpu).train()
        opt = make_optimizer(cfg, model)
        losses = []
        for s in range(2):
            images, tokens, lengths, ids = bench.synth_batch(8, s, gpu, 3, vocab=1000)
            ld = model(images, CaptionBatch(tokens, lengths, ids, max_len=64))
            opt.zero_grad()
            sum(ld.values()).backward()
            opt.step()
            losses.append([float(v) for v in ld.values()])
        outs.append(losses)
    assert outs[0] == outs[1]


def test_captured_train_step_equals_eager_bitwise(gpu):
    """engine.graph.CapturedTrainStep: the hipGraph replay of the train step (forward on four streams, three losses,
    queue push, backward, FusedAdam with per-group lr that an LR scheduler changes between steps) produces the SAME BITS
    as the eager step - losses every step, every parameter / moment / queue entry / BatchNorm buffer at the end - and a
    batch of another shape falls back to the eager step (loudly) without corrupting the recording."""
    import bench
    from textreid_amd.caption import CaptionBatch
    from textreid_amd.config import moco_cfg
    from textreid_amd.engine.graph import CapturedTrainStep
    from textreid_amd.model import build_model
    from textreid_amd.solver import make_optimizer

    B, K, steps = 8, 64, 7
    cfg = moco_cfg("m_resnet50", K=K)
    table = torch.randn(3000, 512, generator=torch.Generator().manual_seed(1)) * 0.02
    batches = [bench.synth_batch(B, s, gpu, 5, vocab=3000) for s in range(steps)]
    small = bench.synth_batch(4, 99, gpu, 5, vocab=3000)
    runs = {}
    for mode in ("eager", "graph", "streams"):
        torch.manual_seed(0)
        model = build_model(cfg, vocab_dict=table).to(gpu).train()
        opt = make_optimizer(cfg, model)
        runner = CapturedTrainStep(model, opt, warmup=2, caption_bound=64, launch="graph" if mode == "eager" else mode)
        losses = []
        for i in range(steps):
            images, tokens, lengths, ids = batches[i]
            cb = CaptionBatch(tokens, lengths, ids % 11003, max_len=64)
            if i == 3:  # a resume in mid-run: the optimizer's moment tensors AND its pointer tables are replaced - the
                # recorded Adam launch has the old tables' addresses baked in, so the runner must notice, run one eager
                # step (which rebuilds them) and record again (never replay against freed tables)
                opt.load_state_dict(opt.state_dict())
            if i == 4:  # an LR scheduler step between two training steps
                for grp in opt.param_groups:
                    grp["lr"] *= 0.5
            if i == 5:  # a ragged last batch of an epoch: other shape
                im2, tk2, ln2, id2 = small
                cb2 = CaptionBatch(tk2, ln2, id2 % 11003, max_len=64)
                ld = runner._eager(im2, cb2) if mode == "eager" else runner(im2, cb2)
                losses.append(torch.stack([v.detach().clone() for v in ld.values()]))
            ld = runner._eager(images, cb) if mode == "eager" else runner(images, cb)
            losses.append(torch.stack([v.detach().clone() for v in ld.values()]))
        torch.cuda.synchronize()
        if mode != "eager":
            assert runner.graph is not None and runner.recaptures == 1 and runner.plan is opt._plan
        if mode == "streams":  # the recording read back and re-issued as stream launches by the library (csrc/step_replay.hip)
            info = runner.replay_info
            assert runner.replayer is not None and info["kernels"] == info["nodes"] > 500 and info["copies"] == 0
            assert 2 <= info["lanes"] <= 8 and info["events"] >= info["lanes"] - 1
        runs[mode] = (torch.stack(losses), {k: v.detach().clone() for k, v in model.state_dict().items()},
                      [opt.state[p]["exp_avg_sq"].clone() for g_ in opt.param_groups for p in g_["params"]],
                      [int(opt.state[p]["step"]) for g_ in opt.param_groups for p in g_["params"]])
        del model, opt, runner
    for other in ("graph", "streams"):
        assert torch.equal(runs["eager"][0], runs[other][0]), (other, (runs["eager"][0] - runs[other][0]).abs().max())
        for k, v in runs["eager"][1].items():
            assert torch.equal(v, runs[other][1][k]), (other, k)
        for a, b in zip(runs["eager"][2], runs[other][2]):
            assert torch.equal(a, b), other
        assert runs["eager"][3] == runs[other][3] and set(runs[other][3]) == {steps + 1}


def test_config1_b128_replay_equals_eager(gpu):
    """The seam between the TIMED launch form and the oracle-checked one at configs[1]'s own size (B = 128, K = 8192: the
    split-count rules, tile shapes and BatchNorm partial counts of the benchmark): from one saved state the same batch goes
    through the recorded step re-issued by csrc/step_replay.hip - FusedAdam included - and through the eager step; the three
    losses, every parameter / buffer (queues, pointer, BatchNorm statistics, key encoders) and every Adam moment agree BIT
    for bit.  `bench.py` runs the same function after its timed region (`replay_equals_eager_b128`); the eager step at this
    size is what tests/test_model_gpu.py::test_config1_b128_k8192_step_vs_oracle holds against the oracle
    (lib/engine/trainer.py:81-91)."""
    import bench
    from textreid_amd.caption import CaptionBatch
    from textreid_amd.config import moco_cfg
    from textreid_amd.engine.graph import CapturedTrainStep
    from textreid_amd.model import build_model
    from textreid_amd.solver import make_optimizer

    B, K = 128, 8192
    torch.manual_seed(0)
    cfg = moco_cfg("m_resnet50", K=K)
    table = torch.randn(49408, 512, generator=torch.Generator().manual_seed(1)) * 0.02
    model = build_model(cfg, vocab_dict=table).to(gpu).train()
    opt = make_optimizer(cfg, model)
    runner = CapturedTrainStep(model, opt, warmup=2, caption_bound=64)
    batches = [bench.synth_batch(B, s, gpu, 1234) for s in range(4)]

    def batch(i):
        images, tokens, lengths, ids = batches[i % 4]
        return images, CaptionBatch(tokens, lengths, (ids + (i // 4) * 4 * (B // 4)) % 11003, max_len=64)

    i = 0
    while runner.graph is None and i < 6:
        runner(*batch(i))
        i += 1
    assert runner.graph is not None and runner.replayer is not None
    for _ in range(2):  # two replayed steps before the comparison: moments and queues are mid-run, not at their start values
        runner(*batch(i))
        i += 1
    res = bench.replay_equals_eager(runner, model, opt, *batch(i))
    print("replay vs eager at B=128:", {k: v for k, v in res.items() if k != "losses"})
    assert res["launch_form"] == "stream replay" and res["adam_moment_pairs"] == 183
    assert res["state_tensors_changed_by_the_step"] > 350  # (query + key encoders, queues, BatchNorm statistics moved)
    assert res["equal"], res["first_differences"]
    # ... and through hipGraphLaunch of the same recording
    runner.force_graph_launch = True
    res = bench.replay_equals_eager(runner, model, opt, *batch(i + 1))
    runner.force_graph_launch = False
    assert res["launch_form"] == "hipgraph replay" and res["equal"], res["first_differences"]


def test_schedule_and_fusion_switches_leave_the_gradients_alone(gpu):
    """The round-5 switches of the backward pass against the forms they replace, on one step of the full model (B = 8):
    SCHEDULING switches - a block's weight gradients behind one event (m_resnet._BATCH_WGRAD), the attention pool's weight
    gradients on the side stream (_SERIAL_ATTN_WGRAD), the data-gradient filter forms packed during the forward (_EARLY_WPT) -
    must not change a bit of any gradient; FUSION switches - BatchNorm-backward sums from the data-gradient GEMM's epilogue
    (ops.USE_BNB_FUSE), bn3 + downsample in one pass (ops.USE_BN_DUAL) - change the order of the per-channel sums only:
    every gradient within 2e-5 of its tensor's largest entry."""
    import bench
    import textreid_amd.backbones.m_resnet as MR
    from textreid_amd import ops
    from textreid_amd.caption import CaptionBatch
    from textreid_amd.config import moco_cfg
    from textreid_amd.model import build_model

    cfg = moco_cfg("m_resnet50", K=64)
    table = torch.randn(3000, 512, generator=torch.Generator().manual_seed(1)) * 0.02
    images, tokens, lengths, ids = bench.synth_batch(8, 3, gpu, 5, vocab=3000)

    def grads():
        torch.manual_seed(0)
        model = build_model(cfg, vocab_dict=table).to(gpu).train()
        loss = sum(model(images, CaptionBatch(tokens, lengths, ids % 11003, max_len=64)).values())
        loss.backward()
        torch.cuda.synchronize()
        return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}

    base = grads()
    assert len(base) > 150
    switches = [(MR, "_BATCH_WGRAD", 0, True), (MR, "_SERIAL_ATTN_WGRAD", True, True), (MR, "_EARLY_WPT", False, True),
                (ops, "USE_BNB_FUSE", False, False), (ops, "USE_BN_DUAL", False, False), (MR, "_BN3_FUSE", False, False)]
    for mod, name, off, exact in switches:
        was = getattr(mod, name)
        try:
            setattr(mod, name, off)
            other = grads()
        finally:
            setattr(mod, name, was)
        assert other.keys() == base.keys(), name
        for k, g in base.items():
            if exact:
                assert torch.equal(g, other[k]), (name, k)
            else:
                assert float((g - other[k]).abs().max()) <= 2e-5 * float(g.abs().max()) + 1e-12, (name, k)


def test_step_replay_of_a_small_forked_recording(gpu):
    """csrc/step_replay.hip on a recording small enough to check by hand: a chain on the capturing stream, a fork to a side
    stream, a join - the plan must keep two lanes and one event per cross-lane edge, re-running it must recompute the outputs
    from the CURRENT contents of the static inputs, and work enqueued on the caller's stream before / after the call must be
    ordered before / after the step.  A recording that holds a host-to-device copy (the one node type this runtime does not
    let the library read back) is refused, not guessed at."""
    import ctypes

    from textreid_amd import ops

    x = torch.zeros(1 << 16, device=gpu)
    y = torch.zeros_like(x)
    side = torch.cuda.Stream(device=gpu)
    g = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(g):
        a = x * 2.0
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            b = a + 1.0
            b2 = b * b
        c = a - 3.0
        main.wait_stream(side)
        out = b2 + c + y
    h = ctypes.c_void_p()
    ops.call("trid_step_replay_build", int(g.raw_cuda_graph()), 8, ctypes.byref(h))
    counts = (ctypes.c_int * 8)()
    ops.call("trid_step_replay_info", h, counts)
    nodes, kernels, copies, memsets, lanes, events, waits, empty = [int(c) for c in counts]
    assert nodes == kernels == 6 and copies == 0 and lanes == 2 and events == 2 and waits == 2
    for step in range(3):
        x.fill_(float(step + 1))          # enqueued before the call: the step must see it
        y.fill_(10.0 * step)
        ops.call("trid_step_replay_run", h, ops.stream())
        got = out.clone()                  # enqueued after the call: must see the step's result
        xv = float(step + 1)
        want = (2 * xv + 1) ** 2 + (2 * xv - 3) + 10.0 * step
        assert torch.equal(got, torch.full_like(got, want)), (step, float(got[0]), want)
    torch.cuda.synchronize()
    ops.call("trid_step_replay_destroy", h)

    pinned = torch.arange(1024, dtype=torch.float32).pin_memory()
    dev = torch.zeros(1024, device=gpu)
    g2 = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(g2):
        dev.copy_(pinned, non_blocking=True)
        z = dev + 1.0
    # a recorded host-to-device copy: on this runtime its node usually reads back EMPTY (no pointers, no extent) and the plan is
    # refused rather than guessed; where the runtime does hand the copy's parameters out (seen on one box of the pool), the plan
    # must replay it correctly - either is right, a plan that replays something else is not
    h2 = ctypes.c_void_p()
    try:
        ops.call("trid_step_replay_build", int(g2.raw_cuda_graph()), 8, ctypes.byref(h2))
    except RuntimeError as e:
        assert "cannot be read back" in str(e) or "cannot be replayed" in str(e), str(e)
        assert not h2.value
    else:
        assert h2.value
        pinned.mul_(2.0)
        dev.zero_()
        torch.cuda.synchronize()
        ops.call("trid_step_replay_run", h2, ops.stream())
        torch.cuda.synchronize()
        assert torch.equal(z.cpu(), pinned + 1.0)
        ops.call("trid_step_replay_destroy", h2)


def test_failed_capture_falls_back_to_eager(gpu):
    """A capture that fails (a non-capturable call, a recording invalidated by another thread) must not end the run:
    the runner restores the gradients, disables further attempts and keeps training eagerly."""
    import bench
    from textreid_amd.caption import CaptionBatch
    from textreid_amd.config import moco_cfg
    from textreid_amd.engine.graph import CapturedTrainStep
    from textreid_amd.model import build_model
    from textreid_amd.solver import make_optimizer

    cfg = moco_cfg("m_resnet50", K=64)
    torch.manual_seed(0)
    model = build_model(cfg, vocab_dict=torch.randn(1000, 512) * 0.02).to(gpu).train()
    opt = make_optimizer(cfg, model)
    runner = CapturedTrainStep(model, opt, warmup=1, caption_bound=64)

    def boom(images, cb):
        with torch.cuda.graph(torch.cuda.CUDAGraph(), capture_error_mode="thread_local"):
            torch.zeros(4, device=gpu).sum().item()  # a host read inside a capture: the runtime refuses it

    runner._capture = boom
    # host state an aborted recording leaves behind: collectives staged inside the dead capture
    from textreid_amd.parallel import GradReducer

    runner.reducer = GradReducer()
    runner.reducer._stages.append((None, torch.zeros(1, device=gpu), [], []))
    runner.reducer._pending.append((None, torch.zeros(1, device=gpu), []))
    runner.reducer._staged.add(123)
    out = []
    for s in range(3):
        images, tokens, lengths, ids = bench.synth_batch(4, s, gpu, 3, vocab=1000)
        ld = runner(images, CaptionBatch(tokens, lengths, ids, max_len=64))
        out.append(float(sum(ld.values())))
    torch.cuda.synchronize()
    assert runner.disabled and runner.graph is None and all(np.isfinite(out))
    assert runner.reducer._stages == [] and runner.reducer._pending == [] and runner.reducer._staged == set()
    assert {int(opt.state[p]["step"]) for g_ in opt.param_groups for p in g_["params"]} == {3}  # three real optimizer steps


def test_captured_step_with_a_loose_caption_bound(gpu):
    """The recorded step as engine.trainer.do_train builds it (caption_bound=None: the text encoder's recorded loop runs
    the token tensor's full width, 105 steps, and reads the batch's true maximum from the device - the reference's
    zero-pad-enters-the-max quirk, gru.py:63, depends on it) against the eager step that knows the maximum on the host:
    same parameters (no optimizer), captions of at most 40 tokens with a different maximum per batch.  Losses to fp32
    rounding, every gradient to 1e-4 of its maximum (the text encoder's weight-gradient GEMMs split their (t, b) rows
    differently for 105 and for <= 40 steps: summation order only)."""
    import bench
    from textreid_amd.caption import CaptionBatch
    from textreid_amd.config import moco_cfg
    from textreid_amd.engine.graph import CapturedTrainStep
    from textreid_amd.model import build_model

    B, K = 8, 64
    cfg = moco_cfg("m_resnet50", K=K)
    torch.manual_seed(0)
    model = build_model(cfg, vocab_dict=torch.randn(3000, 512) * 0.02).to(gpu).train()
    state = {k: v.clone() for k, v in model.state_dict().items()}
    gen = torch.Generator().manual_seed(8)

    def batch(s, top):
        im, tk, ln, ids = bench.synth_batch(B, s, gpu, 7, vocab=3000)
        ln = torch.randint(2, top + 1, (B,), generator=gen)
        ln[s % B] = top
        tk = tk.cpu()
        for i, n in enumerate(ln.tolist()):
            tk[i, n:] = 0
        return im, tk.to(gpu), ln.to(gpu), ids

    batches = [batch(0, 33), batch(1, 40), batch(2, 17)]
    runner = CapturedTrainStep(model, None, warmup=1)  # caption_bound=None, as do_train
    got = []
    for im, tk, ln, ids in batches:
        model.load_state_dict(state)  # same parameters, queues and BatchNorm buffers for every batch and both paths
        ld = runner(im, CaptionBatch(tk, ln, ids))
        torch.cuda.synchronize()
        got.append(({k: float(v) for k, v in ld.items()}, {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}))
    assert runner.graph is not None and runner.bound == 105 and runner.calls == 3
    for (im, tk, ln, ids), (gl, gg) in zip(batches, got):
        model.load_state_dict(state)
        ld = runner._eager(im, CaptionBatch(tk, ln, ids))  # host-side maximum: the loop runs max(ln) steps
        torch.cuda.synchronize()
        for k, v in ld.items():
            assert abs(float(v) - gl[k]) <= 1e-5 * abs(float(v)), (k, float(v), gl[k])
        worst = max(float((gg[n] - p.grad).abs().max() / (p.grad.abs().max() + 1e-30)) for n, p in model.named_parameters() if p.grad is not None)
        assert worst <= 1e-4, worst
    assert len(got[0][1]) == 183


def test_bucketed_recordings_for_ragged_captions(gpu):
    """engine.graph.BucketedTrainStep (what do_train builds): captions of 8-64 tokens in a 105-wide token tensor replay on
    recordings of AT MOST 64 recurrence steps - one per caption bucket (32 / 48 / 64), each recorded on the second batch of its
    bucket - instead of one 105-step recording (gru.py:66-82 packs to the batch maximum; build.py:26 pads to 105).  Against the
    eager step that runs exactly max(length) steps: losses to fp32 rounding, every gradient to 1e-4 of its maximum (the bound
    changes the text encoder's launch shapes, i.e. summation order, nothing else)."""
    import bench
    from textreid_amd.caption import CaptionBatch
    from textreid_amd.config import moco_cfg
    from textreid_amd.engine.graph import BucketedTrainStep
    from textreid_amd.model import build_model

    B, K = 8, 64
    cfg = moco_cfg("m_resnet50", K=K)
    torch.manual_seed(0)
    model = build_model(cfg, vocab_dict=torch.randn(3000, 512) * 0.02).to(gpu).train()
    state = {k: v.clone() for k, v in model.state_dict().items()}
    gen = torch.Generator().manual_seed(9)

    def batch(s, top):
        im, tk, ln, ids = bench.synth_batch(B, s, gpu, 7, vocab=3000)
        ln = torch.randint(8, top + 1, (B,), generator=gen)
        ln[s % B] = top
        tk = tk.cpu()
        for i, n in enumerate(ln.tolist()):
            tk[i, n:] = 0
        return im, tk.to(gpu), ln.to(gpu), ids, top

    tops = [20, 40, 60, 25, 45, 62, 30, 47, 64]
    batches = [batch(s, t) for s, t in enumerate(tops)]
    runner = BucketedTrainStep(model, None, warmup=1)
    got = []
    for im, tk, ln, ids, top in batches:
        model.load_state_dict(state)
        ld = runner(im, CaptionBatch(tk, ln, ids, max_len=top))
        torch.cuda.synchronize()
        got.append(({k: float(v) for k, v in ld.items()}, {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}))
    assert runner.recorded == {32: 32, 48: 48, 64: 64}, runner.recorded
    assert all(r.calls == 3 and not r.disabled for r in runner.runners.values())
    eager = runner.runners[64]
    for (im, tk, ln, ids, top), (gl, gg) in zip(batches, got):
        model.load_state_dict(state)
        ld = eager._eager(im, CaptionBatch(tk, ln, ids, max_len=top))
        torch.cuda.synchronize()
        for k, v in ld.items():
            assert abs(float(v) - gl[k]) <= 1e-5 * abs(float(v)), (top, k, float(v), gl[k])
        worst = max(float((gg[n] - p.grad).abs().max() / (p.grad.abs().max() + 1e-30)) for n, p in model.named_parameters() if p.grad is not None)
        assert worst <= 1e-4, (top, worst)
    # a caption beyond the last bucket of a NARROW token tensor takes the tensor's width; an empty bucket list is refused
    assert runner.bucket_of(CaptionBatch(batches[0][1][:, :50], batches[0][2].clamp(max=50), None, max_len=50)) == 50
    with pytest.raises(ValueError):
        BucketedTrainStep(model, None, buckets=())


def test_do_train_captured_matches_eager(gpu):
    """engine.trainer.do_train with its DEFAULT capture=True against capture=False on the same data: seven
    steps - two eager warm-ups, the recording, replays, a RAGGED last batch (other shape: eager fall-back) - with
    captions of at most 40 tokens inside 105-wide token tensors, i.e. the recorded text encoder loops over a bound
    (105) far above every batch's true maximum and reads that maximum from the device.  Losses of every step agree to
    1e-4 (the two paths differ only in summation order: the text encoder's weight-gradient GEMMs split their (t, b) rows
    differently for 105 and for <= 40 steps); queue ids / pointer bit-exact; parameters within Adam's own
    sign-amplified rounding (a gradient entry at rounding level moves its weight by lr whatever its size)."""
    import bench
    from textreid_amd.caption import CaptionBatch
    from textreid_amd.config import moco_cfg
    from textreid_amd.engine.trainer import do_train
    from textreid_amd.model import build_model
    from textreid_amd.solver import make_lr_scheduler, make_optimizer

    B, K, full = 16, 64, 6
    cfg = moco_cfg("m_resnet50", K=K)
    table = torch.randn(3000, 512, generator=torch.Generator().manual_seed(1)) * 0.02
    gen = torch.Generator().manual_seed(4)
    batches = []
    for s in range(full + 1):
        b = B if s < full else B // 2
        im, tk, ln, ids = bench.synth_batch(b, s, "cpu", 6, vocab=3000)
        ln = torch.randint(3, 41, (b,), generator=gen)
        for i, n in enumerate(ln.tolist()):
            tk[i, n:] = 0
        batches.append((im, tk, ln, ids % 11003))
    assert batches[0][1].shape[1] == 105 and max(int(b_[2].max()) for b_ in batches) <= 40

    class Loader:
        def __len__(self):
            return len(batches)

        def __iter__(self):
            for im, tk, ln, ids in batches:
                yield im, CaptionBatch(tk, ln, ids), None  # (no host-side max_len: as the reference's collate hands captions over)

    runs = {}
    for capture in (False, True):
        torch.manual_seed(0)
        model = build_model(cfg, vocab_dict=table).to(gpu)
        opt = make_optimizer(cfg, model)
        sched = make_lr_scheduler(cfg, opt)
        seen = []

        class Meters:
            def update(self, **kw):
                seen.append(kw)

            def __str__(self):
                return ""

        args = {"max_epoch": 1, "epoch": 0, "iteration": 0}
        do_train(model, Loader(), None, opt, sched, None, Meters(), gpu, checkpoint_period=10, evaluate_period=10, arguments=args,
                 log_period=3, capture=capture)
        torch.cuda.synchronize()
        assert args["iteration"] == full + 1
        runs[capture] = ([kw for kw in seen if "loss" in kw], {k: v.detach().clone() for k, v in model.state_dict().items()}, opt.param_groups[0]["lr"])
        del model, opt
    (le, se, lr), (lg, sg, _) = runs[False], runs[True]
    assert len(le) == len(lg) == full + 1
    print("do_train captured vs eager, total loss per step:", [("%.6f" % a["loss"], "%.6f" % b["loss"]) for a, b in zip(le, lg)])
    for i, (a, b) in enumerate(zip(le, lg)):
        # steps 1-2 are eager in both runs: identical bits.  Step 3 is the first REPLAY: it starts from identical parameters,
        # so its losses (the forward of the recorded step, text encoder looping 105 steps over <= 40-token captions) must
        # agree to fp32 rounding.  From step 4 on the two runs differ by what Adam makes of rounding-level gradient
        # differences (an update of lr whatever the gradient's size): a few 1e-4 of the loss on this random-init model.
        tol = 0.0 if i < 2 else (1e-5 if i == 2 else 5e-3)
        for k in a:
            assert abs(a[k] - b[k]) <= tol * abs(a[k]), (i, k, a[k], b[k])
    head = "embed_model."
    assert torch.equal(se[head + "id_queue"], sg[head + "id_queue"]) and torch.equal(se[head + "queue_ptr"], sg[head + "queue_ptr"])
    worst = 0.0
    for k, v in se.items():
        if v.dtype.is_floating_point and "queue" not in k and "running" not in k:
            worst = max(worst, float((v - sg[k]).abs().max()))
    print("do_train captured vs eager: %d steps, largest parameter difference %.2e (lr %.1e)" % (full + 1, worst, lr))
    assert worst <= 2.5 * (full + 1) * 2e-4  # every step may move an entry by its lr (bias lr = 2 x base) in either path


def test_k_reciprocal_rerank_matches_reference_golden(gpu, golden_dir):
    """evaluation.py:40-65,122-124,144-163: Jaccard re-rank matrices and re-ranked CMC/mAP."""
    from textreid_amd.evaluation import k_reciprocal, l2_normalize_rows, rank

    g = load(golden_dir, "rank.npz")
    te, ie = torch.from_numpy(g["te"]).to(gpu), torch.from_numpy(g["ie"]).to(gpu)
    tn, im = l2_normalize_rows(te), l2_normalize_rows(ie)
    rtn = k_reciprocal(im, tn)
    rvn = k_reciprocal(tn, im)
    assert np.allclose(rtn.cpu().numpy(), g["rtn"], atol=1e-7) and np.allclose(rvn.cpu().numpy(), g["rvn"], atol=1e-7)
    sim = torch.from_numpy(g["sim_ti"]).to(gpu)
    tp, ip = torch.from_numpy(g["tp"]).to(gpu), torch.from_numpy(g["ip"]).to(gpu)
    cmc, mAP, idx = rank(k_reciprocal(tn, im, base=sim), tp, ip, (1, 5, 10), get_mAP=True)
    assert np.allclose(cmc.cpu().numpy(), g["re_t2i_cmc"], atol=1e-4) and abs(float(mAP) - float(g["re_t2i_map"])) < 1e-3
    cmc, mAP, _ = rank(k_reciprocal(im, tn, base=sim.t().contiguous()), ip, tp, (1, 5, 10), get_mAP=True)
    assert np.allclose(cmc.cpu().numpy(), g["re_i2t_cmc"], atol=1e-4) and abs(float(mAP) - float(g["re_i2t_map"])) < 1e-3


def test_evaluation_entry_point(gpu, tmp_path):
    """evaluation(dataset, predictions, ...) with duplicate images (get_unique) vs the oracle."""
    from textreid_amd.evaluation import evaluation

    N, C = 60, 32
    img = OF.randn("ev:img", (20, C), 2)
    image_ids = [i // 3 for i in range(N)]  # 3 captions per image
    pids = [i // 6 for i in range(N)]       # 2 images per identity

    class DS:
        def get_id_info(self, idx):
            return image_ids[idx], pids[idx]

    txt = OF.randn("ev:txt", (N, C), 2)
    preds = {i: [img[image_ids[i]].to(gpu), txt[i].to(gpu)] for i in range(N)}
    top1 = evaluation(DS(), preds, str(tmp_path), [1, 5, 10], save_data=True, rerank=True)
    res = evaluation.last_results
    keep = torch.tensor([i * 3 for i in range(20)])
    sim = OE.similarity(txt, img)
    tpid, ipid = torch.tensor(pids), torch.tensor(pids)[keep]
    cmc, mAP, _ = OE.rank(sim, tpid, ipid, (1, 5, 10), True)
    assert np.allclose(res["t2i"][0].cpu().numpy(), cmc.numpy(), atol=1e-4) and abs(float(res["t2i"][1]) - float(mAP)) < 1e-3
    assert abs(float(top1) - float(cmc[0])) < 1e-4
    import torch.nn.functional as F

    rvn = OE.k_reciprocal(F.normalize(txt, dim=1), F.normalize(img, dim=1))
    cmc2, mAP2, _ = OE.rank(rvn + sim, tpid, ipid, (1, 5, 10), True)
    assert np.allclose(res["re-t2i"][0].cpu().numpy(), cmc2.numpy(), atol=1e-4) and abs(float(res["re-t2i"][1]) - float(mAP2)) < 1e-3
    saved = np.load(os.path.join(str(tmp_path), "inference_data.npz"))
    assert set(saved.files) == {"image_pid", "text_pid", "similarity", "rvn_mat", "rtn_mat"}  # evaluation.py:126-143
    assert np.allclose(saved["rvn_mat"], rvn.numpy(), atol=1e-6)
    # `test_net.py --load-result`: predictions=None re-reads the saved arrays and reproduces every metric
    first = {k: (v[0].cpu().numpy().copy(), float(v[1])) for k, v in res.items()}
    evaluation(DS(), None, str(tmp_path), [1, 5, 10], save_data=False, rerank=True)
    for k, (c0, m0) in first.items():
        assert np.allclose(evaluation.last_results[k][0].cpu().numpy(), c0, atol=1e-5) and abs(float(evaluation.last_results[k][1]) - m0) < 1e-4, k
    evaluation(DS(), preds, str(tmp_path), [1, 5, 10], save_data=False, rerank=False)
    cmc3, _ = OE.rank(sim, tpid, ipid, (1, 5, 10), False)
    assert np.allclose(evaluation.last_results["t2i"][0].cpu().numpy(), cmc3.numpy(), atol=1e-4)


def test_inference_dedupes_image_encodes(gpu):
    """engine.inference: each distinct image is encoded once (row f2); predictions equal the
    encode-per-caption path of the reference (inference.py:14-26)."""
    import types

    from textreid_amd.backbones.gru import GRU
    from textreid_amd.backbones.m_resnet import ModifiedResNet
    from textreid_amd.caption import CaptionBatch
    from textreid_amd.embeddings.moco_head.head import MoCoHead
    from textreid_amd.engine.inference import compute_on_dataset

    ns = types.SimpleNamespace
    table = OF.randn("inf:tab", (50, 64), 0, 0.5)
    vis = ModifiedResNet([1, 1, 1, 1], 64, 4, 1, (96, 32), 16)
    txt = GRU(64, 64, 64, 1, 0.0, True, "clip_vit", "./", vocab_dict=table)
    cfg = ns(MODEL=ns(EMBEDDING=ns(FEATURE_SIZE=32, EPSILON=0.1), MOCO=ns(K=32, M=0.9, FC=False), NUM_CLASSES=53))
    model = ns(embed_model=MoCoHead(cfg, vis, txt).to(gpu))
    model.eval = lambda: model.embed_model.eval()
    N = 12
    imgs = OF.randn("inf:img", (4, 3, 96, 32), 0)
    image_ids = [i // 3 for i in range(N)]
    tok = OF.randint("inf:tok", 1, 50, (N, 105), 0)
    ln = OF.randint("inf:len", 2, 20, (N,), 0)

    class DS:
        def get_id_info(self, idx):
            return image_ids[idx], image_ids[idx] // 2

        def __len__(self):
            return N

    class Loader:
        dataset = DS()

        def __iter__(self):
            for s in range(0, N, 4):
                idx = list(range(s, s + 4))
                yield imgs[[image_ids[i] for i in idx]], CaptionBatch(tok[idx], ln[idx]), idx

    a = compute_on_dataset(model, Loader(), gpu, dedupe=True)
    assert compute_on_dataset.last_stats == {"images_encoded": 4, "samples": N, "encoder_passes": 1}  # (coalesced across the 3 loader batches)
    b = compute_on_dataset(model, Loader(), gpu, dedupe=False)
    assert compute_on_dataset.last_stats["images_encoded"] == N
    c = compute_on_dataset(model, Loader(), gpu, dedupe=True, encode_batch=0)  # one encoder pass per loader batch, as the reference
    assert compute_on_dataset.last_stats == {"images_encoded": 4, "samples": N, "encoder_passes": 3}
    d = compute_on_dataset(model, Loader(), gpu, dedupe=True, encode_batch=2)   # flushes in the middle of the run
    assert compute_on_dataset.last_stats["images_encoded"] == 4 and sorted(d) == list(range(N))
    for i in range(N):
        assert torch.allclose(a[i][0], c[i][0], rtol=1e-5, atol=1e-6) and torch.equal(a[i][1], c[i][1])
        assert torch.allclose(a[i][0], d[i][0], rtol=1e-5, atol=1e-6) and torch.equal(a[i][1], d[i][1])
    for i in range(N):
        # eval-mode BatchNorm is per-sample independent: identical up to GEMM tile-edge effects
        assert torch.allclose(a[i][0], b[i][0], rtol=1e-5, atol=1e-6) and torch.equal(a[i][1], b[i][1])


def test_train_step_has_no_host_device_sync(gpu):
    """SURVEY 8 b2 ("no host reads of device scalars"): after warm-up a whole training step - encoders, losses, queue
    push, backward, FusedAdam - must not synchronise the host with the device (torch's sync debug mode raises on any
    blocking copy / .item()).  Round 2 found nine such copies per step in the optimizer's pointer-table upload."""
    import bench
    from textreid_amd.caption import CaptionBatch
    from textreid_amd.config import moco_cfg
    from textreid_amd.model import build_model
    from textreid_amd.solver import make_optimizer

    B = 8
    torch.manual_seed(0)
    cfg = moco_cfg("m_resnet50", K=64)
    model = build_model(cfg, vocab_dict=torch.randn(49408, 512) * 0.02).to(gpu)
    model.train()
    opt = make_optimizer(cfg, model)
    batches = [bench.synth_batch(B, s, gpu, 5) for s in range(2)]

    def step(i):
        images, tokens, lengths, ids = batches[i % 2]
        cb = CaptionBatch(tokens, lengths, ids % 11003, max_len=64)
        loss = sum(model(images, cb).values())
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss

    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        last = step(3)
        last = step(4)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    assert bool(torch.isfinite(last))
    # ... and leaves no garbage that only the cyclic collector would free (an autograd node holding its own output kept
    # ~150 KB of device memory per step alive in round 2): live allocation count is flat with the collector switched off
    import gc

    del last
    gc.collect()
    gc.disable()
    try:
        step(5)
        torch.cuda.synchronize()
        n0 = torch.cuda.memory_stats()["allocation.all.current"]
        for i in range(6, 10):
            step(i)
        torch.cuda.synchronize()
        n1 = torch.cuda.memory_stats()["allocation.all.current"]
    finally:
        gc.enable()
    assert n1 <= n0 + 2, (n0, n1)


def test_do_train_config0_plumbing(gpu, tmp_path):
    """BASELINE configs[0] on the GPU box: moco_gru_cliprn50 at bs128, K=2048, 256 synthetic 384x128 images +
    64-token captions, ONE epoch (2 steps) of engine.trainer.do_train + the per-epoch evaluation through
    engine.inference, as train_net.py drives them (trainer.py:38-139).  Plumbing: losses finite and
    decreasing-ish, queue pointer / ids advanced by exactly two batches, LR schedule stepped, R@1 returned."""
    import bench
    from textreid_amd.caption import CaptionBatch
    from textreid_amd.config import moco_cfg
    from textreid_amd.engine.trainer import do_train
    from textreid_amd.model import build_model
    from textreid_amd.solver import make_lr_scheduler, make_optimizer

    B, K, N = 128, 2048, 256
    torch.manual_seed(0)
    cfg = moco_cfg("m_resnet50", K=K)
    model = build_model(cfg, vocab_dict=torch.randn(49408, 512) * 0.02).to(gpu)
    opt = make_optimizer(cfg, model)
    sched = make_lr_scheduler(cfg, opt)
    batches = [bench.synth_batch(B, s, "cpu", 3) for s in range(N // B)]

    class TrainLoader:
        def __len__(self):
            return len(batches)

        def __iter__(self):
            for im, tk, ln, ids in batches:
                yield im, CaptionBatch(tk, ln, ids, max_len=64), None

    vn = 32
    vim, vtk, vln, vids = bench.synth_batch(vn, 9, "cpu", 4)

    class ValDS:
        def get_id_info(self, idx):
            return idx // 2, int(vids[idx])

        def __len__(self):
            return vn

    class ValLoader:
        dataset = ValDS()

        def __iter__(self):
            for s in range(0, vn, 16):
                idx = list(range(s, s + 16))
                yield vim[[i // 2 * 2 for i in idx]], CaptionBatch(vtk[idx], vln[idx]), idx

    seen = []

    class Meters:
        def update(self, **kw):
            seen.append(kw)

        def __str__(self):
            return str(seen[-1])

    lr0 = opt.param_groups[0]["lr"]
    args = {"max_epoch": 1, "epoch": 0, "iteration": 0}
    do_train(model, TrainLoader(), [ValLoader()], opt, sched, None, Meters(), gpu, checkpoint_period=10, evaluate_period=1,
             arguments=args, log_period=1)
    head = model.embed_model
    steps_seen = [kw for kw in seen if "loss" in kw]
    assert args["iteration"] == 2 and args["epoch"] == 1 and len(steps_seen) == 2
    assert any("top1" in kw for kw in seen)  # the evaluation's R@1 reaches the meters (trainer.py:124)
    assert all(np.isfinite(v) for kw in seen for v in kw.values())
    assert int(head.queue_ptr) == (2 * B) % K
    assert torch.equal(head.id_queue[0, : 2 * B].cpu(), torch.cat([batches[0][3], batches[1][3]]))
    assert opt.param_groups[0]["lr"] != lr0 or sched.last_epoch == 1



def test_device_image_pipeline_matches_oracle(gpu):
    """raw uint8 images of mixed sizes -> fp32 NCHW batch on the device (resize as Pillow, flip, pad + crop,
    ToTensor, Normalize, RandomErasing with value = PIXEL_MEAN) against the oracle chain (lib/data/transforms.py:15-27
    order), BIT-exact: train with augmentation (every sample a different flip / crop / erase), train without, eval."""
    import oracle.transforms as OT
    from textreid_amd.transforms import BatchTransform, sample_params

    H, W, mean, std = 384, 128, [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    rs = np.random.RandomState(5)
    sizes = [(97, 53), (384, 128), (500, 200), (60, 30), (701, 333), (384, 53), (201, 128), (130, 64)]
    imgs = [(rs.rand(h, w, 3) * 255).astype(np.uint8) for h, w in sizes]
    # train + augmentation, explicit parameters covering the corners
    params = np.array([[1, 0, 0, 10, 20, 50, 30, 0], [0, 20, 20, 0, 0, 0, 0, 0], [1, 7, 13, 300, 100, 84, 28, 0], [0, 10, 10, 0, 0, 383, 127, 0],
                       [1, 20, 0, 5, 5, 1, 1, 0], [0, 0, 20, 0, 0, 0, 0, 0], [1, 3, 17, 100, 3, 200, 120, 0], [0, 10, 10, 0, 0, 0, 0, 0]], dtype=np.int32)
    t = BatchTransform(H, W, mean, std, is_train=True, use_aug=True, padding=10, device=gpu)
    out = t(imgs, params=params).cpu().numpy()
    for b, im in enumerate(imgs):
        p = params[b]
        ref = OT.pipeline(im, H, W, mean, std, flip=bool(p[0]), padding=10, crop=(int(p[1]), int(p[2])),
                          erase=(int(p[3]), int(p[4]), int(p[5]), int(p[6])) if p[5] > 0 else None, erase_value=mean)
        assert np.array_equal(out[b], ref), (b, float(np.abs(out[b] - ref).max()))
    # train without augmentation (flip only) and eval (nothing random)
    t2 = BatchTransform(H, W, mean, std, is_train=True, use_aug=False, device=gpu)
    p2 = np.zeros((len(imgs), 8), dtype=np.int32)
    p2[::2, 0] = 1
    out2 = t2(imgs, params=p2).cpu().numpy()
    t3 = BatchTransform(H, W, mean, std, is_train=False, device=gpu)
    out3 = t3(imgs).cpu().numpy()
    for b, im in enumerate(imgs):
        assert np.array_equal(out2[b], OT.pipeline(im, H, W, mean, std, flip=bool(p2[b, 0])))
        assert np.array_equal(out3[b], OT.pipeline(im, H, W, mean, std))
    # sampled parameters are valid rectangles / offsets
    sp = sample_params(256, H, W, 10, True, np.random.default_rng(0))
    assert sp[:, 1:3].min() >= 0 and sp[:, 1:3].max() <= 20
    assert bool(((sp[:, 3] + sp[:, 5] <= H) & (sp[:, 4] + sp[:, 6] <= W)).all()) and 0.3 < (sp[:, 5] > 0).mean() < 0.7
