"""Worker for tests/test_dp_gpu.py: W ranks (gloo, possibly sharing one GPU) run the HIP
train step on their shard; rank 0 checks losses and reduced gradients against the CPU oracle's
single-process global-batch computation with per-shard BatchNorm."""
import os
import sys
import types

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle.fill as OF  # noqa: E402
import oracle.head as OH  # noqa: E402
import oracle.visual as OV  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def main():
    dist.init_process_group(os.environ.get("TRID_DIST_BACKEND", "gloo"), init_method="env://")
    W, r = dist.get_world_size(), dist.get_rank()
    dev = torch.device("cuda", r % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    from textreid_amd.backbones.gru import GRU
    from textreid_amd.backbones.m_resnet import ModifiedResNet
    from textreid_amd.caption import CaptionBatch
    from textreid_amd.embeddings.moco_head.head import MoCoHead
    from textreid_amd.parallel import GradReducer

    spec, hidden, embed, vocab, C, K, NC, Bl, seed = OV.TINY, 64, 64, 200, 32, 32, 53, 4, 9
    ns = types.SimpleNamespace
    table = OF.randn("vocab_table_dp", (vocab, embed), seed, 0.5)
    vis = ModifiedResNet(list(spec.layers), spec.output_dim, spec.heads, spec.last_stride, (spec.height, spec.in_width), spec.width)
    txt = GRU(hidden, embed, embed, 1, 0.0, True, "clip_vit", "./", vocab_dict=table)
    fc = os.environ.get("TRID_TEST_FC", "0") == "1"  # MODEL.MOCO.FC: projection heads (six gathered blocks)
    cfg = ns(MODEL=ns(EMBEDDING=ns(FEATURE_SIZE=C, EPSILON=0.1), MOCO=ns(K=K, M=0.9, FC=fc), NUM_CLASSES=NC))
    head = MoCoHead(cfg, vis, txt)
    filled = OF.fill_state(head.state_dict(), seed, "dp.")
    st = {k: v.clone() for k, v in filled.items()}
    OH.init_queues(st, seed)
    for k in ("t_queue", "v_queue", "id_queue", "queue_ptr"):
        filled[k] = st[k].clone()
    head.load_state_dict(filled)
    head.to(dev).train()
    Bg = Bl * W
    x = OF.randn("img:dp", (Bg, 3, spec.height, spec.in_width), seed)
    tok = OF.randint("tok:dp", 1, vocab, (Bg, 105), seed)
    ln = OF.randint("len:dp", 3, 30, (Bg,), seed)
    for i, n in enumerate(ln.tolist()):
        tok[i, n:] = 0
    ids = torch.arange(Bg) // 2
    sl = slice(r * Bl, (r + 1) * Bl)
    red = GradReducer(bucket_mb=1)
    if os.environ.get("TRID_DP_OVERLAP", "1") != "0":  # staged all-reduce from inside the encoder's backward
        head.v_encoder_q.grad_sync = red
    ld = head(x[sl].to(dev), CaptionBatch(tok[sl].to(dev), ln[sl].to(dev), ids[sl].to(dev)))
    sum(ld.values()).backward()
    pre = [p for n, p in head.named_parameters() if p.requires_grad and "loss_evaluator" not in n][::-1]
    red.reduce(pre)
    red.wait()
    torch.cuda.synchronize()
    if r == 0:
        for k in OH.trainable_names(st):
            st[k].requires_grad_(True)
        # oracle: per-shard encoders (own BN statistics), global losses
        vq, tq, vk, tk, vs, ts = [], [], [], [], [], []
        with torch.no_grad():
            OH.momentum_update(st, 0.9)
        for w in range(W):
            s2 = slice(w * Bl, (w + 1) * Bl)
            vf, tf = OH.encode(st, "q", spec, table, x[s2], tok[s2], ln[s2], True)
            ve, te = OH.embed_pair(st, vf, tf)
            vq.append(ve)
            tq.append(te)
            if fc:  # head.py:117-124: the contrastive branch goes through the projection heads
                a, b = OH.fc_pair(st, "q", vf, tf)
                vs.append(a)
                ts.append(b)
            with torch.no_grad():
                vkf, tkf = OH.encode(st, "k", spec, table, x[s2], tok[s2], ln[s2], True)
                a, b = OH.fc_pair(st, "k", vkf, tkf) if fc else OH.embed_pair(st, vkf, tkf)
                vk.append(F.normalize(a, dim=1))
                tk.append(F.normalize(b, dim=1))
        v_embed, t_embed = torch.cat(vq), torch.cat(tq)
        v_src, t_src = (torch.cat(vs), torch.cat(ts)) if fc else (v_embed, t_embed)
        old = OH.losses_from_embeddings(st, v_embed, t_embed, F.normalize(v_src, dim=1), F.normalize(t_src, dim=1),
                                        torch.cat(vk), torch.cat(tk), ids, 0.1)
        sum(old.values()).backward()
        rel = lambda a, b: float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max() / (b.detach().double().abs().max() + 1e-30))
        errs = {k: rel(ld[k], old[k]) for k in old}
        named = dict(head.named_parameters())
        for k in (("v_fc_q.0.weight", "t_fc_q.2.bias") if fc else ()) + ("v_embed_layer.weight", "t_embed_layer.weight", "loss_evaluator.projection", "t_encoder_q.gru.weight_ih_l0",
                  "v_encoder_q.attnpool.c_proj.weight", "v_encoder_q.layer4.0.conv3.weight",
                  "v_encoder_q.layer3.0.conv2.weight", "v_encoder_q.layer1.0.bn2.bias", "v_encoder_q.conv1.weight"):
            errs["grad:" + k] = rel(named[k].grad, st[k].grad)
        OH.enqueue(st, torch.cat(vk), torch.cat(tk), ids)
        sd = head.state_dict()
        errs["v_queue"] = rel(sd["v_queue"], st["v_queue"])
        assert int(sd["queue_ptr"]) == int(st["queue_ptr"]) and torch.equal(sd["id_queue"].cpu(), st["id_queue"])
        print("DP_ERRS", {k: "%.1e" % v for k, v in errs.items()})
        bad = {k: v for k, v in errs.items() if not v < 1e-3}
        assert not bad, bad
        print("DP_OK")
    # ---- replicated state must be IDENTICAL on every rank after the step: queues, ids, pointer (every rank pushed the
    # same gathered keys) and the unreduced post-gather parameter's gradient
    sdr = head.state_dict()
    for name, ten in (("v_queue", sdr["v_queue"]), ("t_queue", sdr["t_queue"]), ("id_queue", sdr["id_queue"].double()),
                      ("queue_ptr", sdr["queue_ptr"].double()), ("projection.grad", head.loss_evaluator.projection.grad)):
        mine = ten.detach().double().contiguous()
        mine = mine if dist.get_backend() == "nccl" else mine.cpu()  # RCCL moves device buffers, gloo host buffers
        parts = [torch.empty_like(mine) for _ in range(W)]
        dist.all_gather(parts, mine)
        for w in range(W):
            assert torch.equal(parts[w], parts[0]), "rank %d: %s differs between ranks 0 and %d" % (r, name, w)
    if r == 0:
        print("DP_REPLICAS_IDENTICAL")
        print("DP_TRANSPORT backend=%s world=%d staged_bytes=%d post_bytes=%d" % (dist.get_backend(), W, red.bytes_staged, red.bytes_post))
    # ---- engine.trainer.do_train under data parallelism: ranks whose models were drawn from DIFFERENT seeds are made replicas by
    # the initial broadcast (DDP's broadcast at wrap, train_net.py:50-56), train on their shards of the same global batches and
    # end bit-identical; the periodic replica digest (log_period = 1 here) passes on every step
    if os.environ.get("TRID_DP_TRAINER", "0") == "1":
        import hashlib

        from textreid_amd.engine.trainer import do_train
        from textreid_amd.solver import FusedAdam

        ld = None  # (the first step's autograd graph must be gone before a step is RECORDED: a stale AccumulateGrad node drags ITS
        torch.cuda.synchronize()  # stream into the recording, which then cannot be ended - tools/exp/dpc_debug.sh)
        torch.manual_seed(1000 + r)
        head.load_state_dict(filled)
        for p_ in head.parameters():  # every rank its own weights (rank 0 keeps the filled ones) ...
            p_.data.add_(0.02 * r * torch.randn_like(p_) * p_.abs().mean())
        head.v_queue.copy_(F.normalize(torch.rand_like(head.v_queue), dim=0))  # ... its own queues, ids and pointer
        head.id_queue.fill_(7 + r)
        head.queue_ptr.fill_(Bl * r)
        head.v_encoder_q.grad_sync = None
        for p_ in head.parameters():
            p_.grad = None

        class Wrapper(torch.nn.Module):  # (textreid_amd.model.Model's surface: `embed_model` + forward)
            def __init__(self, h):
                super().__init__()
                self.embed_model = h

            def forward(self, images, captions):
                return self.embed_model(images, captions)

        wrapper = Wrapper(head)
        opt = FusedAdam([{"params": [p_], "lr": 1e-3} for p_ in head.parameters() if p_.requires_grad], lr=1e-3)

        n_it = int(os.environ.get("TRID_DP_TRAINER_ITERS", "3"))  # (tools/dp_soak.py: a longer run of the same loop)
        if r == 0 and os.environ.get("TRID_DP_TRAINER_CAPTURE", "0") == "1":  # the recording's own report ("train step captured: N segments ...")
            import logging

            h_ = logging.StreamHandler(sys.stdout)
            h_.setFormatter(logging.Formatter("DP_TRAINER_LOG %(message)s"))
            h_.addFilter(lambda rec: rec.getMessage().startswith("train step"))
            logging.getLogger("PersonSearch.trainer").addHandler(h_)
            logging.getLogger("PersonSearch.trainer").setLevel(logging.INFO)

        class Loader:
            dataset = list(range(n_it))

            def __len__(self):
                return n_it

            def __iter__(self):
                for i in range(n_it):
                    yield x[sl].roll(i, 0), CaptionBatch(tok[sl].roll(i, 0), ln[sl].roll(i, 0), (ids[sl] + i) % NC), None

        class Sched:
            def step(self):
                pass

        # (TRID_DP_TRAINER_CAPTURE=1: do_train's default - two eager steps, then the step recorded and replayed in segments)
        do_train(wrapper, Loader(), None, opt, Sched(), None, None, dev, checkpoint_period=10, evaluate_period=10,
                 arguments={"max_epoch": 1, "epoch": 0, "iteration": 0}, log_period=1, capture=os.environ.get("TRID_DP_TRAINER_CAPTURE", "0") == "1")
        torch.cuda.synchronize()
        sha = hashlib.sha256()
        for k_, v_ in head.state_dict().items():
            if "running" in k_:
                continue  # (BatchNorm statistics are rank-local by design, as `broadcast_buffers=False`)
            sha.update(v_.detach().contiguous().cpu().numpy().tobytes())
        allh = [None] * W
        dist.all_gather_object(allh, sha.hexdigest())
        assert all(h == allh[0] for h in allh), "ranks that started from different seeds did not end as replicas"
        assert int(head.queue_ptr) == (n_it * Bg) % K
        if r == 0:
            print("DP_TRAINER_REPLICAS_IDENTICAL iterations=%d" % n_it)
    # ---- the data-parallel step RECORDED (engine.graph.CapturedTrainStep with the run's GradReducer) and replayed in SEGMENTS: the
    # packed all-gather of the forward, the all-reduces staged from inside backward and the bucketed ones after it are cut points of
    # the recording - the library re-issues the recorded kernels between them, torch.distributed runs each collective on the stream
    # its marker was recorded on (RCCL's own launch path; gloo stages through the host as in the eager step) - bit for bit the
    # eager data-parallel step, fused Adam included
    if os.environ.get("TRID_DP_CAPTURED", "0") == "1":
        from textreid_amd.engine.graph import CapturedTrainStep
        from textreid_amd.solver import FusedAdam

        runs = {}
        ld = None  # (the first step's autograd graph must be gone: a stale AccumulateGrad node drags ITS stream into the recording)
        torch.cuda.synchronize()
        for mode in ("eager", "graph"):
            head.load_state_dict(filled)
            for p_ in head.parameters():
                p_.grad = None
            red2 = GradReducer(bucket_mb=1)
            head.v_encoder_q.grad_sync = red2
            opt = FusedAdam([{"params": [p_], "lr": 2e-3 if n.endswith("bias") else 1e-3} for n, p_ in head.named_parameters() if p_.requires_grad], lr=1e-3)
            # (caption_bound = the batches' own maximum: the recorded text encoder then runs the SAME launch shapes as the eager
            # one - with the token tensor's width as bound its [t x b, E] products change size, hence kernel and rounding)
            runner = CapturedTrainStep(head, opt, warmup=2, caption_bound=int(ln[sl].max()), reducer=red2, pre_gather=pre)
            losses = []
            for i in range(5):
                xi = x[sl].roll(i, 0).to(dev)
                cb = CaptionBatch(tok[sl].roll(i, 0).to(dev), ln[sl].roll(i, 0).to(dev), ((ids[sl] + i) % NC).to(dev))
                out = runner._eager(xi, cb) if mode == "eager" else runner(xi, cb)
                losses.append(torch.stack([v.detach().clone() for v in out.values()]))
            torch.cuda.synchronize()
            if mode == "graph":
                assert runner.graph is not None and not runner.disabled, "the data-parallel step was not recorded"
                assert red2.bytes_staged > 0 and red2.bytes_post > 0
                assert len(runner.cuts) >= 3 and runner.replayer is not None, "the collectives are cut points of the stream plan"
                assert [c.kind for c in runner.cuts].count("all_gather") == 1
                n_cuts = len(runner.cuts)
            runs[mode] = (torch.stack(losses), {k: v.detach().clone() for k, v in head.state_dict().items()})
            del runner, opt
        assert torch.equal(runs["eager"][0], runs["graph"][0]), (runs["eager"][0] - runs["graph"][0]).abs().max()
        for k, v in runs["eager"][1].items():
            assert torch.equal(v, runs["graph"][1][k]), k
        if r == 0:
            print("DP_CAPTURED_OK backend=%s world=%d segments=%d" % (dist.get_backend(), W, n_cuts + 1))
    # ---- sharded retrieval: every rank scores its own (unevenly sized) gallery shard, lists are merged
    from textreid_amd.evaluation import similarity_topk

    sizes = [700 + 300 * w for w in range(W)]
    qe = OF.randn("dp:q", (33, 64), seed)
    ge = OF.randn("dp:g", (sum(sizes), 64), seed)
    lo = sum(sizes[:r])
    vals, idx = similarity_topk(qe.to(dev), ge[lo:lo + sizes[r]].to(dev), 10)
    torch.cuda.synchronize()
    sim = F.normalize(qe, dim=1) @ F.normalize(ge, dim=1).t()
    rv, ri = torch.topk(sim, 10, dim=1)
    assert torch.equal(idx.cpu(), ri), "sharded top-k indices differ from the dense top-k"
    assert torch.allclose(vals.cpu(), rv, atol=2e-6)
    if r == 0:
        print("DP_RETRIEVAL_OK")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
