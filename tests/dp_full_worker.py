"""Worker for tests/test_dp_gpu.py's FULL-SIZE cases: BASELINE configs[2] (CLIP-RN50 + BiGRU, 4 ranks x 128 = global
batch 512, queue 8192, fp32-class) and configs[3] (CLIP-RN101 + BiGRU, 8 ranks x 128 = 1024, queue 65536, bf16 mode)
in their N-rank form.  The ranks share whatever GPUs are visible (one on the build pool: `gloo` transport; one rank
per GPU over `nccl` = RCCL when the box has enough of them); every rank runs the product's data-parallel train step
(`textreid_amd.parallel`: packed all-gather of the embeddings, global losses on every rank, SUM all-reduce of the
pre-gather gradients staged from inside backward) on its 128-row shard.  Checked, SURVEY 8e:

  1. replicas: queues, ids, pointer and the un-reduced post-gather gradient (`projection.grad`) are bit-identical on
     every rank after the step (sha256 of the bytes);
  2. the three GLOBAL losses against the CPU oracle's `losses_from_embeddings` (reference `head.py:148-170`,
     `losses.py:42-62,102-128,206-217`) evaluated on the gathered [B_global, 256] blocks - the 512 x 8192 and
     1024 x 65536 contrastive matrices as the reference would form them;
  3. rank 0's REDUCED gradients against a single-process evaluation of the same global batch: the shards run through
     the encoders one after another (own BatchNorm statistics each, as `broadcast_buffers=False`,
     `train_net.py:54-55`), the embeddings are concatenated and the losses taken once - no collective anywhere;
  4. the POST-GATHER gradients against the fp64 CPU ORACLE (not against the HIP path itself): dL/d(v_embed), dL/d(t_embed) of
     the whole 512- / 1024-row global batch and `projection.grad`, by autograd through `oracle.head.losses_from_embeddings`
     on the gathered blocks (normalisation included) - the arithmetic of the global contrastive / instance / alignment
     losses at the size the reference would form them, pinned independently of the kernels under test.
"""
import hashlib
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle.head as OH  # noqa: E402
import oracle.visual as OV  # noqa: E402


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _digest(t):
    return hashlib.sha256(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()


def main():
    backend = os.environ.get("TRID_DIST_BACKEND", "gloo")
    dist.init_process_group(backend, init_method="env://")
    W, r = dist.get_world_size(), dist.get_rank()
    ndev = torch.cuda.device_count()
    dev = torch.device("cuda", r % ndev)
    torch.cuda.set_device(dev)
    from oracle.cases import full_step_case
    from textreid_amd import losses as L
    from textreid_amd import ops
    from textreid_amd.caption import CaptionBatch
    from textreid_amd.config import moco_cfg
    from textreid_amd.model import build_model
    from textreid_amd.parallel import GradReducer

    arch = os.environ.get("TRID_DP_ARCH", "m_resnet50")
    K = int(os.environ.get("TRID_DP_K", "8192"))
    Bl = int(os.environ.get("TRID_DP_BLOCAL", "128"))
    seed = int(os.environ.get("TRID_DP_SEED", "30"))
    spec = {"m_resnet50": OV.RN50, "m_resnet101": OV.RN101}[arch]
    Bg = Bl * W
    t0 = time.time()
    st, table, images, tokens, lengths, ids = full_step_case(spec, Bg, K, 3000, seed)
    model = build_model(moco_cfg(arch, K=K), vocab_dict=table)
    head = model.embed_model
    head.load_state_dict({k: v.clone() for k, v in st.items()})
    model.to(dev).train()
    sl = slice(r * Bl, (r + 1) * Bl)
    red = GradReducer()
    head.v_encoder_q.grad_sync = red  # as engine.trainer.do_train arranges under data parallelism
    seen = {}
    fused = head.loss_evaluator.forward_fused

    def keep(x):  # a copy made by a KERNEL (x * 1): a contiguous clone() is a hipMemcpyAsync, and the recorded step must not hold
        return torch.mul(x.detach(), 1)  # copy nodes (csrc/step_replay.hip cannot read them back on this runtime)

    def spy(*a):  # the gathered blocks the losses are evaluated on, and the gradients that come back to them
        seen["args"] = [keep(x) for x in a[:7]]
        a[0].register_hook(lambda g: seen.__setitem__("d_v_embed", keep(g)))
        a[1].register_hook(lambda g: seen.__setitem__("d_t_embed", keep(g)))
        return fused(*a)

    head.loss_evaluator.forward_fused = spy
    pre = [p for n, p in model.named_parameters() if p.requires_grad and "loss_evaluator" not in n][::-1]
    x_l, cb_l = images[sl].to(dev), CaptionBatch(tokens[sl].to(dev), lengths[sl].to(dev), ids[sl].to(dev))
    runner_mode = os.environ.get("TRID_DP_RUNNER", "1") == "1"
    segments = 0
    if runner_mode:
        # the step on the fast host path (engine.graph.CapturedTrainStep under data parallelism: recorded once, replayed in SEGMENTS
        # around its collectives): two eager calls, then the recording and its first replay - each from the SAME initial state
        # (restored in place), so that what is compared below is the REPLAYED step of that state.  The spies' clones are nodes of
        # the recording: after the replay they hold the replayed step's gathered blocks and gradients.
        from textreid_amd.engine.graph import CapturedTrainStep

        runner = CapturedTrainStep(model, None, warmup=2, caption_bound=int(lengths[sl].max()), reducer=red, pre_gather=pre)
        for i in range(3):
            head.load_state_dict({k: v.clone() for k, v in st.items()})
            ops.note_parameter_write()
            seen.clear()
            ld = runner(x_l, cb_l)
        torch.cuda.synchronize()
        assert runner.graph is not None and not runner.disabled and len(runner.cuts) >= 3, "the data-parallel step was not recorded"
        segments = len(runner.cuts) + 1
    else:
        ld = model(x_l, cb_l)
        sum(ld.values()).backward()
        red.reduce(pre)
        red.wait()
    torch.cuda.synchronize()
    head.loss_evaluator.forward_fused = fused
    t_step = time.time() - t0

    # ---- 1. replicated state identical on every rank
    sd = head.state_dict()
    mine = [_digest(sd["v_queue"]), _digest(sd["t_queue"]), _digest(sd["id_queue"]), _digest(sd["queue_ptr"]),
            _digest(head.loss_evaluator.projection.grad)] + [_digest(x) for x in seen["args"]]
    allr = [None] * W
    dist.all_gather_object(allr, mine)
    for w in range(W):
        assert allr[w] == allr[0], "rank %d: replicated state / gathered blocks differ between ranks 0 and %d" % (r, w)
    assert int(sd["queue_ptr"]) == Bg % K and torch.equal(sd["id_queue"][0, :Bg].cpu(), ids)

    named = dict(head.named_parameters())
    probe = ["v_encoder_q.conv1.weight", "v_encoder_q.layer1.0.conv3.weight", "v_encoder_q.layer3.1.conv2.weight",
             "v_encoder_q.layer4.2.bn3.weight", "v_encoder_q.attnpool.c_proj.weight", "t_encoder_q.gru.weight_hh_l0",
             "v_embed_layer.weight", "t_embed_layer.bias", "loss_evaluator.projection"]
    dp_grads = {k: named[k].grad.detach().clone() for k in probe} if r == 0 else None
    dp_losses = {k: v.detach().clone() for k, v in ld.items()}
    seen = {k: ([x.clone() for x in v] if isinstance(v, list) else v.clone()) for k, v in seen.items()}  # (out of the recording's pool)
    # every rank but 0 is done with its activations / gradients: hand the memory back before rank 0's single-process pass
    del ld
    if runner_mode:
        del runner
    for p in model.parameters():
        p.grad = None
    if r != 0:
        del model, head, named, sd
    torch.cuda.empty_cache()
    dist.barrier()

    if r == 0:
        print("DPFULL_REPLICAS_IDENTICAL world=%d backend=%s devices=%d B_global=%d K=%d arch=%s conv_precision=%d launch=%s (%.0f s to the end of the step)"
              % (W, backend, ndev, Bg, K, arch, ops.conv_precision(), ("segmented_replay:%d" % segments) if runner_mode else "eager", t_step))
        # ---- 2. global losses vs the CPU oracle on the gathered blocks
        v_embed, t_embed, v_q, t_q, v_k, t_k, gid = [x.cpu() for x in seen["args"]]
        assert torch.equal(gid, ids) and v_embed.shape == (Bg, 256)
        ost = {k: (st[k].double() if st[k].dtype.is_floating_point else st[k]) for k in ("loss_evaluator.projection", "t_queue", "v_queue", "id_queue")}
        old = OH.losses_from_embeddings(ost, v_embed.double(), t_embed.double(), v_q.double(), t_q.double(), v_k.double(), t_k.double(), gid, 0.1)
        errs = {"loss:" + k: _rel(dp_losses[k], old[k]) for k in old}
        # ---- 4. post-gather gradients vs the ORACLE: autograd through the fp64 head on the gathered blocks
        import torch.nn.functional as F_

        ove, ote = v_embed.double().requires_grad_(True), t_embed.double().requires_grad_(True)
        opr = st["loss_evaluator.projection"].double().clone().requires_grad_(True)
        ost2 = dict(ost)
        ost2["loss_evaluator.projection"] = opr
        og = OH.losses_from_embeddings(ost2, ove, ote, F_.normalize(ove, dim=1), F_.normalize(ote, dim=1), v_k.double(), t_k.double(), gid, 0.1)
        sum(og.values()).backward()
        # (the gather's backward hands every rank the gradient of the WHOLE global batch; rank 0's copy is checked)
        errs["oracle_grad:v_embed"] = _rel(seen["d_v_embed"], ove.grad)
        errs["oracle_grad:t_embed"] = _rel(seen["d_t_embed"], ote.grad)
        errs["oracle_grad:projection"] = _rel(dp_grads["loss_evaluator.projection"], opr.grad)
        # ---- 3. reduced gradients vs the single-process evaluation of the global batch (per-shard BatchNorm)
        head.load_state_dict({k: v.clone() for k, v in st.items()})
        head.v_encoder_q.grad_sync = None
        with torch.no_grad():
            head._momentum_update_key_encoder()
        vf, tf, vkf, tkf = [], [], [], []
        for w in range(W):
            s2 = slice(w * Bl, (w + 1) * Bl)
            cb = CaptionBatch(tokens[s2].to(dev), lengths[s2].to(dev), ids[s2].to(dev))
            x = images[s2].to(dev)
            vf.append(head.v_encoder_q(x))
            tf.append(head.t_encoder_q(cb))
            with torch.no_grad():
                vkf.append(head.v_encoder_k(x))
                tkf.append(head.t_encoder_k(cb))
        vw, vb, tw, tb = head.v_embed_layer.weight, head.v_embed_layer.bias, head.t_embed_layer.weight, head.t_embed_layer.bias
        # (the embedding layers per shard, as every rank applies them BEFORE the gather: the same launch shapes - a 128-row
        # product runs on another kernel than a 1024-row one, and in the bf16 mode a last-bit difference in dL/d(feature)
        # flips bf16 roundings further down)
        ve, te = torch.cat([L.linear(f_, vw, vb) for f_ in vf]), torch.cat([L.linear(f_, tw, tb) for f_ in tf])
        with torch.no_grad():
            vk = L.l2_normalize(torch.cat([L.linear(f_, vw, vb) for f_ in vkf]))
            tk = L.l2_normalize(torch.cat([L.linear(f_, tw, tb) for f_ in tkf]))
        one = head.loss_evaluator.forward_fused(ve, te, L.l2_normalize(ve), L.l2_normalize(te), vk, tk, ids.to(dev),
                                                head._queue_kc("t_queue"), head._queue_kc("v_queue"), head.id_queue)
        sum(one.values()).backward()
        torch.cuda.synchronize()
        for k in one:
            errs["loss_vs_single_process:" + k] = _rel(dp_losses[k], one[k])
        gmax = max(float(named[k].grad.abs().max()) for k in probe)
        for k in probe:
            ref = named[k].grad
            errs["grad:" + k] = float((dp_grads[k].double() - ref.double()).abs().max() / max(float(ref.abs().max()), 1e-3 * gmax))
        print("DPFULL_ERRS", {k: "%.1e" % v for k, v in errs.items()})
        tol_loss = 1e-3
        bad = {k: v for k, v in errs.items() if not v < (tol_loss if k.startswith("loss") else 1e-3)}
        assert not bad, bad
        print("DPFULL_OK peak_mem_GB=%.1f" % (torch.cuda.max_memory_allocated() / 2 ** 30))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
