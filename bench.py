#!/usr/bin/env python3
"""Headline benchmark: image-text pairs/s of the MoCo train step (CLIP-RN50 +
BiGRU, bs128/GPU, 384x128 images, 64-token captions, 8192-slot queue, fp32).

  python bench.py --gpus N --steps K --warmup W

One process per GPU (torchrun sets RANK/LOCAL_RANK/WORLD_SIZE); rank 0 prints ONE
JSON line.  A step = model(images, captions) -> sum of the three losses ->
zero_grad -> backward -> (gradient SUM all-reduce over RCCL when N>1) -> fused
Adam step, on synthetic inputs already resident in HBM.
`roofline` is measured live with events around the launches of the dominant
kernel (the 3x3 implicit-GEMM convolution) during the timed steps;
`cpu_baseline` times the CPU oracle (a port of the reference step) on a bounded
sample of the same workload on rank 0's host cores.
"""

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak
F16_SPLIT_PEAK_TFLOPS = 2500.0 / 3  # fp32-class two-plane fp16 arithmetic: 3 dense fp16 MFMA products per multiply-add
VISUAL_FWD_GFLOP = {"m_resnet50": 17.24, "m_resnet101": 24.51}  # algorithmic forward work per image (convs + token-0 attention pool)


def log(msg):
    if os.environ.get("RANK", "0") == "0":
        print("[bench %.1fs] %s" % (time.time() - T_START, msg), file=sys.stderr, flush=True)


T_START = time.time()


def synth_batch(B, step, device, seed, vocab=49408, Lpad=105, L=64):
    L = int(os.environ.get("TRID_BENCH_CAPTION_LEN", L))  # (experiments only: the marginal cost of the recurrence; the metric is quoted on 64)
    g = torch.Generator(device="cpu").manual_seed(seed + 7919 * step)
    images = torch.randn(B, 3, 384, 128, generator=g)
    tokens = torch.zeros(B, Lpad, dtype=torch.int64)
    tokens[:, :L] = torch.randint(1, vocab, (B, L), generator=g)
    lengths = torch.full((B,), L, dtype=torch.int64)
    ids = torch.arange(B, dtype=torch.int64) // 4 + step * (B // 4)
    return images.to(device), tokens.to(device), lengths.to(device), ids.to(device)


def cpu_baseline(sample_b=128, steps=3, warm_b=32):
    """Oracle (port of the reference train step incl. Adam) on the host cores: configs[1]'s own batch
    (B = 128), `steps` timed steps after one warm-up step at B = `warm_b` (thread pool,
    allocator); the warm-up size is also timed for one more step and reported as a second point.

    The FIRST timed step doubles as the parity reference of the bench line: it starts from the `margin`-style state and
    batch of tests/test_model_gpu.py::test_config1_b128_k8192_step_vs_oracle (oracle/cases.py), and its losses, all 183
    gradients and the post-step state are returned under "_parity_ref" for `parity_vs_oracle()` below."""
    import oracle.head as OH
    import oracle.visual as OV
    from oracle.cases import full_step_case, synth_batch as cpu_batch

    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    ncpu = min(ncpu, int(os.environ.get("TRID_CPU_THREADS", "32")))  # more threads only adds sync overhead at this size
    torch.set_num_threads(ncpu)
    log("cpu_baseline: %d usable cores (os.cpu_count=%s)" % (ncpu, os.cpu_count()))
    spec, K, vocab, seed = OV.RN50, 8192, 3000, PARITY_SEED
    st, table, images0, tokens0, lengths0, ids0 = full_step_case(spec, sample_b, K, vocab, seed)
    st0 = {k: v.clone() for k, v in st.items()}
    names = OH.trainable_names(st)

    def fwd_bwd(state, bsz, s, first=False):
        images, tokens, lengths, ids = (images0, tokens0, lengths0, ids0) if first else cpu_batch(bsz, s, 1234, vocab=vocab)
        taps = {} if first else None
        ld = OH.train_forward(state, spec, table, images, tokens, lengths, ids, m=0.999, epsilon=0.1, taps=taps)
        for k in names:
            state[k].grad = None
        sum(ld.values()).backward()
        return ld, taps

    # warm-up (thread pool, allocator) on a throw-away copy of the state: the parity step must start from st0 itself
    t0 = time.time()
    sw = {k: v.clone() for k, v in st0.items()}
    for k in names:
        sw[k].requires_grad_(True)
    fwd_bwd(sw, warm_b, 0)
    del sw
    log("cpu_baseline warm-up step B=%d: %.1fs" % (warm_b, time.time() - t0))
    groups = []
    for k in names:
        st[k].requires_grad_(True)
        groups.append({"params": [st[k]], "lr": 2e-4 if "bias" in k else 1e-4, "weight_decay": 0.0 if "bias" in k else 4e-5})
    opt = torch.optim.Adam(groups, lr=1e-4)
    ref = None

    def one(bsz, s, first=False):
        nonlocal ref
        t0 = time.time()
        ld, taps = fwd_bwd(st, bsz, s, first)
        t1 = time.time()
        if first:  # (copies, not timed)
            ref = ({k: v.detach().clone() for k, v in ld.items()}, {k: st[k].grad.clone() for k in names},
                   {k: v.detach().clone() for k, v in st.items()}, float(taps["visual_q"]["relu_min"]))
        t2 = time.time()
        opt.step()
        dt = (t1 - t0) + (time.time() - t2)
        log("cpu_baseline step B=%d: %.1fs" % (bsz, dt))
        return dt

    times = [one(sample_b, 2 + s, first=(s == 0)) for s in range(steps)]
    small = one(warm_b, 1)
    dt = sum(times) / len(times)
    return {
        "value": sample_b / dt,
        "unit": "pairs/s",
        "cores": torch.get_num_threads(),
        "kind": "port",
        "sample": "%d timed train steps (fwd q+k, bwd, Adam) at B=%d (configs[1]'s batch), K=8192, fp32, after 1 warm-up step at B=%d; CPU oracle = port of the reference step" % (steps, sample_b, warm_b),
        "value_at_B%d" % warm_b: warm_b / small,
        "_parity_ref": (st0, table, (images0, tokens0, lengths0, ids0), ref),
    }


PARITY_SEED = 100  # tools/pick_fullstep_seed.py rn50 128 8192 100 (oracle/cases.py: RELU_MIN_BY_BATCH)


def replay_equals_eager(runner, model, opt, images, cb):
    """The seam between the TIMED path and the oracle-checked path, closed at the benchmarked size: from one saved state
    (every parameter and buffer of the model - queues, queue pointer, BatchNorm statistics -, every Adam moment and step
    count) the SAME batch goes once through the launch form the timed region uses (the recorded step re-issued by
    csrc/step_replay.hip, optimizer included) and once through the eager step (the form `parity_vs_oracle` and
    tests/test_model_gpu.py::test_config1_b128_k8192_step_vs_oracle compare with the oracle); the three losses and a digest of
    every state tensor must agree BIT for bit (counterpart of lib/engine/trainer.py:81-91).  The state is restored in place
    (same addresses: the recording stays valid) and left as the replayed step made it."""
    from textreid_amd import ops

    if runner is None or runner.graph is None:
        return None
    params = [p for g in opt.param_groups for p in g["params"] if p in opt.state and "exp_avg" in opt.state[p]]

    def snapshot():
        torch.cuda.synchronize()
        return ({k: v.detach().clone() for k, v in model.state_dict().items()},
                [(opt.state[p]["exp_avg"].clone(), opt.state[p]["exp_avg_sq"].clone(), int(opt.state[p]["step"])) for p in params])

    def restore(snap):
        with torch.no_grad():
            cur = model.state_dict()
            for k, v in snap[0].items():
                cur[k].copy_(v)
            for p, (m, v, t) in zip(params, snap[1]):
                opt.state[p]["exp_avg"].copy_(m)
                opt.state[p]["exp_avg_sq"].copy_(v)
                opt.state[p]["step"] = t
        ops.note_parameter_write()
        torch.cuda.synchronize()

    before = snapshot()
    form = "hipgraph replay" if (runner.force_graph_launch or runner.replayer is None) else "stream replay"
    l_replay = {k: v.detach().clone() for k, v in runner(images, cb).items()}
    after_replay = snapshot()
    restore(before)
    l_eager = {k: v.detach().clone() for k, v in runner._eager(images, cb).items()}
    after_eager = snapshot()
    diffs = [("loss:" + k) for k in l_eager if not torch.equal(l_eager[k], l_replay[k])]
    diffs += [k for k, v in after_eager[0].items() if not torch.equal(v, after_replay[0][k])]
    for i, (a, b) in enumerate(zip(after_eager[1], after_replay[1])):
        if not (torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[2] == b[2]):
            diffs.append("adam[%d]" % i)
    moved = sum(1 for k, v in after_eager[0].items() if v.is_floating_point() and not torch.equal(v, before[0][k]))
    return {"equal": not diffs, "launch_form": form, "state_tensors": len(after_eager[0]), "adam_moment_pairs": len(params),
            "state_tensors_changed_by_the_step": moved, "losses": {k: float(v) for k, v in l_replay.items()}, "first_differences": diffs[:8]}


def parity_vs_oracle(device, pref, tol=1e-3):
    """ONE HIP train step (forward of the four encoders, three losses, backward - no optimizer) from the state and batch
    of the CPU-baseline leg's first timed step, compared with that oracle step: the three losses, all 183 trainable
    gradients in full, both queues after the push, every momentum-updated key parameter, every BatchNorm running
    statistic.  The oracle is the checker here, never the thing timed."""
    from oracle.cases import relu_floor, step_errors
    from textreid_amd.caption import CaptionBatch
    from textreid_amd.config import moco_cfg
    from textreid_amd.model import build_model

    st0, table, (images, tokens, lengths, ids), (rl, rg, rs, relu_min) = pref
    B, K = images.shape[0], st0["v_queue"].shape[1]
    model = build_model(moco_cfg("m_resnet50", K=K), vocab_dict=table)
    head = model.embed_model
    head.load_state_dict({k: v.clone() for k, v in st0.items()})
    model.to(device).train()
    ld = model(images.to(device), CaptionBatch(tokens.to(device), lengths.to(device), ids.to(device)))
    sum(ld.values()).backward()
    torch.cuda.synchronize()
    named = dict(head.named_parameters())
    errs = step_errors(ld, lambda k: named[k].grad, head.state_dict(), (rl, rg, rs))
    worst = max(errs, key=errs.get)
    out = {
        "worst_rel_err": errs[worst],
        "worst_quantity": worst,
        "quantities": len(errs),
        "gradients_compared": sum(k.startswith("grad:") for k in errs),
        "loss_rel_err": {k[5:]: v for k, v in errs.items() if k.startswith("loss:")},
        "tolerance": tol,
        "within_tolerance": bool(errs[worst] <= tol),
        "launch_form": "eager (one step, no optimizer); the timed region's launch form is tied to it bitwise by replay_vs_eager",
        "case": "configs[1] at its exact size: CLIP-RN50 + BiGRU, B=%d, K=%d, margin-style state (oracle.fill seed %d: SELECTED for its ReLU margin by tools/pick_fullstep_seed.py - a well-conditioned step, on which two correct fp32 evaluations take the same side of every ReLU), ragged captions; smallest |ReLU input| of the oracle's query encoder %.1e (floor %.0e).  The complement - an UNSELECTED seed with He-style weights and 1.5e9 unstructured ReLU decisions at this size, in the decision-count form - is tests/test_model_gpu.py::test_config1_b128_unselected_seed_decision_count" % (
            B, K, PARITY_SEED, relu_min, relu_floor(B)),
        "seed_selected_for_relu_margin": True,
    }
    del model, head, named, ld
    torch.cuda.empty_cache()
    return out


def queue_similarity_bench(device, B=128, C=256, K=8192, reps=20, bf16=False):
    """The batch x queue similarity / masked-InfoNCE block on its own (head.py:148-170 + losses.py:206-217),
    forward AND gradient: the fused single-pass kernel of csrc/queue_nce.hip.  Algorithmic HBM bytes = both
    queues + ids + queries + gradients (17.4 MB at K=8192); no [B,K] matrix exists."""
    from textreid_amd import losses

    g = torch.Generator(device="cpu").manual_seed(11)
    nrm = lambda t: torch.nn.functional.normalize(t, dim=1).to(device)
    vq, tq, vk, tk = (nrm(torch.randn(B, C, generator=g)) for _ in range(4))
    tqueue, vqueue = nrm(torch.randn(K, C, generator=g)), nrm(torch.randn(K, C, generator=g))
    ids = torch.arange(B, device=device) // 4
    idq = torch.randint(0, 11003, (1, K), generator=g).to(device)
    from textreid_amd import ops

    fn = lambda: losses.queue_infonce_loss(vq, tq, vk, tk, ids, tqueue, vqueue, idq)
    old_prec = ops.GEMM_PRECISION
    if bf16:
        ops.GEMM_PRECISION = 1  # one bf16 plane of everything (configs[3]'s autocast arithmetic)
    try:
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
    finally:
        ops.GEMM_PRECISION = old_prec
    ms = e0.elapsed_time(e1) / reps
    nbytes = 2 * C * K * 4 + 8 * K + 4 * B * C * 4 + 2 * B * C * 4
    flops = 2 * 2 * 2.0 * B * C * K  # similarity + gradient GEMMs, both modalities
    products = 1 if bf16 else 3
    if bf16:
        return {"K": K, "arithmetic": "bf16 (one plane, 1 MFMA product per multiply-add)", "ms": ms, "algorithmic_MB": nbytes / 1e6,
                "achieved_GB_per_s": nbytes / ms / 1e6, "achieved_TFLOP_per_s": flops / ms / 1e9, "launches": 4,
                "note": "same single pass in bf16-autocast arithmetic (trid_queue_nce_f32 precision 1, the queues still fp32 in HBM): a third of the matrix work of the fp32-class kernel"}
    return {"K": K, "ms": ms, "algorithmic_MB": nbytes / 1e6, "achieved_GB_per_s": nbytes / ms / 1e6,
            "achieved_TFLOP_per_s": flops / ms / 1e9,
            "mfma_issue_TFLOP_per_s": products * flops / ms / 1e9, "mfma_issue_frac_of_dense_peak": products * flops / ms / 1e9 / 2500.0,
            "launches": 3,
            "note": "fused single pass over both queues (queue_nce.hip): ONE kernel for the batch-wide negative filter (hashed id set in LDS) + similarity + masked softmax + dL/dq of both modalities (no [B,K] matrix), partial fold, loss sum; fp32-class arithmetic = 6 fp16 MFMA products per (query, row, channel) (two-plane split, fixed scales), so the block is MFMA-issue-bound at B=128 (mfma_issue_*: the matrix rate actually sustained; random-data MFMA kernels on this part are power-limited to ~0.5-0.6 of the dense peak, profiles/r03j_zero_vs_random.txt): HBM time of the algorithmic bytes at 8 TB/s would be %.1f us" % (nbytes / 8e12 * 1e6)}


def configs3_bench(device, B=128, K=65536, steps=6, warmup=3):
    """configs[3]'s per-GPU workload on this one GPU (CLIP-RN101 + BiGRU, B=128, MoCo queue 65536): the same train step
    in the fp32-class default and with bf16 convolution operands (the residual blocks' activations, filters and
    incoming gradients live in HBM as bf16 tensors, TRID_CONV_PRECISION=1; parity: tests/test_model_gpu.py::
    test_config3_rn101_k65536_bf16 against the bf16-emulating oracle)."""
    from textreid_amd import ops
    from textreid_amd.caption import CaptionBatch
    from textreid_amd.config import moco_cfg
    from textreid_amd.model import build_model
    from textreid_amd.solver import make_optimizer

    torch.manual_seed(0)
    cfg = moco_cfg("m_resnet101", K=K)
    model = build_model(cfg, vocab_dict=torch.randn(49408, 512) * 0.02).to(device).train()
    opt = make_optimizer(cfg, model)
    batches = [synth_batch(B, s, device, 4321) for s in range(2)]
    out = {"workload": "configs[3] per-GPU share: CLIP-RN101 + GRU, bs%d, MoCo queue %d, 1xMI355X, eager launches" % (B, K)}
    old = ops.CONV_PRECISION

    def step(i):
        images, tokens, lengths, ids = batches[i % 2]
        ld = model(images, CaptionBatch(tokens, lengths, (ids + i * (B // 4)) % 11003, max_len=64))
        opt.zero_grad()
        sum(ld.values()).backward()
        opt.step()

    try:
        for name, prec in (("fp32_class", 16), ("bf16_conv_operands", 1)):
            ops.CONV_PRECISION = prec
            for i in range(warmup):
                step(i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                step(warmup + i)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / steps * 1e3
            out[name] = {"ms_per_step": ms, "pairs_per_s": B / ms * 1e3}
    finally:
        ops.CONV_PRECISION = old
    out["bf16_speedup"] = out["fp32_class"]["ms_per_step"] / out["bf16_conv_operands"]["ms_per_step"]
    out["bf16_parity_note"] = ("builder-defined comparator: the reference has no bf16 path, so the bf16 mode is checked against this repository's own "
                               "bf16-emulating oracle (oracle.visual.bf16_conv) within 3x (<= 0.5 % of the quantities: 6x) of that oracle's fp32-vs-fp64 spread - "
                               "tests/test_model_gpu.py::test_config3_rn101_k65536_bf16; it is outside the fp32 parity contract and never the default")
    del model, opt
    torch.cuda.empty_cache()
    return out


def encode_bench(model, images, tokens, lengths, reps=5, arch="m_resnet50"):
    """Eval-mode encode rates (test_net.py path: running-stat BatchNorm, no key encoders): gallery images/s
    and query captions/s of ONE GPU at the training batch size; `gallery_encode` carries the MFMA roofline of the image pass."""
    from textreid_amd.caption import CaptionBatch

    head = model.embed_model
    was_training = model.training
    model.eval()
    cb = CaptionBatch(tokens, lengths, max_len=64)
    out = {}
    with torch.no_grad():
        for name, fn in (("gallery_encode_imgs_per_s", lambda: head.encode_images(images)),
                         ("query_encode_captions_per_s", lambda: head.encode_captions(cb))):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            out[name] = reps * images.shape[0] / (time.perf_counter() - t0)
    # the gallery pass at larger batches (TEST.IMS_PER_BATCH is the user's choice; the M = 24 576-row launches of layer3 / layer4
    # leave the last round of resident workgroups partly empty at 128 images and fill it at 256)
    by_batch = {int(images.shape[0]): out["gallery_encode_imgs_per_s"]}
    with torch.no_grad():
        for mult in (2, 4):
            try:
                big = images.repeat(mult, 1, 1, 1)
                head.encode_images(big)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(max(reps // mult, 2)):
                    head.encode_images(big)
                torch.cuda.synchronize()
                by_batch[int(big.shape[0])] = max(reps // mult, 2) * big.shape[0] / (time.perf_counter() - t0)
                del big
            except RuntimeError:  # (out of memory beside the training state: report what fitted)
                break
    torch.cuda.empty_cache()
    model.train(was_training)
    gf = VISUAL_FWD_GFLOP.get(arch)
    if gf:
        tfl = out["gallery_encode_imgs_per_s"] * gf / 1e3
        roof = {
            "bound": "mfma",
            "kernel": "the eval-mode image encoder as a whole (inference.py:14-26): P16 tile / streaming / ring-of-rows kernels with the running-statistics BatchNorm, residual and ReLU fused into their epilogues",
            "achieved": tfl,
            "peak": F16_SPLIT_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": tfl / F16_SPLIT_PEAK_TFLOPS,
            "traffic": None,
            "algorithmic_gflop_per_image": gf,
            "note": "images/s x algorithmic forward GFLOP per image (SURVEY.md 8d) over the whole pass's wall time, launch gaps and the HBM-bound stem / layer1 kernels included",
        }
        for label, pat in (("tile_3x3", "gemm_p16_kernel<2, 128, 128, 2, 4, 2, 2"), ("tile_1x1", "gemm_p16_kernel<0, 128, 128, 2, 4, 2, 2"), ("stream_1x1", "gemm_p16_stream_kernel")):
            k_ms, k_src = stored_kernel_avg_ms(pat, "_eval_encode_kernel_stats.csv")
            if k_ms:
                roof.setdefault("rocprof_kernels", {})[label] = {"avg_launch_ms": k_ms, "source": k_src}
        best = max(by_batch, key=by_batch.get)
        out["gallery_encode"] = {"value": out["gallery_encode_imgs_per_s"], "unit": "imgs/s", "batch": int(images.shape[0]), "roofline": roof,
                                 "by_batch": {str(k): v for k, v in sorted(by_batch.items())},
                                 "best": {"batch": best, "value": by_batch[best], "frac": by_batch[best] * gf / 1e3 / F16_SPLIT_PEAK_TFLOPS},
                                 "engine_default": {"batch": 512, "value": by_batch.get(512), "note": "engine.inference.compute_on_dataset collects the images that still need encoding across loader batches and runs the encoder 512 at a time (encode_batch; results unchanged): the rate the inference engine delivers; `value` above stays at the training batch size for continuity with earlier rounds"}}
    return out


def retrieval_bench(device, world, rank, G_total=1000000, Q=10000, k=10, shard_rows=None):
    """configs[4]: Q=1e4 text queries against a 1e6-image gallery whose rows are sharded over the ranks
    (G/world per GPU; ONE GPU scores the whole 1e6-row gallery, 1 GB of fp32 embeddings): similarity +
    per-query top-10 on device; with world > 1 the per-shard lists are all-gathered and merged
    (evaluation.similarity_topk).  `shard_rows` times a single shard of that size instead (the per-GPU
    work of the 8-GPU configuration).  Called on EVERY rank (it contains collectives)."""
    import torch.distributed as dist
    from textreid_amd.evaluation import similarity_topk

    shard = G_total // world if shard_rows is None else shard_rows
    gen = torch.Generator(device="cpu").manual_seed(7)
    q = torch.nn.functional.normalize(torch.randn(Q, 256, generator=gen), dim=1).to(device)  # replicated queries
    gen.manual_seed(70 + rank)
    g = torch.nn.functional.normalize(torch.randn(shard, 256, generator=gen), dim=1).to(device)
    similarity_topk(q, g, k, normalize=False)  # warm-up (also sizes the collectives' buffers)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    vals, idx = similarity_topk(q, g, k, normalize=False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tm = torch.tensor([dt], dtype=torch.float64)
        tm = tm.to(device) if dist.get_backend() != "gloo" else tm
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        dt = float(tm.item())
    scored = shard * world
    tfl = 2.0 * Q * scored * 256 / dt / 1e12
    k_ms, k_src = stored_kernel_avg_ms("gemm_p16_stream_kernel<256, 8, 2, false, 4>", "_retrieval_kernel_stats.csv")
    # launches of the filter kernel per match: the rows beyond the first 8192 go in segments of growing length (x 8, no short last
    # one: csrc/retrieval.hip sim_topk), each merged before the next starts
    nseg, s0 = 0, 8192
    while s0 < shard:
        s1 = s0 * 8
        if s1 >= shard or shard - s1 < s1 // 4:
            s1 = shard
        nseg, s0 = nseg + 1, s1
    tr, tr_src = stored_traffic("gemm_p16_stream_kernel<256, 8, 2, false, 4>", infer="rt")
    roof = {
        "bound": "mfma",
        "kernel": "trid::gemm_p16_stream_kernel<256, 8, 2, false, 4> (similarity of pre-split operands with the top-k admission filter as epilogue: 256 queries resident in a workgroup's registers, the gallery streamed through LDS once per query panel; fp16 two-plane arithmetic, 3 MFMA products per multiply-add)",
        "achieved": tfl / world,
        "peak": F16_SPLIT_PEAK_TFLOPS,
        "unit": "TFLOP/s",
        "frac": tfl / world / F16_SPLIT_PEAK_TFLOPS,
        "traffic": tr * nseg if (tr and shard_rows is None and world == 1) else None,
        "traffic_source": tr_src if (shard_rows is None and world == 1) else None,
        "traffic_note": "HBM-side bytes of the filter kernel per MATCH = per-launch average of the stored PMC passes (tools/retrieval_time.py, same Q and G) x its %d launches (gallery segments): the 1 GB gallery crosses the fabric about three times although 40 query panels stream it - the panels of one worker share an L2; the writes are the candidate appends" % nseg,
        "note": "achieved = algorithmic FLOPs of the WHOLE match (2 Q G C, per GPU) / wall time of the whole call (amax + split of both operands, first-panel GEMM + row scan, the filter pass in %d segments, list merges, one 4-byte host read): the filter kernel is ~93 %% of it" % nseg,
    }
    if k_ms and shard_rows is None and world == 1:
        gf = 2.0 * Q * (shard - 8192) * 256 / 1e9
        roof["rocprof_kernel"] = {"launches_per_match": nseg, "avg_launch_ms": k_ms, "kernel_ms_per_match": k_ms * nseg, "achieved": gf / (k_ms * nseg), "frac": gf / (k_ms * nseg) / F16_SPLIT_PEAK_TFLOPS, "source": k_src + ": average duration of this kernel in rocprofv3 --kernel-trace --stats of tools/retrieval_time.py (same Q, G) x launches per match"}
    return {
        "roofline": roof,
        "metric": "gallery imgs/sec (retrieval: similarity + top-10, Q=1e4 queries)",
        "value": scored / dt,
        "unit": "gallery imgs/s",
        "n_gpus": world,
        "gallery_rows_scored": scored,
        "gallery_shard_per_gpu": shard,
        "queries": Q,
        "seconds": dt,
        "tflops": tfl,
        "algorithmic_bytes_per_gpu": shard * 256 * 4 + Q * 256 * 4 + Q * k * 12,
        "note": "fp32 embeddings in, split ONCE into fp16 planes (fp32-class two-plane arithmetic), exact top-10: first 8192 gallery rows via a [Q,8192] panel + streaming scan, the rest via the admission-filter epilogue of the streaming kernel in segments of growing length, thresholds tightened in between (no similarity matrix in HBM)",
    }


def newest_profiles(suffix, infer=None):
    """File names under profiles/ that end in `suffix`, newest round first (names are r<round><letter>_...: descending name order).
    infer: None = the summaries of THIS command's train step only (names without "_infer_"); "rt" / "ev" = those of the
    inference-side tools (r*_infer_rt_* : tools/retrieval_time.py, r*_infer_ev_* : tools/eval_time.py) - the eval pass launches
    the same kernel templates as the train step, and its numbers must not stand in for the step's."""
    here = os.path.dirname(os.path.abspath(__file__))
    try:
        names = (f for f in os.listdir(os.path.join(here, "profiles")) if f.endswith(suffix) and f.startswith("r"))
        return sorted((f for f in names if (("_infer_%s_" % infer) in f if infer else "_infer_" not in f)), reverse=True)
    except OSError:
        return []


def stored_kernel_avg_ms(pattern, suffix):
    """(average duration in ms, file) of the kernels whose name contains `pattern` in the newest committed rocprofv3
    --kernel-trace --stats summary profiles/r*<suffix> that lists them, or (None, None)."""
    import csv

    here = os.path.dirname(os.path.abspath(__file__))
    for fn in newest_profiles(suffix):
        try:
            tot, calls = 0.0, 0
            for r in csv.DictReader(open(os.path.join(here, "profiles", fn))):
                if pattern in r["Name"]:
                    tot += float(r["TotalDurationNs"])
                    calls += int(r["Calls"])
            if calls:
                return tot / calls / 1e6, "profiles/" + fn
        except (OSError, ValueError, KeyError):
            pass
    return None, None


def stored_traffic(pattern, infer=None):
    """(HBM-side bytes per launch, file) of the kernels whose name contains `pattern` in the NEWEST committed PMC summary that
    lists them (profiles/r*_pmc_hbm_traffic.txt, newest first by name; calls-weighted over the matching lines; separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes with the gfx950 2x FETCH_SIZE correction, tools/pmc_summary.py), or (None, None)."""
    here = os.path.dirname(os.path.abspath(__file__))
    for fn in newest_profiles("_pmc_hbm_traffic.txt", infer):
        try:
            tot, calls = 0.0, 0
            for ln in open(os.path.join(here, "profiles", fn)):
                f = ln.split()
                if pattern in ln and len(f) > 5:
                    tot += int(f[1]) * (float(f[2]) + float(f[3])) * 1e6
                    calls += int(f[1])
            if calls:
                return tot / calls, "profiles/" + fn
        except (OSError, ValueError):
            pass
    return None, None


def visible_gpus():
    """Number of GPUs this process could use, WITHOUT touching HIP: the KFD topology in sysfs (nodes with SIMDs), cut by
    the usual visibility masks.  None when sysfs cannot be read (the ranks then check for themselves).
    torch.cuda.device_count() is not used here: without amdsmi it falls through to hipGetDeviceCount, which brings the
    HIP / HSA runtime up in the parent - harmless today (ranks are fresh child processes, never an exec of this one),
    but an initialised parent must never be followed by a re-exec or launcher hop on this pool."""
    import glob

    n = 0
    if not os.path.isdir("/sys/class/kfd"):
        return 0  # no KFD driver: no AMD GPU on this machine
    try:
        for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
            for line in open(f):
                k, _, v = line.partition(" ")
                if k == "simd_count" and int(v) > 0:
                    n += 1
    except OSError:
        return None
    if n == 0:
        return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        m = os.environ.get(var)
        if m is not None:
            n = min(n, len([x for x in m.split(",") if x.strip() != ""]))
    return n


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N rank processes (one per GPU, env:// rendezvous on
    127.0.0.1) and return the worst exit code.  The parent only COUNTS devices (sysfs, no HIP call); the ranks are
    always fresh child processes of it, never an exec of this process."""
    import socket
    import subprocess

    have = visible_gpus()
    if have is not None and have >= 1 and have < n and os.environ.get("TRID_DIST_BACKEND", "nccl") == "gloo":
        have = n  # (debug transport: the ranks share the visible GPU(s) - RCCL needs one GPU per rank, gloo does not)
    if have is not None and have < n:
        print("bench.py: --gpus %d requested but only %d GPU(s) are visible; refusing to print a smaller job's line" % (n, have),
              file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    codes = [p.wait() for p in procs]
    return max(abs(c) for c in codes)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs (default: the launcher's WORLD_SIZE, else 1)")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=128, help="per-GPU batch (weak scaling)")
    ap.add_argument("--queue", type=int, default=8192)
    ap.add_argument("--model", default="m_resnet50")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--retrieval", action="store_true", help="(default on; kept for compatibility)")
    ap.add_argument("--no-retrieval", action="store_true", help="skip the configs[4] retrieval timing ('retrieval' object)")
    ap.add_argument("--no-configs3", action="store_true", help="skip the secondary configs[3]-shape timing ('configs3_1gpu' object)")
    args = ap.parse_args()

    import torch.distributed as dist

    if args.gpus is None:  # `torchrun --nproc-per-node N bench.py` without --gpus: the launcher's world is the job
        args.gpus = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))  # no launcher: become one (before anything touches the GPU)
    # The contract is ONE JSON line on stdout.  Libraries loaded below print there too (RCCL's version banner goes to fd 1 from C
    # when a process group initialises): in a RANK process fd 1 is pointed at stderr for the whole run and the line is written to the
    # real one at the end.  (After the spawn decision above: the ranks a launcher-less `--gpus N` starts must inherit the real stdout.)
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("TRID_DIST_BACKEND", "nccl")
    if world != args.gpus:
        print("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    ndev = torch.cuda.device_count()
    if backend == "gloo":
        local = local % max(ndev, 1)  # (debug transport: several ranks may share one GPU)
    elif local >= ndev:
        print("bench.py: rank %d needs GPU %d but only %d GPU(s) are visible (one RCCL rank per GPU)" % (rank, local, ndev), file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    # dp: the data-parallel exchange steps run.  TRID_DP_FORCE=1 with ONE rank: a one-rank `nccl` group drives every RCCL call of
    # the step on a single-GPU box (the host path of the data-parallel step - eager or segmented replay - is then measurable there)
    dp = world > 1 or os.environ.get("TRID_DP_FORCE", "0") == "1"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, init_method="env://")
    elif dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29591")
        dist.init_process_group(backend=backend, init_method="env://", rank=0, world_size=1)

    from textreid_amd import ops
    from textreid_amd.caption import CaptionBatch
    from textreid_amd.config import moco_cfg
    from textreid_amd.model import build_model
    from textreid_amd.parallel import GradReducer
    from textreid_amd.solver import make_optimizer

    torch.manual_seed(0)  # identical initial weights / queues on every rank
    cfg = moco_cfg(args.model, K=args.queue)
    table = torch.randn(49408, 512) * 0.02
    model = build_model(cfg, vocab_dict=table).to(device)
    model.train()
    log("model built")
    opt = make_optimizer(cfg, model)
    reducer = GradReducer()
    if dp and os.environ.get("TRID_DP_OVERLAP", "1") != "0":
        # image-encoder gradients are all-reduced per residual layer from inside its backward (overlap)
        model.embed_model.v_encoder_q.grad_sync = reducer
    pre_gather = [p for n, p in model.named_parameters() if p.requires_grad and "loss_evaluator" not in n]
    pre_gather.reverse()
    B = args.batch

    batches = [synth_batch(B, s, device, 1234 + rank) for s in range(4)]
    if world > 1:  # global ids so positives match across ranks' queue pushes
        batches = [(im, tk, ln, ids + rank * (B // 4) + s * (B // 4) * (world - 1)) for s, (im, tk, ln, ids) in enumerate(batches)]

    # the whole step (four streams, ~1100 launches) is recorded once and re-issued by the library (engine/graph.py); data parallel:
    # the recording's collectives are cut points, the replay runs in segments around them (TRID_DP_CAPTURE=0: the eager DP step)
    runner = None
    dp_capture = dp and os.environ.get("TRID_DP_CAPTURE", "1") != "0"
    if (not dp or dp_capture) and os.environ.get("TRID_CAPTURE", "1") != "0":
        from textreid_amd.engine.graph import CapturedTrainStep

        runner = CapturedTrainStep(model, opt, warmup=2, caption_bound=64,  # 64-token captions (padded to 105)
                                   reducer=reducer if dp else None, pre_gather=pre_gather)

    def batch(i):
        images, tokens, lengths, ids = batches[i % len(batches)]
        return images, CaptionBatch(tokens, lengths, (ids + (i // len(batches)) * len(batches) * (B // 4) * world) % 11003, max_len=int(os.environ.get("TRID_BENCH_CAPTION_LEN", 64)))

    def eager_step(i):
        images, cb = batch(i)
        loss_dict = model(images, cb)
        losses = sum(loss_dict.values())
        opt.zero_grad()
        losses.backward()
        if dp:
            reducer.reduce(pre_gather)
            reducer.wait()
        opt.step()
        return loss_dict

    def step(i):
        if runner is None:
            return eager_step(i)
        return runner(*batch(i))

    n_prep = 0
    if runner is not None:  # set-up, untimed and not counted as warm-up: two eager steps build every cached table, the third call records
        while runner.graph is None and not runner.disabled and n_prep < 8:
            step(n_prep)
            n_prep += 1
        torch.cuda.synchronize()
        if runner.graph is None:  # (a recording that failed stays eager - CapturedTrainStep said why; the bench goes on with eager steps)
            log("the train step could not be recorded: eager launches")
        else:
            log("train step captured after %d set-up steps" % n_prep)
    for i in range(args.warmup):
        step(n_prep + i)
        torch.cuda.synchronize()
        log("warmup step %d done" % i)
    # recorded or eager?  Both run the same kernels on the same streams with bit-identical results (tests/test_match_state_gpu.py);
    # which is faster depends on the box: the replay saves host time (one hipGraphLaunch instead of ~1000 launches) but its
    # executor adds a dependency edge per node (~0.3 ms per step), the eager step needs the host to stay ahead of a 44 ms GPU
    # step.  A short probe of each decides; both figures go into the line (TRID_BENCH_LAUNCH=graph / eager forces one).
    launch_probe = None
    use_eager = False
    if runner is not None and runner.graph is not None:
        def probe(fn, n=6):
            torch.cuda.synchronize()
            t0p = time.perf_counter()
            for j in range(n):
                fn(j)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0p) / n * 1e3

        base = n_prep + args.warmup
        probe(lambda j: runner._eager(*batch(base + j)), 2)  # (the eager path's own warm-up after the recording)
        ms_streams = probe(lambda j: step(base + 2 + j)) if runner.replayer is not None else None
        ms_graph = None
        if not runner.cuts:  # (a data-parallel recording holds markers in place of its collectives: it has no hipGraphLaunch form)
            try:  # hipGraphLaunch of the same recording
                runner.force_graph_launch = True
                step(base + 8)
                ms_graph = probe(lambda j: step(base + 9 + j))
            except RuntimeError as e:
                log("launch probe: hipGraphLaunch failed (%s)" % (str(e).splitlines()[0] if str(e) else type(e).__name__))
            finally:
                runner.force_graph_launch = False
        ms_eager = probe(lambda j: runner._eager(*batch(base + 15 + j)))
        forced = os.environ.get("TRID_BENCH_LAUNCH", "")
        cands = {"stream replay": ms_streams, "hipgraph replay": ms_graph, "eager": ms_eager}
        if forced in ("eager", "graph", "streams"):
            chosen = {"eager": "eager", "graph": "hipgraph replay", "streams": "stream replay"}[forced]
        else:
            chosen = min((k for k in cands if cands[k] is not None), key=lambda k: cands[k] * (0.995 if k == "eager" else 1.0))
        if cands[chosen] is None:
            chosen = "eager"
        launch_probe = {"stream_replay_ms_per_step": ms_streams, "hipgraph_replay_ms_per_step": ms_graph, "eager_ms_per_step": ms_eager,
                        "chosen": chosen, "stream_replay_plan": runner.replay_info,
                        "collectives_as_cut_points": len(runner.cuts)}
        if world > 1:  # every rank must take the same path: rank 0 decides
            names = ["stream replay", "hipgraph replay", "eager"]
            flag = torch.tensor([names.index(chosen)], device=device)
            dist.broadcast(flag, 0)
            chosen = names[int(flag.item())]
        use_eager = chosen == "eager"
        runner.force_graph_launch = chosen == "hipgraph replay"
        log("launch probe: stream replay %s, hipGraphLaunch %s, eager %.2f ms per step -> %s" % (
            "%.2f" % ms_streams if ms_streams else "-", "%.2f" % ms_graph if ms_graph else "-", ms_eager, chosen))
        n_prep += 23
    timed_step = (lambda i: runner._eager(*batch(i))) if use_eager else step
    # live roofline of the dominant kernel: 3x3 implicit-GEMM conv, 128x128 tiles
    split = ops.GEMM_PRECISION in (1, 3, 6)
    dom = (ops.A_CONV, ops.B_KC, 128, 128, split)
    dom2 = (ops.A_KC, ops.B_KC, 128, 128, split)  # 1x1 convs / linears: the largest TOTAL time of any kernel
    labels = {dom: "conv3x3", dom2: "gemm1x1", "stream1x1": "stream1x1"}  # (the streaming short-K kernel: HBM-bound, events carry bytes)
    if runner is None:  # eager steps: events bracket every launch of the two kernels inside the timed region
        ops.PROFILE = {"match": labels.get, "events": []}
    reducer.reset_stats()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        last = timed_step(n_prep + args.warmup + i)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    log("timed region: %.3fs for %d steps (host enqueue time %.3fs)" % (dt, args.steps, t_host))
    last = {k: v.detach().clone() for k, v in last.items()}
    # host WORK per step, without back-pressure: over the 20 steps of the timed region the launch calls block on full hardware queues
    # (host_enqueue_ms_per_step ~ the GPU step time whatever the launch form); three steps issued onto an idle GPU do not, so the time to
    # get them enqueued is what the host itself needs per step - the figure that decides whether N ranks keep their GPUs fed
    torch.cuda.synchronize()
    t0w = time.perf_counter()
    for i in range(3):
        timed_step(n_prep + args.warmup + args.steps + 100 + i)
    host_work_ms = (time.perf_counter() - t0w) / 3 * 1e3
    torch.cuda.synchronize()
    log("host work per step, no back-pressure: %.2f ms" % host_work_ms)
    seam = None
    if runner is not None and not dp:  # (also when the probe chose eager launches for the timed region: the recording is what do_train replays)
        seam = replay_equals_eager(runner, model, opt, *batch(n_prep + args.warmup + args.steps))
        if seam is not None:
            seam["timed_region_launch_form"] = "eager" if use_eager else seam["launch_form"]
    if seam is not None:
        log("replay vs eager at B=%d (%s, optimizer included): %s" % (B, seam["launch_form"], "bit-identical" if seam["equal"] else "DIFFERENT: %s" % seam["first_differences"]))
    profiled_eager = 0
    if runner is not None:
        # events cannot bracket individual nodes of a replayed graph: the per-launch timing of the dominant kernels comes
        # from eager re-runs of the SAME step (all four streams active) right after the timed region
        # ... on ONE stream (TRID_SERIAL): with the four streams of the product step an event pair also counts the time a
        # kernel WAITS for the CUs another stream's kernel holds (measured: 194 us per launch against 136 us in the rocprofv3
        # trace of the replayed step) - on one stream the pair brackets the kernel's own duration (113-130 us)
        import textreid_amd.backbones.m_resnet as _mr

        ops.PROFILE = {"match": labels.get, "events": []}
        profiled_eager = 3
        os.environ["TRID_SERIAL"] = "1"
        serial_was, _mr._SERIAL_WGRAD = _mr._SERIAL_WGRAD, True
        try:
            for i in range(profiled_eager):
                runner._eager(*batch(n_prep + args.warmup + args.steps + i))
            torch.cuda.synchronize()
        finally:
            os.environ.pop("TRID_SERIAL", None)
            _mr._SERIAL_WGRAD = serial_was
    prof, ops.PROFILE = ops.PROFILE, None
    tmax = torch.tensor([dt], device=device)
    if world > 1:
        if dist.get_backend() == "gloo":
            tc = tmax.cpu()
            dist.all_reduce(tc, op=dist.ReduceOp.MAX)
            tmax = tc
        else:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    loss_val = float(sum(last.values()).item())

    def live(label):
        ev = [e for e in prof["events"] if e[0] == label]
        fl = sum(e[1] for e in ev)
        t = sum(e[2].elapsed_time(e[3]) for e in ev)
        return fl, t, len(ev)

    flops, ms, nlaunch = live("conv3x3")
    achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    flops2, ms2, nlaunch2 = live("gemm1x1")
    bytes3, ms3, nlaunch3 = live("stream1x1")
    dp_stats = reducer.stats() if dp else None

    # the same kernel alone on the GPU (no side-stream work sharing the CUs): the 3x3 layers of layer2-4
    # ... and once more on ZERO-filled operands: MFMA timing does not depend on the data, the power drawn - hence the
    # sustained clock - does; the gap between the two is the part of the distance to the nominal peak that no schedule
    # can close on real data
    iso_fl, iso_ms, iso_ms_zero = 0.0, 0.0, 0.0
    p16 = ops.USE_P16 and ops.conv_precision() == 16  # the residual blocks' convolutions run on pre-split operands
    for (Hh, Ww, Cc) in ((96, 32, 128), (48, 16, 128), (48, 16, 256), (24, 8, 256), (24, 8, 512)):
        for zero in (False, True):
            xx = torch.zeros(B, Hh, Ww, Cc, device=device) if zero else torch.randn(B, Hh, Ww, Cc, device=device)
            ww = torch.zeros(Cc, 9 * Cc, device=device) if zero else torch.randn(Cc, 9 * Cc, device=device)
            cp = ops.conv_precision()
            if p16:
                one = torch.ones(1, device=device)
                xp, wp = (ops.p16_pack(xx, one), ops.p16_pack(ww, one)) if zero else (ops.p16_pack(xx), ops.p16_pack(ww))
                run = lambda: ops.conv_p16(xp, wp, conv3=True)
            else:
                kw = dict(prec=cp, aa=ops.amax(xx), ba=ops.amax(ww)) if cp == 16 else {}  # the model's conv arithmetic
                run = lambda: ops.conv3x3(xx, ww, stats=True, **kw)
            run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                run()
            e1.record()
            torch.cuda.synchronize()
            if zero:
                iso_ms_zero += e0.elapsed_time(e1) / 3
            else:
                iso_ms += e0.elapsed_time(e1) / 3
                iso_fl += 2.0 * B * Hh * Ww * Cc * 9 * Cc
            del xx, ww
    achieved_isolated = iso_fl / (iso_ms * 1e-3) / 1e12
    achieved_isolated_zero = iso_fl / (iso_ms_zero * 1e-3) / 1e12

    prec = ops.GEMM_PRECISION
    cprec = ops.conv_precision()
    if cprec == 16 and p16:
        peak = BF16_MFMA_PEAK_TFLOPS / 3
        kname = "trid::gemm_p16_kernel<A_CONV,128,128,2,4,2> (3x3 implicit-GEMM conv fwd+dgrad on PRE-SPLIT operands: activations / filters live in HBM as two fp16 planes of x*2^s (4 B per element, written split by their producers), LDS-DMA staged, 3 fp16 MFMA 32x32x16 products per multiply-add in one fp32 accumulator)"
        peak_note = "dense bf16/fp16 MFMA peak 2500 TFLOP/s / 3 products = fp32-equivalent peak; SQ counters of this kernel: MFMA pipe busy 58 % of the elapsed cycles at the ~1.95 GHz the chip sustains under this load (profiles/r03e_pmc_sq_p16.txt)"
        arith = "fp32-class: conv products evaluated as 3 fp16 MFMA terms of a scaled 2-way fp16 split (representation + dropped term <= 3*2^-22 per product), operands split ONCE by their producers (P16 tensors); everything else as 6 bf16 MFMA terms of a 3-way bf16 split (<= 2^-26); fp32 accumulation, BatchNorm, losses, optimizer"
    elif cprec == 16:
        # the image encoder's convolutions run the fp16 two-plane split: 3 fp16 MFMA products per fp32 multiply-add
        peak = BF16_MFMA_PEAK_TFLOPS / 3
        kname = "trid::gemm_bf16s_kernel<A_CONV,B_KC,16,128> (3x3 implicit-GEMM conv fwd+dgrad; fp32 operands scaled per tensor by a power of two and split into 2 fp16 planes (11+11 significand bits), 3 fp16 MFMA 32x32x16 products per multiply-add in one fp32 accumulator)"
        peak_note = "dense bf16/fp16 MFMA peak 2500 TFLOP/s / 3 products = fp32-equivalent peak; PMC MfmaUtil of this kernel 29-47 % at 1.8-2.2 GHz (profiles/r02d_pmc_mfma_util.txt)"
        arith = "fp32 operands and accumulation; conv products evaluated as 3 fp16 MFMA terms of a scaled 2-way fp16 split (representation + dropped term <= 3*2^-22 per product), everything else as 6 bf16 MFMA terms of a 3-way bf16 split (<= 2^-26)"
    elif prec in (3, 6):
        # the dominant kernel executes `prec` bf16 MFMA products per fp32 multiply-add
        peak = BF16_MFMA_PEAK_TFLOPS / prec
        kname = "trid::gemm_bf16s_kernel<A_CONV,B_KC,%d planes> (3x3 implicit-GEMM conv fwd+dgrad; fp32 operands split into bf16 planes, %d bf16 MFMA 32x32x16 products per multiply-add, fp32 accumulate)" % (prec // 2, prec)
        peak_note = "dense bf16 MFMA peak 2500 TFLOP/s / %d products = fp32-equivalent peak (a 1 s pure-MFMA loop with random bf16 operands sustains 1858 TFLOP/s on this box: power-limited ~1.8 GHz, profiles/r01i_mfma_bf16_sustained.txt)" % prec
        arith = "fp32 operands and accumulation; products evaluated as %d bf16 MFMA terms of a %d-way bf16 split (dropped terms <= 2^-%d)" % (prec, prec // 2, 26 if prec == 6 else 17)
    elif prec == 1:
        peak = BF16_MFMA_PEAK_TFLOPS
        kname = "trid::gemm_bf16s_kernel<A_CONV,B_KC,1 plane> (3x3 implicit-GEMM conv fwd+dgrad; fp32 tensors in HBM, operands rounded to bf16 while staged into LDS, one bf16 MFMA 32x32x16 per product, fp32 accumulate)"
        peak_note = "dense bf16 MFMA peak"
        arith = "bf16 operands (rounded on the fly), fp32 accumulation, fp32 tensors / BatchNorm / losses / optimizer (bf16-autocast arithmetic); NOT the configs[1] fp32 line"
    else:
        peak = F32_MFMA_PEAK_TFLOPS
        kname = "trid::gemm_kernel<A_CONV,B_KC,128,128,2,4> (3x3 implicit-GEMM conv fwd+dgrad, fp32 MFMA 32x32x2)"
        peak_note = "fp32-input MFMA dense peak"
        arith = "exact fp32-input MFMA"
    # HBM-side bytes per launch of the dominant kernel: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same
    # command, summarised by tools/pmc_summary.py into profiles/ (gfx950 2x FETCH_SIZE correction applied).  A STORED
    # measurement of the committed build, not a counter read of this run: `traffic_source` names the file.
    traffic, traffic_note, traffic_src = None, None, None
    here = os.path.dirname(os.path.abspath(__file__))

    def stored_rocprof_avg_ms(pattern):
        """(average duration in ms, file) of the kernels whose name contains `pattern` in the newest committed rocprofv3
        --kernel-trace --stats summary of this command (profiles/r*_bench_kernel_stats.csv), or (None, None)."""
        return stored_kernel_avg_ms(pattern, "_bench_kernel_stats.csv")

    for fn in newest_profiles("_pmc_hbm_traffic.txt"):
        try:
            for line in open(os.path.join(here, "profiles", fn)):
                f = line.split()
                want = "0, 16, 128>(trid::GemmParams)" if cprec == 16 else "0, 3, 128>(trid::GemmParams)"
                hit = ("gemm_p16_kernel<2, 128, 128, 2, 4, 2, 2" in line) if p16 else ("gemm_bf16s_kernel<2," in line and line.rstrip().endswith(want))
                if prec == 6 and len(f) > 5 and hit:
                    traffic = (float(f[2]) + float(f[3])) * 1e6
                    traffic_src = "profiles/" + fn
                    traffic_note = "read %s MB + write %s MB per launch (separate --pmc passes, stored; average over all launches of this kernel incl. layer1); algorithmic input + weights + output of the layer2-4 shapes ~ 58 + 9 + 58 MB; + the (mean, M2, min, max) BatchNorm partials" % (f[2], f[3])
                    break
        except OSError:
            pass
        if traffic is not None:
            break
    roofline = {
        "bound": "mfma",
        "kernel": kname,
        "achieved": achieved,
        "peak": peak,
        "unit": "TFLOP/s",
        "frac": achieved / peak,
        "traffic": traffic,
        "traffic_source": traffic_src,
        "traffic_note": traffic_note,
        "achieved_isolated": achieved_isolated,
        "frac_isolated": achieved_isolated / peak,
        "achieved_isolated_zero_operands": achieved_isolated_zero,
        "frac_isolated_zero_operands": achieved_isolated_zero / peak,
        "power_note": "the *_zero_operands figures are the same launches on zero-filled tensors: identical instruction stream, lower switching power, higher sustained clock - the gap to *_isolated is set by the part's power limit under random fp16 operands, not by the kernel's schedule (profiles/r03j_zero_vs_random.txt; the guide's own best plain-HIP bf16 GEMM sustains 0.53-0.59 of the dense peak on random data)",
        "note": ("achieved/frac: events around every launch of this kernel " + ("in %d eager re-runs of the step right after the timed region, all kernels on ONE stream (events cannot bracket the nodes of a replayed graph, and with the product's four streams an event pair also counts the wait for CUs held by another stream's kernel; rocprofv3 --kernel-trace of this command gives the durations inside the timed step: rocprof_in_graph)" % profiled_eager if profiled_eager else "during the timed steps, while the text / key-encoder / weight-gradient streams share the CUs") + "; *_isolated: the same kernel on the five layer2-4 3x3 shapes with the GPU to itself"),
        "peak_note": peak_note,
        "launches": nlaunch,
        "avg_launch_ms": ms / max(nlaunch, 1),
        "algorithmic_gflop_per_launch": flops / max(nlaunch, 1) / 1e9,
    }
    # second roofline object: the 1x1-conv / linear kernel <A_KC,B_KC> has the largest TOTAL time per step
    roofline_1x1 = {
        "bound": "mfma",
        "kernel": (kname.split(" (")[0].replace("A_CONV", "A_KC")) + " (1x1 convs fwd + dgrad with K >= 512 - layer3 / layer4 - and the 256 -> 64 / 128 ones; the short-K layers run on the streaming kernel, roofline_stream)",
        "achieved": flops2 / (ms2 * 1e-3) / 1e12 if ms2 > 0 else 0.0,
        "peak": peak,
        "unit": "TFLOP/s",
        "frac": (flops2 / (ms2 * 1e-3) / 1e12 / peak) if ms2 > 0 else 0.0,
        "traffic": None,
        "launches": nlaunch2,
        "avg_launch_ms": ms2 / max(nlaunch2, 1),
        "algorithmic_gflop_per_launch": flops2 / max(nlaunch2, 1) / 1e9,
        "note": "live events around every launch of this kernel " + ("in the %d one-stream eager re-runs of the step right after the timed region" % profiled_eager if profiled_eager else "during the timed steps (all streams running)"),
    }
    t1, t1_src = stored_traffic("gemm_p16_kernel<0, 128, 128, 2, 4, 2, 2") if p16 else (None, None)
    roofline_1x1["traffic"], roofline_1x1["traffic_source"] = t1, t1_src
    if p16:  # the rocprofv3 view of the same kernels inside the REPLAYED step (stored summary of this command)
        for obj, pat in ((roofline, "gemm_p16_kernel<2, 128, 128, 2, 4, 2, 2"), (roofline_1x1, "gemm_p16_kernel<0, 128, 128, 2, 4, 2, 2")):
            ms_r, src_r = stored_rocprof_avg_ms(pat)
            if ms_r and obj["launches"]:
                gf = obj["algorithmic_gflop_per_launch"]
                obj["rocprof_in_graph"] = {"avg_launch_ms": ms_r, "achieved": gf / ms_r, "frac": gf / ms_r / peak,
                                           "source": "%s: average duration of this template inside the timed step x the algorithmic GFLOP per launch of the live events" % src_r,
                                           "note": "begin-to-end durations with the step's four streams co-running (under the stream replay all of them are fed at once): a kernel's duration includes the time it shares the CUs with the other streams' kernels - the step got faster while these got longer; rocprof_one_lane is the kernel on its own"}
            ms_1, src_1 = stored_kernel_avg_ms(pat, "_bench_kernel_stats_one_lane.csv")
            if ms_1 and obj["launches"]:
                gf = obj["algorithmic_gflop_per_launch"]
                obj["rocprof_one_lane"] = {"avg_launch_ms": ms_1, "achieved": gf / ms_1, "frac": gf / ms_1 / peak,
                                           "source": "%s: the same command with the recorded step laid out on ONE stream (TRID_STEP_LANES=1, tools/exp/r05_prof_one_lane.sh): every kernel has the chip to itself, as in the live one-stream events above" % src_1}
    # the two MFMA-bound tile kernels as equals, the one with the larger total per step first; and the HBM-bound streaming kernel
    nrun = max(profiled_eager, 1) if runner is not None else args.steps
    roofline["total_ms_per_step"] = ms / nrun
    roofline_1x1["total_ms_per_step"] = ms2 / nrun
    # (the 3x3 form stays in front unless the 1x1 form's total is clearly larger: the two are within a few per cent of each other,
    # and a headline object that changes kernel from run to run compares with nothing)
    if roofline_1x1["total_ms_per_step"] > 1.1 * roofline["total_ms_per_step"]:
        roofline, roofline_1x1 = roofline_1x1, roofline
    roofline_stream = {
        "bound": "hbm",
        "kernel": "trid::gemm_p16_stream_kernel<K,CW,TM,ACC> (short-K 1x1 convolutions of layer1-3 and the data gradients of conv1, K = 64 / 128 / 256: filter panel in registers, activation tiles by LDS-DMA, BatchNorm partials in registers, persistent workgroups)",
        "achieved": (bytes3 / (ms3 * 1e-3) / 1e9) if ms3 > 0 else 0.0,
        "peak": 8000.0,
        "unit": "GB/s",
        "frac": (bytes3 / (ms3 * 1e-3) / 1e9 / 8000.0) if ms3 > 0 else 0.0,
        "traffic": stored_traffic("gemm_p16_stream_kernel")[0],
        "traffic_source": stored_traffic("gemm_p16_stream_kernel")[1],
        "launches": nlaunch3,
        "avg_launch_ms": ms3 / max(nlaunch3, 1),
        "algorithmic_MB_per_launch": bytes3 / max(nlaunch3, 1) / 1e6,
        "total_ms_per_step": ms3 / nrun,
        "note": "algorithmic bytes = activations once + output once (read and written when accumulating) + filter, over live event time (one-stream eager re-runs); HBM peak 8 TB/s (a float4 copy reaches 6.29: MI355X_MICROARCH.md)",
    }
    retr = None
    if not args.no_retrieval:
        retr = retrieval_bench(device, world, rank)
        if world == 1:  # the per-GPU work of configs[4]'s 8-GPU form: one 125 000-row shard
            sh = retrieval_bench(device, world, rank, shard_rows=125000)
            retr["one_of_8_shards"] = {k: sh[k] for k in ("value", "unit", "gallery_rows_scored", "seconds", "tflops")}
        retr.update(encode_bench(model, batches[0][0], batches[0][1], batches[0][2], arch=args.model))
        log("retrieval: %.1f M gallery imgs/s" % (retr["value"] / 1e6))
    qsim = [queue_similarity_bench(device, B=B, K=k) for k in sorted({args.queue, 65536})]
    qsim.append(queue_similarity_bench(device, B=B, K=65536, bf16=True))
    step_launch = ("eager launches, four streams (chosen by the probe over the recorded step)" if use_eager else
                   ("hipGraph replay (one launch per step)" if (runner.force_graph_launch or runner.replayer is None) else
                    ("segmented stream replay: the recorded step re-issued by csrc/step_replay.hip in %d segments around its %d collectives, which run through torch.distributed on the recorded streams (chosen by the probe)" % (len(runner.cuts) + 1, len(runner.cuts)) if runner.cuts else
                     "stream replay: the recorded step re-issued as stream launches by one library call per step (csrc/step_replay.hip; chosen by the probe)"))) if runner is not None else "eager (%d rank(s)%s)" % (world, ": RCCL collectives inside backward" if dp else "")
    c3 = None
    if world == 1 and not args.no_configs3 and ops.conv_precision() == 16 and args.model == "m_resnet50":
        del model, opt, runner
        torch.cuda.empty_cache()
        c3 = configs3_bench(device)
        log("configs[3] shape on one GPU: %.1f ms fp32-class, %.1f ms bf16 conv operands" % (c3["fp32_class"]["ms_per_step"], c3["bf16_conv_operands"]["ms_per_step"]))
    if rank == 0:
        out = {
            "metric": "image-text pairs/sec (train), CLIP-RN50 + BiGRU MoCo step, bs128/GPU",
            "value": args.steps * B * world / dt,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16" if prec == 1 else "f32",
            "data": "synthetic",
            "config": {
                "workload": "configs[1]: CLIP-%s + GRU, bs%d/GPU, 384x128 images, 64-token captions, MoCo queue %d, %s, %dxMI355X; random-init weights, synthetic inputs resident in HBM" % (
                    "RN50" if args.model == "m_resnet50" else "RN101", B, args.queue, "bf16 MFMA (TRID_GEMM_PRECISION=1)" if prec == 1 else "fp32", world),
                "global_batch": B * world,
                "parallelism": "dp%d" % world,
                "optimizer": "Adam (fused multi-tensor)",
                "step_launch": step_launch,
                "launch_probe": launch_probe,
                "host_enqueue_ms_per_step": t_host / args.steps * 1e3,
                "host_work_ms_per_step_no_backpressure": host_work_ms,
                "gemm_arithmetic": arith,
                "final_loss": loss_val,
            },
            "roofline": roofline,
            "roofline_second": roofline_1x1,
            "roofline_stream": roofline_stream,
        }
        if dp:
            # data-parallel accounting of the timed steps: one RCCL rank per GPU, gradient bytes all-reduced per step,
            # the fraction of them issued from INSIDE backward (overlapped) and the device time left exposed after it
            out["data_parallel"] = dict(dp_stats, rccl_ranks=world, backend=dist.get_backend(),
                                        embedding_allgather_bytes_per_rank=B * (4 * 256 + 2) * 4)
        out["replay_equals_eager_b%d" % B] = seam["equal"] if seam else None
        out["replay_vs_eager"] = seam
        out["gallery_encode"] = retr.pop("gallery_encode", None) if retr else None
        out["retrieval"] = retr
        out["queue_similarity"] = qsim
        out["configs3_1gpu"] = c3
        if not args.no_cpu_baseline and not dp:
            cb = cpu_baseline()
            pref = cb.pop("_parity_ref")
            out["cpu_baseline"] = cb
            out["parity_vs_oracle"] = parity_vs_oracle(device, pref)
            log("parity vs oracle at B=128: worst %.1e (%s) over %d quantities" % (out["parity_vs_oracle"]["worst_rel_err"],
                                                                                 out["parity_vs_oracle"]["worst_quantity"], out["parity_vs_oracle"]["quantities"]))
        else:
            out["cpu_baseline"] = None
            out["parity_vs_oracle"] = None
        real_stdout.write(json.dumps(out) + "\n")
        real_stdout.flush()
    if dp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
