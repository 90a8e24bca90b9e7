"""TEST INFRASTRUCTURE (CPU, oracle side): construction of the full-size one-step cases and the comparison of a
train step against the oracle's.  Shared by tests/test_model_gpu.py, tests/dp_full_worker.py, tools/pick_fullstep_seed.py
and the `cpu_baseline` leg of bench.py (which reports `parity_vs_oracle` from its first timed step) - never imported
by the product (`textreid_amd/`).

A case = the state a reference `MoCoHead` would hold (`lib/models/embeddings/moco_head/head.py:20-71`) filled from
`oracle.fill` in the `margin` style, a frozen token-embedding table (`lib/models/backbones/gru.py:34`), and a batch
shaped like the reference's collate output (`lib/data/collate_batch.py:4-9`): images [B,3,384,128], token ids padded to
105 with ragged lengths, person ids with 4 captions per id (`lib/config/defaults.py:21`)."""

import torch

from . import fill as OF
from . import head as OH

# Smallest |ReLU input| a full-size case must keep on the fp32 oracle, by batch size.  The margin style puts the BULK of
# the pre-activations 8 sigma from the kink; what is left near zero are zero-padding border pixels, and their number
# grows with the batch: at B = 128 (1.5e9 ReLU inputs) no seed of the first eight reaches 2e-4, the best holds 7e-5 -
# still several times the forward error of either side there (~1e-5), which is all the margin is for.
RELU_MIN_BY_BATCH = {128: 5e-5}


def synth_batch(B, step, seed, vocab=49408, Lpad=105, L=64):
    """bench.py's synthetic batch, on the CPU."""
    g = torch.Generator(device="cpu").manual_seed(seed + 7919 * step)
    images = torch.randn(B, 3, 384, 128, generator=g)
    tokens = torch.zeros(B, Lpad, dtype=torch.int64)
    tokens[:, :L] = torch.randint(1, vocab, (B, L), generator=g)
    lengths = torch.full((B,), L, dtype=torch.int64)
    ids = torch.arange(B, dtype=torch.int64) // 4 + step * (B // 4)
    return images, tokens, lengths, ids


def full_step_case(spec, B, K, vocab, seed, style="margin"):
    """(`style`-filled head state, embedding table, images, tokens, lengths, ids).  tools/pick_fullstep_seed.py chooses
    the seeds so that the query encoder's smallest |ReLU input| clears `relu_floor(B)`."""
    torch.manual_seed(0)
    table = torch.randn(vocab, 512) * 0.02
    shapes = OH.state_shapes(spec, K)
    st = {}
    for k, s_ in shapes.items():
        if k.endswith("num_batches_tracked"):
            st[k] = torch.zeros((), dtype=torch.int64)
        elif k in ("id_queue", "queue_ptr"):
            st[k] = torch.zeros(s_, dtype=torch.int64)
        else:
            st[k] = OF.fill("full." + k, s_, seed, style=style)
    OH.init_queues(st, seed)
    images, tokens, lengths, ids = synth_batch(B, 0, 5, vocab=vocab)
    lengths = torch.tensor(([64, 40, 64, 9, 33, 64, 12, 64, 50, 64, 21, 64, 64, 7, 64, 30] * ((B + 15) // 16))[:B])
    for i, n in enumerate(lengths.tolist()):
        tokens[i, n:] = 0
    return st, table, images, tokens, lengths, ids


def relu_floor(B):
    return RELU_MIN_BY_BATCH.get(B, OF.RELU_MIN)


def oracle_step(spec, st0, table, images, tokens, lengths, ids, dtype=torch.float32, taps=None):
    """One oracle train step (forward of all four encoders, the three losses, backward; no optimizer) from state st0
    in `dtype`: (losses, {name: gradient}, state after the step).  `taps` as in `oracle.head.train_forward`."""
    st = {k: (v.to(dtype).clone() if v.dtype.is_floating_point else v.clone()) for k, v in st0.items()}
    tr = OH.trainable_names(st)
    for k in tr:
        st[k].requires_grad_(True)
    ld = OH.train_forward(st, spec, table.to(dtype), images.to(dtype), tokens, lengths, ids, m=0.999, epsilon=0.1, taps=taps)
    sum(ld.values()).backward()
    return {k: v.detach() for k, v in ld.items()}, {k: st[k].grad for k in tr}, {k: v.detach() for k, v in st.items()}


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def step_errors(losses, grad_of, state, ref):
    """{name: relative error} of a step's (losses, gradients via grad_of(name), post-step state dict) against the
    oracle triple `ref`: the three losses, EVERY trainable gradient in full (against max(max|ref|, gradient floor):
    analytically-zero gradients hold rounding noise on both sides, see oracle.fill.grad_floor), both queues, every
    momentum-updated key parameter, every BatchNorm running statistic of all four encoders."""
    rl, rg, rs = ref
    errs = {"loss:" + k: rel(losses[k], rl[k]) for k in rl}
    gfl = 1e-3 * max(float(g.abs().max()) for g in rg.values())
    for k, g in rg.items():
        fl = gfl * (100.0 if k.endswith("attnpool.k_proj.bias") else 1.0)  # softmax shift invariance: exactly zero
        errs["grad:" + k] = float((grad_of(k).detach().cpu().double() - g.double()).abs().max() / max(float(g.abs().max()), fl))
    for k, v in rs.items():
        if k.startswith(("v_encoder_k.", "t_encoder_k.")) and v.dtype.is_floating_point or k.endswith(("running_mean", "running_var")) or k in ("v_queue", "t_queue"):
            errs["state:" + k] = rel(state[k], v)
    return errs
