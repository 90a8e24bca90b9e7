"""Losses of the MoCo configs on the HIP kernel library.

Same call surface as the reference ``lib/models/losses.py``:
``instance_loss`` (:42-62, with ``CrossEntropyLabelSmooth`` :6-39),
``global_align_loss`` (:102-128) and ``infonce_loss`` (:206-217); plus
``queue_infonce_loss``, the fused form the MoCo head uses (queue similarity +
batch-wide negative filter + InfoNCE, reference head.py:148-170 followed by
losses.py:206-217) that never builds the gathered negative matrices.

Each loss is one ``autograd.Function``: the forward launches the similarity /
logit GEMMs and a row kernel that emits the per-row loss and, in place,
dL/dlogits; the gradient GEMMs run right away (the logits never outlive the
forward) and backward only scales by the incoming gradient on device.
"""

import os

import torch

from . import ops
from .ops import call, _p, stream


def _pad16(n):
    return (n + 15) // 16 * 16


def _scaled(g, a):
    """g (0-d device tensor) * a, through the axpby3 kernel (no host sync)."""
    out = torch.empty_like(a)
    gs = g.reshape(1).float().contiguous()  # (a local: temporaries must outlive the call that takes their address)
    call("trid_axpby3_f32", _p(out), _p(a), None, None, _p(gs), a.numel(), stream())
    return out


def _check(*ts):
    for t in ts:
        if not t.is_cuda:
            raise RuntimeError("textreid_amd losses run on the HIP kernel library only (CUDA tensors); no CPU fallback")


class _InstanceLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, projection, v, t, labels, epsilon, scale, norm):
        C, N = projection.shape
        B = v.shape[0]
        ldn = _pad16(N)
        proj = projection.detach().contiguous()
        pnt = ops.empty((ldn, C), v)
        invn = ops.empty((N,), v)
        call("trid_colnorm_f32", _p(proj), None, _p(pnt), _p(invn), C, N, ldn, stream())
        E2 = torch.cat([v.detach(), t.detach()], dim=0).contiguous()
        if norm:  # losses.py:45-47: the embeddings L2-normalised as well
            E2, inv_e = ops.l2norm_rows(E2)
        logits = ops.linear(E2, pnt, alpha=float(scale))  # [2B, ldn] (losses.py:52-53: scale * embed @ normalize(projection))
        labels2 = torch.cat([labels, labels]).long().contiguous()
        rows = ops.empty((2 * B,), v)
        call("trid_smooth_ce_rows_f32", _p(logits), _p(labels2), _p(rows), 2 * B, N, ldn, float(epsilon), 1.0 / B,
             stream())
        loss = ops.empty((1,), v)
        ops.sum_to(rows, loss, 1.0 / B)
        dE2 = ops.matmul_nn(logits, pnt, alpha=float(scale))
        dpnt = ops.matmul_tn(logits, E2, alpha=float(scale))
        if norm:
            dE2 = ops.l2norm_rows_bwd(dE2, E2, inv_e)
        dproj = ops.empty((C, N), v)
        call("trid_colnorm_bwd_f32", _p(dpnt), _p(pnt), _p(invn), _p(dproj), C, N, ldn, stream())
        ctx.saved = (dproj, dE2, B)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        dproj, dE2, B = ctx.saved
        ctx.saved = None
        return _scaled(g, dproj), _scaled(g, dE2[:B]), _scaled(g, dE2[B:]), None, None, None, None


def instance_loss(projection, visual_embed, textual_embed, labels, scale=1, norm=False, epsilon=0.0):
    """lib/models/losses.py:42-62.  The reference builds CrossEntropyLabelSmooth(num_classes=...) WITHOUT passing its own
    `epsilon` on (losses.py:56): any epsilon > 0 smooths with that class's default weight 0.1 (losses.py:18) - reproduced."""
    _check(projection, visual_embed, textual_embed, labels)
    eps_eff = 0.1 if epsilon > 0 else 0.0
    return _InstanceLossFn.apply(projection, visual_embed, textual_embed, labels, eps_eff, scale, bool(norm))


class _GlobalAlignFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v, t, labels, alpha, beta, scale_pos, scale_neg):
        B, C = v.shape
        vn, inv_v = ops.l2norm_rows(v.detach().contiguous())
        tn, inv_t = ops.l2norm_rows(t.detach().contiguous())
        Bp = (B + 3) // 4 * 4
        if Bp != B:  # the [B,B] matrix is a GEMM operand later (leading dimension % 4): zero-pad the batch
            vp, tp = torch.zeros(Bp, C, device=v.device), torch.zeros(Bp, C, device=v.device)
            vp[:B].copy_(vn)
            tp[:B].copy_(tn)
        else:
            vp, tp = vn, tn
        S = ops.linear(vp, tp)  # [Bp,Bp] cosine (padding rows / columns are zero and stay zero)
        rows = ops.empty((B,), v)
        lab = labels.long().contiguous()
        call("trid_global_align_rows_f32", _p(S), _p(lab), _p(rows), B, Bp, float(alpha),
             float(beta), float(scale_pos), float(scale_neg), 1.0, stream())
        loss = ops.empty((1,), v)
        ops.sum_to(rows, loss, 1.0)
        dvn = ops.matmul_nn(S, tp)[:B]
        dtn = ops.matmul_tn(S, vp)[:B]
        dv = ops.l2norm_rows_bwd(dvn.contiguous(), vn, inv_v)
        dt = ops.l2norm_rows_bwd(dtn.contiguous(), tn, inv_t)
        ctx.saved = (dv, dt)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        dv, dt = ctx.saved
        ctx.saved = None
        return _scaled(g, dv), _scaled(g, dt), None, None, None, None, None


def global_align_loss(visual_embed, textual_embed, labels, alpha=0.6, beta=0.4, scale_pos=10, scale_neg=40):
    _check(visual_embed, textual_embed, labels)
    return _GlobalAlignFn.apply(visual_embed, textual_embed, labels, alpha, beta, scale_pos, scale_neg)


class _InfoNCEFn(torch.autograd.Function):
    """CE over [pos | neg]/T with label 0 for materialised logits (reference call surface)."""

    @staticmethod
    def forward(ctx, v_pos, v_neg, t_pos, t_neg, T):
        B = v_pos.shape[0]
        loss = ops.empty((1,), v_pos)
        outs = []
        for i, (pos, neg) in enumerate(((v_pos, v_neg), (t_pos, t_neg))):
            S = neg.detach().clone().contiguous()
            K = S.shape[1]
            hit = torch.zeros(K, dtype=torch.uint8, device=S.device)
            rows = ops.empty((B,), S)
            dpos = ops.empty((B,), S)
            ws = ops.empty((ops.L.load().trid_infonce_ws_floats(B, K),), S)
            posv = pos.detach().reshape(-1).contiguous()
            call("trid_infonce_rows_f32", _p(S), _p(posv), _p(hit), _p(rows), _p(dpos),
                 B, K, K, 1.0 / T, 1.0, _p(ws), stream())
            ops.sum_to(rows, loss, 1.0 / B, accumulate=i > 0)
            outs += [dpos.view(B, 1), S]
        ctx.saved = outs
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        dvp, dvn, dtp, dtn = ctx.saved
        ctx.saved = None
        return _scaled(g, dvp), _scaled(g, dvn), _scaled(g, dtp), _scaled(g, dtn), None


def infonce_loss(v_pos, v_neg, t_pos, t_neg, T=0.07):
    _check(v_pos, v_neg, t_pos, t_neg)
    return _InfoNCEFn.apply(v_pos, v_neg, t_pos, t_neg, T)


FUSED_QUEUE_NCE = os.environ.get("TRID_FUSED_QNCE", "1") != "0"  # A/B switch: 0 = GEMM + row kernels
QUEUE_NCE_WGS = int(os.environ.get("TRID_QNCE_WGS", "0"))            # workgroups per modality (0 = library default)
# loss rows folded by the finish launch's last workgroup (a ticket word + one agent-scope release per workgroup) instead of a
# third launch: MEASURED SLOWER on the MI355X - 38.0 vs 35.7 us at K = 8192, 74.6 vs 72.0 us at K = 65536
# (profiles/r04f_qsim_tune.txt): 256 release fences cost more than the 4.5 us trid_sum_f32 launch they replace.  Off.
FUSED_QUEUE_SUM = os.environ.get("TRID_QNCE_FUSED_SUM", "0") != "0"


class _QueueInfoNCEFn(torch.autograd.Function):
    """v_q,t_q [B,C] normalised queries; v_k,t_k keys; queues row-major [K,C]."""

    @staticmethod
    def forward(ctx, v_q, t_q, v_k, t_k, ids, t_queue, v_queue, id_queue, T, unit_norm):
        B, C = v_q.shape
        K = t_queue.shape[0]
        loss = ops.empty((1,), v_q)
        rows = ops.empty((2, B), v_q)  # per-row losses of both modalities, summed by one launch
        ids = ids.long().contiguous()
        nws = ops.L.load().trid_queue_nce_ws_floats(B, K, C, QUEUE_NCE_WGS) if (FUSED_QUEUE_NCE and unit_norm) else 0
        if nws > 0:
            # ONE pass over both queues: negative filter, similarity, softmax and dL/dq fused, no [B,K] anywhere (queue_nce.hip)
            dq = ops.empty((2, B, C), v_q)
            ws = ops.empty((nws,), v_q)
            vq_, tq_, vk_, tk_ = (x.detach().contiguous() for x in (v_q, t_q, v_k, t_k))  # alive until the call returns
            # (the ticket word: a fresh zero from the amax-slot pool - zero-filled once per 4096 calls, re-recorded with a
            # captured step - so the finish launch folds the loss rows itself: two launches for the whole block)
            ticket = ops.amax_slot(v_q.device) if FUSED_QUEUE_SUM else None
            call("trid_queue_nce_f32", _p(vq_), _p(tq_), _p(vk_),
                 _p(tk_), _p(t_queue), _p(v_queue), _p(id_queue), _p(ids), _p(rows), _p(dq), B, K, C, 1.0 / T,
                 1.0, 1.0, {1: 1, 3: 3}.get(ops.GEMM_PRECISION, 6), QUEUE_NCE_WGS, _p(ws), _p(ticket), _p(loss) if ticket is not None else None,
                 1.0 / B, stream())
            if ticket is None:
                ops.sum_to(rows.view(-1), loss, 1.0 / B)
            ctx.saved = [dq[0], dq[1]]
            return loss[0]
        hit = torch.empty(K, dtype=torch.uint8, device=v_q.device)
        call("trid_queue_hit_mask", _p(id_queue), _p(ids), _p(hit), K, B, stream())
        grads = []
        for i, (q, key, queue) in enumerate(((v_q, t_k, t_queue), (t_q, v_k, v_queue))):
            q = q.detach().contiguous()
            key = key.detach().contiguous()
            S = ops.linear(q, queue)  # [B,K] raw dot products
            dq = ops.empty((B, C), q)
            ws = ops.empty((ops.L.load().trid_infonce_ws_floats(B, K),), q)
            # loss rows, dL/dS in place, and dq = dL/dpos * key (the positive-pair term)
            call("trid_infonce_queue_rows_f32", _p(S), _p(q), _p(key), _p(hit), _p(rows[i]), _p(dq), B, K, K, C, 1.0 / T, 1.0,
                 _p(ws), stream())
            ops.matmul_nn(S, queue, out=dq, accumulate=True)
            grads.append(dq)
        ops.sum_to(rows.view(-1), loss, 1.0 / B)
        ctx.saved = grads
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        dv, dt = ctx.saved
        ctx.saved = None
        return _scaled(g, dv), _scaled(g, dt), None, None, None, None, None, None, None, None


def queue_infonce_loss(v_q, t_q, v_k, t_k, ids, t_queue, v_queue, id_queue, T=0.07, unit_norm=True):
    """InfoNCE of both modalities against the [K,C] row-major queues with the batch-wide negative filter.
    ``unit_norm``: queries, keys and queue rows are L2-normalised (they are in the MoCo head, head.py:128-145) -
    the fused single-pass kernel relies on |logit| <= 1/T; pass False for arbitrary inputs (GEMM + row kernels)."""
    _check(v_q, t_q, v_k, t_k, ids, t_queue, v_queue, id_queue)
    if not (t_queue.is_contiguous() and v_queue.is_contiguous()):
        raise RuntimeError("queue_infonce_loss: queues must be row-major [K,C] contiguous")
    return _QueueInfoNCEFn.apply(v_q, t_q, v_k, t_k, ids, t_queue, v_queue, id_queue, T, unit_norm)


class _L2NormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y, inv = ops.l2norm_rows(x.detach().contiguous())
        ctx.saved = (y, inv)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, inv = ctx.saved
        ctx.saved = None  # y is this node's own output: holding it would keep a reference cycle alive until the cyclic GC runs
        return ops.l2norm_rows_bwd(dy.contiguous(), y, inv)


def l2_normalize(x):
    """F.normalize(x, dim=1) (head.py:128-129)."""
    _check(x)
    return _L2NormFn.apply(x)


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        xd = x.detach().contiguous()
        ctx.saved = (xd, w)
        return ops.linear(xd, w.detach(), b.detach() if b is not None else None)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved
        ctx.saved = None
        dy = dy.contiguous()
        dx = ops.matmul_nn(dy, w.detach()) if ctx.needs_input_grad[0] else None
        dw = ops.matmul_tn(dy, x) if ctx.needs_input_grad[1] else None
        db = ops.colsum(dy) if ctx.needs_input_grad[2] else None
        return dx, dw, db


def linear(x, weight, bias=None):
    """F.linear on the MFMA GEMM (embed layers, head.py:50-51)."""
    _check(x, weight)
    return _LinearFn.apply(x, weight, bias)


class _MlpFn(torch.autograd.Function):
    """Linear -> ReLU -> Linear (the `*_fc_q` / `*_fc_k` projection heads of MODEL.MOCO.FC, head.py:33-42): the ReLU
    rides in the first GEMM's epilogue, its backward is one masking kernel."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        xd = x.detach().contiguous()
        h = ops.linear(xd, w1.detach(), b1.detach(), relu=True)
        ctx.saved = (xd, h, w1, w2)
        return ops.linear(h, w2.detach(), b2.detach())

    @staticmethod
    def backward(ctx, dy):
        x, h, w1, w2 = ctx.saved
        ctx.saved = None
        dy = dy.contiguous()
        dh = ops.matmul_nn(dy, w2.detach())
        dhm = torch.empty_like(dh)
        call("trid_relu_bwd_f32", _p(dh), _p(h), _p(dhm), dh.numel(), stream())
        dx = ops.matmul_nn(dhm, w1.detach()) if ctx.needs_input_grad[0] else None
        return dx, ops.matmul_tn(dhm, x), ops.colsum(dhm), ops.matmul_tn(dy, h), ops.colsum(dy)


def mlp(x, fc):
    """fc: nn.Sequential(Linear, ReLU, Linear) used as a parameter holder."""
    _check(x)
    return _MlpFn.apply(x, fc[0].weight, fc[0].bias, fc[2].weight, fc[2].bias)
