"""Token-embedding + bias-free BiGRU + max-over-time text encoder on the HIP library.

Operator surface of the reference ``lib/models/backbones/gru.py`` (``GRU``
:8-88, ``build_gru`` :91-117): ``forward(list[Caption]) -> [B, 2*hidden]``,
``out_channels``, parameters ``gru.weight_{ih,hh}_l0[_reverse]`` (the nn.GRU is
a parameter holder; its forward is never called), frozen ``vocab_dict`` table as
a plain attribute.  Instead of sort/pack/pad (gru.py:66-82, one host sync), the
recurrence runs as a masked time loop: per step one batched fp32 MFMA GEMM
(h @ W_hh^T, both directions) and one fused gate/state/max kernel; the
zero-pad-enters-the-max behaviour of gru.py:63 is reproduced by the max init.
"""

import os

import numpy as np
import torch
from torch import nn

from .. import ops
from ..caption import CaptionBatch


def load_vocab_dict(root, use_onehot):
    names = {"bert_c4": "bert_vocab_c4.npy", "bert_l2": "bert_vocab_l2.npy", "clip_vit": "clip_vocab_vit.npy",
             "clip_rn50x4": "clip_vocab_rn50x4.npy"}
    if use_onehot not in names:
        raise NotImplementedError(use_onehot)
    return np.load(os.path.join(root, "./datasets/cuhkpedes", names[use_onehot]))


FUSED_GRU_STEP = os.environ.get("TRID_FUSED_GRU", "1") != "0"  # A/B switch: 0 = one GEMM + one cell kernel per step


class _GRUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, tokens, lengths, lmax, lmax_dev, save, w_ih_f, w_hh_f, w_ih_r, w_hh_r, emb_w=None, emb_b=None):
        # `save` comes from the caller: ctx.needs_input_grad ignores torch.no_grad()
        B = tokens.shape[0]
        H = w_hh_f.shape[1]
        E = w_ih_f.shape[1]
        L = lmax
        st = ops.stream()
        # the three input forms of gru.py:22-31,55-60: 0 = rows of the frozen table (every MoCo config); 1 = a TRAINABLE
        # nn.Embedding (`use_onehot == "yes"`); 2 = rows of the frozen table through nn.Linear(vocab_size, embed_size)
        mode = mod.embed_mode
        table = emb_w.detach() if mode == 1 else mod.vocab_dict
        x0 = None
        x = ops.empty((B * L, table.shape[1]), table)
        ops.call("trid_embedding_gather_f32", ops._p(table), ops._p(tokens), ops._p(x), B, L, tokens.stride(0), table.shape[1],
                 table.shape[0], st)
        if mode == 2:
            x0 = x
            x = ops.linear(x0, emb_w.detach(), emb_b.detach())  # [B*L, E]
        gi = ops.empty((B * L, 6 * H), table)
        # fp16-split arithmetic for the big text GEMMs while that is the library's conv mode: every gathered row is a row
        # of the frozen table, so max|x| <= max|table| (computed once per table)
        P = 16 if ops.conv_precision() == 16 else None
        a_x = None
        if P and mode != 0:
            a_x = ops.amax(x)  # (a trained table / the Linear's output: no bound carries over from step to step)
        elif P:
            if getattr(mod, "_table_amax", None) is None or mod._table_amax[0] is not table:
                mod._table_amax = (table, ops.amax(table))
            a_x = mod._table_amax[1]
        ops.gemm(x, w_ih_f, gi, B * L, 3 * H, E, E, E, 6 * H, precision=P, a_amax=a_x, b_amax=ops.amax(w_ih_f) if P else None)
        ops.gemm(x, w_ih_r, gi, B * L, 3 * H, E, E, E, 6 * H, c_off=3 * H, precision=P, a_amax=a_x,
                 b_amax=ops.amax(w_ih_r) if P else None)
        whh = torch.stack([w_hh_f.detach(), w_hh_r.detach()])  # [2,3H,H] (plumbing copy)
        h = torch.zeros(2, B, H, device=table.device)
        maxv = ops.empty((B, 2 * H), table)
        argt = torch.empty(B, 2 * H, dtype=torch.int32, device=table.device)
        ops.call("trid_gru_max_init_f32", ops._p(maxv), ops._p(argt), ops._p(lengths), L, ops._p(lmax_dev), B, H, st)
        gates = ops.empty((2, L, B, 4 * H), table) if save else None
        hprev = ops.empty((2, L, B, H), table) if save else None
        img_bytes = ops.L.load().trid_gru_whh_image_bytes(H) if FUSED_GRU_STEP else 0
        fused = None
        if img_bytes > 0:
            # ONE launch per step (gru_step.hip): W_hh split once into MFMA fragment images, the state carried as
            # packed fp16 planes between the steps
            wamax = ops.amax(whh)
            img_f = torch.empty(img_bytes, dtype=torch.uint8, device=table.device)
            img_b = torch.empty(img_bytes, dtype=torch.uint8, device=table.device)
            ops.call("trid_gru_pack_whh_f16", ops._p(whh), ops._p(wamax), ops._p(img_f), ops._p(img_b), H, st)
            Bp = (B + 15) // 16 * 16
            hp = torch.zeros(2, 2, Bp, H, dtype=torch.int32, device=table.device)
            for s in range(L):
                ops.call("trid_gru_step_fwd_f32", ops._p(img_f), ops._p(wamax), ops._p(hp[s & 1]), ops._p(hp[(s + 1) & 1]),
                         ops._p(h), ops._p(gi), ops._p(lengths), (ops._p(gates) + 4 * s * B * 4 * H) if save else None,
                         (ops._p(hprev) + 4 * s * B * H) if save else None, ops._p(maxv), ops._p(argt), s, L, L, B, Bp, H,
                         L * B * 4 * H, L * B * H, st)
            fused = (img_b, wamax)
        else:
            gh = ops.empty((2, B, 3 * H), table)
            for s in range(L):
                ops.gemm(h, whh, gh, B, 3 * H, H, H, H, 3 * H, batch=2, strideA=B * H, strideB=3 * H * H,
                         strideC=B * 3 * H)
                ops.call("trid_gru_cell_fwd_f32", ops._p(gi), ops._p(gh), ops._p(h), ops._p(lengths),
                         (ops._p(gates) + 4 * s * B * 4 * H) if save else None,
                         (ops._p(hprev) + 4 * s * B * H) if save else None, ops._p(maxv), ops._p(argt), s, L, L, B, H,
                         L * B * 4 * H, L * B * H, st)
        if save:
            ctx.saved = (x, whh, gates, hprev, argt, lengths, B, H, E, L, fused, a_x)
            ctx.embed = (mode, tokens, x0, table.shape[0], w_ih_f.detach(), w_ih_r.detach(), emb_w.detach() if emb_w is not None else None)
        return maxv

    @staticmethod
    def backward(ctx, dout):
        x, whh, gates, hprev, argt, lengths, B, H, E, L, fused, a_x = ctx.saved
        ctx.saved = None
        dout = dout.contiguous()
        st = ops.stream()
        dh = torch.zeros(2, B, H, device=dout.device)
        dGi = ops.empty((B * L, 6 * H), dout)
        dgh = ops.empty((2, L, B, 3 * H), dout)
        if fused is not None:
            img_b, wamax = fused
            nwg = ops.L.load().trid_gru_step_workgroups(B, H)
            amax = ops.empty((L + 1, nwg), dout)  # per-workgroup max|dgh| published by each step
            for s in range(L - 1, -1, -1):
                more = s + 1 < L
                ops.call("trid_gru_step_bwd_f32", ops._p(img_b), ops._p(wamax),
                         (ops._p(dgh) + 4 * (s + 1) * B * 3 * H) if more else None, (ops._p(amax) + 4 * (s + 1) * nwg) if more else None,
                         ops._p(amax) + 4 * s * nwg, ops._p(dout), ops._p(argt), ops._p(gates) + 4 * s * B * 4 * H,
                         ops._p(hprev) + 4 * s * B * H, ops._p(lengths), ops._p(dh), ops._p(dGi), ops._p(dgh) + 4 * s * B * 3 * H,
                         s, L, L, B, H, L * B * 4 * H, L * B * H, L * B * 3 * H, st)
        else:
            for s in range(L - 1, -1, -1):
                ops.call("trid_gru_cell_bwd_f32", ops._p(dout), ops._p(argt), ops._p(gates) + 4 * s * B * 4 * H,
                         ops._p(hprev) + 4 * s * B * H, ops._p(lengths), ops._p(dh), ops._p(dGi),
                         ops._p(dgh) + 4 * s * B * 3 * H, s, L, L, B, H, L * B * 4 * H, L * B * H, L * B * 3 * H, st)
                # dh[d] += dgh[d,s] @ W_hh[d]
                ops.gemm(dgh, whh, dh, B, H, 3 * H, 3 * H, H, H, b_mode=ops.B_NC, batch=2, strideA=L * B * 3 * H,
                         strideB=3 * H * H, strideC=B * H, accumulate=True, a_off=s * B * 3 * H)
        # dW_hh[d] = sum_{s,b} dgh[d,s,b,:]^T hprev[d,s,b,:]
        KK = L * B
        splits = max(1, min(8, KK // 512))
        dwhh = ops.empty((2, 3 * H, H), dout)
        dwih = ops.empty((2, 3 * H, E), dout)
        # operand magnitudes of the fp16-split form: |hprev| < 1 by construction; max|dgh| is the maximum the backward
        # steps published (dGi's r / z rows equal dgh's, its n row is dgh's divided by a gate <= 1: one streaming pass)
        P = 16 if (a_x is not None and fused is not None) else None
        kw_hh = kw_ih = {}
        if P:
            one = getattr(_GRUFn, "_one", None)
            if one is None or one.device != dout.device:
                one = _GRUFn._one = torch.ones(1, device=dout.device)
            a_dgh = torch.amax(amax[:L]).reshape(1)
            kw_hh = dict(precision=P, a_amax=a_dgh, b_amax=one)
            kw_ih = dict(precision=P, a_amax=ops.amax(dGi), b_amax=a_x)
        if splits == 1:
            ops.gemm(dgh, hprev, dwhh, 3 * H, H, KK, 3 * H, H, H, a_mode=ops.A_MC, b_mode=ops.B_NC, batch=2,
                     strideA=KK * 3 * H, strideB=KK * H, strideC=3 * H * H, **kw_hh)
            ops.gemm(dGi, x, dwih, 3 * H, E, KK, 6 * H, E, E, a_mode=ops.A_MC, b_mode=ops.B_NC, batch=2,
                     strideA=3 * H, strideB=0, strideC=3 * H * E, **kw_ih)
        else:
            slab = ops.empty((splits, 2, 3 * H, max(H, E)), dout)
            n = 2 * 3 * H * H
            ops.gemm(dgh, hprev, slab, 3 * H, H, KK, 3 * H, H, H, a_mode=ops.A_MC, b_mode=ops.B_NC, batch=2,
                     strideA=KK * 3 * H, strideB=KK * H, strideC=3 * H * H, splits=splits, strideSplit=n, **kw_hh)
            ops.call("trid_slab_reduce_f32", ops._p(slab), ops._p(dwhh), n, splits, n, 0, st)
            n = 2 * 3 * H * E
            ops.gemm(dGi, x, slab, 3 * H, E, KK, 6 * H, E, E, a_mode=ops.A_MC, b_mode=ops.B_NC, batch=2,
                     strideA=3 * H, strideB=0, strideC=3 * H * E, splits=splits, strideSplit=n, **kw_ih)
            ops.call("trid_slab_reduce_f32", ops._p(slab), ops._p(dwih), n, splits, n, 0, st)
        mode, tokens, x0, vocab, w_ih_f, w_ih_r, emb_w = ctx.embed
        ctx.embed = None
        d_emb_w = d_emb_b = None
        if mode != 0:
            # the input of the recurrence is itself a function of parameters: dX = dGi_fwd W_ih + dGi_rev W_ih_reverse
            dX = ops.matmul_nn(dGi[:, : 3 * H], w_ih_f)
            ops.matmul_nn(dGi[:, 3 * H :], w_ih_r, out=dX, accumulate=True)
            if mode == 1:  # nn.Embedding(padding_idx=0): rows summed per token, deterministic (csrc/attn_text.hip)
                d_emb_w = ops.empty((vocab, E), dout)
                ops.call("trid_embedding_bwd_f32", ops._p(dX), ops._p(tokens), B, L, tokens.stride(0), E, ops._p(d_emb_w), vocab, 0, st)
            else:          # nn.Linear on the frozen rows: dW = dX^T x0, db = column sums
                d_emb_w = ops.matmul_tn(dX, x0)
                d_emb_b = ops.colsum(dX)
        return None, None, None, None, None, None, dwih[0], dwhh[0], dwih[1], dwhh[1], d_emb_w, d_emb_b


class GRU(nn.Module):
    def __init__(self, hidden_dim, vocab_size, embed_size, num_layers, drop_out, bidirectional, use_onehot, root,
                 vocab_dict=None):
        super().__init__()
        if num_layers != 1 or not bidirectional:
            raise NotImplementedError("HIP text encoder covers the 1-layer bidirectional GRU of the shipped configs")
        self.use_onehot = use_onehot
        # word embedding: the three forms of gru.py:22-34
        if use_onehot == "yes":
            self.embed = nn.Embedding(vocab_size, embed_size, padding_idx=0)  # a trainable table (parameter holder: its forward is never called)
            self.vocab_dict = None
            self.embed_mode = 1
        else:
            self.embed = None if vocab_size == embed_size else nn.Linear(vocab_size, embed_size)
            self.embed_mode = 0 if self.embed is None else 2
            if vocab_dict is None:
                vocab_dict = load_vocab_dict(root, use_onehot)
            vocab_dict = torch.as_tensor(vocab_dict).float()
            assert vocab_size == vocab_dict.shape[1]
            dev = "cuda" if torch.cuda.is_available() else "cpu"
            self.vocab_dict = vocab_dict.to(dev).contiguous()  # plain attribute, as gru.py:34
        self.gru = nn.GRU(embed_size, hidden_dim, num_layers=num_layers, dropout=drop_out,
                          bidirectional=bidirectional, bias=False)
        self.out_channels = hidden_dim * 2

    def forward(self, captions):
        cb = CaptionBatch.from_list(captions)
        if not cb.tokens.is_cuda:
            raise RuntimeError("textreid_amd.GRU runs on the HIP kernel library only (CUDA tensors); no CPU fallback")
        if self.vocab_dict is not None and self.vocab_dict.device != cb.tokens.device:
            self.vocab_dict = self.vocab_dict.to(cb.tokens.device)
        g = self.gru
        ws = (g.weight_ih_l0, g.weight_hh_l0, g.weight_ih_l0_reverse, g.weight_hh_l0_reverse)
        extra = () if self.embed is None else ((self.embed.weight, None) if self.embed_mode == 1 else (self.embed.weight, self.embed.bias))
        save = torch.is_grad_enabled() and any(w.requires_grad for w in ws + tuple(e for e in extra if e is not None))
        # cb.max_len is the time-loop length; when it is only an upper BOUND of the batch maximum (bound_only: a recorded
        # step replayed on other captions) the zero-pad quirk of gru.py:63 takes the true maximum from the device
        lmax_dev = cb.lengths.max().reshape(1) if getattr(cb, "bound_only", False) else None
        return _GRUFn.apply(self, cb.tokens.contiguous(), cb.lengths.contiguous(), cb.max_len, lmax_dev, save, *ws, *extra)


def build_gru(cfg, bidirectional, vocab_dict=None):
    model = GRU(cfg.MODEL.GRU.NUM_UNITS, cfg.MODEL.GRU.VOCABULARY_SIZE, cfg.MODEL.GRU.EMBEDDING_SIZE,
                cfg.MODEL.GRU.NUM_LAYER, 1 - cfg.MODEL.GRU.DROPOUT_KEEP_PROB, bidirectional, cfg.MODEL.GRU.ONEHOT,
                cfg.ROOT, vocab_dict=vocab_dict)
    if cfg.MODEL.FREEZE:
        model.gru.eval()
        for p in model.gru.parameters():
            p.requires_grad = False
    return model
