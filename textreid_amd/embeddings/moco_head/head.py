"""Cross-modal MoCo head on the HIP kernel library.

Operator surface of the reference ``lib/models/embeddings/moco_head/head.py``
(``MoCoHead`` :10-183, ``build_moco_head`` :186): ``forward(images, captions)``
returns ``{"instance_loss","infonce_loss","global_align_loss"}`` in training and
``[v_embed, t_embed]`` in eval; same parameter / buffer names and shapes
(queues ``[C,K]``, ``id_queue [1,K]``, ``queue_ptr [1]``) so reference
checkpoints load.  ``FC=False`` (the shipped MoCo configs).

MI355X design points: the feature queues live row-major ``[K,C]`` in HBM (the
registered ``[C,K]`` buffers are transposed views of that storage), so the
similarity GEMM streams 1 KB rows and the enqueue is one contiguous slab write
at a device-resident pointer (no ``int(queue_ptr)`` sync); the momentum update
is a single multi-tensor kernel; the batch-wide negative filter is a per-column
flag consumed by the InfoNCE row kernel instead of nonzero/unique/gather.
"""

import copy

import numpy as np
import os

import torch
import torch.nn as nn

from ... import losses, ops
from ...caption import CaptionBatch
from ...parallel import dp_active, gather_embeddings
from .loss import make_loss_evaluator

EMA_CHUNK = 65536


class _EmaPlan:
    """Device pointer / chunk tables for the multi-tensor EMA kernel."""

    def __init__(self, pairs, device):
        k_ptrs = np.array([k.data_ptr() for _, k in pairs], dtype=np.uint64)
        q_ptrs = np.array([q.data_ptr() for q, _ in pairs], dtype=np.uint64)
        sizes = np.array([k.numel() for _, k in pairs], dtype=np.int64)
        ct, co = [], []
        for i, n in enumerate(sizes):
            for off in range(0, int(n), EMA_CHUNK):
                ct.append(i)
                co.append(off)
        self.key = (tuple(k_ptrs.tolist()), tuple(q_ptrs.tolist()))
        self.k_ptrs = torch.from_numpy(k_ptrs.view(np.int64)).to(device)
        self.q_ptrs = torch.from_numpy(q_ptrs.view(np.int64)).to(device)
        self.sizes = torch.from_numpy(sizes).to(device)
        self.chunk_tensor = torch.tensor(ct, dtype=torch.int32, device=device)
        self.chunk_off = torch.tensor(co, dtype=torch.int64, device=device)
        self.n_chunks = len(ct)


class MoCoHead(nn.Module):
    def __init__(self, cfg, visual_model, textual_model):
        super().__init__()
        self.embed_size = cfg.MODEL.EMBEDDING.FEATURE_SIZE
        self.K = cfg.MODEL.MOCO.K
        self.m = cfg.MODEL.MOCO.M
        self.fc = cfg.MODEL.MOCO.FC
        self.v_encoder_q = visual_model
        self.t_encoder_q = textual_model
        self.v_encoder_k = copy.deepcopy(visual_model)
        self.t_encoder_k = copy.deepcopy(textual_model)
        for p in self.v_encoder_k.parameters():
            p.requires_grad = False
        for p in self.t_encoder_k.parameters():
            p.requires_grad = False
        if self.fc:  # head.py:32-49: two-layer projection heads for the contrastive branch, momentum copies for the keys
            mk = lambda cin: nn.Sequential(nn.Linear(cin, self.embed_size), nn.ReLU(), nn.Linear(self.embed_size, self.embed_size))
            self.v_fc_q = mk(visual_model.out_channels)
            self.t_fc_q = mk(textual_model.out_channels)
            self.v_fc_k = copy.deepcopy(self.v_fc_q)
            self.t_fc_k = copy.deepcopy(self.t_fc_q)
            for p in list(self.v_fc_k.parameters()) + list(self.t_fc_k.parameters()):
                p.requires_grad = False
        self.v_embed_layer = nn.Linear(visual_model.out_channels, self.embed_size)
        self.t_embed_layer = nn.Linear(textual_model.out_channels, self.embed_size)
        # queues: storage [K,C] row-major, registered as the reference-shaped [C,K] transposed views
        tq = torch.nn.functional.normalize(torch.rand(self.embed_size, self.K), dim=0)
        vq = torch.nn.functional.normalize(torch.rand(self.embed_size, self.K), dim=0)
        self.register_buffer("t_queue", tq.t().contiguous().t())
        self.register_buffer("v_queue", vq.t().contiguous().t())
        self.register_buffer("id_queue", -torch.ones((1, self.K), dtype=torch.long))
        self.register_buffer("queue_ptr", torch.zeros(1, dtype=torch.long))
        self.loss_evaluator = make_loss_evaluator(cfg)
        self._ema_plan = None
        self._init_weight()

    def _init_weight(self):
        # head.py:64-71 iterates self.modules(), which includes the encoders' attnpool projections
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.kaiming_normal_(m.weight, a=0, mode="fan_out")
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm1d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    # ---------------------------------------------------------------- state helpers
    def _queue_kc(self, name):
        """Row-major [K,C] storage behind the registered [C,K] buffer."""
        buf = getattr(self, name)
        kc = buf.t()
        if not kc.is_contiguous():  # e.g. after a load that replaced the buffer: re-layout once
            kc = kc.contiguous()
            setattr(self, name, kc.t())
        return kc

    def _ema_pairs(self):
        pairs = list(zip(self.v_encoder_q.parameters(), self.v_encoder_k.parameters()))
        pairs += list(zip(self.t_encoder_q.parameters(), self.t_encoder_k.parameters()))
        if self.fc:  # head.py:86-94
            pairs += list(zip(self.v_fc_q.parameters(), self.v_fc_k.parameters()))
            pairs += list(zip(self.t_fc_q.parameters(), self.t_fc_k.parameters()))
        return pairs

    @torch.no_grad()
    def _momentum_update_key_encoder(self):
        pairs = self._ema_pairs()
        key = (tuple(k.data_ptr() for _, k in pairs), tuple(q.data_ptr() for q, _ in pairs))
        if self._ema_plan is None or self._ema_plan.key != key:
            for q, k in pairs:
                if not (q.is_contiguous() or q.is_contiguous(memory_format=torch.channels_last)) or q.stride() != k.stride():
                    raise RuntimeError("EMA pairs must share a dense memory layout")
            self._ema_plan = _EmaPlan(pairs, pairs[0][0].device)
        pl = self._ema_plan
        ops.call("trid_ema_multi_f32", ops._p(pl.k_ptrs), ops._p(pl.q_ptrs), ops._p(pl.sizes), ops._p(pl.chunk_tensor),
                 ops._p(pl.chunk_off), pl.n_chunks, EMA_CHUNK, float(self.m), 1.0 - float(self.m), ops.stream())
        ops.note_parameter_write()  # raw-pointer write: torch's tensor._version does not move

    @torch.no_grad()
    def _dequeue_and_enqueue(self, v_keys, t_keys, id_keys):
        B = v_keys.shape[0]
        assert self.K % B == 0  # head.py:101
        vq, tq = self._queue_kc("v_queue"), self._queue_kc("t_queue")
        vk, tk, ik = v_keys.contiguous(), t_keys.contiguous(), id_keys.long().contiguous()  # alive until the call returns
        ops.call("trid_enqueue_f32", ops._p(vq), ops._p(tq), ops._p(self.id_queue), ops._p(self.queue_ptr),
                 ops._p(vk), ops._p(tk), ops._p(ik), self.K,
                 self.embed_size, B, ops.stream())

    # ---------------------------------------------------------------- forward
    def _side_stream(self, device, which="_text_stream"):
        if os.environ.get("TRID_SERIAL", "0") == "1":  # experiment (tools/exp/r04_trace.sh): one stream, un-overlapped kernel durations
            return torch.cuda.current_stream(device)
        st = getattr(self, which, None)
        if st is None or st.device != device:
            st = torch.cuda.Stream(device=device)
            setattr(self, which, st)
        return st

    def forward(self, images, captions):
        cb = CaptionBatch.from_list(captions)
        if not images.is_cuda:
            raise RuntimeError("textreid_amd.MoCoHead runs on the HIP kernel library only (CUDA tensors); no CPU fallback")
        if self.training:
            # The text encoders are a chain of ~130 tiny launch-bound kernels per pass: they run on a
            # side HIP stream underneath the (CU-filling) image encoders.  The momentum update only
            # reads query parameters, which do not change during the forward, so doing it first is
            # equivalent to the reference order (head.py:133) and lets the key text encoder start early.
            main = torch.cuda.current_stream()
            side = self._side_stream(images.device)
            with torch.no_grad():
                self._momentum_update_key_encoder()
            # Issue order = how soon each stream has work: the query image encoder (the head of the critical path:
            # its backward follows) is enqueued FIRST, so the side streams wait on an event recorded right after the
            # momentum update instead of on whatever the main stream holds by the time the host gets to them.
            ema_done = torch.cuda.Event()
            ema_done.record(main)
            order = os.environ.get("TRID_ISSUE_ORDER", "qtk")
            side_k = self._side_stream(images.device, "_key_stream")

            def issue(which):
                nonlocal t_feat, tk_feat, vk_feat, v_feat
                if which == "q":
                    v_feat = self.v_encoder_q(images)
                elif which == "k":
                    # The key image encoder (no_grad) runs on a second side stream: its HBM-bound
                    # BatchNorm / pooling passes overlap the MFMA-bound GEMMs of the query encoder.
                    if os.environ.get("TRID_SERIAL_K", "0") == "1":  # experiment: key encoder on the main stream
                        with torch.no_grad():
                            vk_feat = self.v_encoder_k(images)
                        return
                    side_k.wait_event(ema_done)
                    with torch.cuda.stream(side_k), torch.no_grad():
                        vk_feat = self.v_encoder_k(images)
                else:
                    side.wait_event(ema_done)
                    with torch.cuda.stream(side):
                        t_feat = self.t_encoder_q(cb)
                        with torch.no_grad():
                            tk_feat = self.t_encoder_k(cb)

            t_feat = tk_feat = vk_feat = v_feat = None
            for which in order:
                issue(which)
            main.wait_stream(side)
            main.wait_stream(side_k)
            t_feat.record_stream(main)
            tk_feat.record_stream(main)
            vk_feat.record_stream(main)
            v_embed = losses.linear(v_feat, self.v_embed_layer.weight, self.v_embed_layer.bias)
            t_embed = losses.linear(t_feat, self.t_embed_layer.weight, self.t_embed_layer.bias)
            id_q = cb.ids.long()
            with torch.no_grad():
                if self.fc:  # keys through the momentum projection heads (head.py:135-144)
                    v_embed_k = losses.l2_normalize(losses.mlp(vk_feat, self.v_fc_k))
                    t_embed_k = losses.l2_normalize(losses.mlp(tk_feat, self.t_fc_k))
                else:
                    v_embed_k = losses.l2_normalize(losses.linear(vk_feat, self.v_embed_layer.weight, self.v_embed_layer.bias))
                    t_embed_k = losses.l2_normalize(losses.linear(tk_feat, self.t_embed_layer.weight, self.t_embed_layer.bias))
            if self.fc:
                v_src, t_src = losses.mlp(v_feat, self.v_fc_q), losses.mlp(t_feat, self.t_fc_q)  # head.py:118-124
                if dp_active():
                    # the packed gather carries the two projection-head query embeddings as well (six blocks + ids)
                    v_embed, t_embed, v_embed_k, t_embed_k, id_q, v_src, t_src = gather_embeddings(
                        v_embed, t_embed, v_embed_k, t_embed_k, id_q, extra=(v_src, t_src))
                v_embed_q = losses.l2_normalize(v_src)
                t_embed_q = losses.l2_normalize(t_src)
            else:
                if dp_active():
                    # one packed RCCL all-gather; every rank then evaluates the GLOBAL losses
                    v_embed, t_embed, v_embed_k, t_embed_k, id_q = gather_embeddings(v_embed, t_embed, v_embed_k, t_embed_k, id_q)
                v_embed_q = losses.l2_normalize(v_embed)
                t_embed_q = losses.l2_normalize(t_embed)
            out = self.loss_evaluator.forward_fused(
                v_embed, t_embed, v_embed_q, t_embed_q, v_embed_k, t_embed_k, id_q,
                self._queue_kc("t_queue"), self._queue_kc("v_queue"), self.id_queue,
            )
            self._dequeue_and_enqueue(v_embed_k, t_embed_k, id_q)
            return out
        v_feat = self.v_encoder_q(images)
        t_feat = self.t_encoder_q(cb)
        v_embed = losses.linear(v_feat, self.v_embed_layer.weight, self.v_embed_layer.bias)
        t_embed = losses.linear(t_feat, self.t_embed_layer.weight, self.t_embed_layer.bias)
        return [v_embed, t_embed]


    @torch.no_grad()
    def encode_images(self, images):
        """eval-mode image embeddings [N,C] (head.py:114,178)."""
        return losses.linear(self.v_encoder_q(images), self.v_embed_layer.weight, self.v_embed_layer.bias)

    @torch.no_grad()
    def encode_captions(self, captions):
        """eval-mode caption embeddings [N,C] (head.py:115,179)."""
        return losses.linear(self.t_encoder_q(CaptionBatch.from_list(captions)), self.t_embed_layer.weight,
                             self.t_embed_layer.bias)


def build_moco_head(cfg, visual_model, textual_model):
    return MoCoHead(cfg, visual_model, textual_model)
