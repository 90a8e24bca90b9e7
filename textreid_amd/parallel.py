"""Data-parallel pieces: one process per GPU, torch.distributed over RCCL/xGMI.

The train step shards the mini-batch across ranks.  Each rank encodes its
shard (BatchNorm statistics stay rank-local, as the reference's
``broadcast_buffers=False``, train_net.py:54-55), then ONE packed all-gather
moves ``[B_local, 4*C + 1]`` floats per rank (v_embed, t_embed, v_key, t_key,
id) so every rank evaluates the three losses on the GLOBAL batch against its
replicated queue.  Backward needs no collective for the embeddings: each rank
already holds dL/d(e_local), the local rows of the gathered gradient.
Gradients of pre-gather parameters are then SUM-reduced across ranks
(``GradReducer``), post-gather parameters (the instance-loss projection) are
identical on every rank and are not reduced.  SURVEY.md section 8e.
"""

import os

import torch
import torch.distributed as dist


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def dp_active():
    """True when the data-parallel exchange steps (embedding all-gather, gradient all-reduce) run: a process group
    with more than one rank - or, with TRID_DP_FORCE=1, ANY initialised process group, so that a one-rank `nccl`
    group drives every RCCL call of the step on a single-GPU box (tests/test_dp_gpu.py)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("TRID_DP_FORCE", "0") == "1"


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def _backend():
    return dist.get_backend() if dist.is_available() and dist.is_initialized() else None


# ---- cut points of a RECORDED step (engine/graph.py, csrc/step_replay.hip).  While the train step is being captured the
# collectives below are not executed: each leaves a MARKER node in the recording (trid_step_marker, on the stream and between
# the events the collective would have had) and an entry in the recorder's list; the replay re-issues the recorded kernels in
# segments and calls `Cut.run()` - the very functions below, on the marker's stream - at every marker.  The communication
# library's kernels therefore always go through its own launch path (an RCCL kernel is never re-issued from launch parameters
# read back from a graph), and a transport that stages through host memory (gloo) replays as well.
_recorder = None


class Cut:
    """One collective of a recorded step: where the recording holds a marker kernel, the segmented replay
    (engine.graph.CapturedTrainStep) calls run() on the marker's stream."""

    def __init__(self, kind, src, dst):
        self.kind, self.src, self.dst = kind, src, dst  # static tensors of the recording's memory pool

    def run(self):
        if self.kind == "all_gather":
            _all_gather_into(self.dst, self.src)
        else:
            w = _all_reduce_sum_impl(self.src)
            if w is not None:
                w.wait()  # the marker's stream waits for the collective (the host does not)

    def nbytes(self):
        return self.src.numel() * self.src.element_size()


class _CutWork:
    """What an asynchronous collective returns while the step is being recorded: `wait()` joins the side stream the marker
    went to back into the current stream (the recorded edge the replay turns into an event)."""

    def __init__(self, side):
        self.side = side

    def wait(self):
        torch.cuda.current_stream().wait_stream(self.side)


class CutRecorder:
    def __init__(self):
        self.cuts = []
        self._side = None

    def __enter__(self):
        global _recorder
        if _recorder is not None:
            raise RuntimeError("nested CutRecorder")
        _recorder = self
        return self

    def __exit__(self, *exc):
        global _recorder
        _recorder = None
        return False

    def mark(self, kind, src, dst, side=False):
        """A marker for the collective (kind, src -> dst) on the current stream - or, side=True (asynchronous all-reduce), on a
        side stream forked from it; returns the _CutWork to wait on in that case."""
        from . import ops

        cut = Cut(kind, src, dst)
        idx = len(self.cuts)
        self.cuts.append(cut)
        if not side:
            ops.call("trid_step_marker", idx, ops.stream())
            return None
        if self._side is None:
            self._side = torch.cuda.Stream(device=src.device)
        cur = torch.cuda.current_stream()
        self._side.wait_stream(cur)
        with torch.cuda.stream(self._side):
            ops.call("trid_step_marker", idx, ops.stream())
        return _CutWork(self._side)


def _recording(t):
    return _recorder is not None and t.is_cuda and torch.cuda.is_current_stream_capturing()


def _all_gather_into(out, x):
    if _backend() == "gloo" and x.is_cuda:
        parts = [torch.empty(x.shape, dtype=x.dtype) for _ in range(world_size())]
        dist.all_gather(parts, x.cpu())
        out.copy_(torch.cat(parts, dim=0))
    else:
        dist.all_gather_into_tensor(out, x)


def all_gather_rows(x):
    """[n, ...] per rank -> [W*n, ...] (rank-major).  RCCL: one all_gather_into_tensor on the
    current stream; gloo (CPU tests / single-GPU debugging): list all_gather staged through host
    memory for device tensors.  Inside a recording: a cut point (see above)."""
    W = world_size()
    x = x.contiguous()
    out = torch.empty((W * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    if _recording(x):
        _recorder.mark("all_gather", x, out)
        return out
    _all_gather_into(out, x)
    return out


def all_reduce_sum_(t):
    """In-place SUM all-reduce (async handle or None).  Inside a recording: a cut point on a side stream."""
    if _recording(t):
        return _recorder.mark("all_reduce", t, t, side=True)
    return _all_reduce_sum_impl(t)


def _all_reduce_sum_impl(t):
    if _backend() == "gloo" and t.is_cuda:
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        t.copy_(h)
        return None
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)


def broadcast_module_state(module, src=0):
    """Every parameter and buffer of `module` from rank `src` to all ranks, once (what DistributedDataParallel does when it
    wraps the model: train_net.py:50-56 - with `broadcast_buffers=False` the buffers are synchronised at construction only).
    The replicated-queue design needs every rank to START from the same weights, BatchNorm statistics, queues,
    ids and pointer; equal seeds give that by convention, this gives it by construction.  One flat buffer per dtype."""
    if not dp_active():
        return 0
    groups = {}
    for t in list(module.parameters()) + list(module.buffers()):
        groups.setdefault((t.dtype, t.device), []).append(t.data)
    nbytes = 0
    for (dtype, device), ts in groups.items():
        flat = torch.cat([t.reshape(-1) for t in ts])
        if _backend() == "gloo" and flat.is_cuda:
            h = flat.cpu()
            dist.broadcast(h, src)
            flat.copy_(h)
        else:
            dist.broadcast(flat, src)
        off = 0
        for t in ts:
            n = t.numel()
            t.copy_(flat[off : off + n].view_as(t))
            off += n
        nbytes += flat.numel() * flat.element_size()
    if any(t.is_cuda for t in module.parameters()):
        from . import ops

        ops.note_parameter_write()  # (caches keyed on parameter values: the eval path's folded filters / plans)
    return nbytes


def replica_digest(tensors):
    """One int64 per tensor: the wrap-around sum of its words (32-bit lanes; 64-bit tensors: 64-bit words), each weighted by an
    odd multiple of its POSITION - bit-identical replicas have identical digests; a swap of two unequal entries (queue slots,
    ids), or drifts that would cancel in a plain sum, change it.  (A hash, not a proof: a replica that differs in one word is
    told apart unless the difference times its odd weight vanishes mod 2^64, i.e. never for a single word.)"""
    out = []
    for t in tensors:
        b = t.detach().contiguous().reshape(-1)
        if b.element_size() == 8:
            w = b.view(torch.int64)
        elif b.element_size() == 4:
            w = b.view(torch.int32).to(torch.int64)
        else:
            w = b.view(torch.uint8).to(torch.int64)
        pos = torch.arange(w.numel(), dtype=torch.int64, device=w.device)
        out.append((w * (pos * -7046029254386353131 + 1)).sum())  # (0x9E3779B97F4A7C15 as a signed word: odd)
    return torch.stack(out)


def check_replicas(named_tensors, where=""):
    """Raise when the replicated state differs between ranks: a MIN and a MAX all-reduce of the digests (two tiny collectives);
    `named_tensors` = [(name, tensor)] - the MoCo queues, ids and pointer, a pre- and a post-gather parameter: engine.trainer.do_train."""
    if not dp_active():
        return
    d = replica_digest([t for _, t in named_tensors])
    if _backend() == "gloo" and d.is_cuda:
        d = d.cpu()
    lo, hi = d.clone(), d.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    if not torch.equal(lo, hi):
        bad = [n for (n, _), a, b in zip(named_tensors, lo.tolist(), hi.tolist()) if a != b]
        raise RuntimeError("data-parallel replicas have diverged%s: %s differ between ranks (rank %d of %d) - every rank must push the "
                           "same gathered keys and apply the same reduced gradients" % (" " + where if where else "", ", ".join(bad), rank(), world_size()))


class _GatherRows(torch.autograd.Function):
    """all_gather along dim 0; backward returns the local rows of the gradient
    (no collective, no scaling: the loss is evaluated in full on every rank)."""

    @staticmethod
    def forward(ctx, x):
        ctx.rows = x.shape[0]
        return all_gather_rows(x)

    @staticmethod
    def backward(ctx, g):
        r = rank()
        return g[r * ctx.rows : (r + 1) * ctx.rows].contiguous()


def gather_embeddings(v_embed, t_embed, v_key, t_key, ids, extra=()):
    """Packed all-gather of the four [B,C] embedding blocks and the ids (+ `extra`: further differentiable [B,C]
    blocks - with MODEL.MOCO.FC the two projection-head query embeddings, head.py:117-124).
    Gradients flow to v_embed / t_embed / extra only (keys are detached).
    Returns (v, t, v_key, t_key, ids[, *extra]) of the GLOBAL batch, rank-major."""
    B, C = v_embed.shape
    # ids travel as their own BITS: an int64 is two 32-bit lanes of the fp32 payload (a collective only moves
    # bytes), exact for any id - an fp32 VALUE would be exact only below 2^24
    id_lanes = ids.long().contiguous().view(B, 1).view(torch.float32)  # [B, 2]
    packed = torch.cat([v_embed, t_embed, v_key.detach(), t_key.detach(), id_lanes] + list(extra), dim=1)
    g = _GatherRows.apply(packed)
    v, t, vk, tk = g[:, :C], g[:, C : 2 * C], g[:, 2 * C : 3 * C].detach(), g[:, 3 * C : 4 * C].detach()
    gid = g[:, 4 * C : 4 * C + 2].detach().contiguous().view(torch.int64).view(-1)
    more = tuple(g[:, 4 * C + 2 + i * C : 4 * C + 2 + (i + 1) * C].contiguous() for i in range(len(extra)))
    return (v.contiguous(), t.contiguous(), vk.contiguous(), tk.contiguous(), gid) + more


class GradReducer:
    """Bucketed SUM all-reduce of parameter gradients (RCCL runs collectives on its own stream).

    Two entry points: `stage()/finish_stages()` are called from INSIDE the image encoder's backward
    (ModifiedResNet.grad_sync) so that most of the 150 MB of conv gradients is reduced underneath the
    remaining backward; `reduce()/wait()` after backward picks up everything that was not staged.

    ``reduce(params)`` flattens the gradients of ``params`` into ~bucket_mb
    buckets in the given order (call it with parameters in reverse execution
    order so the last layers go first), all-reduces each bucket asynchronously
    and copies the result back; ``wait()`` joins before the optimiser step.
    xGMI is point-to-point (7 links per GPU), so buckets are large (default
    64 MB) to stay bandwidth- rather than latency-bound per link.
    """

    def __init__(self, bucket_mb=64):
        self.bucket_elems = int(bucket_mb * (1 << 20) // 4)
        self._pending = []
        self._stages = []
        self._staged = set()
        # accounting for bench.py (reset_stats() per timed region): bytes handed to all-reduce from inside
        # backward (overlapped) and after it, and device time between "backward done" and "all gradients reduced"
        self.bytes_staged = 0
        self.bytes_post = 0
        self.steps = 0
        self._exposed = []

    def reset_stats(self):
        self.bytes_staged = self.bytes_post = self.steps = 0
        self._exposed = []

    def account_replay(self, staged, post):
        """One REPLAYED step (engine/graph.py): the byte counts the recording of the step produced."""
        self.bytes_staged += staged
        self.bytes_post += post
        self.steps += 1

    def abort(self):
        """Forget every collective in flight WITHOUT waiting for it (engine.graph: a stream capture was invalidated - the
        Work handles and flat buffers staged inside it belong to the dead recording)."""
        self._stages, self._pending, self._staged = [], [], set()
        self._t0 = None

    def stats(self):
        """{"allreduce_bytes_per_step", "staged_fraction", "exposed_ms_per_step"}; synchronises the device."""
        n = max(self.steps, 1)
        ms = 0.0
        if self._exposed:
            torch.cuda.synchronize()
            ms = sum(a.elapsed_time(b) for a, b in self._exposed) / len(self._exposed)
        tot = self.bytes_staged + self.bytes_post
        return {"allreduce_bytes_per_step": tot // n, "staged_fraction": (self.bytes_staged / tot) if tot else 0.0,
                "exposed_ms_per_step": ms}

    # ---- in-backward staging: the image encoder's backward hands over each residual stage's gradients
    # as soon as they are final, so their all-reduce runs under the backward of the earlier stages
    def stage(self, params, grads):
        """Start the SUM all-reduce of `grads` (gradients of `params`, same order).  No-op without data parallelism."""
        if not dp_active() or not grads:
            return
        flats, layouts = [], []
        for g in grads:
            if g.is_contiguous():
                flats.append(g.view(-1))
                layouts.append(("c", g.shape))
            elif g.dim() == 4 and g.is_contiguous(memory_format=torch.channels_last):
                flats.append(g.permute(0, 2, 3, 1).reshape(-1))  # storage order, no copy
                layouts.append(("cl", g.shape))
            else:
                flats.append(g.contiguous().view(-1))
                layouts.append(("c", g.shape))
        flat = torch.cat(flats)
        self.bytes_staged += flat.numel() * 4
        work = all_reduce_sum_(flat)
        self._stages.append((work, flat, list(params), layouts))

    def finish_stages(self):
        """Join every staged all-reduce; {id(param): reduced gradient} in each gradient's original layout."""
        out = {}
        for work, flat, params, layouts in self._stages:
            if work is not None:
                work.wait()
            off = 0
            for p, (kind, shape) in zip(params, layouts):
                n = 1
                for d in shape:
                    n *= d
                piece = flat[off : off + n]
                if kind == "cl":
                    N, C, H, W = shape
                    out[id(p)] = piece.view(N, H, W, C).permute(0, 3, 1, 2)
                else:
                    out[id(p)] = piece.view(shape)
                self._staged.add(id(p))
                off += n
        self._stages = []
        return out

    def reduce(self, params):
        if not dp_active():
            return
        self.steps += 1
        self._t0 = None
        if params and params[0].is_cuda and not torch.cuda.is_current_stream_capturing():  # (timing events cannot be recorded into a graph)
            self._t0 = torch.cuda.Event(enable_timing=True)
            self._t0.record()  # backward is done on this stream
        bucket, n = [], 0
        for p in params:
            if p.grad is None or id(p) in self._staged:  # staged gradients were reduced inside backward
                continue
            bucket.append(p)
            n += p.numel()
            if n >= self.bucket_elems:
                self._launch(bucket)
                bucket, n = [], 0
        if bucket:
            self._launch(bucket)

    def _launch(self, bucket):
        flat = torch.cat([p.grad.reshape(-1) for p in bucket])
        self.bytes_post += flat.numel() * 4
        work = all_reduce_sum_(flat)
        self._pending.append((work, flat, bucket))

    def wait(self):
        for work, flat, bucket in self._pending:
            if work is not None:
                work.wait()
            off = 0
            for p in bucket:
                n = p.numel()
                # (an elementwise KERNEL, not copy_: a contiguous device-to-device copy_ is a hipMemcpyAsync, which a recording of
                # the step keeps as a copy node that csrc/step_replay.hip cannot read back on this runtime; x * 1 is x, bit for bit)
                torch.mul(flat[off : off + n].view_as(p.grad), 1.0, out=p.grad)
                off += n
        self._pending = []
        self._staged = set()
        if getattr(self, "_t0", None) is not None and not torch.cuda.is_current_stream_capturing():
            t1 = torch.cuda.Event(enable_timing=True)
            t1.record()
            self._exposed.append((self._t0, t1))
            self._t0 = None
