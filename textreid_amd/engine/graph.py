"""The training step as ONE hipGraph (counterpart of the loop body ``lib/engine/trainer.py:72-91``:
``model(images, captions)`` -> sum of the losses -> ``zero_grad`` -> ``backward`` -> ``optimizer.step``).

A step is ~1 100 kernel launches on four HIP streams issued from Python through ctypes: ~40 ms of host work per
50 ms step.  With zero host-device synchronisations and a flat allocation profile per step (both pinned by tests)
the whole launch sequence - the side streams included, they fork from and join the capturing stream through events -
is recorded once with ``torch.cuda.graph`` and replayed: one host call per step.

What varies between steps lives in device memory that the replay reads:
  * inputs: copied into static buffers (stream-ordered, before the replay);
  * MoCo queue pointer, BatchNorm counters, amax scalars: device-resident already;
  * optimizer hyper-parameters and bias corrections: ``FusedAdam.advance_for_replay()``.

Shapes are part of the recording: a batch of another shape (or another caption length bound) runs EAGERLY, with a
warning - never silently through a graph recorded for different sizes.

Two ways to run the recording (``launch=``): "graph" = hipGraphLaunch; "streams" (default) = the recorded nodes re-issued as
ordinary stream launches by ONE call into the library (csrc/step_replay.hip: the graph is read back - launch parameters,
edges - and laid out on a few streams, chains in stream order, an event per cross-chain edge).  On this runtime
hipGraphLaunch of the step is slower than the eager step (45.2 vs 44.0 ms); the stream form is the eager step without its
35-39 ms of Python / ctypes time, with every stream fed at once.
"""

import ctypes
import logging
import os

import torch

from ..caption import CaptionBatch
from ..parallel import dp_active


class CapturedTrainStep:
    def __init__(self, model, optimizer, warmup=2, caption_bound=None, reducer=None, pre_gather=None, launch=None, lanes=8):
        """warmup: eager steps before the capture (they build every cached pointer table / workspace / side stream the
        step uses; they are REAL training steps).  optimizer=None: forward + backward only (parity tests).
        caption_bound: number of recurrence steps the recorded text encoder runs (None: the token tensor's width, i.e.
        any caption fits); batches whose longest caption exceeds it run eagerly.
        reducer / pre_gather (data parallel): the parallel.GradReducer of the run and the parameters it SUM-reduces after
        backward.  The step's collectives - the packed embedding all-gather of the forward, the all-reduces staged from
        inside backward, the bucketed ones after it - are CUT POINTS of the recording (parallel.CutRecorder): each leaves a
        marker node where it belongs, and the replay re-issues the recorded kernels in segments, calling the collective
        itself (torch.distributed, i.e. RCCL's own launch path; gloo stages through the host as it always does) on the
        marker's stream between two segments - <= ~8 host round trips per step instead of ~1100 Python launches.  Every
        rank records the same sequence.  A data-parallel recording has no hipGraphLaunch form (the graph holds markers,
        not collectives): when the stream plan cannot be built the step stays eager."""
        self.model, self.optimizer = model, optimizer
        # launch: "streams" | "graph" (None: TRID_STEP_LAUNCH, else "streams"); lanes: streams the recorded nodes are laid out on
        self.launch = launch or os.environ.get("TRID_STEP_LAUNCH", "streams")
        if self.launch not in ("streams", "graph"):
            raise ValueError("CapturedTrainStep: launch must be 'streams' or 'graph', not %r" % (self.launch,))
        self.lanes = int(os.environ.get("TRID_STEP_LANES", lanes))
        self.replayer = None     # handle of csrc/step_replay.hip (launch == "streams")
        self.force_graph_launch = False  # (A/B runs: hipGraphLaunch although a stream plan exists)
        self.replay_info = None
        self.reducer, self.pre_gather = reducer, pre_gather
        self.caption_bound = caption_bound
        self.warmup = max(int(warmup), 1)
        self.calls = 0
        self.graph = None
        self.static = None
        self.out = None
        self.signature = None
        self.plan = None         # the optimizer's pointer tables the recorded Adam launch reads (kept alive with the graph)
        self.disabled = False    # a failed capture: stay eager for the rest of the run
        self.recaptures = 0
        self.adam_cap = None     # FusedAdam's table set of THIS recording
        self.cuts = []           # data parallel: the collectives of the recorded step, in marker order (parallel.Cut)
        self.cut_bytes = (0, 0)  # bytes of them staged from inside backward / issued after it (the reducer's accounting)
        self.log = logging.getLogger("PersonSearch.trainer")
        if dp_active() and reducer is None:
            raise RuntimeError("CapturedTrainStep under data parallelism needs the run's GradReducer (reducer=, pre_gather=): "
                               "the gradient all-reduce is part of the recorded step")

    def _sync_grads(self):
        if self.reducer is not None and dp_active():
            self.reducer.reduce(self.pre_gather)
            self.reducer.wait()

    # ------------------------------------------------------------------ eager form (also the fall-back)
    def _eager(self, images, cb):
        loss_dict = self.model(images, cb)
        losses = sum(loss_dict.values())
        if self.optimizer is not None:
            self.optimizer.zero_grad()
        else:
            for p in self.model.parameters():
                p.grad = None
        losses.backward()
        self._sync_grads()
        if self.optimizer is not None:
            self.optimizer.step()
        # detached: a caller that keeps the returned losses must not keep the autograd graph (and its AccumulateGrad
        # nodes, bound to THIS call's stream) alive into the next call - a stale node inside a capture drags the
        # stream it was created on into the recording
        return {k: v.detach() for k, v in loss_dict.items()}

    @staticmethod
    def _sig(images, cb):
        return (tuple(images.shape), tuple(cb.tokens.shape), cb.ids is not None)

    def _capture(self, images, cb):
        from ..solver import FusedAdam

        self.static = {
            "images": images.clone(),
            "tokens": cb.tokens.clone(),
            "lengths": cb.lengths.clone(),
            "ids": cb.ids.clone() if cb.ids is not None else None,
        }
        self.bound = int(self.caption_bound) if self.caption_bound is not None else int(cb.tokens.shape[1])
        scb = CaptionBatch(self.static["tokens"], self.static["lengths"], self.static["ids"], max_len=self.bound, bound_only=True)
        self.adam_cap = None
        if isinstance(self.optimizer, FusedAdam):
            self.adam_cap = self.optimizer.prepare_capture(defer_table_copy=True)  # this recording's own table set
        elif self.optimizer is not None:
            raise RuntimeError("CapturedTrainStep needs textreid_amd.solver.FusedAdam (or optimizer=None)")
        for p in self.model.parameters():
            p.grad = None  # the captured backward allocates its gradients inside the graph's pool
        from .. import ops

        ops.begin_capture()
        torch.cuda.synchronize()
        g = None
        if self.launch == "streams":
            try:
                g = torch.cuda.CUDAGraph(keep_graph=True)  # the hipGraph_t must outlive the recording: the stream plan reads it back
            except TypeError:  # (a PyTorch without keep_graph / raw_cuda_graph)
                self.log.warning("train step: this PyTorch cannot hand out the recorded hipGraph - using hipGraphLaunch")
                self.launch = "graph"
        if g is None:
            g = torch.cuda.CUDAGraph()
        # thread_local: every launch of the step is issued from this thread (the backward runs on the autograd engine's
        # thread for this device, which torch's capture tracks); a HIP call from an UNRELATED thread - a DataLoader's
        # pin_memory thread allocating or polling events - must not invalidate the ~1100-launch recording
        import contextlib

        from ..parallel import CutRecorder

        dp = dp_active()
        if dp and self.launch != "streams":
            raise RuntimeError("a data-parallel step is replayed as stream launches only (its collectives are cut points of the recording)")
        rec = CutRecorder() if dp else None
        b0 = (self.reducer.bytes_staged, self.reducer.bytes_post, self.reducer.steps) if (dp and self.reducer is not None) else None
        with (rec if rec is not None else contextlib.nullcontext()):
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                loss_dict = self.model(self.static["images"], scb)
                losses = sum(loss_dict.values())
                losses.backward()
                self._sync_grads()
                if self.optimizer is not None:
                    self.optimizer.step()
        self.cuts = rec.cuts if rec is not None else []
        if b0 is not None:  # the recording executed nothing: its byte counts are per REPLAY (GradReducer.account_replay)
            self.cut_bytes = (self.reducer.bytes_staged - b0[0], self.reducer.bytes_post - b0[1])
            self.reducer.bytes_staged, self.reducer.bytes_post, self.reducer.steps = b0
        self.graph, self.out = g, {k: v.detach() for k, v in loss_dict.items()}
        del loss_dict, losses
        if isinstance(self.optimizer, FusedAdam):
            self.optimizer.finish_capture(self.adam_cap)  # the gradient address table of the recorded Adam launch: copied once, here
        self.grads = [(p, p.grad) for p in self.model.parameters() if p.grad is not None]  # live in the graph's pool
        self.signature = self._sig(images, cb)
        # the recorded Adam launch has the ADDRESSES of this plan's tables baked in: hold it (so the allocator cannot
        # recycle them under the graph) and replay only while the optimizer still uses this very object
        self.plan = getattr(self.optimizer, "_plan", None)
        if self.launch == "streams":
            self._build_replayer(g)
        if self.cuts and self.replayer is None:
            raise RuntimeError("the data-parallel recording (%d collectives as cut points) has no stream plan" % len(self.cuts))
        if self.cuts:
            from .. import lib as L

            n = int(L.load().trid_step_replay_markers(self.replayer))
            if n != len(self.cuts):
                raise RuntimeError("the stream plan found %d cut points, the recording made %d" % (n, len(self.cuts)))
        self.log.info("train step captured: %s per step from here on", ("%d segments of stream launches around %d collectives" % (len(self.cuts) + 1, len(self.cuts))) if self.cuts
                      else ("one call into the library (stream replay)" if self.replayer else "one graph launch"))

    def _build_replayer(self, g):
        """The recorded graph read back into a stream-launch plan.  A graph holding a node type the replayer refuses (host
        callback, child graph) stays on hipGraphLaunch - unless it is a data-parallel recording (see _capture)."""
        from .. import ops

        self._drop_replayer()
        h = ctypes.c_void_p()
        try:
            ops.call("trid_step_replay_build", int(g.raw_cuda_graph()), self.lanes, ctypes.byref(h))
        except RuntimeError as e:
            self.log.warning("train step: the recorded graph cannot be replayed as stream launches (%s) - using hipGraphLaunch", str(e).splitlines()[0])
            return
        counts = (ctypes.c_int * 8)()
        ops.call("trid_step_replay_info", h, counts)
        self.replayer = h
        self.replay_info = dict(zip(("nodes", "kernels", "copies", "memsets", "lanes", "events", "waits", "empty"), [int(c) for c in counts]))

    def _drop_replayer(self):
        if self.replayer is not None:
            from .. import ops

            torch.cuda.synchronize()
            ops.call("trid_step_replay_destroy", self.replayer)
        self.replayer = self.replay_info = None

    def __del__(self):
        try:
            self._drop_replayer()
        except Exception:  # (interpreter shutdown)
            pass

    def _drop_graph(self, why):
        self.log.warning("train step: %s - dropping the recorded graph", why)
        self._drop_replayer()
        self.graph = self.static = self.out = self.signature = None
        self.grads = []
        self.cuts = []

    def _try_capture(self, images, cb):
        """Record the step; when the RECORDING fails (a non-capturable call, an invalidated capture) keep training eagerly -
        eager is the documented fall-back for everything else, a failed recording must not end the run at step 3.  Only
        capture failures are absorbed: an out-of-memory error, or a device that still refuses a probe launch after the
        aborted recording was torn down (a real kernel fault, a sticky HIP error), is re-raised.  Under data parallelism
        the ranks agree on the outcome (one MAX all-reduce of the failure flag): either all replay or all stay eager."""
        from .. import ops

        saved = [(p, p.grad) for p in self.model.parameters()]
        err, fatal = None, False
        try:
            self._capture(images, cb)
        except torch.cuda.OutOfMemoryError as e:
            err, fatal = e, True
        except RuntimeError as e:  # what torch / the library raise for a failed or invalidated stream capture
            err = e
        except Exception as e:  # an assertion / ValueError / KeyError from inside the recording: not a capture failure - but the
            err, fatal = e, True  # teardown and the exchange below must still happen on THIS rank before it propagates
        if err is not None:
            if not fatal:
                self.log.warning("train step: hipGraph capture failed (%s: %s) - the step stays eager for the rest of the run",
                                 type(err).__name__, str(err).splitlines()[0] if str(err) else "")
            try:
                self._abort_capture(saved)
            except Exception:  # (the teardown is best effort when the device itself is gone)
                if not fatal:
                    raise
            if not fatal:
                # is the device itself still healthy?  (a capture error leaves it usable; a kernel fault does not)
                torch.cuda.synchronize()
                probe = ops.amax_slot(images.device)
                ops.call("trid_amax_f32", ops._p(self.static_probe(images)), 1, ops._p(probe), ops.stream())
                torch.cuda.synchronize()
        ok = err is None
        if dp_active():
            # every rank learns the outcome BEFORE anything is re-raised: a rank that left here alone would leave the others
            # blocked in this all-reduce until the collective timeout (flag = 1: stay eager, 2: a rank is about to raise)
            import torch.distributed as dist

            flag = torch.tensor([2.0 if fatal else (0.0 if ok else 1.0)], device=images.device if dist.get_backend() != "gloo" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            worst = float(flag.item())
            if ok and worst > 0:
                self.log.warning("train step: another rank could not record the step - dropping this rank's graph, all ranks stay eager")
                self._abort_capture(saved)
                ok = False
            if worst >= 2 and not fatal:
                raise RuntimeError("train step: another rank raised a non-capture error while recording the step")
        if fatal:
            raise err
        if not ok:
            self.disabled = True
        return ok

    @staticmethod
    def static_probe(images):
        return images.reshape(-1)[:1].float().contiguous()

    def _abort_capture(self, saved_grads):
        """Tear down everything an aborted (or discarded) recording left on the host: the reducer's Work handles and flat
        buffers were created inside the invalidated capture (waiting on them, or merging their slices, would crash or corrupt
        the next eager backward), the amax / finalize pools of the capture hold slots of its dead memory pool."""
        from .. import ops

        try:
            torch.cuda.synchronize()
        except RuntimeError:
            pass
        for p, g in saved_grads:
            p.grad = g
        if self.reducer is not None:
            self.reducer.abort()
        ops.begin_capture()  # (clears the capture-private pools)
        self._drop_replayer()
        self.graph = self.static = self.out = self.signature = None
        self.grads = []
        self.cuts = []

    def __call__(self, images, captions):
        cb = CaptionBatch.from_list(captions)
        self.calls += 1
        if self.disabled:
            return self._eager(images, cb)
        if self.graph is not None and self.optimizer is not None and getattr(self.optimizer, "_plan", None) is not self.plan:
            # load_state_dict / __setstate__ / a moved parameter replaced the optimizer's pointer tables: the recorded
            # launch would read the old ones.  One eager step rebuilds them, the next call records again.
            self._drop_graph("the optimizer's pointer tables were rebuilt (state loaded or a tensor moved)")
            self.recaptures += 1
            return self._eager(images, cb)
        if self.graph is None:
            if self.calls <= self.warmup:
                return self._eager(images, cb)
            if self.optimizer is not None and getattr(self.optimizer, "_plan", None) is None:
                return self._eager(images, cb)  # (tables dropped since the warm-up: this eager step rebuilds them)
            if not self._try_capture(images, cb):
                return self._eager(images, cb)
        if self._sig(images, cb) != self.signature or cb.max_len > self.bound:
            self.log.warning("train step: batch signature %s (longest caption %d) does not fit the captured one %s (bound %d) - running this step eagerly",
                             self._sig(images, cb), cb.max_len, self.signature, self.bound)
            return self._eager(images, cb)
        st = self.static
        st["images"].copy_(images, non_blocking=True)
        st["tokens"].copy_(cb.tokens, non_blocking=True)
        st["lengths"].copy_(cb.lengths, non_blocking=True)
        if st["ids"] is not None:
            st["ids"].copy_(cb.ids, non_blocking=True)
        for p, g in self.grads:  # (an eager fall-back step in between re-bound .grad)
            if p.grad is not g:
                p.grad = g
        if self.optimizer is not None:
            self.optimizer.advance_for_replay(self.adam_cap)
        from .. import ops

        if self.cuts:
            self._replay_segments()
        elif self.replayer is not None and not self.force_graph_launch:
            try:
                ops.call("trid_step_replay_run", self.replayer, ops.stream())
            except RuntimeError:
                # part of the step was enqueued (and joined back into this stream by the library), the optimizer tables were
                # advanced for it: the state is not the recorded step's - drop the poisoned plan and say so
                self._drop_graph("the stream replay failed in mid-step")
                self.disabled = True
                raise
        else:
            self.graph.replay()
        return self.out

    def _replay_segments(self):
        """The data-parallel step: segments of recorded launches (one library call each) around the collectives, which run
        through torch.distributed on the stream their marker was recorded on."""
        from .. import lib as L
        from .. import ops

        origin = ops.stream()
        mid, lane = ctypes.c_int(-1), ctypes.c_void_p()
        dev = self.static["images"].device
        run = L.load().trid_step_replay_run_segment
        try:
            while True:
                rc = int(run(self.replayer, origin, ctypes.byref(mid), ctypes.byref(lane)))
                if rc == 0:
                    break
                if rc != 1:
                    raise RuntimeError("trid_step_replay_run_segment failed (rc=%d): %s" % (rc, L.last_error()))
                with torch.cuda.stream(torch.cuda.ExternalStream(lane.value, device=dev)):
                    self.cuts[mid.value].run()
        except RuntimeError:
            self._drop_graph("the segmented replay failed in mid-step")
            self.disabled = True
            raise
        if self.reducer is not None:
            self.reducer.account_replay(*self.cut_bytes)


class BucketedTrainStep:
    """One recording of the step per CAPTION BUCKET (counterpart of `lib/models/backbones/gru.py:66-82`: the reference packs every
    batch to its own longest caption; `lib/data/build.py:26` pads the token tensor to 105).  A recorded text encoder runs a fixed
    number of recurrence steps, and CUHK-PEDES captions average ~25 tokens: a single recording at the tensor's width would run 105
    steps for every batch.  This keeps up to len(buckets) recordings, keyed by their recurrence bound, and sends a batch to the
    smallest one that fits its longest caption (`CaptionBatch.max_len`: one host integer the collate already knows) - recorded
    on first use, after one eager step of that bucket.  Results are those of the eager step (the text encoder's outputs do not
    depend on the bound, only its launch count does).  Data parallel: ONE recording at the token tensor's width (see bucket_of)."""

    def __init__(self, model, optimizer, buckets=(32, 48, 64, 105), warmup=2, **kw):
        env = os.environ.get("TRID_CAPTION_BUCKETS")
        if env:
            buckets = tuple(int(b) for b in env.split(",") if b.strip())
        self.buckets = sorted(set(int(b) for b in buckets))
        if not self.buckets or self.buckets[0] < 1:
            raise ValueError("BucketedTrainStep: caption buckets must be positive, got %r" % (buckets,))
        self.model, self.optimizer, self.warmup, self.kw = model, optimizer, warmup, kw
        self.runners = {}
        self.last = None

    def bucket_of(self, cb):
        width = int(cb.tokens.shape[1])
        if dp_active():
            # every rank must run the SAME launch form in the same call (a rank that records exchanges a flag with the others,
            # a rank that replays does not): the ranks' batches have different longest captions, and agreeing on a bucket would
            # take a host-synchronising collective per step - under data parallelism there is one recording, at the tensor's width
            return width
        b = next((x for x in self.buckets if x >= cb.max_len), width)
        return min(b, width)

    def __call__(self, images, captions):
        cb = CaptionBatch.from_list(captions)
        b = self.bucket_of(cb)
        r = self.runners.get(b)
        if r is None:
            # (the first bucket's eager steps build every cached table of the step; a later bucket needs one eager call only)
            r = CapturedTrainStep(self.model, self.optimizer, warmup=self.warmup if not self.runners else 1, caption_bound=b, **self.kw)
            self.runners[b] = r
        self.last = r
        return r(images, cb)

    @property
    def recorded(self):
        """{bucket: recurrence bound of its recording} for the buckets that have one."""
        return {b: r.bound for b, r in self.runners.items() if r.graph is not None}
