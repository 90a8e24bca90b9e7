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
        reducer / pre_gather (data parallel, `nccl` backend): the parallel.GradReducer of the run and the parameters it
        SUM-reduces after backward; the packed embedding all-gather of the forward, the all-reduces staged from inside
        backward and the bucketed ones after it are RCCL launches on RCCL's stream - stream-ordered, hence recordable
        (probed: tools/exp/rccl_capture_probe.py).  Every rank records the same sequence; a transport that stages
        through the host (gloo) fails the recording and the step stays eager."""
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
        if isinstance(self.optimizer, FusedAdam):
            self.optimizer.prepare_capture(defer_table_copy=True)
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
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            loss_dict = self.model(self.static["images"], scb)
            losses = sum(loss_dict.values())
            losses.backward()
            self._sync_grads()
            if self.optimizer is not None:
                self.optimizer.step()
        self.graph, self.out = g, {k: v.detach() for k, v in loss_dict.items()}
        del loss_dict, losses
        if isinstance(self.optimizer, FusedAdam):
            self.optimizer.finish_capture()  # the gradient address table of the recorded Adam launch: copied once, here
        self.grads = [(p, p.grad) for p in self.model.parameters() if p.grad is not None]  # live in the graph's pool
        self.signature = self._sig(images, cb)
        # the recorded Adam launch has the ADDRESSES of this plan's tables baked in: hold it (so the allocator cannot
        # recycle them under the graph) and replay only while the optimizer still uses this very object
        self.plan = getattr(self.optimizer, "_plan", None)
        if self.launch == "streams":
            self._build_replayer(g)
        self.log.info("train step captured: one %s per step from here on", "call into the library (stream replay)" if self.replayer else "graph launch")

    def _build_replayer(self, g):
        """The recorded graph read back into a stream-launch plan.  Under data parallelism the RCCL nodes of the recording are
        kernels on RCCL's stream like any other; a graph holding a node type the replayer refuses (host callback, child graph)
        stays on hipGraphLaunch."""
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

    def _try_capture(self, images, cb):
        """Record the step; when the RECORDING fails (a non-capturable call, an invalidated capture) keep training eagerly -
        eager is the documented fall-back for everything else, a failed recording must not end the run at step 3.  Only
        capture failures are absorbed: an out-of-memory error, or a device that still refuses a probe launch after the
        aborted recording was torn down (a real kernel fault, a sticky HIP error), is re-raised.  Under data parallelism
        the ranks agree on the outcome (one MAX all-reduce of the failure flag): either all replay or all stay eager."""
        from .. import ops

        saved = [(p, p.grad) for p in self.model.parameters()]
        ok = True
        try:
            self._capture(images, cb)
        except torch.cuda.OutOfMemoryError:
            raise
        except RuntimeError as e:  # what torch / the library raise for a failed or invalidated stream capture
            ok = False
            self.log.warning("train step: hipGraph capture failed (%s: %s) - the step stays eager for the rest of the run",
                             type(e).__name__, str(e).splitlines()[0] if str(e) else "")
            self._abort_capture(saved)
            # is the device itself still healthy?  (a capture error leaves it usable; a kernel fault does not)
            torch.cuda.synchronize()
            probe = ops.amax_slot(images.device)
            ops.call("trid_amax_f32", ops._p(self.static_probe(images)), 1, ops._p(probe), ops.stream())
            torch.cuda.synchronize()
        if dp_active():
            import torch.distributed as dist

            flag = torch.tensor([0.0 if ok else 1.0], device=images.device if dist.get_backend() != "gloo" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if ok and float(flag.item()) > 0:
                self.log.warning("train step: another rank could not record the step - dropping this rank's graph, all ranks stay eager")
                self._abort_capture(saved)
                ok = False
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
            self.optimizer.advance_for_replay()
        if self.replayer is not None and not self.force_graph_launch:
            from .. import ops

            ops.call("trid_step_replay_run", self.replayer, ops.stream())
        else:
            self.graph.replay()
        return self.out
