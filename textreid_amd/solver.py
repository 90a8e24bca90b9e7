"""Optimiser / LR schedule with the reference's rules (``lib/solver/build.py:6-58``,
``lib/solver/lr_scheduler.py``): one param group per tensor, bias lr x
BIAS_LR_FACTOR and bias weight decay WEIGHT_DECAY_BIAS, Adam / AdamW, warm-up +
step / exp / poly / cosine / linear decay per epoch.

``FusedAdam`` keeps ``torch.optim.Adam``'s state layout and per-group
hyper-parameters (so LR schedulers and checkpoints work) but executes the whole
step as ONE multi-tensor HIP kernel over a device pointer table instead of one
launch set per parameter group (the reference builds 183 groups).
"""

from bisect import bisect_right
from math import cos, pi

import numpy as np
import torch

from . import ops

ADAM_CHUNK = 65536


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, decoupled=False):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.decoupled = decoupled
        self._plan = None  # device pointer tables; rebuilt whenever a parameter, gradient or moment tensor moves

    def load_state_dict(self, state_dict):
        """torch.optim.Adam-compatible state (resume: Checkpointer.load -> optimizer.load_state_dict).  The loaded
        moment tensors replace the ones the cached pointer tables refer to, and the bias-correction step count
        lives in the per-parameter ``state[p]["step"]`` entries, which are restored here - nothing else to keep."""
        super().load_state_dict(state_dict)
        for st in self.state.values():  # torch saves the step as a tensor: one conversion here, no host sync per step later
            if "step" in st and not isinstance(st["step"], int):
                st["step"] = int(st["step"])
        self._plan = None

    def __setstate__(self, state):
        super().__setstate__(state)
        self._plan = None

    def _build(self, items):
        """Static part of the plan: parameter / moment pointer tables and the chunk map (re-made only when a parameter
        or moment tensor moves).  Gradient tensors are fresh allocations every step: their table goes up per step
        through pinned staging buffers with an asynchronous copy - a pageable host-to-device copy would wait for the
        whole backward pass and stop the host from running ahead of the GPU."""
        dev = items[0][0].device
        for p, g, st in items:
            if not (p.is_contiguous() or p.is_contiguous(memory_format=torch.channels_last)):
                raise RuntimeError("FusedAdam needs dense parameters")
        ct, co = [], []
        for i, (p, _, _) in enumerate(items):
            for off in range(0, p.numel(), ADAM_CHUNK):
                ct.append(i)
                co.append(off)
        n = len(items)

        def up(values, dtype):
            return torch.from_numpy(np.asarray(values, dtype=dtype)).to(dev)

        self._plan = {
            "key": self._key(items),
            "p": up(np.array([p.data_ptr() for p, _, _ in items], dtype=np.uint64).view(np.int64), np.int64),
            "m": up(np.array([s["exp_avg"].data_ptr() for _, _, s in items], dtype=np.uint64).view(np.int64), np.int64),
            "v": up(np.array([s["exp_avg_sq"].data_ptr() for _, _, s in items], dtype=np.uint64).view(np.int64), np.int64),
            "sizes": up([p.numel() for p, _, _ in items], np.int64),
            "ct": up(ct, np.int32),
            "co": up(co, np.int64),
            "n": len(ct),
            # per-step tables in one row of 8-byte slots: [n gradient pointers | n fp32 lr | n fp32 weight decay |
            # n fp32 1-b1^t | n fp32 sqrt(1-b2^t)], double-buffered
            "dyn": [torch.empty(3 * n + 1, dtype=torch.int64, device=dev) for _ in range(2)],
            "stage": [torch.empty(3 * n + 1, dtype=torch.int64).pin_memory() if dev.type == "cuda" else torch.empty(3 * n + 1, dtype=torch.int64)
                      for _ in range(2)],
            "done": [None, None],
            "turn": 0,
        }

    @staticmethod
    def _key(items):
        return tuple((p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()) for p, _, st in items)

    def _upload(self, items, lrs, wds, bc1, bc2):
        """This step's gradient pointers, per-group lr / weight decay and per-parameter bias corrections -> device,
        without a host sync."""
        pl = self._plan
        n = len(items)
        k = pl["turn"]
        pl["turn"] = k ^ 1
        if pl["done"][k] is not None:
            pl["done"][k].synchronize()  # the copy issued two steps ago read this staging buffer (long finished)
        host = pl["stage"][k].numpy()
        host[:n] = np.array([g.data_ptr() for _, g, _ in items], dtype=np.uint64).view(np.int64)
        f = host[n:].view(np.float32)
        f[:n] = np.asarray(lrs, dtype=np.float32)
        f[n : 2 * n] = np.asarray(wds, dtype=np.float32)
        f[2 * n : 3 * n] = np.asarray(bc1, dtype=np.float32)
        f[3 * n : 4 * n] = np.asarray(bc2, dtype=np.float32)
        dyn = pl["dyn"][k]
        dyn.copy_(pl["stage"][k], non_blocking=True)
        if dyn.is_cuda:
            ev = torch.cuda.Event()
            ev.record()
            pl["done"][k] = ev
        return dyn

    # ---- hipGraph capture (engine/graph.py): the step's kernel launch is recorded once and replayed.  Gradient tensors of
    # a captured backward live at fixed addresses, so their pointer table is written by ONE captured copy from a pinned
    # buffer that never changes afterwards; what changes from replay to replay - lr / weight decay of every group (LR
    # schedulers), the per-parameter bias corrections - lives in the tail of the same device table and is refreshed by
    # `advance_for_replay()` with a stream-ordered copy from a small ring of pinned buffers right before each replay
    # (no host sync, no captured host memory that changes).
    def prepare_capture(self, defer_table_copy=False):
        """OUTSIDE capture (pinned allocations are not capturable), after at least one eager step built the static
        pointer tables: the device / pinned tables of the captured step.
        defer_table_copy: the copy of the captured gradients' addresses to the device is NOT recorded; the caller runs it once,
        right after the recording ended, with finish_capture() (the addresses never change afterwards; a recorded host-to-device
        copy is the one node of the step that csrc/step_replay.hip cannot read back from the graph)."""
        if self._plan is None:
            raise RuntimeError("FusedAdam.prepare_capture(): run one eager step first (static pointer tables)")
        pl = self._plan
        n = len(pl["sizes"])
        # one table set PER RECORDING (a run may hold several - engine.graph.BucketedTrainStep records the step once per caption
        # bucket -, each with its own captured gradient tensors): the recording made next uses `cap`, returned to the caller, who
        # passes it back to finish_capture() / advance_for_replay()
        cap = {
            "graph_n": n,
            "graph_dyn": torch.empty(3 * n + 1, dtype=torch.int64, device=pl["sizes"].device),
            "graph_gptr_host": torch.empty(n, dtype=torch.int64).pin_memory(),
            "graph_ring": [torch.empty(2 * n + 1, dtype=torch.int64).pin_memory() for _ in range(4)],
            "graph_ring_ev": [None] * 4,
            "graph_turn": 0,
            "graph_defer": bool(defer_table_copy),
            "graph_pending": False,
        }
        pl["cap"] = cap
        return cap

    def finish_capture(self, cap=None):
        """After the recording ended (prepare_capture(defer_table_copy=True)): the gradient address table -> device, once."""
        pl = cap if cap is not None else (self._plan or {}).get("cap")
        if pl is not None and pl.get("graph_pending"):
            n = pl["graph_n"]
            pl["graph_dyn"][:n].copy_(pl["graph_gptr_host"], non_blocking=True)
            pl["graph_pending"] = False

    def _capture_tables(self, items):
        """INSIDE the capture: the captured gradients' addresses -> the static table (a captured copy node)."""
        pl = self._plan.get("cap") or {}
        n = len(items)
        if pl.get("graph_n") != n:
            raise RuntimeError("FusedAdam.step() inside a stream capture needs prepare_capture() first (engine.graph.CapturedTrainStep "
                               "does it) and the same set of parameters with gradients as the eager steps")
        pl["graph_gptr_host"].numpy()[:] = np.array([g.data_ptr() for _, g, _ in items], dtype=np.uint64).view(np.int64)
        if pl.get("graph_defer"):
            pl["graph_pending"] = True  # (finish_capture() copies it once the recording has ended)
        else:
            pl["graph_dyn"][:n].copy_(pl["graph_gptr_host"], non_blocking=True)
        return pl["graph_dyn"]

    def advance_for_replay(self, cap=None):
        """Host side of one replayed step: per-parameter step counts, bias corrections, the groups' CURRENT lr / weight
        decay -> the device table of the recording about to be replayed (`cap`: what its prepare_capture() returned; None: the
        latest), stream-ordered."""
        pl = cap if cap is not None else self._plan["cap"]
        n = pl["graph_n"]
        lrs, wds, bc1, bc2 = [], [], [], []
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                t = int(st["step"]) + 1
                st["step"] = t
                lrs.append(group["lr"])
                wds.append(group["weight_decay"])
                bc1.append(1.0 - b1 ** t)
                bc2.append((1.0 - b2 ** t) ** 0.5)
        k = pl["graph_turn"]
        pl["graph_turn"] = (k + 1) % len(pl["graph_ring"])
        if pl["graph_ring_ev"][k] is not None:
            pl["graph_ring_ev"][k].synchronize()  # the copy issued four steps ago (long finished)
        f = pl["graph_ring"][k].numpy().view(np.float32)
        f[:n] = np.asarray(lrs, dtype=np.float32)
        f[n : 2 * n] = np.asarray(wds, dtype=np.float32)
        f[2 * n : 3 * n] = np.asarray(bc1, dtype=np.float32)
        f[3 * n : 4 * n] = np.asarray(bc2, dtype=np.float32)
        pl["graph_dyn"][n:].copy_(pl["graph_ring"][k], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        pl["graph_ring_ev"][k] = ev
        ops.note_parameter_write()

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        items, lrs, wds = [], [], []
        b1 = b2 = eps = None
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                g = p.grad
                if g.stride() != p.stride():  # gradient arrived in another dense layout: re-layout once
                    g = torch.empty_like(p, memory_format=torch.preserve_format).copy_(g)
                    p.grad = g
                items.append((p, g, st))
                lrs.append(group["lr"])
                wds.append(group["weight_decay"])
                gb1, gb2 = group["betas"]
                if b1 is None:
                    b1, b2, eps = gb1, gb2, group["eps"]
                elif (b1, b2, eps) != (gb1, gb2, group["eps"]):
                    raise RuntimeError("FusedAdam: betas/eps must be shared by all groups")
        if not items:
            return loss
        capturing = items[0][0].is_cuda and torch.cuda.is_current_stream_capturing()
        key = self._key(items)
        if self._plan is None or self._plan["key"] != key:
            if capturing:
                raise RuntimeError("FusedAdam: parameter / moment tensors moved since the eager warm-up steps; cannot build tables inside a capture")
            self._build(items)
        pl = self._plan
        n = len(items)
        if capturing:
            # nothing executes during a capture: step counts and hyper-parameters are advanced per REPLAY (advance_for_replay)
            dyn = self._capture_tables(items)
        else:
            # one step count PER PARAMETER, as torch.optim.Adam keeps it (a parameter that starts receiving gradients
            # later, or a resumed state with unequal counts, gets its own bias correction)
            bc1, bc2 = [], []
            for _, _, st in items:
                t = int(st["step"]) + 1
                st["step"] = t
                bc1.append(1.0 - b1 ** t)
                bc2.append((1.0 - b2 ** t) ** 0.5)
            dyn = self._upload(items, lrs, wds, bc1, bc2)
        ops.call("trid_adam_multi_f32", ops._p(pl["p"]), ops._p(dyn), ops._p(pl["m"]), ops._p(pl["v"]),
                 ops._p(pl["sizes"]), ops._p(dyn) + 8 * n, ops._p(dyn) + 12 * n, ops._p(pl["ct"]), ops._p(pl["co"]), pl["n"],
                 ADAM_CHUNK, float(b1), float(b2), float(eps), ops._p(dyn) + 16 * n, ops._p(dyn) + 20 * n,
                 1 if self.decoupled else 0, ops.stream())
        ops.note_parameter_write()  # raw-pointer write: torch's tensor._version does not move
        return loss


def make_optimizer(cfg, model, fused=True):
    params = []
    for key, value in model.named_parameters():
        if not value.requires_grad:
            continue
        lr = cfg.SOLVER.BASE_LR
        weight_decay = cfg.SOLVER.WEIGHT_DECAY
        if "bias" in key:
            lr = cfg.SOLVER.BASE_LR * cfg.SOLVER.BIAS_LR_FACTOR
            weight_decay = cfg.SOLVER.WEIGHT_DECAY_BIAS
        params.append({"params": [value], "lr": lr, "weight_decay": weight_decay})
    betas = (cfg.SOLVER.ADAM_ALPHA, cfg.SOLVER.ADAM_BETA)
    name = cfg.SOLVER.OPTIMIZER
    if name == "SGD":
        return torch.optim.SGD(params, lr=cfg.SOLVER.BASE_LR, momentum=cfg.SOLVER.SGD_MOMENTUM)
    if name in ("Adam", "AdamW"):
        if fused:
            return FusedAdam(params, lr=cfg.SOLVER.BASE_LR, betas=betas, eps=1e-8, decoupled=(name == "AdamW"))
        cls = torch.optim.Adam if name == "Adam" else torch.optim.AdamW
        return cls(params, lr=cfg.SOLVER.BASE_LR, betas=betas, eps=1e-8)
    raise NotImplementedError(name)


class LRSchedulerWithWarmup(torch.optim.lr_scheduler._LRScheduler):
    def __init__(self, optimizer, milestones, gamma=0.1, mode="step", warmup_factor=1.0 / 3, warmup_epochs=10,
                 warmup_method="linear", total_epochs=100, target_lr=0, power=0.9, last_epoch=-1):
        if list(milestones) != sorted(milestones):
            raise ValueError("SOLVER.STEPS must be sorted in increasing order, got %r" % (milestones,))
        if mode not in ("step", "exp", "poly", "cosine", "linear"):
            raise ValueError("unknown lr scheduler mode {}".format(mode))
        if warmup_method not in ("constant", "linear"):
            raise ValueError("unknown warmup_method {}".format(warmup_method))
        self.milestones, self.mode, self.gamma = milestones, mode, gamma
        self.warmup_factor, self.warmup_epochs, self.warmup_method = warmup_factor, warmup_epochs, warmup_method
        self.total_epochs, self.target_lr, self.power = total_epochs, target_lr, power
        super().__init__(optimizer, last_epoch)

    def _factor(self):
        e = self.last_epoch
        if e < self.warmup_epochs:
            if self.warmup_method == "constant":
                return ("mul", self.warmup_factor)
            a = e / self.warmup_epochs
            return ("mul", self.warmup_factor * (1 - a) + a)
        if self.mode == "step":
            return ("mul", self.gamma ** bisect_right(self.milestones, e))
        r = (e - self.warmup_epochs) / (self.total_epochs - self.warmup_epochs)
        if self.mode == "exp":
            return ("mul", self.power ** r)
        if self.mode == "linear":
            return ("mul", 1 - r)
        if self.mode == "poly":
            return ("to", self.power ** (1 - r))
        return ("to", 0.5 * (1 + cos(pi * r)))

    def get_lr(self):
        kind, f = self._factor()
        if kind == "mul":
            return [b * f for b in self.base_lrs]
        return [self.target_lr + (b - self.target_lr) * f for b in self.base_lrs]


def make_lr_scheduler(cfg, optimizer):
    s = cfg.SOLVER
    return LRSchedulerWithWarmup(optimizer, milestones=s.STEPS, gamma=s.GAMMA, warmup_factor=s.WARMUP_FACTOR,
                                 warmup_epochs=s.WARMUP_EPOCHS, warmup_method=s.WARMUP_METHOD, total_epochs=s.NUM_EPOCHS,
                                 mode=s.LRSCHEDULER, target_lr=s.TARGET_LR, power=s.POWER)
