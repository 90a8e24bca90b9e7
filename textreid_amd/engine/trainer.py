"""Device-agnostic counterpart of the reference train loop (``lib/engine/trainer.py:38-139``):
same op order per step -- model(images, captions) -> sum(loss_dict) -> zero_grad ->
backward -> step -- per-epoch scheduler.step(), evaluation every EVALUATE_PERIOD
epochs with rerank=False, best / periodic checkpoints.  Differences: gradients of
pre-gather parameters are SUM-reduced over RCCL by ``GradReducer`` (the reference's
DDP path crashes), and logging keeps the per-step losses ON DEVICE and reads them back once every
``log_period`` steps (one host sync per period instead of one per loss per step); the meters still receive every
step's value, in order, as ``trainer.py:92-93`` feeds them."""

import logging
import os
import time

import torch

from ..parallel import GradReducer, broadcast_module_state, check_replicas, dp_active, world_size
from ..parallel import _backend as parallel_backend
from .inference import inference


def train_step(model, optimizer, images, captions, reducer=None, pre_gather=None):
    loss_dict = model(images, captions)
    losses = sum(loss for loss in loss_dict.values())
    optimizer.zero_grad()
    losses.backward()
    if reducer is not None and dp_active():
        reducer.reduce(pre_gather)
        reducer.wait()
    optimizer.step()
    return loss_dict, losses


def do_train(model, data_loader, data_loader_val, optimizer, scheduler, checkpointer, meters, device,
             checkpoint_period, evaluate_period, arguments, log_period=20, capture=True):
    """capture: record the step as one hipGraph after two eager steps (engine/graph.py) when the optimizer is the fused
    Adam, the model runs on a GPU and there is one process; batches of another shape run eagerly."""
    logger = logging.getLogger("PersonSearch.trainer")
    logger.info("Start training")
    max_epoch, epoch, iteration = arguments["max_epoch"], arguments["epoch"], arguments["iteration"]
    reducer = GradReducer()
    runner = None
    pre_gather = [p for n, p in model.named_parameters() if p.requires_grad and "loss_evaluator" not in n][::-1]
    if dp_active() and hasattr(getattr(model, "embed_model", None), "v_encoder_q"):
        model.embed_model.v_encoder_q.grad_sync = reducer  # conv gradients all-reduced from inside backward
    watch = []
    if dp_active():
        # every rank starts from rank 0's parameters, BatchNorm statistics, queues, ids and pointer (DDP's broadcast at wrap,
        # train_net.py:50-56) - the replicated queues must never depend on the ranks having drawn the same seeds
        nb = broadcast_module_state(model)
        logger.info("data parallel: %.1f MB of parameters and buffers broadcast from rank 0", nb / 2 ** 20)
        em = getattr(model, "embed_model", None)
        watch = [(n, getattr(em, n)) for n in ("queue_ptr", "id_queue", "v_queue", "t_queue") if hasattr(em, n)]
        if pre_gather:
            watch.append(("parameter %s" % next(n for n, p in model.named_parameters() if p is pre_gather[0]), pre_gather[0]))
        post = [(n, p) for n, p in model.named_parameters() if "loss_evaluator" in n]
        watch += [("parameter " + post[0][0], post[0][1])] if post else []
        check_replicas(watch, "after the initial broadcast")
    # data parallel: the step is recorded too - its collectives are cut points of the recording, the replay re-issues the recorded
    # kernels in segments and runs each collective through torch.distributed between two of them (engine/graph.py; ~4 ms of host
    # work per step instead of the eager step's 35-39 ms, so that the ranks feed xGMI instead of waiting for Python).
    # TRID_DP_CAPTURE=0: the eager data-parallel step (A/B runs).
    dp_capture = os.environ.get("TRID_DP_CAPTURE", "1") != "0"
    if capture and (not dp_active() or dp_capture) and torch.device(device).type == "cuda":
        from ..solver import FusedAdam
        from .graph import BucketedTrainStep

        if isinstance(optimizer, FusedAdam):
            # one recording per caption bucket (32 / 48 / 64 / 105 recurrence steps; TRID_CAPTION_BUCKETS overrides): a batch runs
            # the recording of the smallest bucket that holds its longest caption
            runner = BucketedTrainStep(model, optimizer, warmup=2, reducer=reducer if dp_active() else None, pre_gather=pre_gather)
    best_top1 = 0.0
    pending, keys = [], None  # per-step loss vectors still on the device

    def flush_meters():
        """One host read for all pending steps; the meters see every step's value (median / windowed average over
        step values, as in the reference)."""
        if not pending:
            return
        for row in torch.stack(pending).tolist():
            meters.update(loss=row[0], **{k: m for k, m in zip(keys, row[1:])})
        pending.clear()

    start = time.time()
    while epoch < max_epoch:
        epoch += 1
        model.train()
        arguments["epoch"] = epoch
        sampler = getattr(data_loader, "sampler", None)
        if world_size() > 1 and hasattr(sampler, "set_epoch"):
            sampler.set_epoch(epoch)  # trainer.py:66 (guarded: the reference's own call crashes on a batch_sampler loader)
        for step, (images, captions, _) in enumerate(data_loader):
            iteration += 1
            arguments["iteration"] = iteration
            images = images.to(device)
            captions = captions.to(device) if hasattr(captions, "to") else [c.to(device) for c in captions]
            if runner is not None:
                loss_dict = runner(images, captions)
                losses = sum(loss_dict.values())
            else:
                loss_dict, losses = train_step(model, optimizer, images, captions, reducer, pre_gather)
            if watch and iteration % log_period == 0:
                check_replicas(watch, "at iteration %d" % iteration)  # (data parallel: a 64-bit digest per watched tensor)
            if meters is not None:
                # every step counts (trainer.py:92-93 updates the meters per step): kept on device, one read per period
                with torch.no_grad():
                    keys = list(loss_dict.keys())
                    pending.append(torch.stack([losses.detach()] + [v.detach() for v in loss_dict.values()]))
                if iteration % log_period == 0:
                    flush_meters()  # the period's only host read
                    logger.info("epoch [%d][%d/%d] %s lr: %.6f", epoch, step, len(data_loader), str(meters),
                                optimizer.param_groups[-1]["lr"])
        scheduler.step()
        if data_loader_val is not None and epoch % evaluate_period == 0:
            top1 = inference(model, data_loader_val[0], device=device, save_data=False, rerank=False)
            if meters is not None and top1 is not None:
                meters.update(top1=float(top1))  # trainer.py:124
            if top1 is not None and float(top1) > best_top1:
                best_top1 = float(top1)
                if checkpointer is not None:
                    checkpointer.save("best", **arguments)
        if checkpointer is not None and epoch % checkpoint_period == 0:
            checkpointer.save("epoch_{:d}".format(epoch), **arguments)
    if meters is not None:
        flush_meters()  # trailing partial period
    logger.info("Total training time: %.1fs", time.time() - start)
