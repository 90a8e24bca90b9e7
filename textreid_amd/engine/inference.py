"""Counterpart of the reference eval loop (``lib/engine/inference.py:14-96``):
eval-mode encode of every batch -> prediction dict {dataset_idx: [v[256], t[256]]}
-> ``evaluation`` (rerank=False).  Cross-rank accumulation uses one tensor
all-gather of the packed embeddings instead of pickling dicts (comm.py:47-87)."""

import logging
import time

import torch

from ..evaluation import evaluation
from ..parallel import rank, world_size


def compute_on_dataset(model, data_loader, device, dedupe=True, encode_batch=512):
    """Eval encode.  The reference encodes the image once per CAPTION although only unique images
    are kept afterwards (inference.py:17-25, evaluation.py:113-115: 6156 image forwards for 3074
    images on CUHK-PEDES).  With ``dedupe`` each distinct image id is encoded once and its
    embedding reused (row f2); results are identical.
    ``encode_batch``: images that still need encoding are collected ACROSS loader batches and run through the image encoder
    that many at a time (eval-mode BatchNorm is per-sample independent, so the grouping does not change a result): the
    encoder's M = B x 192-row launches of layer3 / layer4 fill the chip from ~256 images on (16.2 k imgs/s at 128 per
    pass, 18.0 k at 512 - DESIGN section 8).  0 / None: one encoder pass per loader batch, as the reference."""
    model.eval()
    head = model.embed_model
    dataset = getattr(data_loader, "dataset", None)
    can_dedupe = dedupe and dataset is not None and hasattr(dataset, "get_id_info")
    results, cache = {}, {}
    stats = {"images_encoded": 0, "samples": 0, "encoder_passes": 0}
    pend_imgs, pend_keys, waiting = [], [], []  # images queued for the encoder, their cache keys, samples waiting for them

    def flush():
        if pend_imgs:
            # in slices of at most encode_batch images: what is pending is up to encode_batch - 1 images plus one loader batch, and
            # a pass beyond the encoder's pre-split (P16) size limit would silently take the slower on-the-fly-split path - and
            # peak activation memory would follow the loader's batch size instead of encode_batch
            imgs = torch.cat(pend_imgs, dim=0)
            step = int(encode_batch) if encode_batch else imgs.shape[0]
            for s0 in range(0, imgs.shape[0], step):
                with torch.no_grad():
                    emb = head.encode_images(imgs[s0 : s0 + step].to(device))
                for r, key in enumerate(pend_keys[s0 : s0 + step]):
                    cache[key] = emb[r]
                stats["encoder_passes"] += 1
            stats["images_encoded"] += len(pend_keys)
            pend_imgs.clear()
            pend_keys.clear()
        for i, key, t_row in waiting:
            results[i] = [cache[key], t_row]
        waiting.clear()
        if not can_dedupe:
            cache.clear()  # (per-sample keys: nothing is reused)

    for images, captions, idxs in data_loader:
        idxs = [int(i) for i in idxs]
        captions = captions.to(device) if hasattr(captions, "to") else [c.to(device) for c in captions]
        with torch.no_grad():
            t = head.encode_captions(captions)
        keys = [dataset.get_id_info(i)[0] for i in idxs] if can_dedupe else [("sample", i) for i in idxs]
        queued = set(pend_keys)
        new = {}
        for j, key in enumerate(keys):
            if key not in cache and key not in queued and key not in new:
                new[key] = j
        if new:
            pend_imgs.append(images[torch.tensor(list(new.values()))])
            pend_keys.extend(new.keys())
        waiting.extend((i, key, t[j]) for j, (i, key) in enumerate(zip(idxs, keys)))
        stats["samples"] += len(idxs)
        if not encode_batch or len(pend_keys) >= encode_batch:
            flush()
    flush()
    compute_on_dataset.last_stats = stats
    return results


def _gather_predictions(pred, device=None):
    """{dataset_idx: [v[C], t[C]]} of every rank -> the merged dict on rank 0 (None elsewhere): ONE packed all-gather of
    [n_max, 2C + 2] rows per rank instead of the reference's pickled dicts (`lib/utils/comm.py:47-87`,
    `lib/engine/inference.py:28-45`).  The dataset index travels as its own bits (an int64 = two fp32 lanes, exact for
    any index), the pack is one stack per rank and the unpack one host transfer - no per-sample launches or syncs.
    The merged embeddings live where the local ones do (`device` for a rank without samples), on every world size."""
    if world_size() == 1:
        return pred
    import torch.distributed as dist

    from ..parallel import all_gather_rows

    W = world_size()
    keys = sorted(pred)
    sizes = [None] * W
    dist.all_gather_object(sizes, len(keys))
    nmax = max(sizes)
    if nmax == 0:
        return {} if rank() == 0 else None
    # a rank without samples still takes part: the embedding width comes from a rank that has some, the device is ITS OWN
    # (the collective runs on this rank's GPU; another rank's device string names a GPU this rank does not own)
    if keys:
        v = torch.stack([pred[k][0].reshape(-1) for k in keys]).float()
        t = torch.stack([pred[k][1].reshape(-1) for k in keys]).float()
        C, dev = v.shape[1], v.device
    widths = [None] * W
    dist.all_gather_object(widths, C if keys else None)
    if not keys:
        C = next(c for c in widths if c is not None)
        if device is not None:
            dev = torch.device(device)
            if dev.type == "cuda" and dev.index is None:
                dev = torch.device("cuda", torch.cuda.current_device())
        else:
            dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        v = t = torch.zeros(0, C, device=dev)
    idx = torch.tensor(keys, dtype=torch.int64, device=dev).view(-1, 1).view(torch.float32)  # [n, 2] bit lanes
    packed = torch.zeros(nmax, 2 * C + 2, device=dev)
    packed[: len(keys)] = torch.cat([idx.view(len(keys), 2), v, t], dim=1)
    out = all_gather_rows(packed)
    if rank() != 0:
        return None
    all_keys = out[:, :2].contiguous().view(torch.int64).view(-1).tolist()  # ONE host read for every rank's block
    merged = {}
    for w, n in enumerate(sizes):
        for j in range(n):
            row = w * nmax + j
            merged[all_keys[row]] = [out[row, 2 : C + 2], out[row, C + 2 :]]
    return merged


def inference(model, data_loader, dataset_name="cuhkpedes-test", device="cuda", output_folder="", save_data=True,
              rerank=False):
    logger = logging.getLogger("PersonSearch.inference")
    dataset = data_loader.dataset
    logger.info("Start evaluation on %s dataset(%d images).", dataset_name, len(dataset))
    t0 = time.time()
    predictions = _gather_predictions(compute_on_dataset(model, data_loader, torch.device(device)), device=device)
    logger.info("Total inference time: %.1fs", time.time() - t0)
    if predictions is None:
        return None
    return evaluation(dataset=dataset, predictions=predictions, output_folder=output_folder, save_data=save_data,
                      rerank=rerank, topk=[1, 5, 10])
