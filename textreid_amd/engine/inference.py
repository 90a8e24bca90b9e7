"""Counterpart of the reference eval loop (``lib/engine/inference.py:14-96``):
eval-mode encode of every batch -> prediction dict {dataset_idx: [v[256], t[256]]}
-> ``evaluation`` (rerank=False).  Cross-rank accumulation uses one tensor
all-gather of the packed embeddings instead of pickling dicts (comm.py:47-87)."""

import logging
import time

import torch

from ..evaluation import evaluation
from ..parallel import rank, world_size


def compute_on_dataset(model, data_loader, device, dedupe=True):
    """Eval encode.  The reference encodes the image once per CAPTION although only unique images
    are kept afterwards (inference.py:17-25, evaluation.py:113-115: 6156 image forwards for 3074
    images on CUHK-PEDES).  With ``dedupe`` each distinct image id is encoded once and its
    embedding reused (row f2); results are identical."""
    model.eval()
    head = model.embed_model
    dataset = getattr(data_loader, "dataset", None)
    can_dedupe = dedupe and dataset is not None and hasattr(dataset, "get_id_info")
    results, cache = {}, {}
    stats = {"images_encoded": 0, "samples": 0}
    for images, captions, idxs in data_loader:
        idxs = [int(i) for i in idxs]
        captions = captions.to(device) if hasattr(captions, "to") else [c.to(device) for c in captions]
        with torch.no_grad():
            t = head.encode_captions(captions)
            if can_dedupe:
                iids = [dataset.get_id_info(i)[0] for i in idxs]
                todo = {}
                for j, iid in enumerate(iids):
                    if iid not in cache and iid not in todo:
                        todo[iid] = j
                if todo:
                    sel = torch.tensor(list(todo.values()))
                    emb = head.encode_images(images[sel].to(device))
                    for r, iid in enumerate(todo):
                        cache[iid] = emb[r]
                    stats["images_encoded"] += len(todo)
                v = [cache[iid] for iid in iids]
            else:
                emb = head.encode_images(images.to(device))
                v = [emb[j] for j in range(len(idxs))]
                stats["images_encoded"] += len(idxs)
        for j, i in enumerate(idxs):
            results[i] = [v[j], t[j]]
        stats["samples"] += len(idxs)
    compute_on_dataset.last_stats = stats
    return results


def _gather_predictions(pred):
    if world_size() == 1:
        return pred
    import torch.distributed as dist

    keys = sorted(pred)
    dev = pred[keys[0]][0].device
    n = torch.tensor([len(keys)], device=dev)
    sizes = [torch.zeros_like(n) for _ in range(world_size())]
    dist.all_gather(sizes, n)
    nmax = int(max(int(s) for s in sizes))
    C = pred[keys[0]][0].numel()
    packed = torch.zeros(nmax, 2 * C + 1, device=dev)
    for r, k in enumerate(keys):
        packed[r, 0] = k
        packed[r, 1 : C + 1] = pred[k][0]
        packed[r, C + 1 :] = pred[k][1]
    out = torch.empty(world_size() * nmax, 2 * C + 1, device=dev)
    dist.all_gather_into_tensor(out, packed)
    if rank() != 0:
        return None
    merged = {}
    for w, s in enumerate(sizes):
        blk = out[w * nmax : w * nmax + int(s)]
        for row in blk:
            merged[int(row[0].round())] = [row[1 : C + 1], row[C + 1 :]]
    return merged


def inference(model, data_loader, dataset_name="cuhkpedes-test", device="cuda", output_folder="", save_data=True,
              rerank=False):
    logger = logging.getLogger("PersonSearch.inference")
    dataset = data_loader.dataset
    logger.info("Start evaluation on %s dataset(%d images).", dataset_name, len(dataset))
    t0 = time.time()
    predictions = _gather_predictions(compute_on_dataset(model, data_loader, torch.device(device)))
    logger.info("Total inference time: %.1fs", time.time() - t0)
    if predictions is None:
        return None
    return evaluation(dataset=dataset, predictions=predictions, output_folder=output_folder, save_data=save_data,
                      rerank=rerank, topk=[1, 5, 10])
