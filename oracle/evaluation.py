"""Oracle: retrieval match + rank metric.

Follows reference ``lib/data/metrics/evaluation.py``:
  similarity  :117-120   (L2-normalise, text @ image.T)
  rank        :11-37     (top-k or full argsort, CMC, mAP; AP is NaN for a
                          query with no relevant gallery item, reproduced)
Test infrastructure only.
"""

import torch
import torch.nn.functional as F


def similarity(text_embed, image_embed):
    return F.normalize(text_embed, p=2, dim=1) @ F.normalize(image_embed, p=2, dim=1).t()


def rank(sim, q_pids, g_pids, topk=(1, 5, 10), get_mAP=True):
    topk = torch.as_tensor(topk)
    max_rank = int(topk.max())
    if get_mAP:
        indices = torch.argsort(sim, dim=1, descending=True)
    else:
        indices = torch.topk(sim, k=max_rank, dim=1, largest=True, sorted=True)[1]
    matches = g_pids[indices].eq(q_pids.view(-1, 1))
    cmc = matches[:, :max_rank].cumsum(1).clamp(max=1).float().mean(0) * 100
    cmc = cmc[topk - 1]
    if not get_mAP:
        return cmc, indices
    num_rel = matches.sum(1)
    cum = matches.cumsum(1)
    ranks = torch.arange(1, matches.shape[1] + 1, dtype=torch.float32)
    prec = cum / ranks[None] * matches
    AP = prec.sum(1) / num_rel
    return cmc, AP.mean() * 100, indices


def k_reciprocal(q_feats, g_feats, neighbor_num=5, alpha=0.05):
    """evaluation.py:53-65 (+ jaccard_mat :44-50): alpha * Jaccard similarity of the top-k
    neighbour sets (in the gallery) of every query i and every gallery item j.  float64 as the
    reference (its numpy matrix is float64)."""
    qg_nn = torch.argsort(q_feats @ g_feats.t(), dim=1, descending=True)[:, :neighbor_num]
    gg_nn = torch.argsort(g_feats @ g_feats.t(), dim=1, descending=True)[:, :neighbor_num]
    eq = (qg_nn[:, None, :, None] == gg_nn[None, :, None, :]).sum(dim=(2, 3)).double()  # |A n B|
    return alpha * eq / (2 * neighbor_num - eq)
