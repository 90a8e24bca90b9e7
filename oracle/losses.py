"""Oracle: the three losses the MoCo configs use, device-agnostic restatement.

Follows reference ``lib/models/losses.py``:
  instance_loss + CrossEntropyLabelSmooth  :42-62, :6-39
  global_align_loss                        :102-128
  infonce_loss                             :206-217
(The reference hard-codes ``.cuda()`` at :36 and :215; dropped here.)
Test infrastructure only.
"""

import torch
import torch.nn.functional as F


def label_smooth_ce(logits, labels, epsilon):
    # losses.py:25-39: targets=(1-eps)*onehot+eps/C ; (-t*logp).mean(0).sum()
    logp = F.log_softmax(logits, dim=1)
    C = logits.shape[1]
    t = torch.zeros_like(logp).scatter_(1, labels.view(-1, 1), 1.0)
    t = (1.0 - epsilon) * t + epsilon / C
    return (-t * logp).mean(0).sum()


def instance_loss(projection, v_embed, t_embed, labels, epsilon=0.0, scale=1, norm=False):
    # losses.py:42-62 (moco_head/loss.py:23-29 calls it with scale=1, norm=False).  Quirk kept: the reference constructs
    # CrossEntropyLabelSmooth(num_classes=...) without handing its `epsilon` on (losses.py:56), so ANY epsilon > 0 smooths
    # with that class's default weight 0.1 (losses.py:18)
    if norm:
        v_embed, t_embed = F.normalize(v_embed, p=2, dim=-1), F.normalize(t_embed, p=2, dim=-1)
    pn = F.normalize(projection, p=2, dim=0)
    lv = scale * (v_embed @ pn)
    lt = scale * (t_embed @ pn)
    if epsilon > 0:
        return label_smooth_ce(lv, labels, 0.1) + label_smooth_ce(lt, labels, 0.1)
    return F.cross_entropy(lv, labels) + F.cross_entropy(lt, labels)


def global_align_loss(v_embed, t_embed, labels, alpha=0.6, beta=0.4, scale_pos=10, scale_neg=40):
    # losses.py:102-128
    B = labels.shape[0]
    s = F.normalize(v_embed, dim=1) @ F.normalize(t_embed, dim=1).t()
    same = labels.view(-1, 1) == labels.view(1, -1)
    lp = torch.log(1 + torch.exp(-scale_pos * (s[same] - alpha)))
    ln = torch.log(1 + torch.exp(scale_neg * (s[~same] - beta)))
    return (lp.sum() + ln.sum()) * 2.0 / B


def infonce_loss(v_pos, v_neg, t_pos, t_neg, T=0.07):
    # losses.py:206-217: CE over [pos | negs]/T with label 0, v + t
    lv = torch.cat([v_pos, v_neg], dim=1) / T
    lt = torch.cat([t_pos, t_neg], dim=1) / T
    z = torch.zeros(lv.shape[0], dtype=torch.long)
    return F.cross_entropy(lv, z) + F.cross_entropy(lt, z)
