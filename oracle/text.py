"""Oracle: frozen token-embedding lookup + 1-layer bias-free BiGRU + max-over-time.

Follows reference ``lib/models/backbones/gru.py``:
  table gather        :55-58
  sort/pack/GRU/pad   :66-82   (restated as an explicit masked time loop)
  max over time       :63      (zero pad rows up to the BATCH-max length enter
                                the max; reproduced here)
GRU cell math is torch.nn.GRU's (gate order r,z,n; bias=False):
  r = sigmoid(W_ir x + W_hr h); z = sigmoid(W_iz x + W_hz h)
  n = tanh(W_in x + r * (W_hn h)); h' = (1-z)*n + z*h
Test infrastructure only.
"""

import torch

GRU_KEYS = ("gru.weight_ih_l0", "gru.weight_hh_l0", "gru.weight_ih_l0_reverse", "gru.weight_hh_l0_reverse")


def state_shapes(hidden=512, embed=512):
    return {
        "gru.weight_ih_l0": (3 * hidden, embed),
        "gru.weight_hh_l0": (3 * hidden, hidden),
        "gru.weight_ih_l0_reverse": (3 * hidden, embed),
        "gru.weight_hh_l0_reverse": (3 * hidden, hidden),
    }


def _cell(gi, h, w_hh):
    H = h.shape[1]
    gh = h @ w_hh.t()
    r = torch.sigmoid(gi[:, :H] + gh[:, :H])
    z = torch.sigmoid(gi[:, H : 2 * H] + gh[:, H : 2 * H])
    n = torch.tanh(gi[:, 2 * H :] + r * gh[:, 2 * H :])
    return (1.0 - z) * n + z * h


def _direction(x, lengths, w_ih, w_hh, reverse, lmax):
    """x [B,L,E]; returns outputs [B,lmax,H] with zeros at t >= length (what
    pad_packed_sequence produces, gru.py:78-79)."""
    B = x.shape[0]
    H = w_hh.shape[1]
    gi_all = x[:, :lmax] @ w_ih.t()  # [B,lmax,3H]
    h = x.new_zeros(B, H)
    outs = [None] * lmax
    steps = range(lmax - 1, -1, -1) if reverse else range(lmax)
    for t in steps:
        m = (lengths > t).to(x.dtype).unsqueeze(1)  # packed sequence: sample active iff t < len
        hn = _cell(gi_all[:, t], h, w_hh)
        h = m * hn + (1.0 - m) * h  # inactive: state frozen (fwd) / still zero (rev)
        outs[t] = m * h
    return torch.stack(outs, dim=1)


def text_forward(st, table, tokens, lengths):
    """tokens [B,Lpad] i64, lengths [B] i64, table [V,E] f32 (frozen, not a
    parameter: gru.py:34) -> [B,2H].  The non-default input forms of gru.py:22-31,59-60: table None and "embed.weight"
    [V,E] in `st` = a trainable nn.Embedding(padding_idx=0) (`use_onehot == "yes"`); "embed.weight" [E,V'] + "embed.bias"
    in `st` beside a table [V,V'] = the frozen rows through nn.Linear(vocab_size, embed_size)."""
    lengths = lengths.view(-1)
    lmax = int(lengths.max())
    if table is None:
        x = torch.nn.functional.embedding(tokens, st["embed.weight"], padding_idx=0)  # gru.py:23-24,59-60
    else:
        x = table[tokens.reshape(-1)].reshape(tokens.shape[0], tokens.shape[1], -1)  # gru.py:55-58
        if "embed.weight" in st:
            x = torch.nn.functional.linear(x, st["embed.weight"], st["embed.bias"])  # gru.py:29-30,59-60
    of = _direction(x, lengths, st["gru.weight_ih_l0"], st["gru.weight_hh_l0"], False, lmax)
    ob = _direction(x, lengths, st["gru.weight_ih_l0_reverse"], st["gru.weight_hh_l0_reverse"], True, lmax)
    out = torch.cat([of, ob], dim=2)  # [B,lmax,2H]
    return out.max(dim=1)[0]  # gru.py:63
