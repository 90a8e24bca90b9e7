"""Caption containers at the L2 -> L3 boundary.

``Caption`` keeps the reference container's surface (``lib/utils/caption.py``:
padded token tensor ``text`` [1,L], ``length`` [1], ``extra_fields`` with ``id``,
``.to(device)``, ``get_field``), so the reference data pipeline plugs in.
``CaptionBatch`` is the struct-of-arrays form the kernels consume (tokens
[B,L], lengths [B], ids [B], host-known max length): one H2D copy per batch
instead of B per-sample ``Caption.to(device)`` calls, and no device->host sync
to learn the batch-max length (the reference syncs at gru.py:74).
"""

import torch


class Caption(object):
    def __init__(self, text, length=None, max_length=None, padded=False, dtype=torch.int64):
        device = text.device if isinstance(text, torch.Tensor) else torch.device("cpu")
        if isinstance(text, list):
            text = [torch.as_tensor(line, dtype=dtype, device=device) for line in text]
            if length is None:
                length = torch.stack([torch.tensor(line.size(0), dtype=torch.int64, device=device) for line in text])
            if max_length is None:
                max_length = max(line.size(-1) for line in text)
        elif not isinstance(text, str):
            text = torch.as_tensor(text, dtype=dtype, device=device)
            if length is None:
                length = torch.tensor(text.size(-1), dtype=torch.int64, device=device)
            if max_length is None:
                max_length = text.size(-1)
        elif length is None:
            length = len(text.split())
        if not padded and not isinstance(text, str):
            text = self.pad(text, max_length, device)
        self.text = text
        self.length = length
        self.max_length = max_length
        self.padded = True
        self.dtype = dtype
        self.extra_fields = {}

    @staticmethod
    def pad(text, max_length, device):
        rows = []
        for line in text:
            n = line.size(0)
            if n < max_length:
                rows.append(torch.cat((line, torch.zeros(max_length - n, dtype=torch.int64, device=device))))
            else:
                rows.append(line[:max_length])
        return torch.stack(rows)

    def add_field(self, field, field_data):
        self.extra_fields[field] = field_data

    def get_field(self, field):
        return self.extra_fields[field]

    def has_field(self, field):
        return field in self.extra_fields

    def fields(self):
        return list(self.extra_fields.keys())

    def to(self, device):
        cap = Caption(self.text, self.length, self.max_length, self.padded, self.dtype)
        if not isinstance(self.text, str):
            cap.text = cap.text.to(device)
            cap.length = cap.length.to(device)
        for k, v in self.extra_fields.items():
            cap.add_field(k, v.to(device) if hasattr(v, "to") else v)
        return cap

    def __len__(self):
        return len(self.text)

    def __repr__(self):
        return "Caption(length={}, max_length={}, padded={})".format(self.length, self.max_length, self.padded)


class CaptionBatch(object):
    """tokens [B,L] i64, lengths [B] i64, ids [B] i64 (or None), max_len: host int."""

    def __init__(self, tokens, lengths, ids=None, max_len=None):
        self.tokens = tokens
        self.lengths = lengths.view(-1)
        self.ids = ids.view(-1) if ids is not None else None
        self.max_len = int(max_len) if max_len is not None else int(self.lengths.max())

    @classmethod
    def from_list(cls, captions):
        """Stack a list of reference-style Caption objects (gru.py:49-53, head.py:130)."""
        if isinstance(captions, CaptionBatch):
            return captions
        tokens = torch.stack([c.text.view(-1) for c in captions], dim=0)
        lengths = torch.stack([c.length.view(-1)[0] for c in captions], dim=0)
        ids = None
        if len(captions) and captions[0].has_field("id"):
            ids = torch.stack([torch.as_tensor(c.get_field("id")).view(-1)[0] for c in captions], dim=0).long()
            ids = ids.to(tokens.device)
        return cls(tokens, lengths, ids)

    def to(self, device):
        return CaptionBatch(self.tokens.to(device), self.lengths.to(device),
                            self.ids.to(device) if self.ids is not None else None, self.max_len)

    def __len__(self):
        return self.tokens.shape[0]
