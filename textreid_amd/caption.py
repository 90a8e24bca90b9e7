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


def _as_rows(text, dtype, device):
    """Token input in any reference-accepted form -> list of 1-D tensors."""
    if isinstance(text, torch.Tensor):
        t = text.to(dtype)
        return [t] if t.dim() == 1 else list(t.reshape(-1, t.shape[-1]))
    if len(text) and not isinstance(text[0], (list, tuple, torch.Tensor)):
        text = [text]
    return [torch.as_tensor(row, dtype=dtype, device=device) for row in text]


class Caption:
    """One caption record: ``text`` is the zero-padded (or truncated) token row(s) [n, max_length],
    ``length`` the unpadded token count(s); free-form per-sample fields (``id``, ``image_id`` ...)
    ride along in ``extra_fields``.  A raw ``str`` caption is kept as is (word count as length)."""

    __slots__ = ("text", "length", "max_length", "padded", "dtype", "extra_fields", "host_length")

    def __init__(self, text, length=None, max_length=None, padded=False, dtype=torch.int64):
        self.dtype, self.padded, self.extra_fields = dtype, True, {}
        self.host_length = None  # largest unpadded token count, known on the host when built from host data
        if isinstance(text, str):
            self.text, self.max_length = text, max_length
            self.length = len(text.split()) if length is None else length
            return
        device = text.device if isinstance(text, torch.Tensor) else torch.device("cpu")
        if padded:  # already a padded tensor (the .to() path): adopt it
            self.text = torch.as_tensor(text, dtype=dtype, device=device)
            self.length = length
            self.max_length = self.text.shape[-1] if max_length is None else max_length
            return
        rows = _as_rows(text, dtype, device)
        counts = [int(r.numel()) for r in rows]
        single = isinstance(text, torch.Tensor) and text.dim() == 1
        if length is None:
            length = torch.tensor(counts[0] if single else counts, dtype=torch.int64, device=device)
            self.host_length = max(counts)  # token counts came from host data: no device read needed later
        self.length = length
        self.max_length = max(counts) if max_length is None else max_length
        if self.host_length is not None:
            self.host_length = min(self.host_length, self.max_length)
        self.text = self.pad(rows, self.max_length, device)

    @staticmethod
    def pad(rows, max_length, device):
        out = torch.zeros(len(rows), max_length, dtype=torch.int64, device=device)
        for i, row in enumerate(rows):
            n = min(int(row.numel()), max_length)
            out[i, :n] = row[:n]
        return out

    # -- per-sample side data ------------------------------------------------------------
    def add_field(self, field, field_data):
        self.extra_fields[field] = field_data

    def get_field(self, field):
        return self.extra_fields[field]

    def has_field(self, field):
        return field in self.extra_fields

    def fields(self):
        return list(self.extra_fields)

    def to(self, device):
        if isinstance(self.text, str):
            moved = Caption(self.text, self.length, self.max_length, dtype=self.dtype)
        else:
            moved = Caption(self.text.to(device), self.length.to(device), self.max_length, padded=True, dtype=self.dtype)
        moved.extra_fields = {k: (v.to(device) if hasattr(v, "to") else v) for k, v in self.extra_fields.items()}
        moved.host_length = self.host_length
        return moved

    def __len__(self):
        return len(self.text)

    def __repr__(self):
        return f"Caption(length={self.length}, max_length={self.max_length}, fields={self.fields()})"


class CaptionBatch:
    """tokens [B,L] i64, lengths [B] i64, ids [B] i64 (or None), max_len: host int."""

    def __init__(self, tokens, lengths, ids=None, max_len=None, bound_only=False):
        """max_len: the batch-maximum token count when the host knows it (else one device read); bound_only: max_len
        is merely an upper bound of it (the recorded train step, engine/graph.py) - the text encoder then loops max_len
        steps and takes the true maximum, which the reference's padding semantics depend on, from the device."""
        self.tokens = tokens
        self.lengths = lengths.view(-1)
        self.ids = ids.view(-1) if ids is not None else None
        self.max_len = int(max_len) if max_len is not None else int(self.lengths.max())
        self.bound_only = bool(bound_only)

    @classmethod
    def from_list(cls, captions):
        """Stack a list of reference-style Caption objects (gru.py:49-53, head.py:130)."""
        if isinstance(captions, CaptionBatch):
            return captions
        tokens = torch.stack([c.text.view(-1) for c in captions], dim=0)
        lengths = torch.stack([c.length.view(-1)[0] for c in captions], dim=0)
        # batch-max length on the HOST when the captions carry it (no device round trip per encoder call)
        host_max = None
        if all(getattr(c, "host_length", None) is not None for c in captions):
            host_max = max(int(c.host_length) for c in captions)
        ids = None
        if len(captions) and captions[0].has_field("id"):
            ids = torch.stack([torch.as_tensor(c.get_field("id")).view(-1)[0] for c in captions], dim=0).long()
            ids = ids.to(tokens.device)
        return cls(tokens, lengths, ids, max_len=host_max)

    def to(self, device):
        return CaptionBatch(self.tokens.to(device), self.lengths.to(device),
                            self.ids.to(device) if self.ids is not None else None, self.max_len, self.bound_only)

    def __len__(self):
        return self.tokens.shape[0]
