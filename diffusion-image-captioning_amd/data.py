"""Data side of the path -- SURVEY.md section 8(f) row 4: the batch-dict schema the training step consumes
(ref CLIP-DDPM.py:167-221): pre-extracted CLIP image/text features resident on the device + tokenised captions,
an 80/20 random split, DataLoader(shuffle, drop_last=True).  The Flickr/COCO files themselves are not in the reference repo;
`ClipCaptionDataset` takes the tensors (however they were produced), `synthetic_dataset` builds them from synth.py.
"""
from __future__ import annotations

import torch

from . import synth
from .config import cfg


class ClipCaptionDataset:
    """image_clip/text_clip [n,512] f32, input_ids/attention_mask [n,L] i64 (+ optional text / image name lists)."""

    def __init__(self, image_clip, text_clip, input_ids, attention_mask, text=None, image=None, device="cuda:0"):
        dev = torch.device(device)
        self.t = dict(image_clip=torch.as_tensor(image_clip, dtype=torch.float32).to(dev),
                      text_clip=torch.as_tensor(text_clip, dtype=torch.float32).to(dev),
                      input_ids=torch.as_tensor(input_ids, dtype=torch.int64).to(dev),
                      attention_mask=torch.as_tensor(attention_mask, dtype=torch.int64).to(dev))
        n = len(self.t["input_ids"])
        assert all(len(v) == n for v in self.t.values())
        self.text = list(text) if text is not None else None
        self.image = list(image) if image is not None else None

    def __len__(self):
        return len(self.t["input_ids"])

    def batch(self, idx):
        b = {k: v[idx] for k, v in self.t.items()}
        if self.text is not None:
            b["text"] = [self.text[i] for i in idx.tolist()]
        if self.image is not None:
            b["image"] = [self.image[i] for i in idx.tolist()]
        return b


class Loader:
    """`DataLoader(dataset_subset, shuffle, batch_size, drop_last=True)` (ref :220-221) over device-resident tensors."""

    def __init__(self, dataset: ClipCaptionDataset, indices, batch_size=None, shuffle=False, seed=0):
        self.ds, self.idx = dataset, torch.as_tensor(indices, dtype=torch.int64)
        self.bs = batch_size or cfg.BATCH_SIZE
        self.shuffle = shuffle
        self.gen = torch.Generator().manual_seed(seed)

    def __len__(self):
        return len(self.idx) // self.bs                      # drop_last=True

    def __iter__(self):
        order = self.idx[torch.randperm(len(self.idx), generator=self.gen)] if self.shuffle else self.idx
        for i in range(len(self)):
            yield self.ds.batch(order[i * self.bs:(i + 1) * self.bs].to(self.ds.t["input_ids"].device))


def random_split(n, train_ratio=None, seed=0):
    """ref :218-219: int(len * TRAIN_SET_RATIO) train items, rest validation, random permutation."""
    ratio = cfg.TRAIN_SET_RATIO if train_ratio is None else train_ratio
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(seed))
    k = int(n * ratio)
    return perm[:k], perm[k:]


def synthetic_dataset(n, max_length=None, vocab=None, seed=1, device="cuda:0"):
    b = synth.batch(n, max_length or cfg.MAX_LENGTH, vocab or cfg.VOCAB_SIZE, seed)
    return ClipCaptionDataset(b["image_clip"], b["text_clip"], b["input_ids"], b["attention_mask"], device=device)
