"""Dataset + collate functions (reference: src/loader.py), re-batched for data parallelism.

`StyleDataset` tokenises and truncates every line at load (loader.py:19-26) -- here through one
batched call into the tokenizer -- and can cache the token lists as a binary file next to the data.
The collate functions return exactly the reference's tuples (CPU int64 / float32 tensors):
  collate_pretrain -> (x, nx_1, nx_2, nx_3, label, c_label)      loader.py:46-70
  collate_warmup   -> (nx, x, label)                             loader.py:72-82
  collate_optimize -> (x, label)                                 loader.py:84-90
The Matcher regression label `c_label` is the reference's word-mover distance between the two
noised sentences computed with gensim/pyemd (src/wmd.py:31-45, out of scope -- gensim is not
installable here): `collate_pretrain` takes any `label_fn(noised_1, noised_2, vocab) -> list[float]`;
`overlap_distance_label` is a dependency-free stand-in (NOT numerically the reference's label).

Data parallelism: `GlobalBatchSampler` + a collate function build the GLOBAL batch on every rank
from the same seed (the noise functions mix tokens across the whole batch, so noise is applied
before sharding), pad to the global maximum length, and `parallel.shard_batch` takes this rank's rows.
"""
import os
import pickle
import random

import numpy as np
import torch
from torch.utils.data import Dataset

from .data_util import align, pth_tensor, rand_perm, transfer_noise
from .vocab import BOS_ID, EOS_ID, PAD_ID, BPETokenizer  # noqa: F401


class StyleDataset(Dataset):
    def __init__(self, files, vocab, max_len, load_func, cache=False):
        super().__init__()
        self.files, self.vocab, self.max_len, self.load_func = files, vocab, max_len, load_func
        self.cache = cache
        self.samples = self._load()

    def truncate(self, sentence):
        return self.vocab.encode(sentence)[: self.max_len]

    def _load(self):
        samples = []
        for file in self.files:
            cpath = f"{file}.tok{self.max_len}.pkl"
            if self.cache and os.path.exists(cpath) and os.path.getmtime(cpath) >= os.path.getmtime(file):
                with open(cpath, "rb") as f:
                    samples += pickle.load(f)
                continue
            part = self.load_func(file, self.truncate, self.vocab, self.max_len) if self.load_func is load_s2l \
                else self.load_func(file, self.truncate)
            if self.cache:
                with open(cpath, "wb") as f:
                    pickle.dump(part, f)
            samples += part
        return samples

    def __getitem__(self, idx):
        return self.samples[idx]

    def __len__(self):
        return len(self.samples)


def load_s2l(file_name, parse_func, vocab=None, max_len=None):
    """loader.py:34-40: label from the file suffix, empty lines dropped.  With `vocab` the whole
    file goes through one encode_batch call (same ids as parse_func line by line)."""
    assert os.path.exists(file_name)
    label = int(file_name.split(".")[-1])
    with open(file_name, "r", encoding="utf-8") as f:
        sentences = [line.strip() for line in f]
    sentences = [s for s in sentences if s]
    if vocab is not None and hasattr(vocab, "encode_batch"):
        return [(ids[:max_len], label) for ids in vocab.encode_batch(sentences)]
    return [(parse_func(s), label) for s in sentences]


def overlap_distance_label(noised_1, noised_2, vocab=None):
    """Stand-in for WMDdistance.cal_wmd_label (src/wmd.py:31-45): 1.5 * (1 - Jaccard overlap of the
    two token bags), in the same [0, 1.5]-ish range the real labels occupy.  Not the reference's
    number -- supply a gensim-backed `label_fn` for that."""
    out = []
    for a, b in zip(noised_1, noised_2):
        sa, sb = set(a), set(b)
        out.append(1.5 * (1.0 - len(sa & sb) / max(1, len(sa | sb))))
    return out


def collate_pretrain(vocab, w2v=None, label_fn=None):
    if label_fn is None:
        label_fn = w2v.cal_wmd_label if w2v is not None else overlap_distance_label

    def collate_func(batch_samples):
        sentences, labels = zip(*batch_samples)
        noised_1 = transfer_noise(sentences, p=0.15)
        noised_2 = transfer_noise(sentences, p=0.15)
        noised_3 = rand_perm(sentences, p=0.15)
        x, _, _ = align(sentences, PAD_ID)
        nx_1, _, _ = align(noised_1, PAD_ID)
        nx_2, _, _ = align(noised_2, PAD_ID)
        nx_3, _, _ = align(noised_3, PAD_ID)
        c_label = label_fn(noised_1, noised_2, vocab)
        return (pth_tensor(x, torch.long), pth_tensor(nx_1, torch.long), pth_tensor(nx_2, torch.long),
                pth_tensor(nx_3, torch.long), pth_tensor(labels, torch.long), pth_tensor(c_label, torch.float))
    return collate_func


def collate_warmup(batch_samples):
    sentences, labels = zip(*batch_samples)
    noised = transfer_noise(sentences, p=0.1)
    x, _, _ = align(sentences, PAD_ID)
    nx, _, _ = align(noised, PAD_ID)
    return (pth_tensor(nx, torch.long), pth_tensor(x, torch.long), pth_tensor(labels, torch.long))


def collate_optimize(batch_samples):
    sentences, labels = zip(*batch_samples)
    x, _, _ = align(sentences, PAD_ID)
    return (pth_tensor(x, torch.long), pth_tensor(labels, torch.long))


class GlobalBatchSampler:
    """Index batches of the GLOBAL batch size, identical on every rank (seeded by epoch), dropping
    nothing: the last batch may be short, and is trimmed to a multiple of `world` rows."""

    def __init__(self, n, global_batch, shuffle, seed=0, world=1):
        self.n, self.bs, self.shuffle, self.seed, self.world = n, global_batch, shuffle, seed, world
        self.epoch = 0

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __iter__(self):
        idx = list(range(self.n))
        if self.shuffle:
            random.Random(self.seed + 1000003 * self.epoch).shuffle(idx)
        for s in range(0, self.n, self.bs):
            b = idx[s:s + self.bs]
            b = b[: len(b) // self.world * self.world]
            if b:
                yield b

    def __len__(self):
        return (self.n + self.bs - 1) // self.bs


def iterate_batches(dataset, sampler, collate, seed=0):
    """Yields (batch_idx, collated global batch).  Seeds numpy/random per batch so that every rank
    draws the same noise for the same global batch."""
    for bi, idx in enumerate(sampler):
        s = (seed + 7919 * sampler.epoch + bi) % (2 ** 31 - 1)
        np.random.seed(s)
        random.seed(s)
        yield bi, collate([dataset[i] for i in idx])
