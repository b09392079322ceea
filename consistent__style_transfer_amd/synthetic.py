"""Synthetic Yelp-shaped batches: ids uniform in [4, V) (0..3 are the
reserved <pad>/<s>/</s>/<unk>, vocab.py:5-11), sentence lengths drawn from the word-length
histogram of data/yelp/style.dev.* capped at max_len, right-padded with PAD=0 exactly like
loader.align; labels Bernoulli(0.5); c_label ~ U(0, 1.5) float32 (stand-in for the WMD labels of
wmd.py:34-45, which need gensim)."""
import numpy as np
import torch

# counts of sentences with 1..15 words in the 4000 Yelp dev sentences
YELP_LEN_HIST = np.array([12, 90, 195, 227, 340, 348, 323, 320, 324, 346, 327, 313, 284, 278, 273], dtype=np.float64)


def _ids(rs, B, L, V, full):
    x = rs.randint(4, V, size=(B, L)).astype(np.int64)
    if not full:
        lens = rs.choice(np.arange(1, len(YELP_LEN_HIST) + 1), size=B, p=YELP_LEN_HIST / YELP_LEN_HIST.sum())
        lens = np.minimum(np.maximum(lens + 2, 1), L)      # BPE lengthens sentences slightly
        lens[0] = L                                        # the batch maximum reaches max_len (global padding)
        for b in range(B):
            x[b, lens[b]:] = 0
    return torch.from_numpy(x)


def pretrain_batch(B, L, V, seed=0, full=False):
    """(x, nx_1, nx_2, nx_3, label, c_label) as collate_pretrain returns them (loader.py:62-69)."""
    rs = np.random.RandomState(seed)
    label = torch.from_numpy(rs.randint(0, 2, size=(B,)).astype(np.int64))
    c_label = torch.from_numpy(rs.uniform(0, 1.5, size=(B,)).astype(np.float32))
    return (_ids(rs, B, L, V, full), _ids(rs, B, L, V, full), _ids(rs, B, L, V, full), _ids(rs, B, L, V, full), label, c_label)


def warmup_batch(B, L, V, seed=0, full=False):
    """(nx, x, label) as collate_warmup returns them (loader.py:78-82)."""
    rs = np.random.RandomState(seed + 1)
    label = torch.from_numpy(rs.randint(0, 2, size=(B,)).astype(np.int64))
    return (_ids(rs, B, L, V, full), _ids(rs, B, L, V, full), label)


def optimize_batch(B, L, V, seed=0, full=False):
    """(x, label) as collate_optimize returns them (loader.py:87-90)."""
    rs = np.random.RandomState(seed + 2)
    label = torch.from_numpy(rs.randint(0, 2, size=(B,)).astype(np.int64))
    return (_ids(rs, B, L, V, full), label)
