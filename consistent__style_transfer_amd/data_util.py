"""Host-side batch utilities (reference: src/data_util.py).

`align` right-pads with PAD (no BOS/EOS is ever added); `transfer_noise` pulls each token out with
probability p and re-inserts it into a sentence drawn with probability proportional to the original
sentence lengths, at a uniform position (tokens cross sentences, lengths change); `rand_perm` shuffles
a p-fraction of the tokens of the flattened batch among themselves (lengths preserved).  The random
draws are made in the reference's order from the same generators (numpy global RNG for the masks
and the sentence choice, Python `random` for positions / the shuffle), so a seeded run reproduces
the reference's noise bit for bit (tests/golden/host.json).
"""
import os
import random

import numpy as np
import torch


def path_cat(*parts):
    return os.path.join(*parts)


def pth_tensor(tensor, dtype):
    return torch.tensor(tensor, dtype=dtype)


def add_borders(tokens_list, start=None, end=None):
    head = [] if start is None else [start]
    tail = [] if end is None else [end]
    if start is None and end is None:
        return None                                   # the reference falls off the end of its if/elif chain
    return [head + list(tokens) + tail for tokens in tokens_list]


def align(sentences, pad_value, max_len=None):
    """data_util.py:25-30 -> (padded lists, lengths, max_len)."""
    if max_len is None:
        max_len = max(len(s) for s in sentences)
    lengths = [min(len(s), max_len) for s in sentences]
    padded = [list(s[:max_len]) + [pad_value] * (max_len - len(s)) for s in sentences]
    return padded, lengths, max_len


def transfer_noise(sentences, p):
    """data_util.py:32-54."""
    kept, bag = [], []
    for s in sentences:
        take = np.random.uniform(size=(len(s))) < p          # one draw per sentence, in order
        kept.append([tok for tok, t in zip(s, take) if not t])
        bag.extend(tok for tok, t in zip(s, take) if t)
    lens = np.array([len(s) for s in sentences], dtype=float)
    dest = np.random.choice(list(range(len(sentences))), size=(len(bag),), p=lens / lens.sum())
    for tok, d in zip(bag, dest):
        kept[d].insert(random.randint(0, len(kept[d])), tok)
    return kept


def rand_perm(sentences, p=0.15):
    """data_util.py:56-74."""
    flat = [tok for s in sentences for tok in s]
    hit = np.flatnonzero(np.random.uniform(size=(len(flat))) < p)
    words = [flat[i] for i in hit]
    random.shuffle(words)
    for i, w in zip(hit, words):
        flat[i] = w
    out, pos = [], 0
    for s in sentences:
        out.append(flat[pos:pos + len(s)])
        pos += len(s)
    return out
