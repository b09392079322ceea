"""Dataset + collate functions (reference: src/loader.py), re-batched for data parallelism.

`StyleDataset` tokenises and truncates every line at load (loader.py:19-26) -- here through one
batched call into the tokenizer -- and can keep the result as a binary token cache next to the data
(`--token_cache`; format: TokenCache below), so later runs and every data-parallel rank skip tokenisation.
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
import json
import os
import random
import struct

import numpy as np
import torch
from torch.utils.data import Dataset

from .data_util import align, pth_tensor, rand_perm, transfer_noise
from .vocab import BOS_ID, EOS_ID, PAD_ID, BPETokenizer  # noqa: F401


class TokenCache:
    """Binary token cache of one data file, `<file>.tok<max_len>.cstt`:

        bytes 0-7    magic  b"CSTTOK1\0"
        int64        n            number of sentences (empty lines already dropped, loader.py:37-39)
        int64        max_len      truncation applied (loader.py:25-26)
        int64        vocab_size   len(vocab) the ids were produced with (a different tokenizer invalidates the cache)
        int32[n]     label        from the file suffix (loader.py:36)
        int64[n+1]   offset       sentence i = tokens[offset[i] : offset[i+1]]
        int32[...]   tokens       BPE ids, already truncated to max_len

    Little endian, no padding.  Valid while its mtime is not older than the text file's."""
    MAGIC = b"CSTTOK1\0"

    @staticmethod
    def path(file, max_len):
        return f"{file}.tok{max_len}.cstt"

    @classmethod
    def write(cls, path, samples, max_len, vocab_size):
        lab = np.array([l for _, l in samples], dtype="<i4")
        off = np.zeros(len(samples) + 1, dtype="<i8")
        np.cumsum([len(t) for t, _ in samples], out=off[1:])
        tok = np.fromiter((i for t, _ in samples for i in t), dtype="<i4", count=int(off[-1]))
        tmp = f"{path}.tmp{os.getpid()}"
        with open(tmp, "wb") as f:
            f.write(cls.MAGIC)
            f.write(struct.pack("<qqq", len(samples), max_len, vocab_size))
            f.write(lab.tobytes())
            f.write(off.tobytes())
            f.write(tok.tobytes())
        os.replace(tmp, path)                              # atomic: ranks racing to write the same cache never see half a file

    @classmethod
    def read(cls, path, max_len, vocab_size):
        """-> list of (token list, label), or None when the file is missing / for another max_len or vocabulary."""
        try:
            with open(path, "rb") as f:
                blob = f.read()
        except OSError:
            return None
        if blob[:8] != cls.MAGIC or len(blob) < 32:
            return None
        n, ml, vs = struct.unpack_from("<qqq", blob, 8)
        if ml != max_len or vs != vocab_size:
            return None
        p = 32
        lab = np.frombuffer(blob, dtype="<i4", count=n, offset=p)
        p += 4 * n
        off = np.frombuffer(blob, dtype="<i8", count=n + 1, offset=p)
        p += 8 * (n + 1)
        tok = np.frombuffer(blob, dtype="<i4", count=int(off[-1]), offset=p)
        tl = tok.tolist()
        return [(tl[off[i]:off[i + 1]], int(lab[i])) for i in range(n)]


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
            cpath = TokenCache.path(file, self.max_len)
            if self.cache and os.path.exists(cpath) and os.path.getmtime(cpath) >= os.path.getmtime(file):
                part = TokenCache.read(cpath, self.max_len, len(self.vocab))
                if part is not None:
                    samples += part
                    continue
            part = self.load_func(file, self.truncate, self.vocab, self.max_len) if self.load_func is load_s2l \
                else self.load_func(file, self.truncate)
            if self.cache:
                TokenCache.write(cpath, part, self.max_len, len(self.vocab))
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


class LabelCache:
    """Content-distance labels of a SEEDED pretrain run, `--label_cache <file>` (numpy .npz container):

        meta            JSON string: {"format": "CSTLBL1", "seed", "global_batch", "n_sentences", "noise_p", "label_fn"}
        e<epoch>_b<bi>  float32 [rows of that global batch]    for every batch the producing run saw

    Every batch of a run is seeded by (seed, epoch, batch index) (iterate_batches), so the noised sentences -- hence their
    labels -- are reproducible: `tools/make_label_cache.py` replays the sampler + noise of a run and stores what
    `label_fn` (the WMD of wmd.py) returns; training then reads the labels back instead of solving 256 transportation
    problems per batch on the training thread.  A missing key, or a cache made for another seed / batch size / corpus,
    raises: silently training the Matcher on stale targets would be worse than stopping."""

    def __init__(self, path=None, meta=None):
        self.tables, self.meta, self.path = {}, dict(meta or {}), path
        if path is not None and os.path.exists(path):
            z = np.load(path, allow_pickle=False)
            self.meta = json.loads(str(z["meta"]))
            if self.meta.get("format") != "CSTLBL1":
                raise ValueError(f"{path}: not a CSTLBL1 label cache")
            self.tables = {k: z[k] for k in z.files if k != "meta"}

    def check(self, seed, global_batch, n_sentences):
        want = {"seed": seed, "global_batch": global_batch, "n_sentences": n_sentences}
        bad = {k: (self.meta.get(k), v) for k, v in want.items() if self.meta.get(k) != v}
        if bad:
            raise ValueError(f"label cache {self.path} was made for another run (cache value, this run): {bad}")

    @staticmethod
    def key(epoch, bi):
        return f"e{epoch}_b{bi}"

    def put(self, epoch, bi, labels):
        self.tables[self.key(epoch, bi)] = np.asarray(labels, dtype=np.float32)

    def get(self, epoch, bi, rows):
        t = self.tables.get(self.key(epoch, bi))
        if t is None or len(t) != rows:
            raise KeyError(f"label cache {self.path}: no labels for epoch {epoch} batch {bi} with {rows} rows")
        return t.tolist()

    def save(self, path=None):
        meta = dict(self.meta, format="CSTLBL1")
        np.savez_compressed(path or self.path, meta=np.array(json.dumps(meta)), **self.tables)


class PretrainCollate:
    """loader.py:46-70 as a picklable callable (PrefetchBatches ships it to worker processes).  `position` = (epoch, batch
    index) of the batch being built; iterate_batches / the prefetch workers set it, the label cache reads it."""

    def __init__(self, vocab, label_fn, label_cache=None, shard=None):
        """shard = (rank, world): compute the content-distance labels of this rank's rows only (parallel.shard_batch gives rank r the
        rows [r n, (r + 1) n) of the global batch) -- the noise is drawn for the whole global batch on every rank, the transportation
        problems are not solved `world` times over.  Only label functions that take `rows=` are narrowed."""
        self.vocab, self.label_fn, self.label_cache, self.shard = vocab, label_fn, label_cache, shard
        self.position = (0, 0)
        import inspect
        try:
            self._takes_rows = "rows" in inspect.signature(label_fn).parameters
        except (TypeError, ValueError):
            self._takes_rows = False

    def without_cache(self):
        """The collate for anything that is NOT the seeded training stream the cache was made for (validation: Trainer.fit).  The cache
        is keyed by (epoch, batch index) of the TRAINING sampler only; read through it, validation batch i would silently get the
        labels of training batch i of epoch 0 (same row count), and those targets decide best_eval, the freeze flags and mat.pth."""
        return PretrainCollate(self.vocab, self.label_fn, None, self.shard) if self.label_cache is not None else self

    def __call__(self, batch_samples):
        sentences, labels = zip(*batch_samples)
        noised_1 = transfer_noise(sentences, p=0.15)
        noised_2 = transfer_noise(sentences, p=0.15)
        noised_3 = rand_perm(sentences, p=0.15)
        x, _, _ = align(sentences, PAD_ID)
        nx_1, _, _ = align(noised_1, PAD_ID)
        nx_2, _, _ = align(noised_2, PAD_ID)
        nx_3, _, _ = align(noised_3, PAD_ID)
        if self.label_cache is not None:
            c_label = self.label_cache.get(*self.position, len(sentences))
        elif self.shard is not None and self.shard[1] > 1 and self._takes_rows and len(sentences) % self.shard[1] == 0:
            per = len(sentences) // self.shard[1]
            c_label = self.label_fn(noised_1, noised_2, self.vocab, rows=(self.shard[0] * per, (self.shard[0] + 1) * per))
        else:
            c_label = self.label_fn(noised_1, noised_2, self.vocab)
        return (pth_tensor(x, torch.long), pth_tensor(nx_1, torch.long), pth_tensor(nx_2, torch.long),
                pth_tensor(nx_3, torch.long), pth_tensor(labels, torch.long), pth_tensor(c_label, torch.float))


def collate_pretrain(vocab, w2v=None, label_fn=None, label_cache=None, shard=None):
    """`w2v`: an object with the reference's `cal_wmd_label(xs1, xs2, tokenizer)` (wmd.WMDdistance); `label_fn`: any
    function of (noised_1, noised_2, vocab); `label_cache` (a LabelCache): labels are read from it instead of computed."""
    if label_fn is None:
        label_fn = w2v.cal_wmd_label if w2v is not None else overlap_distance_label
    return PretrainCollate(vocab, label_fn, label_cache, shard)


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
    """Index batches of the GLOBAL batch size, identical on every rank (seeded by epoch).  The last batch may be short; under data
    parallelism (world > 1) it is TRIMMED to a multiple of `world` rows so that every rank gets the same number of rows (every loss
    is a batch mean over equal shards): up to world - 1 sentences of the LAST batch of an epoch are not trained on in that epoch
    (at most 7 of ~444 000 at 8 ranks; with shuffle they are different sentences every epoch).  The transfer writer
    (main_optimize --mode test) does not use this trimming: it pads the last shard instead and drops nothing."""

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


def _build_batch(dataset, collate, idx, seed, epoch, bi):
    s = (seed + 7919 * epoch + bi) % (2 ** 31 - 1)
    np.random.seed(s)
    random.seed(s)
    if hasattr(collate, "position"):
        collate.position = (epoch, bi)
    return collate([dataset[i] for i in idx])


def iterate_batches(dataset, sampler, collate, seed=0):
    """Yields (batch_idx, collated global batch).  Seeds numpy/random per batch so that every rank
    draws the same noise for the same global batch."""
    for bi, idx in enumerate(sampler):
        yield bi, _build_batch(dataset, collate, idx, seed, sampler.epoch, bi)


_PF = {}


def _pf_init(dataset, collate):
    _PF["dataset"], _PF["collate"] = dataset, collate
    torch.set_num_threads(1)


def _pf_work(job):
    idx, seed, epoch, bi = job
    return bi, _build_batch(_PF["dataset"], _PF["collate"], idx, seed, epoch, bi)


class PrefetchBatches:
    """iterate_batches with the batch construction (token noise, padding, and for pretrain the content-distance labels
    -- the reference's host hot spot, loader.py:60) moved to `workers` forked processes that run `depth` batches ahead of
    the consumer.  Same batches, same order, bit for bit: a batch depends only on (dataset, indices, seed, epoch, batch
    index).  The workers never touch the GPU (numpy / Python / scipy on CPU tensors only); forking a process that has
    initialised HIP is not safe, so the pool uses the 'spawn' start method and receives the dataset once at start-up."""

    def __init__(self, dataset, sampler, collate, seed=0, workers=4, depth=8):
        self.dataset, self.sampler, self.collate, self.seed = dataset, sampler, collate, seed
        self.workers, self.depth = max(1, workers), max(1, depth)
        self.pool = None

    def _pool(self):
        if self.pool is None:
            import multiprocessing as mp
            self.pool = mp.get_context("spawn").Pool(self.workers, initializer=_pf_init, initargs=(self.dataset, self.collate))
        return self.pool

    def __iter__(self):
        pool = self._pool()
        jobs = ((idx, self.seed, self.sampler.epoch, bi) for bi, idx in enumerate(self.sampler))
        pending = []
        for job in jobs:
            pending.append(pool.apply_async(_pf_work, (job,)))
            if len(pending) >= self.depth:
                yield pending.pop(0).get()
        while pending:
            yield pending.pop(0).get()

    def close(self):
        if self.pool is not None:
            self.pool.terminate()
            self.pool = None
