"""Content-distance labels of the pretrain stage (reference: src/wmd.py:31-45, called from src/loader.py:60).

The reference asks gensim for the Word Mover's Distance between the two noised copies of every sentence of a batch --
256 pyemd solves per batch on the main thread, which is what bounds its pretrain throughput (SURVEY.md section 8a row 14,
8f row 1).  gensim / pyemd are third-party (gensim 3.8-era `KeyedVectors.wmdistance`, unpinned by the reference) and are not
installable here, so the published algorithm is restated on numpy + scipy:

    wmdistance(d1, d2):  drop out-of-vocabulary tokens; either side empty -> inf; one distinct token in total -> 0;
                         D[i, j] = ||v_i - v_j||_2 over the distinct tokens (vectors L2-normalised: wmd.py:54 init_sims(replace=True));
                         all distances 0 -> inf; nBOW weights = count / document length;
                         result = min sum_ij F_ij D_ij  s.t.  F >= 0, F 1 = nbow(d1), F^T 1 = nbow(d2)      (earth mover's distance)

The transportation problem is solved exactly by this build's own C++ solver (csrc/host_wmd.cpp, libcst_host.so: successive shortest
augmenting paths; pyemd's Pele-Werman solver is exact too, so the two agree to rounding), a whole batch per call
(`WMDdistance.cal_wmd_label` -> cst_host_wmd_labels); scipy's HiGHS LP (`emd_lp`) is kept as the independent check of the tests.  Parity with gensim itself is UNPINNED offline; what IS pinned (tests/test_host_cpu.py) is the
algorithm: hand-solvable cases, symmetry, the triangle inequality, agreement with an independent min-cost-flow formulation,
and the reference's three special cases in `cal_wmd_label` (empty sentence, inf, normal).

Word vectors: the reference trains gensim Word2Vec on the BPE token strings and pickles it as `<ds>-w2v.bin` (wmd.py:19,
73-77), which only gensim can read.  This module reads / writes a plain container instead:

    <ds>-w2v.npz :  tokens  (unicode array, the BPE token strings)      vectors (float32 [n, dim], any norm)

`tools/convert_w2v.py` writes it from a gensim model where gensim exists; `WordVectors.from_cooccurrence` builds a small
PPMI + SVD embedding without any third-party trainer (tests, smoke runs).

Throughput: labels need the noised sentences, which are drawn per batch, so they cannot be tabulated per sentence.  Two
routes take the solves off the training thread: `loader.PrefetchBatches` builds whole batches (noise + labels) in worker
processes ahead of the GPU, bit-identical to the in-line path because every batch is seeded by (seed, epoch, batch index);
and `loader.LabelCache` stores the labels of such a seeded run in a file so that later runs (and every data-parallel rank)
read them back instead of solving.
"""
import ctypes
import math
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.path.join(_HERE, "csrc", "libcst_host.so")
_host = None


def host_lib():
    """libcst_host.so (csrc/host_wmd.cpp, ABI include/cst_host.h): the exact transportation solver and the batch label entry point.
    No fallback: the per-pair scipy LP of round 2 (281 labels / s / core) cannot keep up with the GPU and is kept only as `emd_lp`, the
    independent check the tests compare this library against."""
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise ImportError(f"{HOST_LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (g++ -O3 -shared)")
        L = ctypes.CDLL(HOST_LIB_PATH)
        vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
        L.cst_host_abi_version.restype = ctypes.c_int
        L.cst_host_emd.restype = ctypes.c_int
        L.cst_host_emd.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp, vp, vp]
        L.cst_host_wmd_labels.restype = ctypes.c_int
        L.cst_host_wmd_labels.argtypes = [vp, vp, vp, vp, i64, i64, i64, vp, i32, vp, i32, i32, vp]
        _host = L
    return _host


def _ragged(seqs):
    off = np.zeros(len(seqs) + 1, dtype=np.int64)
    np.cumsum([len(s) for s in seqs], out=off[1:])
    ids = np.fromiter((t for s in seqs for t in s), dtype=np.int32, count=int(off[-1]))
    return ids, off


class WordVectors:
    def __init__(self, tokens, vectors, normalise=True):
        self.index = {t: i for i, t in enumerate(tokens)}
        v = np.asarray(vectors, dtype=np.float64)
        if normalise:                                     # Word2Vec.load(...).wv.init_sims(replace=True)   (wmd.py:54)
            n = np.linalg.norm(v, axis=1, keepdims=True)
            v = v / np.where(n > 0, n, 1.0)
        self.vectors = v
        self.tokens = list(tokens)

    def __contains__(self, tok):
        return tok in self.index

    def __getitem__(self, tok):
        return self.vectors[self.index[tok]]

    def save(self, path):
        np.savez_compressed(path, tokens=np.array(self.tokens), vectors=self.vectors.astype(np.float32))

    @classmethod
    def load(cls, path):
        z = np.load(path, allow_pickle=False)
        return cls([str(t) for t in z["tokens"]], z["vectors"])

    @classmethod
    def from_cooccurrence(cls, token_sentences, dim=32, window=5, min_count=5):
        """Dependency-free embedding: positive PMI of windowed co-occurrence counts, truncated SVD.  Tokens rarer than
        `min_count` stay out of the vocabulary, as Word2Vec's default min_count=5 leaves them (wmd.py:19)."""
        freq = {}
        for s in token_sentences:
            for t in s:
                freq[t] = freq.get(t, 0) + 1
        toks = sorted(t for t, c in freq.items() if c >= min_count)
        idx = {t: i for i, t in enumerate(toks)}
        n = len(toks)
        C = np.zeros((n, n), dtype=np.float64)
        for s in token_sentences:
            ids = [idx[t] for t in s if t in idx]
            for a, i in enumerate(ids):
                for j in ids[max(0, a - window):a]:
                    C[i, j] += 1.0
                    C[j, i] += 1.0
        tot = C.sum()
        if tot == 0 or n == 0:
            return cls(toks, np.zeros((n, max(1, dim))))
        row = C.sum(1, keepdims=True)
        with np.errstate(divide="ignore", invalid="ignore"):
            pmi = np.log(C * tot / (row * row.T))
        pmi[~np.isfinite(pmi)] = 0.0
        pmi = np.maximum(pmi, 0.0)
        U, S, _ = np.linalg.svd(pmi, full_matrices=False)
        k = min(dim, n)
        return cls(toks, U[:, :k] * np.sqrt(S[:k]))


def emd(w1, w2, D):
    """Exact earth mover's distance between two histograms of equal mass over the same support (libcst_host.so: successive shortest
    augmenting paths).  Rows / columns of zero weight are dropped first (the problem is over the occupied bins only)."""
    i1, i2 = np.flatnonzero(w1 > 0), np.flatnonzero(w2 > 0)
    a = np.ascontiguousarray(w1[i1], dtype=np.float64)
    b = np.ascontiguousarray(w2[i2], dtype=np.float64)
    cost = np.ascontiguousarray(D[np.ix_(i1, i2)], dtype=np.float64)
    out = ctypes.c_double()
    rc = host_lib().cst_host_emd(len(a), len(b), a.ctypes.data, b.ctypes.data, cost.ctypes.data, ctypes.addressof(out))
    if rc != 0:
        raise RuntimeError("cst_host_emd: bad arguments")
    return float(out.value)


def emd_lp(w1, w2, D):
    """The same transportation problem as a linear program (scipy HiGHS): the independent formulation tests/test_host_cpu.py holds
    `emd` to.  Not on the training path."""
    from scipy.optimize import linprog
    i1, i2 = np.flatnonzero(w1 > 0), np.flatnonzero(w2 > 0)
    a, b = w1[i1], w2[i2]
    cost = D[np.ix_(i1, i2)]
    n, m = len(a), len(b)
    if n == 1:
        return float((cost[0] * b).sum())
    if m == 1:
        return float((cost[:, 0] * a).sum())
    A = np.zeros((n + m, n * m))
    for i in range(n):
        A[i, i * m:(i + 1) * m] = 1.0
    for j in range(m):
        A[n + j, j::m] = 1.0
    res = linprog(cost.reshape(-1), A_eq=A[:-1], b_eq=np.concatenate([a, b])[:-1], bounds=(0, None), method="highs")
    if res.status != 0:
        raise RuntimeError(f"emd_lp: LP solver failed ({res.message})")
    return float(res.fun)


def wmdistance(wv, document1, document2):
    """gensim KeyedVectors.wmdistance restated (see the module docstring)."""
    d1 = [t for t in document1 if t in wv]
    d2 = [t for t in document2 if t in wv]
    if not d1 or not d2:
        return float("inf")
    vocab = sorted(set(d1) | set(d2))
    if len(vocab) == 1:
        return 0.0
    pos = {t: i for i, t in enumerate(vocab)}
    V = np.stack([wv[t] for t in vocab])
    D = np.sqrt(np.maximum(((V[:, None, :] - V[None, :, :]) ** 2).sum(-1), 0.0))
    s1, s2 = set(d1), set(d2)
    keep = np.array([[(a in s1 and b in s2) or (a in s2 and b in s1) for b in vocab] for a in vocab])
    D = np.where(keep, D, 0.0)                            # gensim fills only the (doc1 token, doc2 token) pairs
    if D.sum() == 0.0:
        return float("inf")
    w1, w2 = np.zeros(len(vocab)), np.zeros(len(vocab))
    for t in d1:
        w1[pos[t]] += 1.0 / len(d1)
    for t in d2:
        w2[pos[t]] += 1.0 / len(d2)
    return emd(w1, w2, D)                                 # every (doc1 bin, doc2 bin) pair the flow can use is filled; D[i, i] = 0


class WMDdistance:
    """The reference's class surface (wmd.py:11-56) over WordVectors: `cal_wmd`, `cal_wmd_label`, `save`, `load`."""

    def __init__(self, word_vectors=None):
        self.wv = word_vectors

    def cal_wmd(self, x1, x2):
        return wmdistance(self.wv, x1, x2)

    def _rows_of_ids(self, tokenizer):
        """vocabulary id -> row of the word-vector table, -1 for tokens without a vector (cached per tokenizer)."""
        key = id(tokenizer)
        hit = getattr(self, "_row_cache", None)
        if hit is None or hit[0] != key:
            toks = tokenizer.ids_to_tokens(list(range(len(tokenizer))))
            rows = np.array([self.wv.index.get(t, -1) for t in toks], dtype=np.int32)
            self._row_cache = hit = (key, rows, np.ascontiguousarray(self.wv.vectors, dtype=np.float64))
        return hit[1], hit[2]

    def cal_wmd_label(self, xs1, xs2, tokenizer, rows=None, nthreads=1):
        """src/wmd.py:34-45 for a whole batch in ONE call of libcst_host.so (cst_host_wmd_labels).  rows = (lo, hi): only those pairs
        are computed (a data-parallel rank needs the labels of its own rows only), the others are returned as 0.0."""
        n = len(xs1)
        lo, hi = (0, n) if rows is None else rows
        row_of_id, vec = self._rows_of_ids(tokenizer)
        ids1, off1 = _ragged(xs1)
        ids2, off2 = _ragged(xs2)
        out = np.zeros(n, dtype=np.float64)
        rc = host_lib().cst_host_wmd_labels(ids1.ctypes.data, off1.ctypes.data, ids2.ctypes.data, off2.ctypes.data, n, int(lo), int(hi),
                                            row_of_id.ctypes.data, len(row_of_id), vec.ctypes.data, vec.shape[1], int(nthreads),
                                            out.ctypes.data)
        if rc != 0:
            raise RuntimeError("cst_host_wmd_labels: bad arguments")
        return out.tolist()

    def cal_wmd_label_py(self, xs1, xs2, tokenizer):
        """The per-pair Python restatement of round 2 (numpy distances + one transportation solve per pair): what the tests compare
        the batch entry point with.  Not on the training path."""
        label = []
        for x1, x2 in zip(xs1, xs2):
            if len(x1) == 0 or len(x2) == 0:              # wmd.py:37-38
                label.append(max([float(len(x1)), float(len(x2))]))
            else:
                distance = self.cal_wmd(tokenizer.ids_to_tokens(x1), tokenizer.ids_to_tokens(x2))
                if distance == float("inf"):              # wmd.py:41-42
                    label.append((len(x1) + len(x2)) / 2)
                else:
                    label.append(distance)
        return label

    def save(self, path):
        self.wv.save(path)

    @classmethod
    def load(cls, path):
        return cls(WordVectors.load(path))

    @classmethod
    def train(cls, file_lists, tokenizer, dim=32):
        """Stand-in for `Word2Vec(sentences, iter=10)` (wmd.py:13-19) without gensim: PPMI + SVD on the same token strings."""
        sents = []
        for f in file_lists:
            with open(f, "r", encoding="utf-8") as fh:
                sents += [tokenizer.ids_to_tokens(tokenizer.encode(line.strip())) for line in fh if line.strip()]
        return cls(WordVectors.from_cooccurrence(sents, dim=dim))


def is_finite_label(x):
    return not (math.isinf(x) or math.isnan(x))
