"""Offline quality metrics of finished transfers -- SURVEY 8(f) row 4 (reference: evaluate/prepare.py, evaluate/eval.py,
evaluate/auto/{transfer_intensity,content_preserve,naturalness,style_lexicon,nt_classifier}.py).

    python -m consistent__style_transfer_amd.evaluate prepare <dataset> [<model>]
    python -m consistent__style_transfer_amd.evaluate eval    <dataset> <model>

The three numbers the reference reports for `output/<dataset>-<model>/style.test.{0,1}.tsf` against `data/<dataset>/style.test.{0,1}`:

    STI  style-transfer intensity (higher is better): per sentence, the earth mover's distance between the style classifier's class
         distribution of the input and of the output under a ground distance of 1 between any two classes, signed +/- by whether the
         TARGET class became more probable (transfer_intensity.py:9-23).  With that ground distance the EMD is the total-variation
         distance 1/2 sum |p - q| (pyemd cancels the mass two histograms share bin by bin before it moves anything).
    CP   content preservation (lower is better): Word Mover's Distance between the output and the input after every style-lexicon
         word was replaced by the token MASK (content_preserve.py:13-28,43-50; eval.py:38-43 passes (transfer, origin)).
    NT   naturalness (higher is better): an adversarial unigram classifier scores P(natural | text); a pair is a success unless the
         INPUT scored strictly higher than the output (naturalness.py:63-98).

This module is CPU-side tooling next to the hot path, not part of it; it never touches the GPU.  What the reference takes from
third-party packages is restated over what this image has:

    fastText supervised classifier (prepare.py:33-58)   -> multinomial logistic regression over binary unigram counts (fastText's default
                                                            supervised model is a linear softmax over a bag of words) -- scikit-learn
    gensim Word2Vec of the masked corpus (prepare.py:75) -> wmd.WordVectors (npz container; from_cooccurrence() where no trained vectors exist)
    gensim wmdistance / pyemd                            -> wmd.wmdistance (exact transportation LP)
    sklearn CountVectorizer(binary) + l1 LR, C = 3       -> the same estimators (style_lexicon.py:14-23,68-96; nt_classifier.py:8-24)
    DeepMoji regex tokenizer (auto/tokenizer.py)         -> tokenize() below: words (with inner ' - _), numbers, runs of one repeated
                                                            symbol, any other character -- the reference's token classes without its
                                                            URL / e-mail / emoticon / emoji patterns (the corpora are pre-tokenised
                                                            lower-case review text)

Parity with the reference's numbers (results.md) is UNPINNED: its dumps (fastText model, lexicon, vectors, classifiers) are not in the
repository and cannot be rebuilt offline.  What is pinned (tests/test_host_cpu.py) is the arithmetic: hand-computed STI / judgments,
masking, the lexicon rule (weights beyond two standard deviations of the non-zero l1 weights), and an end-to-end prepare + eval run on
the dev-sample fixture.
"""
import json
import os
import pickle
import re
import sys

import numpy as np

from .wmd import WordVectors, wmdistance

MASK = "MASK"
_TOKEN = re.compile(r"[A-Za-z]+(?:['\-_][A-Za-z]+)+|[0-9]+|[A-Za-z]+|(\S)\1*", re.UNICODE)


def tokenize(text):
    """Words (inner apostrophes / hyphens / underscores kept: "don't", "red-haired", "CUSTOM_TOKEN"), digit runs, runs of one repeated
    non-space character ("!!!", "..."), single other characters; whitespace separates and is dropped."""
    return [m.group(0) for m in _TOKEN.finditer(text)]


def load_dataset(path):
    with open(path, "r", encoding="utf-8") as f:
        return [line.strip() for line in f]


# ---------------------------------------------------------------------------------------------------------------- STI
def unit_emd(p, q):
    """EMD between two histograms of equal mass when every pair of distinct bins is at distance 1."""
    p, q = np.asarray(p, dtype=np.float64), np.asarray(q, dtype=np.float64)
    return 0.5 * float(np.abs(p - q).sum())


def direction_corrected_emd(p_in, p_out, target):
    return unit_emd(p_in, p_out) * (1.0 if p_out[target] >= p_in[target] else -1.0)


def calculate_STIs(inputs, outputs, target_styles, class_probs):
    """class_probs(texts) -> [n, n_class] probabilities, columns in label order (transfer_intensity.py:25-35)."""
    pi, po = np.asarray(class_probs(list(inputs))), np.asarray(class_probs(list(outputs)))
    return [direction_corrected_emd(a, b, t) for a, b, t in zip(pi, po, target_styles)]


# ---------------------------------------------------------------------------------------------------------------- CP
def mask_style_words(texts, lexicon):
    out = []
    for text in texts:
        out.append(" ".join(MASK if tok.lower() in lexicon else tok for tok in tokenize(text)))
    return out


def calculate_wmd_scores(references, candidates, word_vectors):
    return [wmdistance(word_vectors, tokenize(r), tokenize(c)) for r, c in zip(references, candidates)]


def finite_mean(values):
    """Mean over the finite entries (a pair with no in-vocabulary token on one side has WMD = inf: the reference's mean would be inf
    as well; reporting the finite mean next to the count keeps the number usable)."""
    v = [x for x in values if np.isfinite(x)]
    return (sum(v) / len(v) if v else float("inf")), len(values) - len(v)


# ---------------------------------------------------------------------------------------------------------------- NT
def generate_judgments(input_scores, output_scores):
    """1: the input scored as more natural, 0: the output did, None: a tie."""
    return [None if a == b else int(a > b) for a, b in zip(input_scores, output_scores)]


def aggregate_judgments(judgments):
    return sum(1 for j in judgments if j is None or j == 0) / len(judgments)


# ---------------------------------------------------------------------------------------------------------------- estimators
def _sklearn():
    try:
        from sklearn.feature_extraction.text import CountVectorizer
        from sklearn.linear_model import LogisticRegression
    except ImportError as e:                                  # pragma: no cover
        raise RuntimeError("evaluate: scikit-learn is needed for the lexicon / classifiers") from e
    return CountVectorizer, LogisticRegression


def fit_vectorizer(texts):
    CountVectorizer, _ = _sklearn()
    v = CountVectorizer(binary=True, tokenizer=tokenize, lowercase=True, token_pattern=None)
    v.fit(texts)
    return v


def train_l1_lr(X, y, C=3.0):
    _, LogisticRegression = _sklearn()
    lr = LogisticRegression(penalty="l1", C=C, solver="liblinear")
    lr.fit(X, y)
    return lr


def lexicon_from_weights(weights, vocabulary, n_std=2.0):
    """style_lexicon.py:25-66: of the features with a non-zero l1 weight, keep those whose weight lies more than `n_std` standard
    deviations from the mean of the non-zero weights; returns [(feature, weight)] sorted by weight."""
    inv = {i: t for t, i in vocabulary.items()}
    w = np.asarray(weights, dtype=np.float64).reshape(-1)
    nz = np.flatnonzero(np.abs(w) > 0.0)
    if nz.size == 0:
        return []
    vals = w[nz]
    lo, hi = vals.mean() - n_std * vals.std(), vals.mean() + n_std * vals.std()
    keep = nz[(vals < lo) | (vals > hi)]
    return sorted(((inv[int(i)], float(w[i])) for i in keep), key=lambda e: e[1])


def generate_lexicon(negative_texts, positive_texts):
    x = list(negative_texts) + list(positive_texts)
    y = np.concatenate([np.zeros(len(negative_texts)), np.ones(len(positive_texts))])
    vec = fit_vectorizer(x)
    lr = train_l1_lr(vec.transform(x), y)
    return lexicon_from_weights(lr.coef_[0], vec.vocabulary_), vec


class StyleClassifier:
    """Stand-in for the fastText supervised model of prepare.py:33-58: softmax regression over binary unigram counts."""

    def __init__(self, vectorizer, model):
        self.vectorizer, self.model = vectorizer, model

    @classmethod
    def train(cls, texts_by_label):
        _, LogisticRegression = _sklearn()
        labels = sorted(texts_by_label)
        x = [t for l in labels for t in texts_by_label[l]]
        y = np.concatenate([np.full(len(texts_by_label[l]), i) for i, l in enumerate(labels)])
        vec = fit_vectorizer(x)
        return cls(vec, LogisticRegression(C=10.0, max_iter=200).fit(vec.transform(x), y))

    def __call__(self, texts):
        return self.model.predict_proba(self.vectorizer.transform(texts))


class UnigramNaturalness:
    """naturalness.py:44-54 / nt_classifier.py: P(natural) from an l1 logistic regression that separates model outputs (0) from corpus
    sentences (1) on binary unigram counts."""

    def __init__(self, vectorizer, model):
        self.vectorizer, self.model = vectorizer, model

    @classmethod
    def train(cls, transferred, originals, vectorizer):
        x = list(transferred) + list(originals)
        y = np.concatenate([np.zeros(len(transferred)), np.ones(len(originals))])
        return cls(vectorizer, train_l1_lr(vectorizer.transform(x), y))

    def score(self, texts):
        return self.model.predict_proba(self.vectorizer.transform(texts))[:, 1]


# ---------------------------------------------------------------------------------------------------------------- pipeline
def _paths(dataset, model, base_dir, eval_dir):
    dump = os.path.join(eval_dir, "eval_dump")
    return {"data": os.path.join(base_dir, "data", dataset), "out": os.path.join(base_dir, "output", f"{dataset}-{model}"), "dump": dump,
            "clf": os.path.join(dump, f"model_{dataset}.pkl"), "lexicon": os.path.join(dump, f"lexicon_{dataset}.json"),
            "vectorizer": os.path.join(dump, f"vectorizer_{dataset}.pkl"), "w2v": os.path.join(dump, f"mask_w2v_{dataset}.npz"),
            "adv": os.path.join(dump, "adv_models", f"unigram_lr_{model}_{dataset}.pkl")}


def _split_files(directory, key, suffix=""):
    return sorted(os.path.join(directory, n) for n in os.listdir(directory) if key in n and n.endswith(suffix) and ".tok" not in n)


def prepare(dataset, model=None, base_dir="..", eval_dir=".", w2v_dim=64, log=print):
    """prepare.py: the style classifier, the style lexicon + vectorizer, the vectors of the masked corpus, and (given a model name whose
    transferred TRAIN files exist) the adversarial naturalness classifier.  Existing dumps are kept."""
    P = _paths(dataset, model, base_dir, eval_dir)
    os.makedirs(os.path.join(P["dump"], "adv_models"), exist_ok=True)
    train = {n.rsplit(".", 1)[-1]: load_dataset(n) for n in _split_files(P["data"], "style.train.")}
    if not os.path.exists(P["clf"]):
        log("training the style classifier")
        with open(P["clf"], "wb") as f:
            pickle.dump(StyleClassifier.train(train), f)
    if not (os.path.exists(P["lexicon"]) and os.path.exists(P["vectorizer"]) and os.path.exists(P["w2v"])):
        log("generating the style lexicon and the masked-corpus vectors")
        ranked, vec = generate_lexicon(train["0"], train["1"])
        with open(P["lexicon"], "w", encoding="utf-8") as f:
            json.dump({"binary sentiment": ranked}, f)
        with open(P["vectorizer"], "wb") as f:
            pickle.dump(vec, f)
        texts = [t for n in _split_files(P["data"], "style.") if ("train" in n or "dev" in n) for t in load_dataset(n)]
        masked = mask_style_words(texts, {w for w, _ in ranked})
        WordVectors.from_cooccurrence([tokenize(t) for t in masked], dim=w2v_dim).save(P["w2v"])
    if model and not os.path.exists(P["adv"]):
        tsf = [t for n in _split_files(P["out"], "train", ".tsf") for t in load_dataset(n)]
        if tsf:
            log("training the adversarial naturalness classifier")
            with open(P["vectorizer"], "rb") as f:
                vec = pickle.load(f)
            ori = [t for l in sorted(train) for t in train[l]]
            with open(P["adv"], "wb") as f:
                pickle.dump(UnigramNaturalness.train(tsf, ori, vec), f)
    return P


def load_lexicon(path):
    with open(path, "r", encoding="utf-8") as f:
        return {w for w, _ in json.load(f)["binary sentiment"]}


def evaluate(dataset, model, base_dir="..", eval_dir=".", log=print):
    """eval.py: returns {"STI", "CP", "NT"} (+ "CP_undefined": pairs whose WMD is infinite) and prints the reference's three lines."""
    P = _paths(dataset, model, base_dir, eval_dir)
    ori0, ori1 = load_dataset(os.path.join(P["data"], "style.test.0")), load_dataset(os.path.join(P["data"], "style.test.1"))
    tsf0, tsf1 = load_dataset(os.path.join(P["out"], "style.test.0.tsf")), load_dataset(os.path.join(P["out"], "style.test.1.tsf"))
    origin, transfer = ori0 + ori1, tsf0 + tsf1
    assert len(origin) == len(transfer), "evaluate: the transferred files do not line up with the test files"
    labels = [1] * len(tsf0) + [0] * len(tsf1)                 # target style of each transfer (eval.py:31)
    with open(P["clf"], "rb") as f:
        clf = pickle.load(f)
    stis = calculate_STIs(origin, transfer, labels, clf)
    lexicon = load_lexicon(P["lexicon"])
    wv = WordVectors.load(P["w2v"])
    cp, undefined = finite_mean(calculate_wmd_scores(mask_style_words(transfer, lexicon), mask_style_words(origin, lexicon), wv))
    with open(P["adv"], "rb") as f:
        adv = pickle.load(f)
    nt = aggregate_judgments(generate_judgments(adv.score(origin), adv.score(transfer)))
    res = {"STI": sum(stis) / len(stis), "CP": cp, "CP_undefined": undefined, "NT": nt}
    log("STI (higher is better): %.4f" % res["STI"])
    log("CP (lower is better): %.4f" % res["CP"])
    log("NT (higher is better): %.4f" % res["NT"])
    return res


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    if len(argv) < 2 or argv[0] not in ("prepare", "eval"):
        print(__doc__.split("\n\n")[1])
        return 2
    if argv[0] == "prepare":
        prepare(argv[1], argv[2] if len(argv) > 2 else None)
    else:
        evaluate(argv[1], argv[2])
    return 0


if __name__ == "__main__":
    sys.exit(main())
