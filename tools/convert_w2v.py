#!/usr/bin/env python3
"""Convert the reference's gensim Word2Vec pickle (`<ds>-w2v.bin`, src/wmd.py:47-48) into this build's container
`<ds>-w2v.npz` (wmd.WordVectors: tokens + float32 vectors).  Needs gensim, so it runs wherever the reference's own
`python wmd.py <dataset>` ran -- not in the build container.

    python tools/convert_w2v.py ../dump/yelp/yelp-w2v.bin ../dump/yelp/yelp-w2v.npz"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    src, dst = sys.argv[1], sys.argv[2]
    try:
        from gensim.models.word2vec import Word2Vec
    except ImportError:
        sys.exit("convert_w2v.py needs gensim (it reads the reference's pickle); run it where the reference's wmd.py ran")
    from consistent__style_transfer_amd.wmd import WordVectors
    wv = Word2Vec.load(src).wv
    tokens = list(getattr(wv, "index_to_key", None) or wv.index2word)
    vecs = np.stack([wv[t] for t in tokens]).astype(np.float32)
    WordVectors(tokens, vecs, normalise=False).save(dst)
    print(f"{dst}: {len(tokens)} tokens x {vecs.shape[1]}")


if __name__ == "__main__":
    main()
