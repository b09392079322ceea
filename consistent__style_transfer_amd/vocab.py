"""BPE vocabulary indexing (reference: src/vocab.py).

Same contract: HF `tokenizers.CharBPETokenizer` (lower-casing, `</w>` suffix) with the special tokens
pinned to ids <pad>=0, <s>=1, </s>=2, <unk>=3 (vocab.py:5-11,18-20); `encode` returns ids only;
`decode` drops special tokens by default; `len()` is V, the size of every vocabulary-shaped tensor.
File format is the reference's: `<name>-vocab.json` + `<name>-merges.txt`.
"""
import os

from tokenizers import CharBPETokenizer

PAD, BOS, EOS, UNK = "<pad>", "<s>", "</s>", "<unk>"
PAD_ID, BOS_ID, EOS_ID, UNK_ID = 0, 1, 2, 3


class BPETokenizer:
    def __init__(self, text_list, vocab_size, lazy=False):
        self.tokenizer = None
        if not lazy:
            self.tokenizer = CharBPETokenizer()
            self.tokenizer.train(text_list, vocab_size=vocab_size, special_tokens=[PAD, BOS, EOS, UNK])
            self.tokenizer.add_special_tokens([PAD, BOS, EOS])

    def tokens_to_ids(self, tokens):
        return [self.tokenizer.token_to_id(t) for t in tokens]

    def ids_to_tokens(self, ids):
        return [self.tokenizer.id_to_token(i) for i in ids]

    def encode(self, text):
        return self.tokenizer.encode(text).ids

    def encode_batch(self, texts):
        """Additive: one call into the Rust tokenizer for a list of lines (loader ingest path)."""
        return [e.ids for e in self.tokenizer.encode_batch(list(texts))]

    def decode(self, ids, skip_special=True):
        return self.tokenizer.decode(list(ids), skip_special_tokens=skip_special)

    def save(self, path, file_name):
        """Writes `<path>/<file_name>-vocab.json` and `-merges.txt` (vocab.py:36-37; the 2020 API
        `tokenizer.save(path, name)` is `save_model` in tokenizers >= 0.8)."""
        os.makedirs(path, exist_ok=True)
        self.tokenizer.save_model(path, file_name)

    @classmethod
    def load(cls, vocab, merges):
        tkz = cls(None, None, lazy=True)
        tkz.tokenizer = CharBPETokenizer(vocab, merges)
        tkz.tokenizer.add_special_tokens([PAD, BOS, EOS])
        return tkz

    def __len__(self):
        return self.tokenizer.get_vocab_size()
