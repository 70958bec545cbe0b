"""Synthetic natural-language-like text and a TRAINED sub-word tokenizer for the timed surface legs of bench.py.

The reference tokenizes with the model's own HF tokenizer -- for BAAI/bge-m3 and bge-reranker-v2-m3 XLM-R's SentencePiece Unigram
model, 250 002 pieces, pair template ``<s> A </s></s> B </s>`` (``services/model_manager.py:254-260`` -> sentence-transformers).
No tokenizer file exists offline, so one is TRAINED here with the same library (`tokenizers`, Rust) and the same model class:

* ``lexicon()``: 400 000 pronounceable pseudo-words built from syllables (seeded), in a seeded order that is their frequency rank;
* ``make_tokenizer()``: `UnigramTrainer` over a corpus in which every word occurs and frequent words follow Zipf's law, NFKC +
  Metaspace as XLM-R, specials at ids 0-3 (<s>, <pad>, </s>, <unk>), learned pieces + never-matching filler pieces up to exactly
  250 002 (the trainer's seed set saturates at ~240 k on this corpus), XLM-R's post-processor.  ~6 minutes on 8 cores, so the result
  is COMMITTED as a fixture (`tests/golden/unigram250k_tokenizer.json.xz`, 2.4 MB) and this script is its generator:
      python tools/synth_text.py --train
* ``zipf_text(key, n_words)``: deterministic text for corpus row / query ``key``: words drawn from the lexicon by Zipf rank.
  ~2.2 pieces per word under the trained model (XLM-R on English: ~1.3; the pseudo-words carry no morphology to learn).
"""
from __future__ import annotations

import lzma
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURE = os.path.join(ROOT, "tests", "golden", "unigram250k_tokenizer.json.xz")
VOCAB = 250_002
N_WORDS = 400_000
_LEX = None
_CDF = None


def lexicon():
    """The 400 000 pseudo-words, index = frequency rank."""
    global _LEX
    if _LEX is None:
        rnd = random.Random(7)
        cons = ["b", "c", "d", "f", "g", "h", "j", "k", "l", "m", "n", "p", "r", "s", "t", "v", "w", "z", "ch", "sh", "th", "st", "tr", "pr",
                "br", "kr", "pl"]
        vow = ["a", "e", "i", "o", "u", "ai", "ea", "ou", "io", "ee"]
        syl = [c + v for c in cons for v in vow] + [c + v + c2 for c in cons[:12] for v in vow[:5] for c2 in ["n", "r", "s", "t", "l", "m"]]
        words = set()
        while len(words) < N_WORDS:
            k = rnd.choices([1, 2, 3, 4, 5], [5, 30, 35, 20, 10])[0]
            words.add("".join(rnd.choice(syl) for _ in range(k)))
        lex = sorted(words)
        perm = np.random.default_rng(3).permutation(len(lex))
        _LEX = [lex[i] for i in perm]
    return _LEX


def _cdf():
    global _CDF
    if _CDF is None:
        p = 1.0 / np.arange(1, N_WORDS + 1, dtype=np.float64)
        _CDF = np.cumsum(p / p.sum())
    return _CDF


def zipf_words(key: int, n_words: int, top: int = N_WORDS):
    """``top`` < N_WORDS: only the ``top`` most frequent words (Zipf's law renormalised over them) -- running text of a narrower
    vocabulary, fewer pieces per word under the trained model (the training corpus uses all 400 000)."""
    u = np.random.default_rng([int(key) & 0xFFFFFFFF, int(key) >> 32, 91]).random(n_words)
    lex, cdf = lexicon(), _cdf()
    if top < N_WORDS:
        u = u * cdf[top - 1]
    return [lex[i] for i in np.minimum(np.searchsorted(cdf, u), top - 1)]


def zipf_text(key: int, n_words: int, top: int = N_WORDS) -> str:
    return " ".join(zipf_words(key, n_words, top))


def make_tokenizer(n_sentences: int = 100_000) -> str:
    """Train -> tokenizer.json text (see the module docstring)."""
    from tokenizers import Tokenizer, models, normalizers, pre_tokenizers, processors, trainers
    import json

    lex = lexicon()

    def corpus():
        for i in range(0, len(lex), 40):                   # every word at least once
            yield " ".join(lex[i:i + 40])
        for s in range(n_sentences):                       # Zipfian running text
            yield zipf_text(1_000_000_007 + s, 40)

    tk = Tokenizer(models.Unigram())
    tk.normalizer = normalizers.NFKC()
    tk.pre_tokenizer = pre_tokenizers.Metaspace()
    tr = trainers.UnigramTrainer(vocab_size=VOCAB, special_tokens=["<s>", "<pad>", "</s>", "<unk>"], unk_token="<unk>", show_progress=False)
    tk.train_from_iterator(corpus(), trainer=tr)
    blob = json.loads(tk.to_str())
    vocab = blob["model"]["vocab"]
    have = {p for p, _ in vocab}
    rnd, i = random.Random(11), 0
    while len(vocab) < VOCAB:                               # filler pieces: upper-case, never produced by the lower-case lexicon
        piece = "▁" + "".join(rnd.choice("QXZJKVWY") for _ in range(9)) + str(i)
        i += 1
        if piece not in have:
            have.add(piece)
            vocab.append([piece, -40.0])
    blob["model"]["unk_id"] = 3
    tk = Tokenizer.from_str(json.dumps(blob))
    tk.post_processor = processors.TemplateProcessing(single="<s> $A </s>", pair="<s> $A </s> </s> $B </s>",
                                                      special_tokens=[("<s>", 0), ("</s>", 2)])
    assert tk.get_vocab_size() == VOCAB and tk.token_to_id("<s>") == 0 and tk.token_to_id("<pad>") == 1 and tk.token_to_id("</s>") == 2
    return tk.to_str()


def load_tokenizer_json() -> str:
    with lzma.open(FIXTURE, "rt", encoding="utf-8") as f:
        return f.read()


def unigram_tokenizer():
    """-> tensor_truth_amd.tokenization.HFTokenizer over the committed 250 002-piece Unigram model (XLM-R layout)."""
    sys.path.insert(0, ROOT)
    from tensor_truth_amd.tokenization import HFTokenizer

    return HFTokenizer(None, "xlmr", json_str=load_tokenizer_json())


if __name__ == "__main__":
    if "--train" in sys.argv:
        text = make_tokenizer()
        with lzma.open(FIXTURE, "wt", encoding="utf-8", preset=9) as f:
            f.write(text)
        print(f"wrote {FIXTURE}: {os.path.getsize(FIXTURE) / 1e6:.2f} MB")
    tk = unigram_tokenizer()
    words = zipf_words(5, 40)
    ids = tk.encode(" ".join(words))
    print(f"vocab {tk.tk.get_vocab_size()}; 40 words -> {len(ids)} ids; pair -> {tk.encode_pair('hello world', 'second text')[0]}")
