"""Text -> embedding ingest rate: host tokenization overlapped with the encoder.

Builds a bge-m3-shaped embedder (synthetic weights) and a synthetic SentencePiece-style Unigram tokenizer
(250k pieces, the XLM-R vocabulary size; Rust `tokenizers`, same code path as a real tokenizer.json), then times
  tokenize only | encoder only (token ids ready) | tokenize THEN encode (serial) | the pipelined product path.
Usage: python tools/probes/ingest_pipeline.py [n_texts] [words_per_text] [hash|unigram] [window]"""
import os
import random
import sys
import tempfile
import time

import torch

sys.path.insert(0, ".")
from tensor_truth_amd.embedding import HipHuggingFaceEmbedding  # noqa: E402
from tensor_truth_amd.tokenization import HashTokenizer, HFTokenizer  # noqa: E402


def build_unigram_tokenizer(path: str, vocab_size: int = 250_002, seed: int = 5) -> str:
    """tokenizer.json with XLM-R's special-token layout and a random Unigram vocabulary (piece = 1-7 letters)."""
    from tokenizers import Tokenizer, models, normalizers, pre_tokenizers, processors

    rnd = random.Random(seed)
    letters = "abcdefghijklmnopqrstuvwxyz"
    pieces = {"<s>": 0.0, "<pad>": 0.0, "</s>": 0.0, "<unk>": 0.0}
    for ch in letters + "0123456789.,;:!?-":
        pieces[ch] = -12.0
        pieces["▁" + ch] = -11.0
    pieces["▁"] = -9.0
    while len(pieces) < vocab_size:
        n = rnd.randint(2, 7)
        w = "".join(rnd.choice(letters) for _ in range(n))
        if rnd.random() < 0.5:
            w = "▁" + w
        pieces.setdefault(w, -4.0 - 1.2 * n + rnd.random())
    tk = Tokenizer(models.Unigram(list(pieces.items()), unk_id=3, byte_fallback=False))
    tk.normalizer = normalizers.NFKC()
    tk.pre_tokenizer = pre_tokenizers.Metaspace()
    tk.post_processor = processors.TemplateProcessing(single="<s> $A </s>", pair="<s> $A </s> </s> $B </s>",
                                                      special_tokens=[("<s>", 0), ("</s>", 2)])
    os.makedirs(path, exist_ok=True)
    out = os.path.join(path, "tokenizer.json")
    tk.save(out)
    return out


def make_texts(n: int, words: int, seed: int = 11):
    rnd = random.Random(seed)
    letters = "abcdefghijklmnopqrstuvwxyz"
    lex = ["".join(rnd.choice(letters) for _ in range(rnd.randint(2, 10))) for _ in range(20000)]
    return [" ".join(rnd.choice(lex) for _ in range(max(8, int(rnd.gauss(words, words * 0.25))))) for _ in range(n)]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    words = int(sys.argv[2]) if len(sys.argv) > 2 else 110
    kind = sys.argv[3] if len(sys.argv) > 3 else "unigram"
    texts = make_texts(n, words)
    if kind == "unigram":
        tok = HFTokenizer(build_unigram_tokenizer(tempfile.mkdtemp()), "xlmr")
        vocab = 250_002
    else:
        vocab = 250_002
        tok = HashTokenizer("xlmr", vocab)
    emb = HipHuggingFaceEmbedding("BAAI/bge-m3", device="cuda", embed_batch_size=1024,
                                  model_kwargs={"synthetic_seed": 3, "tokenizer": tok}, max_length=512)
    if len(sys.argv) > 4:
        emb.pipeline_window = int(sys.argv[4])
    t0 = time.perf_counter()
    seqs = emb._tokenize(texts, "")
    t_tok = time.perf_counter() - t0
    toks = sum(len(s) for s in seqs)
    emb.embed_token_batches(seqs[:1024])          # warm-up (workspace, pinned staging)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    emb.embed_token_batches(seqs)
    torch.cuda.synchronize()
    t_gpu = time.perf_counter() - t0
    t0 = time.perf_counter()
    ref = emb.embed_token_batches(emb._tokenize(texts, ""))
    torch.cuda.synchronize()
    t_serial = time.perf_counter() - t0
    t0 = time.perf_counter()
    got = emb._embed_texts(texts, "")
    torch.cuda.synchronize()
    t_pipe = time.perf_counter() - t0
    if os.environ.get("TT_PIPE_TRACE"):
        tok_t, enq_t = [], []
        orig_tok, orig_enq = emb._tokenize, emb.embed_token_batches

        def tok(ts, pre):
            a = time.perf_counter()
            r = orig_tok(ts, pre)
            tok_t.append((a - t0, time.perf_counter() - t0, len(ts)))
            return r

        def enq(sq):
            a = time.perf_counter()
            r = orig_enq(sq)
            enq_t.append((a - t0, time.perf_counter() - t0, len(sq)))
            return r

        emb._tokenize, emb.embed_token_batches = tok, enq
        t0 = time.perf_counter()
        emb._embed_texts(texts, "")
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print(f"host loop done at {t_host * 1e3:.0f} ms, GPU done at {t_all * 1e3:.0f} ms")
        print("tokenize spans (ms):", [(round(a * 1e3), round(b * 1e3), n) for a, b, n in tok_t])
        print("enqueue  spans (ms):", [(round(a * 1e3), round(b * 1e3), n) for a, b, n in enq_t])
        emb._tokenize, emb.embed_token_batches = orig_tok, orig_enq
    same = bool(torch.equal(ref, got))
    print(f"{kind} tokenizer, {n} texts, {toks / n:.0f} tokens/text, window {emb.pipeline_window}, host cores {os.cpu_count()}: "
          f"tokenize {n / t_tok:.0f} texts/s | encoder only {n / t_gpu:.0f} | serial {n / t_serial:.0f} | "
          f"pipelined {n / t_pipe:.0f} chunks/s | identical embeddings: {same}")


if __name__ == "__main__":
    main()
