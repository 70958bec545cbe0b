"""Tokenizer adapters for the encoders.

The reference tokenizes inside sentence-transformers with the model's HF tokenizer
(Rust ``tokenizers``).  ``HFTokenizer`` wraps a local ``tokenizer.json`` when
one exists; ``HashTokenizer`` is a deterministic stand-in for environments without tokenizer
files (the build container and the benchmark box have no network): it produces ids in the
model's vocabulary range with the model's special-token layout, which is all the kernels
and the parity tests need.  The kernel boundary itself is token-ids-in.
"""
from __future__ import annotations

import hashlib
import os
import re
import threading
from typing import List, Optional, Sequence, Tuple

_WORD = re.compile(r"\w+|[^\w\s]", re.UNICODE)


_warned_threads = False


def _cap_tokenizer_threads() -> None:
    """The Rust ``tokenizers`` library starts one rayon thread per host core the first time a batch is encoded.  On the 256-thread
    GPU host that buys nothing -- 400 rerank pairs x 292 tokens take 26.5 ms with 256 threads and 27.1 ms with 16
    (tools/probes/tokenizer_threads.py, profiles/r05_tokenizer_threads.log) -- and costs a lot as soon as two request threads encode
    at once (a rerank batch's pairs beside a retrieval batch's queries: 0.4 ms of query tokenisation became 5.5 ms in the surface
    leg).  The size of that pool is the APPLICATION's decision: ``RAYON_NUM_THREADS`` is process-wide, every other rayon user
    (polars ...) and every child process inherits it, and it is read once, when the pool starts.  So this package never sets it on
    its own: ``TT_TOKENIZER_THREADS=N`` is the explicit opt-in (applied only while ``RAYON_NUM_THREADS`` is unset, before the library's
    first parallel call); without it a many-core host gets one log line naming the setting.  The package's own worker processes
    get their thread count in the environment they are started with (``ingest_workers``), never through this process's."""
    global _warned_threads
    if "RAYON_NUM_THREADS" in os.environ:
        return
    want = os.environ.get("TT_TOKENIZER_THREADS", "").strip()
    if want:
        os.environ["RAYON_NUM_THREADS"] = str(max(1, int(want)))
        return
    try:
        cpus = len(os.sched_getaffinity(0))
    except AttributeError:
        cpus = os.cpu_count() or 1
    if cpus > 16 and not _warned_threads:
        _warned_threads = True
        import logging

        logging.getLogger(__name__).info(
            "tokenizers will start %d rayon threads; on many-core hosts RAYON_NUM_THREADS=16 (or TT_TOKENIZER_THREADS=16) in the "
            "application's environment avoids contention between request threads", cpus)


class PreTokenized:
    """A passage whose token ids are already known (the index kept its leaves' ids at ingest: ``HipVectorIndex.leaf_token_ids``):
    the BODY ids, without special tokens.  Stands in for the passage string in a (query, passage) pair handed to the reranker."""

    __slots__ = ("ids",)

    def __init__(self, ids):
        self.ids = ids


def truncate_longest_first(n_a: int, n_b: int, budget: int):
    """The lengths the Rust ``tokenizers`` library keeps of a pair under ``TruncationStrategy::LongestFirst`` with ``budget`` =
    max_length minus the pair's special tokens (tokenizers/src/utils/truncation.rs, restated): nothing is cut while the pair fits;
    otherwise the shorter side keeps what it has up to half the budget and the longer side gets the rest."""
    if n_a + n_b <= budget:
        return n_a, n_b
    n1, n2, swap = n_a, n_b, False
    if n1 > n2:
        n1, n2, swap = n2, n1, True
    n2 = n1 if n1 > budget else max(n1, budget - n1)
    if n1 + n2 > budget:
        n1 = budget // 2
        n2 = n1 + budget % 2
    if swap:
        n1, n2 = n2, n1
    return n1, n2


def tokenizer_signature(tk) -> str:
    """Two components may exchange token ids only if their tokenizers are the same function: a digest of the serialised tokenizer
    (HFTokenizer) or of the stand-in's parameters."""
    if isinstance(tk, HFTokenizer):
        return "hf:" + hashlib.sha256(tk._json.encode("utf-8")).hexdigest()
    if isinstance(tk, HashTokenizer):
        return f"hash:{tk.arch}:{tk.vocab_size}"
    return f"other:{id(tk)}"


def assemble_pairs(tk, pairs, max_length: int):
    """(query string, PreTokenized passage) pairs -> the ids ``tk.encode_pair(query, passage_text, max_length)`` would return, without
    tokenising the passages again: the query is tokenised once per distinct string, the specials come from the tokenizer's layout,
    the cut is the library's longest-first rule.  -> list of int32 arrays."""
    import numpy as np

    sp = tk.sp
    sep = np.asarray(sp.pair_sep, dtype=np.int32)
    budget = max_length - 2 - len(sp.pair_sep)
    qcache = {}
    out = []
    for q, pre in pairs:
        qi = qcache.get(q)
        if qi is None:
            qi = qcache[q] = np.asarray(tk.encode(q, None)[1:-1], dtype=np.int32)
        pi = pre.ids
        n1, n2 = truncate_longest_first(len(qi), len(pi), budget)
        ids = np.empty(n1 + n2 + 2 + len(sep), dtype=np.int32)
        ids[0] = sp.bos
        ids[1:1 + n1] = qi[:n1]
        ids[1 + n1:1 + n1 + len(sep)] = sep
        ids[1 + n1 + len(sep):-1] = pi[:n2]
        ids[-1] = sp.eos
        out.append(ids)
    return out


class SpecialTokens:
    def __init__(self, arch: str):
        if arch == "xlmr":
            self.bos, self.pad, self.eos, self.unk, self.first_free = 0, 1, 2, 3, 4
            self.pair_sep = [2, 2]          # <s> A </s></s> B </s>
        else:
            self.bos, self.pad, self.eos, self.unk, self.first_free = 101, 0, 102, 100, 1000
            self.pair_sep = [102]           # [CLS] A [SEP] B [SEP]


class HashTokenizer:
    """Word-level hashing tokenizer (deterministic, vocabulary-range ids)."""

    def __init__(self, arch: str, vocab_size: int):
        self.arch = arch
        self.vocab_size = vocab_size
        self.sp = SpecialTokens(arch)
        self._cache = {}           # word -> id (words repeat; bounded below)
        self._text_cache = {}      # text -> ids (the same chunks are retrieved again and again; bounded, see _ids)

    def _ids(self, text: str) -> List[int]:
        hit = self._text_cache.get(text)
        if hit is not None:
            return list(hit)
        out = self._ids_uncached(text)
        if len(self._text_cache) >= 65536:      # crude bound: drop everything (a few hundred MB at most before that)
            self._text_cache.clear()
        self._text_cache[text] = tuple(out)
        return out

    def _ids_uncached(self, text: str) -> List[int]:
        lo = self.sp.first_free
        span = self.vocab_size - lo
        cache = self._cache
        out = []
        for w in _WORD.findall(text):
            t = cache.get(w)
            if t is None:
                h = int.from_bytes(hashlib.blake2b(w.encode("utf-8"), digest_size=8).digest(), "little")
                t = lo + h % span
                if len(cache) < (1 << 20):
                    cache[w] = t
            out.append(t)
        return out

    def encode(self, text: str, max_length: Optional[int] = None) -> List[int]:
        body = self._ids(text)
        if max_length is not None:
            body = body[: max(0, max_length - 2)]
        return [self.sp.bos] + body + [self.sp.eos]

    def encode_batch(self, texts: Sequence[str], max_length: Optional[int] = None) -> List[List[int]]:
        return [self.encode(t, max_length) for t in texts]

    def encode_pair_batch(self, pairs: Sequence[Tuple[str, str]], max_length: int = 512):
        return [self.encode_pair(a, b, max_length) for a, b in pairs]

    def encode_pair(self, a: str, b: str, max_length: int = 512) -> Tuple[List[int], List[int]]:
        """-> (ids, token_type_ids); truncation 'longest_first' like CrossEncoder's tokenizer call."""
        ia, ib = self._ids(a), self._ids(b)
        budget = max_length - 2 - len(self.sp.pair_sep)
        na, nb = truncate_longest_first(len(ia), len(ib), max(budget, 0))      # (the Rust library's rule: one definition for both tokenizers)
        ia, ib = ia[:na], ib[:nb]
        ids = [self.sp.bos] + ia + self.sp.pair_sep + ib + [self.sp.eos]
        n_a = 1 + len(ia) + (1 if self.arch != "xlmr" else len(self.sp.pair_sep))
        types = [0] * n_a + [1] * (len(ids) - n_a) if self.arch != "xlmr" else [0] * len(ids)
        return ids, types


class HFTokenizer:
    """Adapter over a local HF ``tokenizer.json`` (``tokenizers`` library).  Two tokenizer objects: one that never
    truncates (single texts; long ones are cut here, keeping the closing special token) and one with
    ``longest_first`` truncation for pairs -- a Rust tokenizer cannot be reconfigured while another thread is
    encoding with it, and the ingest pipeline tokenizes from background threads."""

    def __init__(self, tokenizer_json: Optional[str], arch: str, json_str: Optional[str] = None):
        _cap_tokenizer_threads()
        from tokenizers import Tokenizer  # local import: optional dependency

        self._path = tokenizer_json
        # the serialised tokenizer is kept in memory: the pair tokenizer is built from it later, whatever has happened
        # to the file meanwhile (and a tokenizer converted from sentencepiece files never touches the disk at all)
        self._json = json_str if json_str is not None else open(tokenizer_json, encoding="utf-8").read()
        self.tk = Tokenizer.from_str(self._json)
        self.tk.no_truncation()
        self.tk.no_padding()
        self._pair_tk = None
        self._pair_len = None
        self._lock = threading.Lock()
        self.arch = arch
        self.sp = SpecialTokens(arch)

    @staticmethod
    def _cut(ids: List[int], max_length: Optional[int]) -> List[int]:
        if max_length is not None and len(ids) > max_length:
            return ids[: max_length - 1] + [ids[-1]]
        return ids

    def encode(self, text: str, max_length: Optional[int] = None) -> List[int]:
        return self._cut(self.tk.encode(text).ids, max_length)

    def encode_batch(self, texts: Sequence[str], max_length: Optional[int] = None) -> List[List[int]]:
        """Whole batch in one call: the Rust tokenizer fans it out over the host cores and releases the GIL, so
        background threads can tokenize the next windows while this one is packed and enqueued (embedding.py)."""
        texts = list(texts)
        if len(texts) <= 8:
            # a retrieval batch's few query strings: one by one on this thread -- a batch call would queue behind whatever large
            # job (a rerank batch's pairs) occupies the library's shared thread pool (0.4 ms became 5-8 ms in the surface leg)
            return [self._cut(self.tk.encode(t).ids, max_length) for t in texts]
        return [self._cut(enc.ids, max_length) for enc in self._batch(self.tk, texts)]

    @staticmethod
    def _batch(tk, inputs):
        # encode_batch_fast (tokenizers >= 0.20) skips the character offsets nothing here reads
        fast = getattr(tk, "encode_batch_fast", None)
        return fast(inputs) if fast is not None else tk.encode_batch(inputs)

    def _pairs(self, max_length: int):
        from tokenizers import Tokenizer

        with self._lock:
            if self._pair_tk is None or self._pair_len != max_length:
                tk = Tokenizer.from_str(self._json)
                tk.no_padding()
                tk.enable_truncation(max_length=max_length, strategy="longest_first")
                self._pair_tk, self._pair_len = tk, max_length
            return self._pair_tk

    def encode_pair(self, a: str, b: str, max_length: int = 512):
        enc = self._pairs(max_length).encode(a, b)
        return enc.ids, enc.type_ids

    def encode_pair_batch(self, pairs: Sequence[Tuple[str, str]], max_length: int = 512):
        encs = self._batch(self._pairs(max_length), [(a, b) for a, b in pairs])
        if self.arch == "xlmr":
            # XLM-R has ONE token type (type_vocab 1): the segment ids are all zero and no caller reads them -- building 400 x 292
            # more Python ints per rerank batch under the GIL is a third of this call
            return [(enc.ids, None) for enc in encs]
        return [(enc.ids, enc.type_ids) for enc in encs]


def load_tokenizer(model_dir: Optional[str], arch: str, vocab_size: int):
    """``model_dir`` None (synthetic / state_dict weights: no tokenizer exists) -> the hashing stand-in.  A real
    checkpoint directory must bring a real tokenizer: ``tokenizer.json``, or sentencepiece / WordPiece files that
    ``transformers`` can convert to one -- real weights fed hashed ids would produce meaningless embeddings and
    rerank scores without any error, so anything else raises (the package's no-silent-fallback policy)."""
    if not model_dir:
        return HashTokenizer(arch, vocab_size)
    tj = os.path.join(model_dir, "tokenizer.json")
    if os.path.exists(tj):
        return HFTokenizer(tj, arch)
    slow_files = ("sentencepiece.bpe.model", "spiece.model", "tokenizer.model", "vocab.txt")
    if any(os.path.exists(os.path.join(model_dir, f)) for f in slow_files):
        try:
            from transformers import AutoTokenizer

            fast = AutoTokenizer.from_pretrained(model_dir, use_fast=True, local_files_only=True)
            backend = getattr(fast, "backend_tokenizer", None)
            if backend is None:
                raise RuntimeError("transformers returned a slow tokenizer")
            # The converted tokenizer stays in memory, nothing is written into the model directory: with one process per
            # GPU every rank converts at the same moment (a half-written tokenizer.json would be parsed by its
            # neighbour), and the directory may be a shared or HF-cache snapshot this package has no business editing.
            return HFTokenizer(None, arch, json_str=backend.to_str())
        except Exception as exc:  # noqa: BLE001
            raise FileNotFoundError(
                f"{model_dir}: no tokenizer.json, and converting its sentencepiece / vocab files failed ({exc})") from exc
    raise FileNotFoundError(
        f"{model_dir} holds model weights but no tokenizer (tokenizer.json, sentencepiece.bpe.model or vocab.txt): "
        "refusing to pair real weights with the hashing stand-in tokenizer")
