"""Generate the golden fixtures under tests/golden/ (run in the BUILD container only).

    python tests/golden/make_golden.py

Needs what only the build container has: ``transformers`` (the third-party
implementation the reference's embedder / reranker run on) and the
reference checkout at /root/reference (for ``services/retrieval_metrics.py``,
imported by file path).  Neither travels to the GPU box; the fixtures written
here do.  Fixtures hold inputs and expected outputs only -- weights are
re-derived from their seed by ``oracle.encoder.synth_weights`` and guarded by a
checksum stored in the fixture.
"""
from __future__ import annotations

import importlib.util
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import encoder as oe  # noqa: E402
from oracle import scan as osc  # noqa: E402


def weights_checksum(W) -> float:
    return float(sum(float(v.double().abs().sum()) for v in W.values()))


def hf_state_dict(W, prefix):
    return {prefix + k if not k.startswith("classifier.") else k: v for k, v in W.items()}


def make_xlmr():
    from transformers import XLMRobertaConfig, XLMRobertaModel

    cfg = oe.EncoderConfig(arch="xlmr", vocab_size=1000, hidden=1024, layers=2, heads=16, ffn=4096,
                           max_pos=66, type_vocab=1, pad_id=1, ln_eps=1e-5)
    W = oe.synth_weights(cfg, seed=11)
    ids, mask = oe.synth_tokens(4, 16, cfg, seed=777, lengths=[16, 9, 5, 12])
    hcfg = XLMRobertaConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden, num_hidden_layers=cfg.layers,
                            num_attention_heads=cfg.heads, intermediate_size=cfg.ffn,
                            max_position_embeddings=cfg.max_pos, type_vocab_size=1, pad_token_id=1,
                            layer_norm_eps=cfg.ln_eps, hidden_act="gelu",
                            hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    model = XLMRobertaModel(hcfg, add_pooling_layer=False).eval()
    missing, unexpected = model.load_state_dict(W, strict=False)
    assert not unexpected, unexpected
    assert all("position_ids" in m or "token_type_ids" in m for m in missing), missing
    with torch.no_grad():
        hid = model(input_ids=ids, attention_mask=mask).last_hidden_state
    emb = torch.nn.functional.normalize(hid[:, 0], p=2, dim=1)
    mine = oe.encoder_forward(ids, mask, W, cfg)
    err = (mine - hid)[mask.bool()].abs().max().item()
    print(f"xlmr: oracle vs transformers max abs err (valid tokens) = {err:.3e}")
    assert err < 2e-4
    np.savez_compressed(
        os.path.join(HERE, "xlmr_encoder.npz"),
        cfg=json.dumps(cfg.__dict__), seed=11, wsum=weights_checksum(W),
        ids=ids.numpy(), mask=mask.numpy(), hidden=hid.numpy().astype(np.float32),
        emb=emb.numpy().astype(np.float32))


def make_bert():
    from transformers import BertConfig, BertModel

    cfg = oe.EncoderConfig(arch="bert", vocab_size=1000, hidden=384, layers=2, heads=12, ffn=1536,
                           max_pos=64, type_vocab=2, pad_id=0, ln_eps=1e-12)
    W = oe.synth_weights(cfg, seed=12)
    ids, mask = oe.synth_tokens(4, 16, cfg, seed=778, lengths=[16, 7, 11, 3])
    type_ids = torch.zeros_like(ids)
    type_ids[:, 8:] = 1
    hcfg = BertConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden, num_hidden_layers=cfg.layers,
                      num_attention_heads=cfg.heads, intermediate_size=cfg.ffn,
                      max_position_embeddings=cfg.max_pos, type_vocab_size=2, pad_token_id=0,
                      layer_norm_eps=cfg.ln_eps, hidden_act="gelu",
                      hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    model = BertModel(hcfg, add_pooling_layer=False).eval()
    missing, unexpected = model.load_state_dict(W, strict=False)
    assert not unexpected, unexpected
    assert all("position_ids" in m or "token_type_ids" in m for m in missing), missing
    with torch.no_grad():
        hid = model(input_ids=ids, attention_mask=mask, token_type_ids=type_ids).last_hidden_state
    emb = torch.nn.functional.normalize(hid[:, 0], p=2, dim=1)
    mine = oe.encoder_forward(ids, mask, W, cfg, type_ids=type_ids)
    err = (mine - hid)[mask.bool()].abs().max().item()
    print(f"bert: oracle vs transformers max abs err (valid tokens) = {err:.3e}")
    assert err < 2e-4
    np.savez_compressed(
        os.path.join(HERE, "bert_encoder.npz"),
        cfg=json.dumps(cfg.__dict__), seed=12, wsum=weights_checksum(W),
        ids=ids.numpy(), mask=mask.numpy(), type_ids=type_ids.numpy(),
        hidden=hid.numpy().astype(np.float32), emb=emb.numpy().astype(np.float32))


def make_xenc():
    from transformers import XLMRobertaConfig, XLMRobertaForSequenceClassification

    cfg = oe.EncoderConfig(arch="xlmr", vocab_size=1000, hidden=1024, layers=2, heads=16, ffn=4096,
                           max_pos=66, type_vocab=1, pad_id=1, ln_eps=1e-5, num_labels=1)
    W = oe.synth_weights(cfg, seed=13)
    ids, mask = oe.synth_tokens(8, 24, cfg, seed=779, lengths=[24, 20, 13, 24, 9, 17, 5, 22])
    hcfg = XLMRobertaConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden, num_hidden_layers=cfg.layers,
                            num_attention_heads=cfg.heads, intermediate_size=cfg.ffn,
                            max_position_embeddings=cfg.max_pos, type_vocab_size=1, pad_token_id=1,
                            layer_norm_eps=cfg.ln_eps, hidden_act="gelu", num_labels=1,
                            hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                            classifier_dropout=0.0)
    model = XLMRobertaForSequenceClassification(hcfg).eval()
    sd = {("roberta." + k if not k.startswith("classifier.") else k): v for k, v in W.items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("position_ids" in m or "token_type_ids" in m for m in missing), missing
    with torch.no_grad():
        logits = model(input_ids=ids, attention_mask=mask).logits[:, 0]
    scores = torch.sigmoid(logits)
    mine = oe.rerank_logits(ids, mask, W, cfg)
    err = (mine - logits).abs().max().item()
    print(f"xenc: oracle vs transformers max abs logit err = {err:.3e}")
    assert err < 2e-4
    np.savez_compressed(
        os.path.join(HERE, "xenc_head.npz"),
        cfg=json.dumps(cfg.__dict__), seed=13, wsum=weights_checksum(W),
        ids=ids.numpy(), mask=mask.numpy(), logits=logits.numpy().astype(np.float32),
        scores=scores.numpy().astype(np.float32))


def make_scan():
    """C[4096,1024] bf16, Q[16,1024], K=50 -> fp32 scores + indices + tie gap.
    Produced by oracle/scan.py and cross-checked against oracle/scan_ref.c."""
    import ctypes

    corpus = osc.synth_corpus(4096, 1024, seed=1234)
    queries, planted = osc.synth_queries(corpus, 16, seed=4321)
    vals, idx, gap = osc.scan_topk(corpus, queries, 50)
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle_scan.so"))
    cs = np.empty((16, 50), np.float32)
    ci = np.empty((16, 50), np.int32)
    c16 = corpus.view(torch.int16).numpy()
    q16 = queries.view(torch.int16).numpy()
    rc = lib.tt_oracle_scan_topk(c16.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(4096), 1024,
                                 q16.ctypes.data_as(ctypes.c_void_p), 16, 50,
                                 cs.ctypes.data_as(ctypes.c_void_p), ci.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    tie_free = gap.numpy() > 1e-6
    assert (ci[tie_free] == idx.numpy()[tie_free]).all(), "C and numpy oracles disagree on indices"
    assert np.allclose(cs, vals.numpy(), rtol=1e-5, atol=1e-6)
    for q in range(16):
        if planted[q] >= 0:
            assert idx[q, 0] == planted[q]
    print(f"scan: tie-free queries {int(tie_free.sum())}/16, min gap {gap.min().item():.3e}")
    np.savez_compressed(os.path.join(HERE, "scan_4096x1024_k50.npz"), corpus_seed=1234, query_seed=4321,
                        n=4096, d=1024, k=50, scores=vals.numpy(), idx=idx.numpy().astype(np.int32),
                        gap=gap.numpy(), planted=planted.numpy(),
                        corpus_sum=float(corpus.float().double().sum()),
                        query_sum=float(queries.float().double().sum()))


def make_metrics():
    """Outputs of the reference's own compute_retrieval_metrics on hand-made nodes."""
    path = "/root/reference/src/tensortruth/services/retrieval_metrics.py"
    spec = importlib.util.spec_from_file_location("ref_retrieval_metrics", path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules["ref_retrieval_metrics"] = mod
    spec.loader.exec_module(mod)

    class N:  # minimal TextNode / NodeWithScore stand-ins (duck typing as the reference does)
        def __init__(self, text, metadata):
            self.text, self.metadata = text, metadata

        def get_content(self):
            return self.text

    class NS:
        def __init__(self, score, text, metadata):
            self.score, self.node = score, N(text, metadata)

    cases = {
        "empty": [],
        "single": [[0.83, "alpha " * 10, {"filename": "a.md", "doc_type": "library"}]],
        "mixed": [
            [0.91, "x" * 400, {"filename": "a.md", "doc_type": "library"}],
            [0.72, "y" * 123, {"filename": "a.md", "doc_type": "library"}],
            [0.55, "z" * 999, {"file_name": "b.pdf", "doc_type": "paper"}],
            [0.39, "w" * 10, {"source_url": "http://c", "doc_type": "book"}],
            [0.12, "", {}],
        ],
        "none_scores": [
            [None, "abc", {"filename": "a"}],
            [0.5, "defg", {"filename": "b"}],
        ],
        "even": [[s, "t" * (50 + i), {"filename": f"f{i % 3}", "doc_type": "library"}]
                 for i, s in enumerate([0.1, 0.2, 0.3, 0.4, 0.7, 0.8, 0.9, 0.95])],
    }
    out = {}
    for name, rows in cases.items():
        nodes = [NS(s, t, m) for s, t, m in rows]
        out[name] = {"nodes": rows, "expected": mod.compute_retrieval_metrics(nodes).to_dict()}
    out["entropy"] = [{"counts": c, "expected": mod.calculate_entropy(c)}
                      for c in ([], [5], [1, 1], [3, 1], [2, 2, 2, 2], [0, 0], [10, 1, 1])]
    with open(os.path.join(HERE, "metrics_golden.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("metrics: wrote", len(out), "cases")


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    make_xlmr()
    make_bert()
    make_xenc()
    make_scan()
    make_metrics()
