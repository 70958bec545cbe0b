"""GPU: the reference-precision (fp32) path -- ``model_kwargs={"torch_dtype": "float32"}``, the reference's own default
dtype (``app_utils/config_schema.py:66-76``) -- against the fp32 CPU oracle at north_star's tolerance: scores within 1e-3
relative (measured: ~1e-5), embeddings to 1e-5.  (Full depth, 4 x 50 pairs x 24 layers: test_rank_agreement_gpu.py.)"""
import ctypes

import numpy as np
import pytest
import torch

from oracle import encoder as oe

pytestmark = pytest.mark.gpu

XLMR = dict(arch="xlmr", vocab_size=2000, hidden=256, layers=3, heads=4, ffn=512, max_pos=300, type_vocab=1, pad_id=1,
            ln_eps=1e-5, num_labels=1)
BERT = dict(arch="bert", vocab_size=3000, hidden=384, layers=2, heads=12, ffn=1536, max_pos=128, type_vocab=2, pad_id=0,
            ln_eps=1e-12)


def _pad(seqs, pad):
    L = max(len(s) for s in seqs)
    ids = torch.full((len(seqs), L), pad, dtype=torch.int64)
    mask = torch.zeros(len(seqs), L, dtype=torch.int64)
    for b, s in enumerate(seqs):
        ids[b, : len(s)] = torch.tensor(s)
        mask[b, : len(s)] = 1
    return ids, mask


@pytest.mark.parametrize("m,n,k", [(128, 128, 32), (200, 256, 96), (1, 384, 1024), (333, 1152, 384), (64, 1024, 4096)])
def test_gemm_f32_building_block(dev, built_lib, m, n, k):
    from tensor_truth_amd import _lib

    lib = _lib.load_library()
    g = torch.Generator().manual_seed(m * 7 + n)
    a, w = torch.randn(m, k, generator=g), torch.randn(n, k, generator=g) * 0.05
    bias, res = torch.randn(n, generator=g), torch.randn(m, n, generator=g)
    ref = a.double() @ w.double().T + bias.double()
    wants = {0: ref, 1: 0.5 * ref * (1 + torch.erf(ref / 2 ** 0.5)), 2: ref + res.double(), 3: torch.tanh(ref)}
    da, dw, db, dr = a.to(dev), w.to(dev), bias.to(dev), res.to(dev)
    for epi, want in wants.items():
        c = torch.full((m, n), float("nan"), device=dev)
        rc = lib.tt_gemm_f32(da.data_ptr(), dw.data_ptr(), db.data_ptr(), dr.data_ptr() if epi == 2 else None, c.data_ptr(),
                             m, n, k, epi, torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(rc, "tt_gemm_f32")
        torch.cuda.synchronize()
        err = (c.cpu().double() - want).abs().max().item()
        scale = want.abs().max().item()
        assert err <= 2e-6 * max(scale, 1.0) * (k / 32) ** 0.5, (epi, err, scale)     # fp32 sums of k products
    with pytest.raises(_lib.TTError):
        _lib.check(lib.tt_gemm_f32(da.data_ptr(), dw.data_ptr(), db.data_ptr(), None, c.data_ptr(), m, n + 1, k, 0, None), "tt_gemm_f32")


@pytest.mark.parametrize("shape", [XLMR, BERT], ids=["xlmr", "bert"])
def test_f32_forward_matches_the_oracle(dev, built_lib, shape):
    from tensor_truth_amd.encoder import EncoderConfig, pack_tokens
    from tensor_truth_amd.encoder_f32 import EncoderF32, EncoderWeightsF32

    cfg_o, cfg = oe.EncoderConfig(**shape), EncoderConfig(**shape)
    W = oe.synth_weights(cfg_o, seed=29)
    g = torch.Generator().manual_seed(4)
    lens = [n for n in (cfg.max_seq_len, 65, 129, 33, 7, 200, 100, 1 + 16) if n <= cfg.max_seq_len]
    lo, bos, eos = (4, 0, 2) if cfg.arch == "xlmr" else (1000, 101, 102)
    seqs = [[bos] + torch.randint(lo, cfg.vocab_size, (n - 2,), generator=g).tolist() + [eos] for n in lens]
    types = None
    if cfg.arch == "bert":
        types = [[0] * (len(s) // 2) + [1] * (len(s) - len(s) // 2) for s in seqs]
    enc = EncoderF32(EncoderWeightsF32(cfg, W, dev))
    batch = pack_tokens(seqs, cfg, types)
    hidden, _ = enc.forward_packed(batch)
    emb, emb16 = enc.embed_packed(batch)
    torch.cuda.synchronize()
    ids, mask = _pad(seqs, cfg.pad_id)
    tids = None
    if types is not None:
        tids = torch.zeros_like(ids)
        for b, t in enumerate(types):
            tids[b, : len(t)] = torch.tensor(t)
    with torch.no_grad():
        want_h = oe.encoder_forward(ids, mask, W, cfg_o, type_ids=tids)
    hidden = hidden.cpu()
    worst = 0.0
    for b, s in enumerate(seqs):
        st = int(batch.seq_start[b])
        worst = max(worst, (hidden[st:st + len(s)] - want_h[b, : len(s)]).abs().max().item())
    assert worst <= 2e-4, worst                                        # LayerNorm-scale values (|x| ~ 1-5), fp32 both sides
    want_e = oe.cls_pool_normalize(want_h)
    assert (emb.cpu() - want_e).abs().max().item() <= 2e-5
    assert torch.equal(emb16.cpu(), emb.cpu().to(torch.bfloat16))
    if cfg.num_labels:
        scores, logits = enc.rerank_packed(batch, want_logits=True)
        want_s = oe.rerank_scores(ids, mask, W, cfg_o)
        rel = ((scores.cpu() - want_s).abs() / want_s.abs()).max().item()
        assert rel <= 1e-3, rel                                        # north_star: fp scores within 1e-3 relative
        assert rel <= 1e-4, rel                                        # (what this path actually delivers)
        assert torch.allclose(scores.cpu(), torch.sigmoid(logits.cpu()), atol=1e-6)


def test_float32_through_the_plugin_surface(dev, built_lib, monkeypatch):
    """model_kwargs={"torch_dtype": "float32"} -- the string the reference's config carries (config_schema.py:66-76) and
    the torch dtype its ModelManager maps it to (model_manager.py:218-229) -- on both plugin classes, on the fp32-MFMA
    implementation of the reference precision (TT_REFERENCE_IMPL=fp32; the default for this shape is the split-bf16
    one: tests/test_x3_gpu.py)."""
    monkeypatch.setenv("TT_REFERENCE_IMPL", "fp32")
    from tensor_truth_amd.embedding import HipHuggingFaceEmbedding
    from tensor_truth_amd.encoder import EncoderConfig
    from tensor_truth_amd.encoder_f32 import EncoderF32
    from tensor_truth_amd.rerank import HipSentenceTransformerRerank
    from tensor_truth_amd.schema import NodeWithScore, QueryBundle, TextNode

    cfg, cfg_o = EncoderConfig(**XLMR), oe.EncoderConfig(**XLMR)
    W = oe.synth_weights(cfg_o, seed=31)
    rr = HipSentenceTransformerRerank(model="test/xenc", top_n=3, device="cuda",
                                      model_kwargs={"encoder_config": cfg, "state_dict": W, "torch_dtype": torch.float32})
    assert isinstance(rr._encoder, EncoderF32) and sum(p.numel() for p in rr.model.parameters()) > 0
    texts = [" ".join(f"w{(7 * i + j) % 50}" for j in range(5 + 3 * i)) for i in range(9)]
    query = "w1 w2 w3 which one"
    got = torch.tensor(rr.predict([(query, t) for t in texts]))
    pair_ids = [rr._tokenizer.encode_pair(query, t, rr.max_length)[0] for t in texts]
    ids, mask = _pad(pair_ids, cfg.pad_id)
    want = oe.rerank_scores(ids, mask, W, cfg_o)
    assert ((got - want).abs() / want.abs()).max().item() <= 1e-4
    nodes = [NodeWithScore(node=TextNode(text=t, id_=f"n{i}"), score=0.1) for i, t in enumerate(texts)]
    out = rr.postprocess_nodes(nodes, query_bundle=QueryBundle(query_str=query))
    assert [n.node.id_ for n in out] == [f"n{int(i)}" for i in torch.argsort(want, descending=True)[:3]]   # the fp32 order
    emb = HipHuggingFaceEmbedding("test/embed", device="cuda",
                                  model_kwargs={"encoder_config": EncoderConfig(**{**XLMR, "num_labels": 0}), "state_dict": W,
                                                "torch_dtype": "float32"})
    assert isinstance(emb._encoder, EncoderF32)
    e = torch.tensor(emb.get_text_embedding_batch(texts))
    seqs = [emb._tokenizer.encode(t, emb.max_length) for t in texts]
    ids, mask = _pad(seqs, cfg.pad_id)
    want_e = oe.embed(ids, mask, W, cfg_o)
    assert (e - want_e).abs().max().item() <= 2e-5
    # bf16 stays the default
    from tensor_truth_amd.encoder import Encoder
    assert isinstance(HipSentenceTransformerRerank(model="test/xenc", device="cuda",
                                                   model_kwargs={"encoder_config": cfg, "state_dict": W})._encoder, Encoder)
