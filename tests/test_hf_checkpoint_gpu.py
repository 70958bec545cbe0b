"""GPU: HF checkpoint directories (config.json + model.safetensors, as the reference's models ship) load through
``weights.resolve`` and the HIP forward agrees with the upstream implementation itself -- ``transformers``' XLM-R / BERT
forward in fp32 on the CPU, the code the reference reaches through sentence-transformers.

The checkpoints are tiny random-init models written by ``save_pretrained`` into a temp dir (no network, nothing from
/root/reference): this pins the weight-name mapping (``roberta.`` prefix, ``classifier.dense / out_proj``), config parsing,
XLM-R position ids, CLS pooling + L2 norm and the sigmoid head against upstream, not against our own oracle.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

transformers = pytest.importorskip("transformers")


def _perturb_layernorms(model, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if "LayerNorm.weight" in name:
                p.add_(0.2 * torch.randn(p.shape, generator=g))
            elif "LayerNorm.bias" in name or name.endswith(".bias"):
                p.add_(0.1 * torch.randn(p.shape, generator=g))
            elif p.dim() == 2 and "embeddings" not in name:
                p.mul_(2.5)                     # default init (std 0.02) gives nearly constant outputs: widen the spread


def _ragged(rng, n, lo, hi, vocab, bos, eos):
    seqs = []
    for _ in range(n):
        length = int(rng.integers(lo, hi))
        seqs.append([bos] + rng.integers(5, vocab, size=length - 2).tolist() + [eos])
    return seqs


def _padded(seqs, pad):
    L = max(map(len, seqs))
    ids = torch.full((len(seqs), L), pad, dtype=torch.long)
    mask = torch.zeros((len(seqs), L), dtype=torch.long)
    for i, s in enumerate(seqs):
        ids[i, : len(s)] = torch.tensor(s)
        mask[i, : len(s)] = 1
    return ids, mask


def test_xlmr_cross_encoder_checkpoint_vs_transformers(dev, built_lib, tmp_path):
    from transformers import XLMRobertaConfig, XLMRobertaForSequenceClassification

    from tensor_truth_amd.rerank import HipSentenceTransformerRerank

    torch.manual_seed(3)
    cfg = XLMRobertaConfig(vocab_size=1200, hidden_size=256, num_hidden_layers=3, num_attention_heads=4, intermediate_size=512,
                           max_position_embeddings=200, type_vocab_size=1, pad_token_id=1, bos_token_id=0, eos_token_id=2,
                           num_labels=1, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, classifier_dropout=0.0)
    model = XLMRobertaForSequenceClassification(cfg).eval()
    _perturb_layernorms(model, 1)
    rng = np.random.default_rng(0)
    seqs = _ragged(rng, 24, 6, 150, 1200, 0, 2)
    ids, mask = _padded(seqs, 1)
    with torch.no_grad():
        # a random-init head scores every pair nearly alike: standardise its logits over this batch (mean 0, std 1.5)
        # so the comparison exercises the sigmoid's whole range
        raw = model(input_ids=ids, attention_mask=mask).logits[:, 0]
        k = 1.5 / raw.std()
        model.classifier.out_proj.weight.mul_(k)
        model.classifier.out_proj.bias.mul_(k)
        model.classifier.out_proj.bias.sub_(model(input_ids=ids, attention_mask=mask).logits[:, 0].mean())
        want_logit = model(input_ids=ids, attention_mask=mask).logits[:, 0]
        want = torch.sigmoid(want_logit)
    assert 1.2 < want_logit.std().item() < 1.8 and abs(want_logit.mean().item()) < 1e-3
    model.save_pretrained(str(tmp_path / "xenc"), safe_serialization=True)
    # weights-only checkpoint: a real model directory without tokenizer files is refused (no silent hashing stand-in) ...
    with pytest.raises(FileNotFoundError):
        HipSentenceTransformerRerank(model=str(tmp_path / "xenc"), top_n=3, device="cuda")
    # ... unless the caller hands a tokenizer over explicitly (this block works on token ids)
    from tensor_truth_amd.tokenization import HashTokenizer
    rr = HipSentenceTransformerRerank(model=str(tmp_path / "xenc"), top_n=3, device="cuda",
                                      model_kwargs={"tokenizer": HashTokenizer("xlmr", 1200)})
    assert rr.config.arch == "xlmr" and rr.config.num_labels == 1 and rr.config.layers == 3 and rr.config.max_seq_len == 198
    got = rr.score_token_pairs(seqs).cpu()
    err = (got - want).abs().max().item()
    # bf16 forward vs fp32 upstream; the head was scaled up by k (tens), which scales the hidden-state rounding noise too
    assert err < 4e-2, (err, float(k))
    # order wherever upstream separates two pairs by more than twice the score bound (never skipped: rank_checks.py)
    from rank_checks import assert_order_on_separable
    n_sep = assert_order_on_separable(want.numpy(), got.numpy(), 8e-2, "HF checkpoint rerank order")
    assert n_sep >= 20, n_sep     # the standardised head spreads 24 pairs over the sigmoid: most pairs are separable
    assert torch.corrcoef(torch.stack([got, want]))[0, 1].item() > 0.995

    # ---- from strings, as SentenceTransformerRerank.postprocess_nodes / CrossEncoder.predict run it:
    # a tokenizer.json next to the weights is picked up; pairs are encoded <s> q </s></s> p </s>, truncated longest-first
    from tokenizers import Tokenizer, models, pre_tokenizers, processors

    from tensor_truth_amd.schema import NodeWithScore, QueryBundle, TextNode

    words = [f"w{i}" for i in range(1100)]
    vocab = {"<s>": 0, "<pad>": 1, "</s>": 2, "<unk>": 3, **{w: 4 + i for i, w in enumerate(words)}}
    tk = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tk.pre_tokenizer = pre_tokenizers.Whitespace()
    tk.post_processor = processors.TemplateProcessing(single="<s> $A </s>", pair="<s> $A </s> </s> $B </s>",
                                                      special_tokens=[("<s>", 0), ("</s>", 2)])
    tk.save(str(tmp_path / "xenc" / "tokenizer.json"))
    rr2 = HipSentenceTransformerRerank(model=str(tmp_path / "xenc"), top_n=4, device="cuda")
    query = " ".join(words[i] for i in rng.integers(0, 1100, size=9))
    passages = [" ".join(words[i] for i in rng.integers(0, 1100, size=int(n))) for n in rng.integers(5, 260, size=12)]
    nodes = [NodeWithScore(node=TextNode(text=p, id_=f"p{i}"), score=0.5) for i, p in enumerate(passages)]
    ranked = rr2.postprocess_nodes(nodes, QueryBundle(query_str=query))            # positional bundle, as web_search.py:155
    tk.enable_truncation(max_length=198, strategy="longest_first")                 # the model's limit (200 - 2 positions)
    encs = [tk.encode(query, p).ids for p in passages]
    assert max(map(len, encs)) == 198
    ids2, mask2 = _padded(encs, 1)
    with torch.no_grad():
        want2 = torch.sigmoid(model(input_ids=ids2, attention_mask=mask2).logits[:, 0])
    by_id = {f"p{i}": float(want2[i]) for i in range(12)}
    assert len(ranked) == 4 and all(isinstance(n.score, float) for n in ranked)
    assert [n.score for n in ranked] == sorted((n.score for n in ranked), reverse=True)
    assert max(abs(n.score - by_id[n.node.id_]) for n in ranked) < 4e-2
    best4 = sorted(by_id.values(), reverse=True)[:4]
    assert all(abs(n.score - b) < 6e-2 for n, b in zip(ranked, best4))


@pytest.mark.parametrize("arch", ["xlmr", "bert"])
def test_bi_encoder_checkpoint_vs_transformers(dev, built_lib, tmp_path, arch):
    from tensor_truth_amd.embedding import HipHuggingFaceEmbedding

    torch.manual_seed(4)
    if arch == "xlmr":
        from transformers import XLMRobertaConfig, XLMRobertaModel

        cfg = XLMRobertaConfig(vocab_size=1500, hidden_size=256, num_hidden_layers=2, num_attention_heads=4,
                               intermediate_size=1024, max_position_embeddings=260, type_vocab_size=1, pad_token_id=1,
                               hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
        model = XLMRobertaModel(cfg, add_pooling_layer=False).eval()
        bos, eos, pad = 0, 2, 1
    else:
        from transformers import BertConfig, BertModel

        cfg = BertConfig(vocab_size=1500, hidden_size=384, num_hidden_layers=2, num_attention_heads=12, intermediate_size=1536,
                         max_position_embeddings=256, type_vocab_size=2, pad_token_id=0, hidden_dropout_prob=0.0,
                         attention_probs_dropout_prob=0.0)
        model = BertModel(cfg, add_pooling_layer=True).eval()     # the pooler's weights are in the file and must be ignored
        bos, eos, pad = 101, 102, 0
    _perturb_layernorms(model, 2)
    d = tmp_path / f"bi_{arch}"
    if arch == "xlmr":
        model.save_pretrained(str(d), safe_serialization=True)
    else:                                   # the older checkpoint format: config.json + pytorch_model.bin
        d.mkdir()
        cfg.save_pretrained(str(d))
        torch.save(model.state_dict(), str(d / "pytorch_model.bin"))
    from tensor_truth_amd.tokenization import HashTokenizer
    emb = HipHuggingFaceEmbedding(str(d), device="cuda", embed_batch_size=16,
                                  model_kwargs={"tokenizer": HashTokenizer(arch, cfg.vocab_size)})   # token-id level test
    assert emb.config.arch == arch and emb.config.hidden == cfg.hidden_size and emb.config.num_labels == 0
    rng = np.random.default_rng(1)
    seqs = _ragged(rng, 40, 3, 200, 1500, bos, eos)
    got = emb.embed_token_batches(seqs).cpu()
    ids, mask = _padded(seqs, pad)
    with torch.no_grad():
        hidden = model(input_ids=ids, attention_mask=mask).last_hidden_state
    want = torch.nn.functional.normalize(hidden[:, 0], p=2, dim=1)          # Pooling(cls) + Normalize
    cos = (got * want).sum(dim=1)
    assert cos.min().item() >= 0.999 and (got - want).abs().max().item() <= 1e-2, (cos.min().item(), (got - want).abs().max().item())
    assert (got.norm(dim=1) - 1).abs().max() < 1e-3

    # ---- a checkpoint that declares MEAN pooling (e5 / all-MiniLM / gte style: 1_Pooling/config.json): Pooling(mean) + Normalize,
    #      in the default precision and in the reference precision (split-bf16 for the 256-wide XLM-R shape, fp32 MFMA for BERT-384)
    import json

    assert emb.pooling == "cls"
    (d / "1_Pooling").mkdir()
    (d / "1_Pooling" / "config.json").write_text(json.dumps({
        "word_embedding_dimension": cfg.hidden_size, "pooling_mode_cls_token": False, "pooling_mode_mean_tokens": True,
        "pooling_mode_max_tokens": False, "pooling_mode_mean_sqrt_len_tokens": False}))
    m = mask.unsqueeze(-1).float()
    want_mean = torch.nn.functional.normalize((hidden * m).sum(1) / m.sum(1), p=2, dim=1)
    assert (want_mean - want).abs().max().item() > 0.05                      # the two poolings differ on this checkpoint
    for precision, cos_min, err_max in (("bf16", 0.999, 1e-2), ("reference", 0.999999, 2e-4)):
        emb_m = HipHuggingFaceEmbedding(str(d), device="cuda", embed_batch_size=16,
                                        model_kwargs={"tokenizer": HashTokenizer(arch, cfg.vocab_size), "precision": precision})
        assert emb_m.pooling == "mean"
        got_m = emb_m.embed_token_batches(seqs).cpu()
        cos_m = (got_m * want_mean).sum(dim=1)
        assert cos_m.min().item() >= cos_min and (got_m - want_mean).abs().max().item() <= err_max, \
            (precision, cos_m.min().item(), (got_m - want_mean).abs().max().item())
        assert (got_m.norm(dim=1) - 1).abs().max() < 1e-3
    # anything else (max pooling ...) is refused, not approximated
    (d / "1_Pooling" / "config.json").write_text(json.dumps({"pooling_mode_cls_token": False, "pooling_mode_max_tokens": True}))
    with pytest.raises(NotImplementedError):
        HipHuggingFaceEmbedding(str(d), device="cuda", model_kwargs={"tokenizer": HashTokenizer(arch, cfg.vocab_size)})


def test_bert_cross_encoder_checkpoint_vs_transformers(dev, built_lib, tmp_path):
    """The third reranker the reference offers out of the box (``cross-encoder/ms-marco-MiniLM-L-6-v2``,
    app_utils/config_schema.py:83-87) is a BertForSequenceClassification: pooler.dense -> tanh -> classifier on [CLS], token
    type 1 on the passage segment, and a config.json that tells CrossEncoder to return the raw logit (Identity).  A tiny
    random-init checkpoint of that class, upstream forward in fp32 on the CPU as the reference."""
    import json

    from transformers import BertConfig, BertForSequenceClassification

    from tensor_truth_amd.rerank import HipSentenceTransformerRerank
    from tensor_truth_amd.tokenization import HashTokenizer

    torch.manual_seed(5)
    cfg = BertConfig(vocab_size=1500, hidden_size=384, num_hidden_layers=3, num_attention_heads=12, intermediate_size=1536,
                     max_position_embeddings=160, type_vocab_size=2, pad_token_id=0, num_labels=1,
                     hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, classifier_dropout=0.0)
    model = BertForSequenceClassification(cfg).eval()
    _perturb_layernorms(model, 2)
    with torch.no_grad():
        model.bert.embeddings.token_type_embeddings.weight.mul_(4.0)       # make a wrong / missing segment id visible
    rng = np.random.default_rng(1)
    seqs, types = [], []
    for _ in range(24):
        na, nb = int(rng.integers(3, 20)), int(rng.integers(4, 120))
        a, b = rng.integers(5, 1500, size=na).tolist(), rng.integers(5, 1500, size=nb).tolist()
        seqs.append([101 % 1500] + a + [102 % 1500] + b + [102 % 1500])
        types.append([0] * (na + 2) + [1] * (nb + 1))
    ids, mask = _padded(seqs, 0)
    tt, _ = _padded(types, 0)
    with torch.no_grad():
        raw = model(input_ids=ids, attention_mask=mask, token_type_ids=tt).logits[:, 0]
        k = 1.5 / raw.std()
        model.classifier.weight.mul_(k)
        model.classifier.bias.mul_(k)
        model.classifier.bias.sub_(model(input_ids=ids, attention_mask=mask, token_type_ids=tt).logits[:, 0].mean())
        want_logit = model(input_ids=ids, attention_mask=mask, token_type_ids=tt).logits[:, 0]
        no_types = model(input_ids=ids, attention_mask=mask).logits[:, 0]
    assert (want_logit - no_types).abs().max().item() > 0.3          # the segment ids matter in this checkpoint
    mdir = tmp_path / "minilm"
    model.save_pretrained(str(mdir), safe_serialization=True)
    tk = HashTokenizer("bert", 1500)
    # (a) no activation named in config.json: CrossEncoder's default for one label, sigmoid
    rr = HipSentenceTransformerRerank(model=str(mdir), top_n=3, device="cuda", model_kwargs={"tokenizer": tk})
    assert rr.config.arch == "bert" and rr.config.num_labels == 1 and rr.config.type_vocab == 2 and rr.activation == "sigmoid"
    with pytest.raises(ValueError):
        rr.score_token_pairs(seqs)                                   # a BERT cross-encoder without segment ids: refused
    got = rr.score_token_pairs(seqs, types).cpu()
    # bf16 forward: the standardised head (weights x k, a bias that cancels a mean logit of several units) turns the 2^-9
    # rounding of its weights into a common shift of every logit by ~0.1-0.2; the fp32-grade mode below pins the mapping
    # itself at 2e-3, here the bound is what that shift allows
    assert (got - torch.sigmoid(want_logit)).abs().max().item() < 8e-2
    assert torch.corrcoef(torch.stack([got, torch.sigmoid(want_logit)]))[0, 1].item() > 0.995
    # (b) the ms-marco checkpoints' config.json: Identity -> scores are the raw logits
    conf = json.loads((mdir / "config.json").read_text())
    conf["sbert_ce_default_activation_function"] = "torch.nn.modules.linear.Identity"
    (mdir / "config.json").write_text(json.dumps(conf))
    rr_id = HipSentenceTransformerRerank(model=str(mdir), top_n=3, device="cuda", model_kwargs={"tokenizer": tk})
    assert rr_id.activation == "identity"
    got_l = rr_id.score_token_pairs(seqs, types).cpu()
    d = got_l - want_logit
    assert abs(d.mean().item()) < 0.35 and (d - d.mean()).abs().max().item() < 0.15, (d, float(k))   # common shift / per-pair noise
    assert got_l.min().item() < -0.5 and got_l.max().item() > 1.0    # raw logits, not probabilities
    from rank_checks import assert_order_on_separable
    assert assert_order_on_separable(want_logit.numpy(), got_l.numpy(), 0.3, "BERT cross-encoder order") >= 20
    # the reference precision (TT_PRECISION=reference / torch_dtype=float32) on the same checkpoint
    rr_ref = HipSentenceTransformerRerank(model=str(mdir), top_n=3, device="cuda",
                                          model_kwargs={"tokenizer": tk, "precision": "reference"})
    got_ref = rr_ref.score_token_pairs(seqs, types).cpu()
    assert (got_ref - want_logit).abs().max().item() < 2e-3 * max(1.0, want_logit.abs().max().item())
    # round 6: this 384-wide, 32-wide-head family runs the reference precision on the matrix cores (f16x3), not
    # on the fp32 MFMA; probabilities within north_star's 1e-3 relative of the upstream fp32 forward
    from tensor_truth_amd import precision as _prec
    from tensor_truth_amd.encoder_x3 import EncoderX3

    assert _prec.reference_impl(rr_ref.config) == "f16x3" and isinstance(rr_ref._encoder, EncoderX3) and "split-fp16" in rr_ref.precision
    assert (got_ref - want_logit).abs().max().item() < 2e-4 * max(1.0, want_logit.abs().max().item())
    rr_sig = HipSentenceTransformerRerank(model=str(mdir), top_n=3, device="cuda",
                                          model_kwargs={"tokenizer": tk, "precision": "reference", "activation": "sigmoid"})
    if rr_sig.activation == "sigmoid":
        p_got, p_want = rr_sig.score_token_pairs(seqs, types).cpu(), torch.sigmoid(want_logit)
        assert ((p_got - p_want).abs() / p_want).max().item() <= 1e-3

    # ---- from strings through the postprocessor surface: [CLS] q [SEP] p [SEP], segment 1 on the passage, longest-first
    from tokenizers import Tokenizer, models, pre_tokenizers, processors

    from tensor_truth_amd.schema import NodeWithScore, QueryBundle, TextNode

    words = [f"w{i}" for i in range(1300)]
    vocab = {"[PAD]": 0, "[UNK]": 1, "[CLS]": 2, "[SEP]": 3, **{w: 5 + i for i, w in enumerate(words)}}
    tkj = Tokenizer(models.WordLevel(vocab, unk_token="[UNK]"))
    tkj.pre_tokenizer = pre_tokenizers.Whitespace()
    tkj.post_processor = processors.TemplateProcessing(single="[CLS] $A [SEP]", pair="[CLS] $A [SEP] $B:1 [SEP]:1",
                                                       special_tokens=[("[CLS]", 2), ("[SEP]", 3)])
    tkj.save(str(mdir / "tokenizer.json"))
    rr2 = HipSentenceTransformerRerank(model=str(mdir), top_n=4, device="cuda")
    query = " ".join(words[i] for i in rng.integers(0, 1300, size=7))
    passages = [" ".join(words[i] for i in rng.integers(0, 1300, size=int(n))) for n in rng.integers(5, 220, size=12)]
    nodes = [NodeWithScore(node=TextNode(text=p, id_=f"p{i}"), score=0.5) for i, p in enumerate(passages)]
    ranked = rr2.postprocess_nodes(nodes, query_bundle=QueryBundle(query_str=query))
    tkj.enable_truncation(max_length=160, strategy="longest_first")
    encs = [tkj.encode(query, p) for p in passages]
    assert max(len(e.ids) for e in encs) == 160
    ids2, mask2 = _padded([e.ids for e in encs], 0)
    tt2, _ = _padded([e.type_ids for e in encs], 0)
    with torch.no_grad():
        want2 = model(input_ids=ids2, attention_mask=mask2, token_type_ids=tt2).logits[:, 0]
    by_id = {f"p{i}": float(want2[i]) for i in range(12)}
    assert len(ranked) == 4 and all(isinstance(n.score, float) for n in ranked)
    assert [n.score for n in ranked] == sorted((n.score for n in ranked), reverse=True)
    assert max(abs(n.score - by_id[n.node.id_]) for n in ranked) < 0.4          # (common shift of the bf16 head, see above)
    top4 = sorted(by_id, key=by_id.get, reverse=True)[:4]
    gaps_ok = sorted(by_id.values(), reverse=True)[3] - sorted(by_id.values(), reverse=True)[4] > 0.3
    assert not gaps_ok or [n.node.id_ for n in ranked][:4] == top4 or set(n.node.id_ for n in ranked) == set(top4)


def test_xlmr_base_reranker_shape_vs_oracle(dev, built_lib):
    """``BAAI/bge-reranker-base`` (the second out-of-the-box reranker): XLM-R base -- 768 wide, 12 heads, 3072 FFN; here at
    3 of its 12 layers with seeded synthetic weights against the fp32 CPU oracle (the 768-wide GEMM / LayerNorm / head
    launches are what this pins; depth is covered by the v2-m3 tests)."""
    from dataclasses import replace

    from oracle import encoder as oe
    from tensor_truth_amd.encoder import BGE_RERANKER_BASE, KNOWN_CONFIGS
    from tensor_truth_amd.rerank import HipSentenceTransformerRerank
    from tensor_truth_amd.tokenization import HashTokenizer

    assert KNOWN_CONFIGS["BAAI/bge-reranker-base"] is BGE_RERANKER_BASE
    cfg = replace(BGE_RERANKER_BASE, layers=3, vocab_size=4000)
    ocfg = oe.EncoderConfig(**cfg.__dict__)
    W = oe.synth_weights(ocfg, seed=21)
    rng = np.random.default_rng(4)
    seqs = _ragged(rng, 20, 8, 300, 4000, 0, 2)
    ids, mask = _padded(seqs, 1)
    with torch.no_grad():
        want = oe.rerank_scores(ids, mask, W, ocfg)
    for precision, bound in (("bf16", 2e-2), ("reference", 1e-4)):
        rr = HipSentenceTransformerRerank(model="BAAI/bge-reranker-base", top_n=3, device="cuda",
                                          model_kwargs={"encoder_config": cfg, "state_dict": W, "precision": precision,
                                                        "tokenizer": HashTokenizer("xlmr", 4000)})
        assert rr.activation == "sigmoid" and rr.precision.startswith(precision)
        assert precision != "reference" or "split-fp16" in rr.precision          # 768 = 12 heads x 64: the split-plane kernels take it
        got = rr.score_token_pairs(seqs).cpu()
        assert (got - want).abs().max().item() < bound, precision
