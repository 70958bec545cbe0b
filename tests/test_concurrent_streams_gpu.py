"""GPU: the encoder and the scan stay bit-reproducible when they run BESIDE each other on two streams.

Regression test for the round-3 contention finding (profiles/r03_contention_race.log, r03_pk_mfma_hazard.log): a packed-f32
instruction form the compiler had chosen for embed_ln_kernel returned wrong lanes whenever the scan's MFMA loop shared its SIMDs,
so forwards differed by ~5e-3 as soon as the retrievers got their own stream.  rowops.hip is now built without that form
(tests/test_lib_abi.py checks the disassembly); this is the behavioural half: same bits loaded as idle, in both directions."""
import threading
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    torch.cuda.set_device(0)
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def corpus(dev):
    g = torch.Generator(device=dev).manual_seed(77)
    c = torch.nn.functional.normalize(torch.randn(1_000_000, 1024, device=dev, generator=g), dim=1).to(torch.bfloat16)
    q = torch.nn.functional.normalize(torch.randn(32, 1024, device=dev, generator=g), dim=1).to(torch.bfloat16)
    return c, q


class _Neighbour:
    """Runs ``fn`` in a loop on its own stream from a second thread until stopped."""

    def __init__(self, dev, fn):
        self.dev, self.fn, self.stop, self.count, self.error = dev, fn, False, 0, None
        self.stream = torch.cuda.Stream(device=dev)
        self.thread = threading.Thread(target=self._run)

    def _run(self):
        torch.cuda.set_device(self.dev)
        try:
            with torch.cuda.stream(self.stream):
                while not self.stop:
                    self.fn()
                    self.stream.synchronize()
                    self.count += 1
        except Exception as e:  # noqa: BLE001
            self.error = e

    def __enter__(self):
        self.thread.start()
        while self.count < 2 and self.error is None and self.thread.is_alive():
            time.sleep(0.01)
        return self

    def __exit__(self, *exc):
        self.stop = True
        self.thread.join()
        assert self.error is None, f"neighbour loop failed: {self.error!r}"
        assert self.count >= 2


def _encoder(dev, dtype, layers=2):
    from tensor_truth_amd.encoder import BGE_RERANKER_V2_M3, Encoder, EncoderConfig, EncoderWeights, pack_token_matrix, synthetic_state_device

    cfg = EncoderConfig(**{**BGE_RERANKER_V2_M3.__dict__, "layers": layers})
    enc = Encoder(EncoderWeights(cfg, synthetic_state_device(cfg, dev, seed=2), dev, dtype=dtype))
    rng = np.random.default_rng(3)
    pairs = rng.integers(4, cfg.vocab_size, size=(350, 292), dtype=np.int32)
    pairs[:, 0] = 0
    pairs[:, -1] = 2
    return enc, pack_token_matrix(pairs, cfg)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_encoder_bits_unchanged_beside_a_streaming_scan(dev, corpus, dtype):
    from tensor_truth_amd import scan as tscan

    c, q = corpus
    enc, batch = _encoder(dev, dtype)
    idle_hidden = enc.forward_packed(batch)[0].clone()
    idle_scores = enc.rerank_packed(batch).clone()
    torch.cuda.synchronize()
    with _Neighbour(dev, lambda: tscan.scan_topk(c, q, 50, check_overflow=False)) as nb:
        first = nb.count
        for _ in range(12):
            assert torch.equal(enc.forward_packed(batch)[0], idle_hidden)
            assert torch.equal(enc.rerank_packed(batch), idle_scores)
        torch.cuda.synchronize()
        assert nb.count > first, "the scan loop did not run beside the encoder"


def test_embedding_layer_alone_beside_a_streaming_scan(dev, corpus):
    """layers = 0: the embedding gather + LayerNorm kernel by itself -- the kernel the finding was in."""
    from tensor_truth_amd import scan as tscan

    c, q = corpus
    enc, batch = _encoder(dev, torch.bfloat16, layers=0)
    idle = enc.forward_packed(batch)[0].clone()
    torch.cuda.synchronize()
    with _Neighbour(dev, lambda: tscan.scan_topk(c, q, 50, check_overflow=False)):
        for _ in range(24):
            assert torch.equal(enc.forward_packed(batch)[0], idle)
        torch.cuda.synchronize()


@pytest.mark.parametrize("n_queries", [1, 32, 256])
def test_scan_bits_unchanged_beside_an_encoder_forward(dev, corpus, n_queries):
    from tensor_truth_amd import scan as tscan

    c, q = corpus
    qq = q[:n_queries] if n_queries <= q.shape[0] else q.repeat(n_queries // q.shape[0], 1).roll(1, dims=1).contiguous()
    enc, batch = _encoder(dev, torch.bfloat16)
    s0, i0 = tscan.scan_topk(c, qq, 50)
    s0, i0 = s0.clone(), i0.clone()
    torch.cuda.synchronize()
    with _Neighbour(dev, lambda: enc.rerank_packed(batch)):
        for _ in range(12):
            s, i = tscan.scan_topk(c, qq, 50)
            assert torch.equal(i, i0) and torch.equal(s, s0)
        torch.cuda.synchronize()
