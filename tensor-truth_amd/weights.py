"""Locating model weights for the HIP encoders.

Order of resolution (first hit wins):
  1. ``model_kwargs["state_dict"]``      -- tensors handed over directly (tests);
  2. a local model directory             -- ``model_kwargs["model_dir"]``, ``$TT_AMD_MODEL_DIR/<org>/<name>``
                                            or the HF hub cache, holding ``model.safetensors`` (or ``pytorch_model.bin``)
                                            (HF checkpoint names) + ``config.json``;
  3. ``model_kwargs["synthetic_seed"]``  -- seeded random init of the known architecture
                                            (benchmarks / parity tests; no network here).
Anything else raises, like the reference does when a model cannot be loaded
(``model_manager.py:265-272``: the caller wraps it in ``RuntimeError("Failed to load ...")``).
"""
from __future__ import annotations

import json
import os
from typing import Dict, Optional, Tuple

import torch

from .encoder import KNOWN_CONFIGS, EncoderConfig, synthetic_state, synthetic_state_device


WEIGHT_FILES = ("model.safetensors", "pytorch_model.bin")


def _config_from_hf(d: dict, num_labels_default: int = 0) -> EncoderConfig:
    mt = d.get("model_type", "xlm-roberta")
    arch = "bert" if mt == "bert" else "xlmr"
    archs = " ".join(d.get("architectures", []))
    num_labels = 1 if "SequenceClassification" in archs else num_labels_default
    return EncoderConfig(
        arch=arch, vocab_size=d["vocab_size"], hidden=d["hidden_size"], layers=d["num_hidden_layers"],
        heads=d["num_attention_heads"], ffn=d["intermediate_size"], max_pos=d["max_position_embeddings"],
        type_vocab=d.get("type_vocab_size", 1), pad_id=d.get("pad_token_id", 1 if arch == "xlmr" else 0),
        ln_eps=d.get("layer_norm_eps", 1e-5), num_labels=num_labels)


def head_activation(model_name: str, model_dir: Optional[str], model_kwargs: Optional[dict]) -> str:
    """What ``CrossEncoder.predict`` applies to a single-label head's logit ([UPSTREAM-K], sentence-transformers): the
    activation named in the checkpoint's config.json (``sbert_ce_default_activation_function``, or
    ``sentence_transformers.activation_fn`` in newer exports), else Sigmoid.  The BGE rerankers carry none (sigmoid
    scores in (0, 1)); the ``cross-encoder/ms-marco-*`` checkpoints name ``torch.nn.modules.linear.Identity`` and are
    scored by their raw logits.  ``model_kwargs["activation"]`` ("sigmoid" / "identity") overrides."""
    mk = model_kwargs or {}
    if mk.get("activation"):
        act = str(mk["activation"]).lower()
        if act not in ("sigmoid", "identity"):
            raise ValueError("model_kwargs['activation'] must be 'sigmoid' or 'identity'")
        return act
    name = None
    if model_dir and os.path.exists(os.path.join(model_dir, "config.json")):
        with open(os.path.join(model_dir, "config.json")) as f:
            d = json.load(f)
        name = d.get("sbert_ce_default_activation_function") or (d.get("sentence_transformers") or {}).get("activation_fn")
    elif model_name.startswith("cross-encoder/ms-marco-"):
        name = "torch.nn.modules.linear.Identity"      # synthetic weights under the real name: the real checkpoint's setting
    if name is None:
        return "sigmoid"
    tail = str(name).rsplit(".", 1)[-1].lower()
    if tail == "identity":
        return "identity"
    if tail == "sigmoid":
        return "sigmoid"
    raise ValueError(f"cross-encoder activation '{name}' is not supported (Sigmoid or Identity)")


def pooling_mode(model_dir: Optional[str]) -> str:
    """The sentence-transformers pooling a checkpoint directory declares (``1_Pooling/config.json``): "cls", "mean", ... ;
    "cls" when the directory declares nothing (the BGE models the reference defaults to are CLS + Normalize)."""
    if not model_dir:
        return "cls"
    pc = os.path.join(model_dir, "1_Pooling", "config.json")
    if not os.path.exists(pc):
        return "cls"
    with open(pc) as f:
        d = json.load(f)
    on = [k[len("pooling_mode_"):] for k, v in d.items() if k.startswith("pooling_mode_") and v is True]
    if on == ["cls_token"]:
        return "cls"
    return "+".join(sorted(on)) or "none"


def find_model_dir(model_name: str, model_kwargs: Optional[dict]) -> Optional[str]:
    mk = model_kwargs or {}
    cands = []
    if mk.get("model_dir"):
        cands.append(mk["model_dir"])
    if os.path.isdir(model_name):
        cands.append(model_name)
    root = os.environ.get("TT_AMD_MODEL_DIR")
    if root:
        cands += [os.path.join(root, model_name), os.path.join(root, model_name.split("/")[-1])]
    for c in cands:
        if any(os.path.exists(os.path.join(c, f)) for f in WEIGHT_FILES):
            return c
    try:  # HF hub cache, offline
        from huggingface_hub import try_to_load_from_cache

        for f in WEIGHT_FILES:
            p = try_to_load_from_cache(model_name, f)
            if isinstance(p, str) and os.path.exists(p):
                return os.path.dirname(p)
    except Exception:  # noqa: BLE001
        pass
    return None


def load_state(model_dir: str) -> Dict[str, torch.Tensor]:
    """``model.safetensors`` if present, else the older ``pytorch_model.bin`` (tensors only, ``weights_only=True``)."""
    st = os.path.join(model_dir, "model.safetensors")
    if os.path.exists(st):
        from safetensors.torch import load_file

        return load_file(st)
    return torch.load(os.path.join(model_dir, "pytorch_model.bin"), map_location="cpu", weights_only=True)


def resolve(model_name: str, model_kwargs: Optional[dict], device: torch.device,
            want_head: bool) -> Tuple[EncoderConfig, Dict[str, torch.Tensor], Optional[str]]:
    """-> (config, state dict, model_dir or None)."""
    mk = model_kwargs or {}
    cfg = mk.get("encoder_config") or KNOWN_CONFIGS.get(model_name)
    if "state_dict" in mk:
        if cfg is None:
            raise ValueError(f"no architecture known for '{model_name}': pass model_kwargs['encoder_config']")
        return cfg, mk["state_dict"], None
    mdir = find_model_dir(model_name, mk)
    if mdir is not None:
        with open(os.path.join(mdir, "config.json")) as f:
            cfg = _config_from_hf(json.load(f), 1 if want_head else 0)
        return cfg, load_state(mdir), mdir
    if "synthetic_seed" in mk:
        if cfg is None:
            raise ValueError(f"no architecture known for '{model_name}': pass model_kwargs['encoder_config']")
        seed = int(mk["synthetic_seed"])
        if mk.get("synthetic_on_device", cfg.layers * cfg.hidden >= 12 * 768):
            return cfg, synthetic_state_device(cfg, device, seed), None
        return cfg, synthetic_state(cfg, seed), None
    raise FileNotFoundError(
        f"weights for '{model_name}' not found: no model.safetensors under model_kwargs['model_dir'], "
        f"$TT_AMD_MODEL_DIR or the HF cache, and this environment has no network. "
        f"(Benchmarks/tests: pass model_kwargs={{'synthetic_seed': N}}.)")
