"""Model bundles: ``model.pth`` + ``config.yaml`` + ``model_meta.json`` as the reference's TrainingThread writes them
(backend/workthreads.py:856-886) and its ClassificationThread reads them back (``_load_model``, :372-451).

``load_model_bundle`` returns the MI355X head (not a torch module) and the bundle's metadata, with the reference's
rules: architecture and hyper-parameters come from ``model_meta.json``; ``behaviors`` / ``seq_len`` fall back to
``config.yaml``; ``lstm_hidden_size`` and ``lstm_layers`` missing from the metadata are inferred from the weights
(:415-425); an encoder mismatch between the project and the bundle refuses the load (:390-399); the weights are loaded
non-strictly (:441) - unexpected entries are ignored and the two scalar parameters may be absent (they then keep the
constructor's initial values, classifier_head.py:90,96).  Any other missing tensor is an error here: the reference would
keep torch's random initialisation for it, which no caller can have wanted.
"""
from __future__ import annotations

import json
import os
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from .head import ClassifierLSTMDeltas

LEGACY_ARCH = "ClassifierLegacyLSTM"


def _read_yaml(path: str) -> dict:
    import yaml
    with open(path, "r") as f:
        return yaml.safe_load(f) or {}


def load_model_bundle(model_dir: str, device="cuda", project_encoder: Optional[str] = None,
                      in_features: int = 768) -> Tuple[Optional[ClassifierLSTMDeltas], Optional[dict]]:
    """-> (head, meta) or (None, None) when the bundle must not be used (encoder mismatch, legacy architecture),
    like ``ClassificationThread._load_model``.  ``meta["hyperparameters"]`` is completed as the reference does and
    ``meta["calibration"]["temperature"]`` is what ``infer_file`` should be given (workthreads.py:484)."""
    cfg_path = os.path.join(model_dir, "config.yaml")
    config = _read_yaml(cfg_path) if os.path.exists(cfg_path) else {}
    meta_path = os.path.join(model_dir, "model_meta.json")
    if not os.path.exists(meta_path):          # workthreads.py:381-388: a bundle without metadata is a v2 (legacy) model
        meta = {"head_architecture_version": LEGACY_ARCH, "hyperparameters": dict(config),
                "encoder_model_identifier": project_encoder}
    else:
        with open(meta_path, "r") as f:
            meta = json.load(f)
    model_encoder = meta.get("encoder_model_identifier")
    if project_encoder and model_encoder and model_encoder != project_encoder:
        print(f"Encoder mismatch! Project is for '{project_encoder}', but model was trained with '{model_encoder}'. "
              "Please re-encode videos.")
        return None, None
    arch = meta.get("head_architecture_version", LEGACY_ARCH)
    hp = dict(meta.get("hyperparameters", {}))
    hp.setdefault("behaviors", config.get("behaviors", []))
    hp.setdefault("seq_len", config.get("seq_len", 31))
    meta["hyperparameters"] = hp
    if not arch.startswith("ClassifierLSTMDeltas"):
        # the v2 head returns a 3-tuple that infer_file cannot unpack (cbas.py:544); the reference's README calls such
        # models unusable in v3 - refuse instead of failing later
        print(f"Model bundle {model_dir!r} has architecture {arch!r}; only ClassifierLSTMDeltas runs on the v3 path.")
        return None, None
    weights_path = os.path.join(model_dir, "model.pth")
    try:
        sd = torch.load(weights_path, map_location="cpu", weights_only=True)
    except TypeError:
        sd = torch.load(weights_path, map_location="cpu")
    w: Dict[str, np.ndarray] = {k: v.detach().to(torch.float32).numpy() for k, v in sd.items()}
    if "lstm_hidden_size" not in hp:
        ref = w.get("attention_head.weight", w.get("lin2.weight"))
        hp["lstm_hidden_size"] = int(ref.shape[1] // 2) if ref is not None else 64
    if "lstm_layers" not in hp:
        keys = [int(k.split("weight_ih_l")[1].split("_")[0]) for k in w if "lstm.weight_ih_l" in k]
        hp["lstm_layers"] = max(keys) + 1 if keys else 1
    use_acc = bool(hp.get("use_acceleration", "acc_bottleneck.0.weight" in w))
    head = ClassifierLSTMDeltas(in_features=in_features, out_features=len(hp["behaviors"]), seq_len=int(hp["seq_len"]),
                                lstm_hidden_size=int(hp["lstm_hidden_size"]), lstm_layers=int(hp["lstm_layers"]),
                                use_acceleration=use_acc)
    head.load_state_dict(w, strict=False)
    head.to(device).eval()
    return head, meta


def save_model_bundle(model_dir: str, head: ClassifierLSTMDeltas, behaviors: List[str], name: str,
                      encoder_model_identifier: Optional[str], temperature: float = 1.0,
                      training_run_info: Optional[dict] = None, cbas_commit_hash: str = "unknown") -> None:
    """Write the three files of a bundle with the keys the reference writes (workthreads.py:856-886)."""
    import yaml
    os.makedirs(model_dir, exist_ok=True)
    torch.save(head.state_dict(), os.path.join(model_dir, "model.pth"))
    cfg = head.config
    with open(os.path.join(model_dir, "config.yaml"), "w") as f:
        yaml.dump({"name": name, "behaviors": list(behaviors), "seq_len": int(cfg.seq_len),
                   "architecture": "ClassifierLSTMDeltas"}, f, allow_unicode=True)
    meta = {
        "model_bundle_schema": "1.0",
        "cbas_commit_hash": cbas_commit_hash,
        "encoder_model_identifier": encoder_model_identifier,
        "head_architecture_version": "ClassifierLSTMDeltas",
        "hyperparameters": {"behaviors": list(behaviors), "seq_len": int(cfg.seq_len),
                            "use_acceleration": bool(cfg.use_acceleration),
                            "lstm_hidden_size": int(cfg.lstm_hidden_size), "lstm_layers": int(cfg.lstm_layers)},
        "training_run_info": dict(training_run_info or {}),
        "calibration": {"temperature": float(temperature)},
    }
    with open(os.path.join(model_dir, "model_meta.json"), "w") as f:
        json.dump(meta, f, indent=4)
