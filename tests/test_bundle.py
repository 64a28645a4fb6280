"""Model bundles as the reference writes and reads them (backend/workthreads.py:856-886 write, :372-451 read):
host-side rules checked on CPU; the GPU round trip is in tests/test_gpu_round2.py."""
import json
import os

import numpy as np
import pytest
import torch

from cbas_amd import bundle as B, config as C, weights as W
from cbas_amd.head import ClassifierLSTMDeltas

NAMES = ["eating", "drinking", "rearing", "climbing", "digging", "nesting", "resting", "grooming", "exploring"]
ENC = "facebook/dinov3-vitb16-pretrain-lvd1689m"


def _write_reference_style(d, hcfg, w, meta_hparams, temperature=1.37, encoder=ENC):
    """The three files exactly as TrainingThread writes them (torch.save of the state dict, yaml, json)."""
    import yaml
    os.makedirs(d, exist_ok=True)
    torch.save({k: torch.from_numpy(np.asarray(v).copy()) for k, v in w.items()}, os.path.join(d, "model.pth"))
    with open(os.path.join(d, "config.yaml"), "w") as f:
        yaml.dump({"name": "m1", "behaviors": NAMES, "seq_len": hcfg.seq_len, "architecture": "ClassifierLSTMDeltas"}, f)
    meta = {"model_bundle_schema": "1.0", "cbas_commit_hash": "abc", "encoder_model_identifier": encoder,
            "head_architecture_version": "ClassifierLSTMDeltas", "hyperparameters": meta_hparams,
            "training_run_info": {"num_runs": 1, "optimization_target": "f1"}, "calibration": {"temperature": temperature}}
    with open(os.path.join(d, "model_meta.json"), "w") as f:
        json.dump(meta, f, indent=4)


def test_load_infers_missing_hyperparameters_from_the_weights(tmp_path):
    hcfg = C.HeadConfig(lstm_hidden_size=96, lstm_layers=2, seq_len=63)
    w = W.synth_head_weights(hcfg, 3)
    w["some_future_buffer"] = np.zeros(3, np.float32)                    # strict=False: unexpected entries are ignored
    del w["gate"]                                                        # ... and scalars may be absent (keep init 0.2)
    _write_reference_style(str(tmp_path / "m1"), hcfg, w, {"behaviors": NAMES, "seq_len": 63})
    head, meta = B.load_model_bundle(str(tmp_path / "m1"), device="cpu", project_encoder=ENC)
    assert isinstance(head, ClassifierLSTMDeltas)
    assert head.config.lstm_hidden_size == 96 and head.config.lstm_layers == 2 and head.seq_len == 63
    assert head.out_features == 9 and head.config.use_acceleration
    assert meta["hyperparameters"]["lstm_hidden_size"] == 96 and meta["calibration"]["temperature"] == 1.37
    sd = head.state_dict()
    assert float(sd["gate"]) == pytest.approx(0.2) and "some_future_buffer" not in sd
    np.testing.assert_array_equal(sd["lin1.weight"].numpy(), w["lin1.weight"])


def test_load_refuses_encoder_mismatch_legacy_and_missing_tensors(tmp_path):
    hcfg = C.HeadConfig()
    w = W.synth_head_weights(hcfg, 3)
    _write_reference_style(str(tmp_path / "a"), hcfg, w, {"behaviors": NAMES, "seq_len": 31}, encoder="facebook/dinov2-with-registers-base")
    assert B.load_model_bundle(str(tmp_path / "a"), device="cpu", project_encoder=ENC) == (None, None)
    # no model_meta.json = a v2 (legacy) bundle: not runnable on the v3 path
    _write_reference_style(str(tmp_path / "b"), hcfg, w, {"behaviors": NAMES, "seq_len": 31})
    os.remove(str(tmp_path / "b" / "model_meta.json"))
    assert B.load_model_bundle(str(tmp_path / "b"), device="cpu", project_encoder=ENC) == (None, None)
    # a missing tensor is an error (torch would keep its random initialisation)
    w2 = dict(w)
    del w2["lin2.weight"]
    _write_reference_style(str(tmp_path / "c"), hcfg, w2, {"behaviors": NAMES, "seq_len": 31})
    with pytest.raises(RuntimeError, match="lin2.weight"):
        B.load_model_bundle(str(tmp_path / "c"), device="cpu", project_encoder=ENC)


def test_save_then_load_round_trip_and_no_acceleration(tmp_path):
    hcfg = C.HeadConfig(use_acceleration=False, lstm_hidden_size=48)
    head = ClassifierLSTMDeltas(768, 9, use_acceleration=False, lstm_hidden_size=48)
    head.load_state_dict(W.synth_head_weights(hcfg, 5))
    B.save_model_bundle(str(tmp_path / "m"), head, NAMES, "m", ENC, temperature=0.8, training_run_info={"num_runs": 2})
    meta = json.load(open(str(tmp_path / "m" / "model_meta.json")))
    assert set(meta) == {"model_bundle_schema", "cbas_commit_hash", "encoder_model_identifier", "head_architecture_version",
                         "hyperparameters", "training_run_info", "calibration"}                   # workthreads.py:867-883
    assert meta["hyperparameters"]["use_acceleration"] is False
    h2, m2 = B.load_model_bundle(str(tmp_path / "m"), device="cpu", project_encoder=ENC)
    assert not h2.config.use_acceleration and h2.config.lstm_hidden_size == 48
    for k, v in head.state_dict().items():
        assert torch.equal(v, h2.state_dict()[k]), k
    # the reference's own loader accepts what we wrote: same keys / shapes as its module's state dict
    ref_keys = set(W.head_param_shapes(hcfg))
    assert set(torch.load(str(tmp_path / "m" / "model.pth"), weights_only=True)) == ref_keys
