"""Monkey-patch a running CBAS backend so its unmodified worker threads use the MI355X path.

    import cbas_amd.integration; cbas_amd.integration.install()

Replaces ``cbas.DinoEncoder`` / ``cbas.encode_file`` / ``cbas.infer_file`` (reference
backend/cbas.py:399-572, 650-677) and ``classifier_head.ClassifierLSTMDeltas``
(backend/classifier_head.py:57-172) with the drop-ins of this package.  See INTEGRATION.md.
"""
from __future__ import annotations

import importlib
import os


def install(strict: bool = True) -> bool:
    """Returns True when the CBAS modules were found and patched.  ``strict=False`` makes a missing
    CBAS backend a no-op instead of an ImportError."""
    try:
        cbas = importlib.import_module("cbas")
        classifier_head = importlib.import_module("classifier_head")
    except ImportError:
        if strict:
            raise
        return False
    from .encoder import DinoEncoder
    from .head import ClassifierLSTMDeltas
    from .pipeline import encode_file, infer_file
    from .train import train_lstm_model

    cbas._reference_DinoEncoder = getattr(cbas, "DinoEncoder", None)
    cbas._reference_encode_file = getattr(cbas, "encode_file", None)
    cbas._reference_infer_file = getattr(cbas, "infer_file", None)
    cbas._reference_train_lstm_model = getattr(cbas, "train_lstm_model", None)
    classifier_head._reference_ClassifierLSTMDeltas = getattr(classifier_head, "ClassifierLSTMDeltas", None)
    cbas.DinoEncoder = DinoEncoder
    cbas.encode_file = encode_file
    cbas.infer_file = infer_file
    cbas.train_lstm_model = train_lstm_model          # TrainingThread, workthreads.py:635
    classifier_head.ClassifierLSTMDeltas = ClassifierLSTMDeltas
    return True


def uninstall() -> None:
    cbas = importlib.import_module("cbas")
    classifier_head = importlib.import_module("classifier_head")
    for mod, names in ((cbas, ("DinoEncoder", "encode_file", "infer_file", "train_lstm_model")),
                       (classifier_head, ("ClassifierLSTMDeltas",))):
        for n in names:
            ref = getattr(mod, "_reference_" + n, None)
            if ref is not None:
                setattr(mod, n, ref)


if os.environ.get("CBAS_USE_MI355X") == "1":  # pragma: no cover
    install(strict=False)
