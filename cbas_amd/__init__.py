"""cbas_amd: MI355X-native streamed frame encoder + behaviour classifier for CBAS.

Drop-in for the one hot path of jones-lab-tamu/CBAS (``encode_file`` / ``DinoEncoder`` /
``infer_file`` / ``ClassifierLSTMDeltas``: reference backend/cbas.py:399-572,650-677 and
backend/classifier_head.py:57-172).  The arithmetic lives in hand-written HIP kernels behind the
C-ABI library ``libcbas_mi355x.so`` (see ``include/cbas_mi355x.h``); this package is the host-side
mirror of the reference's Python interface.  There is no CPU fallback: using the encoder or the
head without the built library raises.
"""
from .config import ViTConfig, HeadConfig, VIT_S16, VIT_B16, VIT_L16, VIT_TINY, NAMED_VIT

__all__ = ["ViTConfig", "HeadConfig", "VIT_S16", "VIT_B16", "VIT_L16", "VIT_TINY", "NAMED_VIT"]
__version__ = "0.1.0"
