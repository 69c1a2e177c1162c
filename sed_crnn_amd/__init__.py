"""sed_crnn_amd — MI355X-native SEDnet hot path (hand-written HIP behind a flat C ABI).

(The task names the package ``sed-crnn_amd``; a hyphen is not importable, so the directory is
``sed_crnn_amd``.)  Public surface mirrors the reference: TimePooledCRNN / run_epoch (sed.py),
FocalBCELoss / CRNNLightning (crnn_lightning.py), metrics.compute_scores (metrics.py), get_model
(README.md:44).  Nothing here falls back to torch.nn compute or to the CPU oracle.
"""
from . import metrics  # noqa: F401
from ._lib import LIB_PATH, SedHipError, lib  # noqa: F401
from .fit import EpochTally, fit, fit_folds, run_epoch, run_epoch_device  # noqa: F401
from .lightning import CRNNLightning, fit_lightning  # noqa: F401
from .losses import BCEWithLogitsLoss, FocalBCELoss  # noqa: F401
from .model import (HipCRNN, LightningTimePooledCRNN, SEDNet, TimePooledCRNN,  # noqa: F401
                    get_model)
from .optim import FusedAdam, clip_grad_norm_  # noqa: F401

__version__ = "0.1.0"
