"""``CTrainer``: the historical name of the thin ctypes caller of ``ian_train_step``.  Since round 4 the C++ sequencer
(csrc/ian_trainer.cpp) is the ONLY sequencer of the training step -- single GPU and data parallel -- and
``trainer.Trainer`` is that thin caller; this module keeps the old import path."""
from __future__ import annotations

from .trainer import DISCRIM_KEYS, GEN_KEYS, METRICS, IanTrainError as CTrainerError, Trainer, TrainConfig  # noqa: F401


class CTrainer(Trainer):
    def __init__(self, config_path, params, batch, deconv_flip=True, **kw):
        Trainer.__init__(self, config_path, params, batch, deconv_flip=deconv_flip, **kw)
