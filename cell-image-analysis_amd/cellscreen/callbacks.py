"""The three callbacks of the reference's fit() call as one host-side state machine
(CAE_improved_modeltrain.py:263-283):

    EarlyStopping(monitor='val_loss', patience=10, restore_best_weights=True)       :264-269
    ModelCheckpoint(best_autoencoder.keras, monitor='val_loss', save_best_only=True) :270-275
    ReduceLROnPlateau(monitor='val_loss', factor=0.5, patience=5, min_lr=1e-6)       :276-282

Everything not passed there is a Keras default, restated from the published Keras sources
(keras/src/callbacks/{early_stopping,model_checkpoint,reduce_lr_on_plateau}.py; Keras is not installed here,
so this is unpinned by execution -- tests/test_callbacks_cpu.py pins the rules below with scripted val_loss runs):

* EarlyStopping: min_delta = 0, mode auto -> min: improvement iff `val < best` (the first finite-or-not value
  always counts: best starts as None).  `wait` is incremented BEFORE the test each epoch and reset to 0 on
  improvement; training stops at the end of the epoch where `wait >= patience` (and epoch > 0).  Keras 3 restores
  the best weights in on_train_end whether or not the stop fired (`keras_version=3`, the version that writes the
  `.keras` files the reference saves); Keras 2 restored them only when the stop fired (`keras_version=2`).
* ModelCheckpoint(save_best_only): saves iff `val < best`, best starting at +inf (a NaN never saves).
* ReduceLROnPlateau: min_delta = 1e-4, cooldown = 0, mode auto -> min: improvement iff `val < best - 1e-4`, best
  starting at +inf; otherwise `wait += 1` and, once `wait >= patience`, `lr <- max(lr * factor, min_lr)` ONLY IF
  `lr > min_lr` -- and `wait` is reset only when a reduction actually happened.  The learning rate lives in the
  optimizer as a float32 variable; the callback reads it back as a Python float, so every value here is the float32
  rounding of what was assigned.

Callbacks run in list order at the end of each epoch: EarlyStopping, ModelCheckpoint, ReduceLROnPlateau; a stop
request does not skip the later callbacks of that epoch.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import spec


def _f32(x: float) -> float:
    return float(np.float32(x))


@dataclass
class EpochActions:
    """What the training loop must do at the end of an epoch."""
    snapshot_best_weights: bool = False      # EarlyStopping took model.get_weights()
    save_checkpoint: bool = False            # ModelCheckpoint writes best_autoencoder.keras
    lr: float = spec.ADAM_LR                 # learning rate for the NEXT epoch
    lr_reduced: bool = False
    stop_training: bool = False


@dataclass
class FitCallbacks:
    lr: float = spec.ADAM_LR
    es_patience: int = spec.ES_PATIENCE
    es_min_delta: float = 0.0
    rl_factor: float = spec.RLROP_FACTOR
    rl_patience: int = spec.RLROP_PATIENCE
    rl_min_lr: float = spec.RLROP_MIN_LR
    rl_min_delta: float = spec.RLROP_MIN_DELTA
    rl_cooldown: int = 0
    keras_version: int = 3
    # state
    es_wait: int = 0
    es_best: Optional[float] = None
    es_best_epoch: int = 0
    es_has_snapshot: bool = False
    stopped_epoch: int = 0
    mc_best: float = float("inf")
    mc_saved_epochs: List[int] = field(default_factory=list)
    rl_best: float = float("inf")
    rl_wait: int = 0
    rl_cooldown_counter: int = 0
    lr_reduced_epochs: List[int] = field(default_factory=list)

    def __post_init__(self):
        self.lr = _f32(self.lr)

    def on_epoch_end(self, epoch: int, val_loss: float) -> EpochActions:
        """epoch is 0-based, as Keras passes it."""
        act = EpochActions(lr=self.lr)
        cur = float(val_loss)
        # ---- EarlyStopping.on_epoch_end
        if not self.es_has_snapshot:                     # "if best weights were never set, the current weights are the best"
            self.es_has_snapshot = True
            self.es_best_epoch = epoch
            act.snapshot_best_weights = True
        self.es_wait += 1
        if self.es_best is None or bool(np.less(cur - self.es_min_delta, self.es_best)):
            self.es_best = cur
            self.es_best_epoch = epoch
            act.snapshot_best_weights = True
            self.es_wait = 0
        elif self.es_wait >= self.es_patience and epoch > 0:
            self.stopped_epoch = epoch
            act.stop_training = True
        # ---- ModelCheckpoint(save_best_only=True)
        if bool(np.less(cur, self.mc_best)):
            self.mc_best = cur
            self.mc_saved_epochs.append(epoch)
            act.save_checkpoint = True
        # ---- ReduceLROnPlateau.on_epoch_end
        if self.rl_cooldown_counter > 0:
            self.rl_cooldown_counter -= 1
            self.rl_wait = 0
        if bool(np.less(cur, self.rl_best - self.rl_min_delta)):
            self.rl_best = cur
            self.rl_wait = 0
        elif not self.rl_cooldown_counter > 0:
            self.rl_wait += 1
            if self.rl_wait >= self.rl_patience:
                old_lr = self.lr
                if old_lr > float(np.float32(self.rl_min_lr)):
                    self.lr = _f32(max(old_lr * self.rl_factor, self.rl_min_lr))
                    self.lr_reduced_epochs.append(epoch)
                    act.lr_reduced = True
                    self.rl_cooldown_counter = self.rl_cooldown
                    self.rl_wait = 0
        act.lr = self.lr
        return act

    def restore_best_at_train_end(self) -> bool:
        """EarlyStopping(restore_best_weights=True).on_train_end: does the model end with the best epoch's weights?"""
        if not self.es_has_snapshot:
            return False
        if self.keras_version >= 3:
            return True
        return self.stopped_epoch > 0                    # Keras 2: only when the stop fired
