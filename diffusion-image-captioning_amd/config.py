"""Process-global hyper-parameters, named exactly as the reference's module constants (CLIP-DDPM.py:55-114).

The reference reads these as module globals *at call time* (its asserts in `forward`/`loss` compare against
`BATCH_SIZE`, `SAMPLE_SIZE`, `MAX_LENGTH` ...), so the drop-in keeps one mutable global `cfg` object:
`dic.cfg.BATCH_SIZE = 512` plays the role of editing the constant in the script.
"""
from __future__ import annotations

from dataclasses import dataclass

LOSS_KINDS = {"series_sum_sample_mean": 0, "series_sum": 1, "mse_series_mean": 2, "mse_series_sum": 3}


@dataclass
class Config:
    DEBUG: bool = False
    BATCH_SIZE: int = 8
    MAX_LENGTH: int = 16
    LEARNING_RATE: float = 1e-4
    END_LEARNING_RATE: float = 5e-5
    TRAIN_SET_RATIO: float = 0.8
    EARLY_STOP_RATIO: float = 1.05
    EPOCH_NUM: int = 5
    DYNAMIC_ROUNDING_WEIGHT: float = -1
    ROUNDING_WEIGHT: float = 0.5
    LOSS_FUNC: str = "series_sum_sample_mean"
    CLIP_ADDING_METHOD: str = "concat"
    CLASSIFIER_FREE_WEIGHT: float = 0.0
    CLASSIFIER_FREE_PROB: float = 0.2
    TRAIN_EMBEDDING: bool = False
    IN_CHANNEL: int = 768
    BETA_MIN: float = 0.0001
    BETA_MAX: float = 0.02
    STEP_TOT: int = 1000
    COSIN_SCHEDULE: bool = True
    SAMPLE_SIZE: int = 100
    X_0_PREDICTION: bool = True
    X_T_STEP_INTERVAL: int = 100
    USE_X_T_LOSS: bool = True
    USE_X_1_LOSS: bool = True
    USE_PROB_LOSS: bool = True
    VOCAB_SIZE: int = 30522
    # build-side switch (not a reference constant): skip the provably-unused text row when no sequence is guided
    DROP_UNUSED_TEXT_ROW: bool = True
    # build-side switch: learned timestep embedding added before the embeddings LayerNorm (BASELINE north_star names it; the reference's
    # denoiser is not time-conditioned, ref :271, so the parity value is False)
    TIMESTEP_EMBEDDING: bool = False

    def update(self, **kw):
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError(f"unknown hyper-parameter {k}")
            setattr(self, k, v)
        return self


cfg = Config()
