"""MI355X-native CLIP-conditioned diffusion-LM captioning hot path.

The directory name carries a hyphen (it mirrors the reference repo's name), so import it with
`importlib.import_module("diffusion-image-captioning_amd")`; the first import registers the
alias `dic_amd` so `import dic_amd` works afterwards.

Public surface = the reference's hot-path callables (ref CLIP-DDPM.py):
    cfg                      the hyper-parameter globals (BATCH_SIZE, SAMPLE_SIZE, MAX_LENGTH, ...)
    DistilBertModel          denoiser wrapper (alias of engine.Denoiser)
    AdamW                    fused optimizer with torch.optim.AdamW's interface
    diffuse_t, generate_diffuse_pair, loss, train_func, validate, sample
Importing the package works without a GPU (host logic, synthetic data); constructing a model or calling an op
without the built HIP library / a visible MI355X raises RuntimeError -- there is no CPU fallback.
"""
import sys as _sys

_sys.modules.setdefault("dic_amd", _sys.modules[__name__])

from . import synth, parallel  # noqa: E402,F401
from .config import cfg, Config, LOSS_KINDS  # noqa: E402,F401
from ._lib import build, lib, LIB_PATH  # noqa: E402,F401


def __getattr__(name):
    # torch-dependent pieces are imported lazily so `import` stays cheap for host-only uses
    if name in ("Denoiser", "DistilBertModel"):
        from .engine import Denoiser
        return Denoiser
    if name in ("AdamW", "diffuse_t", "generate_diffuse_pair", "loss", "train_func", "validate", "sample",
                "alpha_cumprod_table", "set_alpha_cumprod", "seed_noise", "seed_timesteps", "seed_guidance", "seed_all", "rng_state",
                "set_rng_state", "set_loaders", "dedup_columns"):
        from . import diffusion
        return getattr(diffusion, name)
    if name == "GraphedTrainStep":
        from .graph import GraphedTrainStep
        return GraphedTrainStep
    if name in ("bleu", "harness"):
        import importlib
        return importlib.import_module(f"{__name__}.{name}")
    raise AttributeError(name)
