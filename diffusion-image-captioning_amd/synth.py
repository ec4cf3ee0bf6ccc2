"""Deterministic synthetic tensors (weights, CLIP features, captions, noise).

There is no network for checkpoints or datasets, so every weight and batch used by the
tests, the golden-vector generator, `smoke()` and `bench.py` is *generated*.  The generator
is pure integer arithmetic (splitmix64 on `(stream, index)`) followed by one float
multiply, so the same `(stream, shape)` gives bit-identical float32 arrays on every
machine and numpy version -- the golden fixtures under `tests/golden/` therefore only
need to store seeds and the reference's *outputs*, never the 94 MB embedding matrix.

Shapes/ranges follow SURVEY.md section 8(d): unit-norm 512-d CLIP features
(COCO_BLEU.py:221), uniform token ids, ones-then-zeros attention masks.
"""
from __future__ import annotations

import zlib
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def stream_id(name: str, seed: int = 0) -> int:
    """Stable 32-bit stream id from a tensor name and a seed."""
    return (zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0xFFFFFFFF


def _bits(stream: int, n: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        base = _splitmix64(np.full(1, np.uint64(stream) * np.uint64(0x100000001B3) + np.uint64(0x51ED), dtype=np.uint64))
        return _splitmix64(idx ^ base)


def normal(stream: int, shape, scale: float = 1.0, shift: float = 0.0) -> np.ndarray:
    """Approximately N(shift, scale^2): Irwin-Hall sum of four 16-bit uniforms (integer math)."""
    n = int(np.prod(shape)) if len(shape) else 1
    b = _bits(stream, n)
    s = ((b & np.uint64(0xFFFF)) + ((b >> np.uint64(16)) & np.uint64(0xFFFF))
         + ((b >> np.uint64(32)) & np.uint64(0xFFFF)) + (b >> np.uint64(48))).astype(np.int64)
    # mean 2*65535, variance 4*(65536^2-1)/12
    z = (s - 131070).astype(np.float64) * (1.0 / 37837.22)
    return (z * scale + shift).astype(np.float32).reshape(shape)


def uniform_int(stream: int, shape, lo: int, hi: int) -> np.ndarray:
    """Integers in [lo, hi)."""
    n = int(np.prod(shape)) if len(shape) else 1
    b = _bits(stream, n) >> np.uint64(11)
    return (lo + (b % np.uint64(hi - lo)).astype(np.int64)).reshape(shape)


def uniform(stream: int, shape) -> np.ndarray:
    """float32 in [0, 1) with 24 bits."""
    n = int(np.prod(shape)) if len(shape) else 1
    b = (_bits(stream, n) >> np.uint64(40)).astype(np.float64)
    return (b * (1.0 / 16777216.0)).astype(np.float32).reshape(shape)


# ----------------------------------------------------------------------------------------
# model state (names follow the reference module tree: CLIP-DDPM.py:227-256 + HF DistilBERT)
# ----------------------------------------------------------------------------------------

def denoiser_param_specs(n_layers: int, dim: int = 768, hidden: int = 3072, max_pos: int = 512, clip_dim: int = 512,
                         train_embedding_vocab: int | None = None, in_channel: int = 16):
    """(name, shape, scale, shift) in the order `DistilBertModel.parameters()` returns them
    (CLIP-DDPM.py:258-269): HF encoder params, image_linear, text_linear, [TRAIN_EMBEDDING: embedding, lm_head,
    input_projection, output_projection (:260-262)], segment_embedding."""
    specs = [
        ("model.distilbert.embeddings.position_embeddings.weight", (max_pos, dim), 0.02, 0.0),
        ("model.distilbert.embeddings.LayerNorm.weight", (dim,), 0.1, 1.0),
        ("model.distilbert.embeddings.LayerNorm.bias", (dim,), 0.05, 0.0),
    ]
    for i in range(n_layers):
        p = f"model.distilbert.transformer.layer.{i}."
        for lin in ("q_lin", "k_lin", "v_lin", "out_lin"):
            specs.append((p + f"attention.{lin}.weight", (dim, dim), 0.03, 0.0))
            specs.append((p + f"attention.{lin}.bias", (dim,), 0.02, 0.0))
        specs.append((p + "sa_layer_norm.weight", (dim,), 0.1, 1.0))
        specs.append((p + "sa_layer_norm.bias", (dim,), 0.05, 0.0))
        specs.append((p + "ffn.lin1.weight", (hidden, dim), 0.03, 0.0))
        specs.append((p + "ffn.lin1.bias", (hidden,), 0.02, 0.0))
        specs.append((p + "ffn.lin2.weight", (dim, hidden), 0.03, 0.0))
        specs.append((p + "ffn.lin2.bias", (dim,), 0.02, 0.0))
        specs.append((p + "output_layer_norm.weight", (dim,), 0.1, 1.0))
        specs.append((p + "output_layer_norm.bias", (dim,), 0.05, 0.0))
    specs += [
        ("model.vocab_transform.weight", (dim, dim), 0.03, 0.0),
        ("model.vocab_transform.bias", (dim,), 0.02, 0.0),
        ("model.vocab_layer_norm.weight", (dim,), 0.1, 1.0),
        ("model.vocab_layer_norm.bias", (dim,), 0.05, 0.0),
        ("image_linear.weight", (dim, clip_dim), 0.04, 0.0),
        ("image_linear.bias", (dim,), 0.02, 0.0),
        ("text_linear.weight", (dim, clip_dim), 0.04, 0.0),
        ("text_linear.bias", (dim,), 0.02, 0.0),
    ]
    if train_embedding_vocab is not None:      # learned 16-d token space (CLIP-DDPM.py:238-243)
        v, c = train_embedding_vocab, in_channel
        specs += [
            ("embedding.weight", (v, c), 1.0, 0.0),
            ("lm_head.weight", (v, c), 0.15, 0.0),
            ("input_projection.weight", (dim, c), 0.15, 0.0),
            ("input_projection.bias", (dim,), 0.1, 0.0),
            ("output_projection.weight", (c, dim), 0.03, 0.0),
            ("output_projection.bias", (c,), 0.02, 0.0),
        ]
    specs.append(("segment_embedding.weight", (2, dim), 0.5, 0.0))
    return specs


def denoiser_state(n_layers: int, seed: int = 0, **kw) -> dict:
    return {name: normal(stream_id(name, seed), shape, scale, shift)
            for name, shape, scale, shift in denoiser_param_specs(n_layers, **kw)}


def vocab_embedding(vocab: int = 30522, dim: int = 768, seed: int = 0) -> np.ndarray:
    """Frozen token embedding E; the rounding head is tied to it (W_lm = E, bias 0),
    as in pretrained DistilBERT (CLIP-DDPM.py:245-247, 331)."""
    return normal(stream_id("embedding.weight", seed), (vocab, dim), 0.05, 0.0)


# ----------------------------------------------------------------------------------------
# batches (schema: CLIP-DDPM.py:190-197)
# ----------------------------------------------------------------------------------------

def batch(batch_size: int, max_length: int = 16, vocab: int = 30522, seed: int = 1, clip_dim: int = 512) -> dict:
    img = normal(stream_id("image_clip", seed), (batch_size, clip_dim)).astype(np.float64)
    txt = normal(stream_id("text_clip", seed), (batch_size, clip_dim)).astype(np.float64)
    img = (img / np.sqrt((img * img).sum(-1, keepdims=True))).astype(np.float32)
    txt = (txt / np.sqrt((txt * txt).sum(-1, keepdims=True))).astype(np.float32)
    ids = uniform_int(stream_id("input_ids", seed), (batch_size, max_length), 0, vocab)
    lens = uniform_int(stream_id("lengths", seed), (batch_size,), 6, max_length + 1)
    mask = (np.arange(max_length)[None, :] < lens[:, None]).astype(np.int64)
    return {"image_clip": img, "text_clip": txt, "input_ids": ids, "attention_mask": mask}


def timesteps(sample_size: int, step_tot: int, seed: int) -> np.ndarray:
    return uniform_int(stream_id("t", seed), (sample_size, 1, 1), 0, step_tot)


def noise(shape, seed: int, tag: str = "eps") -> np.ndarray:
    return normal(stream_id(tag, seed), tuple(shape))
