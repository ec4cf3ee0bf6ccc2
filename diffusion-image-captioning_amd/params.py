"""Flat parameter store of the denoiser.

All trainable tensors of `DistilBertModel` (ref CLIP-DDPM.py:227-269; 108 tensors / 44.3 M elements at 6 layers)
live in ONE contiguous fp32 buffer `P`, with gradients `G` and the AdamW moments `M`, `V` in three more buffers of
the same layout, plus (bf16 mode) a bf16 shadow `Pb` that the GEMMs read and (split-weight mode) its remainder `Pl = bf16(P - Pb)`.  Consequences:
  * AdamW is ONE kernel launch over the flat range (dic_adamw) and writes the shadow in the same pass;
  * the data-parallel gradient exchange is ONE RCCL all-reduce of `G`;
  * q/k/v projection weights are stored stacked ([2304][768]) so the three Linears run as one GEMM, and
    [LayerNorm.weight | LayerNorm.bias | bias of the Linear feeding that LayerNorm] are adjacent so one column-sum
    launch finalises all three gradients.
`parameters()` still returns per-tensor views in the reference's order (CLIP-DDPM.py:258-269), each with `.grad`
set to the matching view of `G`, so `torch.optim.AdamW(model.parameters())` -- the reference's trainer -- also works.
"""
from __future__ import annotations

import numpy as np
import torch

from . import synth

ALIGN = 64  # floats (256 B)


class ParamStore:
    def __init__(self, n_layers: int, device, concat: bool = True, dim: int = 768, hidden: int = 3072,
                 max_pos: int = 512, clip_dim: int = 512, bf16_shadow: bool = True, train_embedding_vocab: int | None = None,
                 in_channel: int = 16, timestep_embedding: int | None = None, split_shadow: bool = False):
        self.n_layers, self.dim, self.hidden, self.concat = n_layers, dim, hidden, concat
        self.te_vocab, self.in_channel = train_embedding_vocab, in_channel
        self.device = torch.device(device)
        self._slots = {}     # internal slot name -> (offset, shape)
        off = 0

        def add(name, shape):
            nonlocal off
            n = int(np.prod(shape))
            self._slots[name] = (off, tuple(shape))
            off += (n + ALIGN - 1) // ALIGN * ALIGN

        for i in range(n_layers):
            add(f"L{i}.Wqkv", (3 * dim, dim)); add(f"L{i}.bqkv", (3 * dim,))
            add(f"L{i}.Wo", (dim, dim))
            add(f"L{i}.ln1g", (dim,)); add(f"L{i}.ln1b", (dim,)); add(f"L{i}.bo", (dim,))
            add(f"L{i}.W1", (hidden, dim)); add(f"L{i}.b1", (hidden,))
            add(f"L{i}.W2", (dim, hidden))
            add(f"L{i}.ln2g", (dim,)); add(f"L{i}.ln2b", (dim,)); add(f"L{i}.b2", (dim,))
        add("pos", (max_pos, dim)); add("eln_g", (dim,)); add("eln_b", (dim,))
        add("Wvt", (dim, dim)); add("vln_g", (dim,)); add("vln_b", (dim,)); add("bvt", (dim,))
        add("Wimg", (dim, clip_dim)); add("bimg", (dim,))
        add("Wtxt", (dim, clip_dim)); add("btxt", (dim,))
        if concat:
            add("seg", (2, dim))
        self.temb_steps = timestep_embedding
        if timestep_embedding:
            add("temb", (int(timestep_embedding), dim))       # optional timestep embedding (cfg.TIMESTEP_EMBEDDING), absent from the reference
        if train_embedding_vocab is not None:
            # TRAIN_EMBEDDING ablation (ref :238-243): learned token embedding / rounding head in a 16-d space and the two
            # projections to and from the encoder width.  The head is stored with its rows padded to a tile multiple
            # (zero rows: no gradient, no decay effect) so its weight-gradient GEMM can write whole tiles.
            v, c = train_embedding_vocab, in_channel
            add("E16", (v, c)); add("Wlm16", ((v + 127) // 128 * 128, c))
            add("Win", (dim, c)); add("bin", (dim,)); add("Wout", (c, dim)); add("bout", (c,))
        self.numel = off
        self.text_unused = False      # set per step by diffusion.loss(): text_linear takes no part in the graph
        self.zero_pending = False     # set by AdamW.zero_grad(): the next backward clears the gradient slots it does not write
        self.P = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.G = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.Pb = torch.zeros(off, dtype=torch.bfloat16, device=self.device) if bf16_shadow else None
        # split weights (engine.Denoiser(split_weights=True)): Pl = bf16(P - Pb), the half the forward GEMMs add back as DicGemmParams.B2
        self.Pl = torch.zeros(off, dtype=torch.bfloat16, device=self.device) if (bf16_shadow and split_shadow) else None

        # reference-named views, in the order of CLIP-DDPM.py:258-269
        te = dict(train_embedding_vocab=train_embedding_vocab, in_channel=in_channel) if train_embedding_vocab is not None else {}
        self.names = [n for n, _, _, _ in synth.denoiser_param_specs(n_layers, **te)]
        if not concat:
            self.names = [n for n in self.names if n != "segment_embedding.weight"]
        if timestep_embedding:
            self.names = self.names + ["timestep_embedding.weight"]
        self._views = {n: self._view(self.P, n) for n in self.names}
        self._gviews = {n: self._view(self.G, n) for n in self.names}
        for n in self.names:
            self._views[n]._dic_store = self
        self.relink_grads()

    # ---- slot helpers
    def off(self, slot):
        return self._slots[slot][0]

    def ptr(self, slot, which="P"):
        buf = {"P": self.P, "G": self.G, "Pb": self.Pb, "Pl": self.Pl}[which]
        return buf.data_ptr() + self._slots[slot][0] * buf.element_size()

    def slot_view(self, buf, slot):
        o, shp = self._slots[slot]
        return buf[o:o + int(np.prod(shp))].view(shp)

    def _view(self, buf, ref_name):
        d = self.dim
        pre = "model.distilbert."
        simple = {
            pre + "embeddings.position_embeddings.weight": "pos", pre + "embeddings.LayerNorm.weight": "eln_g",
            pre + "embeddings.LayerNorm.bias": "eln_b", "model.vocab_transform.weight": "Wvt",
            "model.vocab_transform.bias": "bvt", "model.vocab_layer_norm.weight": "vln_g",
            "model.vocab_layer_norm.bias": "vln_b", "image_linear.weight": "Wimg", "image_linear.bias": "bimg",
            "text_linear.weight": "Wtxt", "text_linear.bias": "btxt", "segment_embedding.weight": "seg",
            "timestep_embedding.weight": "temb",
        }
        if ref_name in simple:
            return self.slot_view(buf, simple[ref_name])
        te = {"embedding.weight": "E16", "input_projection.weight": "Win", "input_projection.bias": "bin",
              "output_projection.weight": "Wout", "output_projection.bias": "bout"}
        if ref_name in te:
            return self.slot_view(buf, te[ref_name])
        if ref_name == "lm_head.weight":
            return self.slot_view(buf, "Wlm16")[:self.te_vocab]
        assert ref_name.startswith(pre + "transformer.layer."), ref_name
        rest = ref_name[len(pre + "transformer.layer."):]
        i, rest = rest.split(".", 1)
        L = f"L{i}."
        qkv = {"q_lin": 0, "k_lin": 1, "v_lin": 2}
        for lin, j in qkv.items():
            if rest == f"attention.{lin}.weight":
                return self.slot_view(buf, L + "Wqkv")[j * d:(j + 1) * d]
            if rest == f"attention.{lin}.bias":
                return self.slot_view(buf, L + "bqkv")[j * d:(j + 1) * d]
        m = {"attention.out_lin.weight": "Wo", "attention.out_lin.bias": "bo", "sa_layer_norm.weight": "ln1g",
             "sa_layer_norm.bias": "ln1b", "ffn.lin1.weight": "W1", "ffn.lin1.bias": "b1", "ffn.lin2.weight": "W2",
             "ffn.lin2.bias": "b2", "output_layer_norm.weight": "ln2g", "output_layer_norm.bias": "ln2b"}
        return self.slot_view(buf, L + m[rest])

    def active_ranges(self):
        """Flat [lo, hi) ranges the optimizer must touch: everything, minus text_linear when it was unused this step."""
        if not self.text_unused:
            return [(0, self.numel)]
        lo = self._slots["Wtxt"][0]
        o, shp = self._slots["btxt"]
        hi = o + (int(np.prod(shp)) + ALIGN - 1) // ALIGN * ALIGN
        return [(0, lo), (hi, self.numel)] if hi < self.numel else [(0, lo)]

    # ---- public
    def relink_grads(self):
        """(Re)attach `.grad` views of the flat gradient buffer (torch optimizers' zero_grad() sets them to None)."""
        for n in self.names:
            self._views[n].grad = self._gviews[n]

    def parameters(self):
        return [self._views[n] for n in self.names]

    def named_parameters(self):
        return [(n, self._views[n]) for n in self.names]

    @torch.no_grad()
    def load_state(self, state: dict):
        """`state`: reference-named tensors/arrays (e.g. synth.denoiser_state or a checkpoint)."""
        for n in self.names:
            if n == "timestep_embedding.weight" and n not in state:
                continue                                    # a reference-shaped state has none: keep the current table
            v = state[n]
            v = torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v
            self._views[n].copy_(v.to(self.device, torch.float32))

    def state_dict(self):
        return {n: self._views[n].detach().clone() for n in self.names}

    @torch.no_grad()
    def init_like_reference(self, seed: int = 0):
        """HF DistilBERT init (N(0, 0.02), zero biases, LayerNorm 1/0; hf `_init_weights`) and torch-default
        Linear/Embedding init for the wrapper's image/text/segment layers (ref :252-256), seeded."""
        g = torch.Generator(device="cpu").manual_seed(seed)
        for n in self.names:
            v = self._views[n]
            if "LayerNorm.weight" in n or "layer_norm.weight" in n:
                v.fill_(1.0)
            elif n in ("input_projection.bias", "output_projection.bias"):
                bound = 1.0 / np.sqrt(self.in_channel if n.startswith("input") else self.dim)
                v.copy_(((torch.rand(v.shape, generator=g) * 2 - 1) * bound).to(self.device))
            elif n.endswith(".bias") and not n.startswith(("image_linear", "text_linear")):
                v.zero_()
            elif n.startswith(("image_linear", "text_linear")):
                bound = 1.0 / np.sqrt(512)
                v.copy_(((torch.rand(v.shape, generator=g) * 2 - 1) * bound).to(self.device))
            elif n in ("segment_embedding.weight", "embedding.weight"):           # nn.Embedding default: N(0, 1)
                v.copy_(torch.randn(v.shape, generator=g).to(self.device))
            elif n in ("lm_head.weight", "input_projection.weight", "output_projection.weight"):   # nn.Linear default init
                bound = 1.0 / np.sqrt(v.shape[1])
                v.copy_(((torch.rand(v.shape, generator=g) * 2 - 1) * bound).to(self.device))
            else:
                v.copy_((torch.randn(v.shape, generator=g) * 0.02).to(self.device))
