#!/usr/bin/env python3
"""Are the parameters after two training steps bit-identical whatever the launch order (one stream / two streams / streamed AdamW / hipGraph)?"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dic = importlib.import_module("diffusion-image-captioning_amd")
OPT = importlib.import_module("diffusion-image-captioning_amd.options").OPT
B, L, V, nl = int(os.environ.get("BATCH", "8")), 16, 3000, int(os.environ.get("LAYERS", "2"))
dic.cfg.update(BATCH_SIZE=B, SAMPLE_SIZE=2, MAX_LENGTH=L, STEP_TOT=100, COSIN_SCHEDULE=False, VOCAB_SIZE=V, CLASSIFIER_FREE_WEIGHT=0.0,
               CLIP_ADDING_METHOD="concat", LOSS_FUNC="series_sum_sample_mean", X_0_PREDICTION=True, ROUNDING_WEIGHT=0.5)
E = dic.synth.vocab_embedding(V, 768, 0)
x = {k: torch.from_numpy(v).cuda() for k, v in dic.synth.batch(B, L, V, 1).items()}
def run(tag, graph=False, **opts):
    keep = {k: getattr(OPT, k) for k in opts}
    for k, v in opts.items():
        setattr(OPT, k, v)
    try:
        m = dic.DistilBertModel(E, E, config=dict(n_layers=nl, dropout=0.1, attention_dropout=0.1), dtype="bf16", seed=3)
        tr = dic.AdamW(m.parameters(), lr=1e-4)
        dic.seed_all(77)
        outs = []
        if graph:
            step = dic.GraphedTrainStep(m, tr, x, warmup=1)
            for _ in range(int(os.environ.get("REPLAYS", "2"))):
                step()
            step.release()
        else:
            for _ in range(1 + int(os.environ.get("REPLAYS", "2"))):
                dic.train_func(m, tr, x)
        torch.cuda.synchronize()
        return m.params.P.clone(), m.params.G.clone()
    finally:
        for k, v in keep.items():
            setattr(OPT, k, v)
ref = run("one stream", wgrad_stream=False, streamed_adamw=False)
for tag, kw in (("two streams", dict(streamed_adamw=False)), ("one stream + streamed AdamW", dict(wgrad_stream=False)), ("default (two streams, streamed AdamW)", {}),
                ("default, hipGraph", dict(graph=True)), ("default again", {})):
    g = kw.pop("graph", False)
    P, G = run(tag, graph=g, **kw)
    dP = (P - ref[0]).abs()
    names = m_names = None
    print(f"{tag:42s} max |dP| {float(dP.max()):.3e}  differing elements {int((dP > 0).sum())}   max |dG| {float((G - ref[1]).abs().max()):.3e}")
    if float(dP.max()) > 0:
        st = dic.DistilBertModel(E, E, config=dict(n_layers=nl), dtype="bf16").params
        dG = (G - ref[1]).abs()
        for (o_, k), nxt in zip(sorted(((v[0], k) for k, v in st._slots.items())), sorted(((v[0], k) for k, v in st._slots.items()))[1:] + [(st.numel, None)]):
            seg = dP[o_:nxt[0]]
            if seg.numel() and float(seg.max()) > 0:
                print(f"      dP in {k:12s} max {float(seg.max()):.3e}")
        for n, p_ in st.named_parameters():
            pass
        slots = sorted(((v[0], k) for k, v in st._slots.items()))
        for (o_, k), nxt in zip(slots, slots[1:] + [(st.numel, None)]):
            seg = dG[o_:nxt[0]]
            if seg.numel() and float(seg.max()) > 0:
                print(f"      dG in {k:12s} max {float(seg.max()):.3e}  (|G| max {float(ref[1][o_:nxt[0]].abs().max()):.3e})")
