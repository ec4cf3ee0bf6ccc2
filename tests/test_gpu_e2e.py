"""End-to-end parity of the HIP path behind the reference's call signatures: golden vectors recorded from the
reference itself (tests/golden, made by oracle/gen_golden.py) and the CPU oracle on fresh seeded inputs.  GPU only.

Tolerances: fp32 mode -- losses 1e-4 relative (north_star), hidden states 2e-4 absolute, token ids bit-exact;
bf16 mode -- losses 3e-3 relative (bf16 operands, fp32 accumulation/statistics), reported in DESIGN.md."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

dic = importlib.import_module("diffusion-image-captioning_amd")
synth = dic.synth
from oracle import ref_model as R          # noqa: E402

TRAIN_CASES = ["base_b4s3l16", "cfg_b2s2l32", "cfg6_b2s2l32", "deep6_b2s2l16", "add_mse_b3s2l16", "xprev_sum_b3s2l16", "addcfg_msesum_b3s2l16"]
TRAIN_STEP_CASES = TRAIN_CASES + ["refdefault_b8s100l16"]      # (+ one AdamW step at the shape the reference trains at: B=8, S=100, 6 layers)


def configure(m):
    dic.cfg.update(BATCH_SIZE=m["B"], SAMPLE_SIZE=m["S"], MAX_LENGTH=m["L"], STEP_TOT=m["step_tot"], COSIN_SCHEDULE=m["cosine"],
                   ROUNDING_WEIGHT=m["rounding_weight"], LOSS_FUNC=m["loss"], CLIP_ADDING_METHOD=m["fusion"],
                   CLASSIFIER_FREE_WEIGHT=m["cfg_w"], CLASSIFIER_FREE_PROB=m["cfg_prob"], X_0_PREDICTION=m["x0_pred"],
                   X_T_STEP_INTERVAL=m["x_t_step_interval"], VOCAB_SIZE=m["vocab"],
                   TRAIN_EMBEDDING=m.get("train_embedding", False), IN_CHANNEL=m.get("in_channel", 768))


def build_model(m, dtype, z=None):
    configure(m)
    # the cosine table depends on the host's torch.cos (not correctly rounded): use the table the reference host produced
    dic.set_alpha_cumprod(torch.from_numpy(z["alpha_cumprod"]) if z is not None else None)
    if m.get("train_embedding"):        # ref :325-327: the model builds its own embedding, head and projections
        model = dic.DistilBertModel(config=dict(n_layers=m["n_layers"], dropout=0.0, attention_dropout=0.0), dtype=dtype)
        model.load_state(synth.denoiser_state(m["n_layers"], m["wseed"], train_embedding_vocab=m["vocab"], in_channel=m["in_channel"]))
    else:
        E = synth.vocab_embedding(m["vocab"], 768, m["wseed"])
        model = dic.DistilBertModel(E, E, config=dict(n_layers=m["n_layers"], dropout=0.0, attention_dropout=0.0), dtype=dtype)
        model.load_state(synth.denoiser_state(m["n_layers"], m["wseed"]))
    x = {k: torch.from_numpy(v).cuda() for k, v in synth.batch(m["B"], m["L"], m["vocab"], m["dseed"]).items()}
    return model, x


def draws(m, seed):
    t = torch.from_numpy(synth.uniform_int(synth.stream_id("t", seed), (m["S"], 1, 1), 0, m["step_tot"]))
    n_noise = 2 if m["x0_pred"] else 3
    noises = [torch.from_numpy(synth.noise((m["B"], m["L"], m.get("in_channel", 768)), seed, f"eps{i}")) for i in range(n_noise)]
    u = torch.from_numpy(synth.uniform(synth.stream_id("cfg", seed), (m["S"] * m["B"], 1)))
    return t, noises, u


def f(x):
    return float(x.detach().float().cpu())


@pytest.mark.parametrize("name", TRAIN_CASES)
def test_golden_eval_forward_fp32(name):
    z, m = load_golden(name)
    model, x = build_model(m, "fp32", z)
    model.eval()
    t, noises, u = draws(m, 123)
    S, B, L = m["S"], m["B"], m["L"]
    with torch.no_grad():
        x_0 = model.embedding(x["input_ids"])
        x_t = dic.diffuse_t(x_0, t.cuda(), noise=noises[0])
        x_1 = dic.diffuse_t(x_0, torch.ones(1, dtype=torch.int64), noise=noises[-1])
        np.testing.assert_array_equal(x_t[:, :2, :8].cpu().numpy(), z["x_t_head"])     # q_sample bit-exact
        np.testing.assert_array_equal(x_1[:, :2, :8].cpu().numpy(), z["x_1_head"])
        cm = R.concat_mask_for(R.Config(SAMPLE_SIZE=S, BATCH_SIZE=B, CLASSIFIER_FREE_WEIGHT=m["cfg_w"], CLASSIFIER_FREE_PROB=m["cfg_prob"]), S * B, u)
        lt, ht = model(x_t, x["image_clip"].unsqueeze(1).repeat(S, 1, 1), x["text_clip"].unsqueeze(1).repeat(S, 1, 1),
                       x["attention_mask"].repeat(S, 1), cm.cuda())
        l1, h1 = model(x_1, x["image_clip"].unsqueeze(1), x["text_clip"].unsqueeze(1), x["attention_mask"],
                       torch.tensor([1, 0]).repeat(B, 1).cuda())
        stride = 1 if z["hid_t"].shape[-1] == ht.shape[-1] else 16
        np.testing.assert_allclose(ht[:, :, ::stride].cpu().numpy(), z["hid_t"], rtol=0, atol=2e-4)
        np.testing.assert_allclose(h1[:, :, ::stride].cpu().numpy(), z["hid_1"], rtol=0, atol=2e-4)
        np.testing.assert_array_equal(lt.argmax(-1).cpu().numpy(), z["argmax_t"])       # token ids bit-exact
        np.testing.assert_array_equal(l1.argmax(-1).cpu().numpy(), z["argmax_1"])
        np.testing.assert_allclose(torch.logsumexp(lt.double(), -1).cpu().numpy(), z["lse_t"], rtol=2e-5)
        l, a, b, c = dic.train_func(model, None, x, train=False, t=t, noises=noises, cfg_uniform=u)
    got = np.array([f(l), f(a), f(b), f(c)])
    np.testing.assert_allclose(got, z["eval_losses"], rtol=1e-4)


def test_golden_reference_default_shape_fp32_and_bf16():
    """The reference's default shape -- B=8, S=100 (808 sequences), 6 layers, cosine T=1000 (ref :57-114) -- against the fixture the
    reference produced at that shape: q_sample (to the reference host's sqrt rounding), hidden states, token ids (bit-exact wherever the reference's own top-2 margin
    exceeds 5e-4: 12 800 rows through 6 layers leave ~1e-5 on a logit), logsumexp, validate()-style losses to 1e-4; then the same step
    in bf16 (the benchmarked dtype) with its loss deltas reported against the same reference numbers."""
    z, m = load_golden("refdefault_b8s100l16")
    model, x = build_model(m, "fp32", z)
    model.eval()
    t, noises, u = draws(m, 123)
    S, B = m["S"], m["B"]
    with torch.no_grad():
        x_0 = model.embedding(x["input_ids"])
        x_t = dic.diffuse_t(x_0, t.cuda(), noise=noises[0])
        # q_sample: the reference host's fp32 sqrt(alpha_bar[t]) is not correctly rounded for 3 of these 100 timesteps (checked against
        # fp64 when the fixture was made), the product's table is: those rows differ by one ulp of the coefficient, all others are bit-equal
        xh, zh = x_t[:, :2, :8].cpu().numpy(), z["x_t_head"]
        assert (xh == zh).mean() > 0.9 and np.abs(xh - zh).max() <= 1.2e-7 * np.abs(zh).max()
        lt, ht = model(x_t, x["image_clip"].unsqueeze(1).repeat(S, 1, 1), x["text_clip"].unsqueeze(1).repeat(S, 1, 1),
                       x["attention_mask"].repeat(S, 1), torch.tensor([1, 0]).repeat(S * B, 1).cuda())
        np.testing.assert_allclose(ht[::50, :, ::64].cpu().numpy(), z["hid_t"], rtol=0, atol=2e-4)
        ids = lt.argmax(-1).cpu().numpy()
        bad = ids != z["argmax_t"]
        assert not (bad & (z["margin_t"] > 5e-4)).any() and bad.mean() < 1e-3, f"{bad.sum()} token ids differ"
        np.testing.assert_allclose(torch.logsumexp(lt.double(), -1).cpu().numpy(), z["lse_t"], rtol=2e-5)
        del lt
        l, a, b, c = dic.train_func(model, None, x, train=False, t=t, noises=noises, cfg_uniform=u)
    np.testing.assert_allclose(np.array([f(l), f(a), f(b), f(c)]), z["eval_losses"], rtol=1e-4)
    del model
    model, x = build_model(m, "bf16", z)
    model.eval()
    with torch.no_grad():
        l, a, b, c = dic.train_func(model, None, x, train=False, t=t, noises=noises, cfg_uniform=u)
    rel = np.abs(np.array([f(l), f(a), f(b), f(c)]) - z["eval_losses"]) / np.abs(z["eval_losses"])
    print("bf16 vs reference fp32 at the reference-default shape, relative loss deltas (l, x_t, x_1, prob):", rel)
    assert rel.max() < 3e-3


@pytest.mark.parametrize("name", TRAIN_STEP_CASES)
def test_golden_two_training_steps_fp32(name):
    z, m = load_golden(name)
    model, x = build_model(m, "fp32", z)
    model.train()
    trainer = dic.AdamW(model.parameters(), lr=m["lr"])
    names = [n for n, _ in model.named_parameters()]
    assert names == m["param_names"]
    keep = np.array([not n.endswith("k_lin.bias") for n in names])      # analytically-zero gradient: see test_oracle_golden
    for step in range(z["step_losses"].shape[0]):
        t, noises, u = draws(m, 123 + step)
        l, a, b, c = dic.train_func(model, trainer, x, train=True, t=t, noises=noises, cfg_uniform=u)
        got = np.array([f(l), f(a), f(b), f(c)])
        np.testing.assert_allclose(got, z["step_losses"][step], rtol=1e-4, err_msg=f"step {step} losses")
        gn = np.array([float(p.grad.double().norm()) for p in model.parameters()])
        np.testing.assert_allclose(gn[keep], z["grad_norms"][step][keep], rtol=2e-3, atol=1e-6, err_msg=f"step {step} grad norms")
        gh = np.stack([np.resize(p.grad.flatten()[:8].cpu().numpy(), 8) for p in model.parameters()])
        scale = np.abs(z["grad_heads"][step]).max(axis=1, keepdims=True) + 1e-8
        assert (np.abs(gh - z["grad_heads"][step])[keep] / scale[keep]).max() < 2e-2, f"step {step} grad heads"
        pn = np.array([float(p.detach().double().norm()) for p in model.parameters()])
        np.testing.assert_allclose(pn[keep], z["param_norms"][step][keep], rtol=2e-6, err_msg=f"step {step} param norms")


@pytest.fixture
def restore_cfg():
    yield
    dic.cfg.update(TRAIN_EMBEDDING=False, IN_CHANNEL=768)


def test_train_embedding_ablation_matches_reference_fp32(restore_cfg):
    """TRAIN_EMBEDDING=True (ref :98-102, 238-243, 292-293, 319-320): learned 16-d embedding / head / projections.  Same bar as
    the main path: eval forward + ids, then two AdamW steps (losses, per-tensor gradient norms -- embedding, lm_head and
    projection gradients included --, parameter norms) against the fixture the reference itself produced."""
    test_golden_eval_forward_fp32("trainemb_b3s2l16")
    test_golden_two_training_steps_fp32("trainemb_b3s2l16")


@pytest.mark.parametrize("name", ["trainemb_cfg_b3s2l16", "trainemb_xprev_add_b3s2l16"])
def test_train_embedding_with_guidance_and_with_xprev_targets(restore_cfg, name):
    """The ablation combined with the other switches: classifier-free guidance (guided copies share the projected input and send
    their input gradient back to the row they copy), and x_{t-1} prediction under "add" fusion with the L2-norm loss (the x_t loss
    then targets a noised copy of x_0, which scales the target gradient by sqrt(abar[t_next]))."""
    test_golden_two_training_steps_fp32(name)
    z, m = load_golden(name)
    model, x = build_model(m, "fp32", z)
    model.eval()
    t, noises, u = draws(m, 123)
    with torch.no_grad():
        l, a, b, c = dic.train_func(model, None, x, train=False, t=t, noises=noises, cfg_uniform=u)
    np.testing.assert_allclose(np.array([f(l), f(a), f(b), f(c)]), z["eval_losses"], rtol=1e-4)


def test_train_embedding_ablation_bf16_encoder_and_sampling(restore_cfg):
    z, m = load_golden("trainemb_b3s2l16")
    model, x = build_model(m, "bf16", z)
    trainer = dic.AdamW(model.parameters(), lr=m["lr"])
    for step in range(2):
        t, noises, u = draws(m, 123 + step)
        l, a, b, c = dic.train_func(model, trainer, x, train=True, t=t, noises=noises, cfg_uniform=u)
        got = np.array([f(l), f(a), f(b), f(c)])
        np.testing.assert_allclose(got, z["step_losses"][step], rtol=5e-3)
    # sampling loop in the 16-d space against the oracle (fp32 encoder: ids bit-exact)
    model32, _ = build_model(m, "fp32", z)
    model32.eval()
    start = torch.from_numpy(synth.noise((m["B"], m["L"] + 2, m["in_channel"]), 77, "restored"))
    ids, hid = dic.sample(model32, x["image_clip"], steps=3, start=start, return_hidden=True)
    ocfg = R.Config(MAX_LENGTH=m["L"], n_layers=m["n_layers"], vocab=m["vocab"], TRAIN_EMBEDDING=True, IN_CHANNEL=m["in_channel"])
    om = R.build(ocfg, synth.denoiser_state(m["n_layers"], m["wseed"], train_embedding_vocab=m["vocab"], in_channel=m["in_channel"]),
                 synth.vocab_embedding(8, 768, 0), requires_grad=False)
    oids, ohid = R.sample(om, x["image_clip"].cpu(), steps=3, start=start)
    np.testing.assert_allclose(hid.cpu().numpy(), ohid.numpy(), atol=3e-4, rtol=0)
    np.testing.assert_array_equal(ids.cpu().numpy(), oids.numpy())


@pytest.mark.parametrize("name", ["base_b4s3l16", "deep6_b2s2l16"])
def test_golden_training_bf16_within_tolerance(name):
    z, m = load_golden(name)
    model, x = build_model(m, "bf16", z)
    trainer = dic.AdamW(model.parameters(), lr=m["lr"])
    for step in range(2):
        t, noises, u = draws(m, 123 + step)
        l, a, b, c = dic.train_func(model, trainer, x, train=True, t=t, noises=noises, cfg_uniform=u)
        got = np.array([f(l), f(a), f(b), f(c)])
        print(name, "bf16 step", step, "rel loss delta", np.abs(got - z["step_losses"][step]) / np.abs(z["step_losses"][step]))
        np.testing.assert_allclose(got, z["step_losses"][step], rtol=3e-3)


def test_sampling_loop_ids_bit_exact_fp32_and_bf16_hidden():
    z, m = load_golden("sample_b3k3")
    dic.cfg.update(MAX_LENGTH=m["L"], CLASSIFIER_FREE_WEIGHT=0.0, CLIP_ADDING_METHOD="concat", VOCAB_SIZE=m["vocab"])
    dic.set_alpha_cumprod(None)
    E = synth.vocab_embedding(m["vocab"], 768, m["wseed"])
    xb = synth.batch(m["B"], m["L"], m["vocab"], m["dseed"])
    start = torch.from_numpy(synth.noise((m["B"], m["L"] + 2, 768), m["start_seed"], "restored"))
    model = dic.DistilBertModel(E, E, config=dict(n_layers=m["n_layers"]), dtype="fp32")
    model.load_state(synth.denoiser_state(m["n_layers"], m["wseed"]))
    model.eval()
    ids, hid = dic.sample(model, torch.from_numpy(xb["image_clip"]), steps=m["steps"], start=start, return_hidden=True)
    np.testing.assert_allclose(hid.cpu().numpy(), z["final_hidden"], atol=3e-4, rtol=0)
    np.testing.assert_array_equal(ids.cpu().numpy(), z["ids"])
    np.testing.assert_array_equal(dic.dedup_columns(ids).cpu().numpy(), z["uniq"])
    model16 = dic.DistilBertModel(E, E, config=dict(n_layers=m["n_layers"]), dtype="bf16")
    model16.load_state(synth.denoiser_state(m["n_layers"], m["wseed"]))
    model16.eval()
    ids16, hid16 = dic.sample(model16, torch.from_numpy(xb["image_clip"]), steps=m["steps"], start=start, return_hidden=True)
    err = float((hid16.cpu() - torch.from_numpy(z["final_hidden"])).abs().max())
    same = ids16.cpu().numpy() == z["ids"]
    agree = float(same.mean())
    # a random-init denoiser gives near-ties between vocabulary rows: report the agreement per top-1 / top-2 logit margin of the REFERENCE
    # hidden state -- ids may only differ where the reference's own decision is within the bf16 drift of the hidden state
    lg = torch.from_numpy(z["final_hidden"])[:, :m["L"]].double() @ torch.from_numpy(E).double().t()
    top2 = lg.topk(2, -1).values
    margin = (top2[..., 0] - top2[..., 1]).numpy()
    drift = float((hid16.cpu()[:, :m["L"]].double() - torch.from_numpy(z["final_hidden"])[:, :m["L"]].double()).norm(dim=-1).max())
    bound = 2.0 * drift * float(torch.from_numpy(E).double().norm(dim=-1).max())          # |delta logit_a - delta logit_b| <= |dx| (|w_a| + |w_b|)
    for lo, hi in ((0, 0.01), (0.01, 0.03), (0.03, 0.1), (0.1, 1e9)):
        sel = (margin >= lo) & (margin < hi)
        if sel.any():
            print(f"bf16 sampling: margin [{lo}, {hi}): {int(sel.sum())} tokens, agreement {float(same[sel].mean()):.3f}")
    print("bf16 sampling: max |hidden - ref| =", err, " id agreement =", agree, " decision bound on the margin =", bound)
    assert err < 0.15 and agree > 0.6
    assert bool(same[margin > bound].all()), "a token whose reference margin exceeds what the hidden-state drift can flip must agree"
    # the other modes run the same loop: the raw engine; the exact form (fp32-residual epilogues inside the captured passes); and the default
    # engine with its mean-row launches + centred residual stream INSIDE the sampling passes (options.sample_raw off): same bar
    engine_mod = importlib.import_module("diffusion-image-captioning_amd.engine")
    for dt in ("bf16r", "bf16w", "bf16+corrections"):
        mp = dic.DistilBertModel(E, E, config=dict(n_layers=m["n_layers"]), dtype=dt.split("+")[0])
        mp.load_state(synth.denoiser_state(m["n_layers"], m["wseed"]))
        mp.eval()
        keep_raw = engine_mod.OPT.sample_raw
        engine_mod.OPT.sample_raw = "+" not in dt and keep_raw
        try:
            idp, hp = dic.sample(mp, torch.from_numpy(xb["image_clip"]), steps=m["steps"], start=start, return_hidden=True)
        finally:
            engine_mod.OPT.sample_raw = keep_raw
        errp = float((hp.cpu() - torch.from_numpy(z["final_hidden"])).abs().max())
        samep = idp.cpu().numpy() == z["ids"]
        print(f"{dt} sampling: max |hidden - ref| = {errp}, id agreement = {float(samep.mean())}")
        assert errp < 0.15 and bool(samep[margin > bound].all())


def test_training_with_dropout_runs_and_is_replayable():
    """Reference default: dropout 0.1 / attention dropout 0.1 in train mode.  Philox masks are keyed by (seed, index):
    two models with the same seed take identical steps; the loss stays finite and moves."""
    dic.cfg.update(BATCH_SIZE=4, SAMPLE_SIZE=2, MAX_LENGTH=16, STEP_TOT=100, COSIN_SCHEDULE=False, ROUNDING_WEIGHT=0.5,
                   LOSS_FUNC="series_sum_sample_mean", CLIP_ADDING_METHOD="concat", CLASSIFIER_FREE_WEIGHT=0.0,
                   X_0_PREDICTION=True, VOCAB_SIZE=2000)
    dic.set_alpha_cumprod(None)
    E = synth.vocab_embedding(2000, 768, 0)
    x = {k: torch.from_numpy(v).cuda() for k, v in synth.batch(4, 16, 2000, 1).items()}
    losses = []
    for rep in range(2):
        model = dic.DistilBertModel(E, E, config=dict(n_layers=2, dropout=0.1, attention_dropout=0.1), dtype="bf16", seed=3)
        model.load_state(synth.denoiser_state(2, 0))
        trainer = dic.AdamW(model.parameters(), lr=1e-4)
        dic.seed_noise(99)
        ls = []
        for step in range(3):
            t = torch.from_numpy(synth.uniform_int(synth.stream_id("t", step), (2, 1, 1), 0, 100))
            l, *_ = dic.train_func(model, trainer, x, t=t)
            ls.append(f(l))
        losses.append(ls)
    assert losses[0] == losses[1], losses
    assert all(np.isfinite(losses[0])) and losses[0][2] < losses[0][0]


def test_data_parallel_code_path_on_one_gpu_matches_the_plain_step(monkeypatch):
    """RCCL at world size 1 with the exchange path forced on: per-layer all-reduce slices issued from the backward, AdamW
    stepping each slice as its exchange finishes.  Must take bit-identical steps to the single-process path."""
    import torch.distributed as dist
    dic.cfg.update(BATCH_SIZE=4, SAMPLE_SIZE=2, MAX_LENGTH=16, STEP_TOT=100, COSIN_SCHEDULE=False, ROUNDING_WEIGHT=0.5,
                   LOSS_FUNC="series_sum_sample_mean", CLIP_ADDING_METHOD="concat", CLASSIFIER_FREE_WEIGHT=0.0,
                   X_0_PREDICTION=True, VOCAB_SIZE=2000)
    dic.set_alpha_cumprod(None)
    E = synth.vocab_embedding(2000, 768, 0)
    x = {k: torch.from_numpy(v).cuda() for k, v in synth.batch(4, 16, 2000, 1).items()}

    def run():
        model = dic.DistilBertModel(E, E, config=dict(n_layers=2, dropout=0.1, attention_dropout=0.1), dtype="bf16", seed=3)
        model.load_state(synth.denoiser_state(2, 0))
        trainer = dic.AdamW(model.parameters(), lr=1e-4)
        dic.seed_noise(99)
        ls = []
        for step in range(3):
            t = torch.from_numpy(synth.uniform_int(synth.stream_id("t", step), (2, 1, 1), 0, 100))
            l, *_ = dic.train_func(model, trainer, x, t=t)
            ls.append(f(l))
        torch.cuda.synchronize()
        return ls, model.params.P.clone()

    plain_losses, plain_P = run()
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", "29731")
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    monkeypatch.setenv("DIC_FORCE_REDUCER", "1")
    try:
        dist.init_process_group("nccl", rank=0, world_size=1)
    except Exception as e:                                  # no usable rendezvous / RCCL on this box: nothing to compare
        pytest.skip(f"RCCL process group unavailable: {e}")
    try:
        dp_losses, dp_P = run()
    finally:
        dist.destroy_process_group()
    assert dp_losses == plain_losses
    assert torch.equal(dp_P, plain_P)


def test_reference_trainer_torch_adamw_also_works():
    z, m = load_golden("base_b4s3l16")
    model, x = build_model(m, "fp32", z)
    trainer = torch.optim.AdamW(model.parameters(), lr=m["lr"])
    for step in range(2):
        t, noises, u = draws(m, 123 + step)
        l, a, b, c = dic.train_func(model, trainer, x, train=True, t=t, noises=noises, cfg_uniform=u)
        np.testing.assert_allclose([f(l), f(a), f(b), f(c)], z["step_losses"][step], rtol=1e-4)


def test_validate_equals_a_hand_loop_of_eval_steps_on_the_same_seeds():
    """validate(model) (ref :488-501) = eval mode, no_grad, train_func(..., train=False) over the loader, the three loss terms averaged -- exactly:
    with the timestep and noise streams re-seeded, a hand-written loop gives bit-identical means; train mode is restored (ref :499)."""
    z, m = load_golden("base_b4s3l16")
    model, x = build_model(m, "fp32", z)
    x2 = {k: torch.from_numpy(v).cuda() for k, v in synth.batch(m["B"], m["L"], m["vocab"], m["dseed"] + 1).items()}
    loader = [x, x2, x]
    dic.set_loaders(val_loader=loader, trainer=None)
    dic.seed_noise(5)
    dic.seed_timesteps(77)
    vt, v1, vp = dic.validate(model)
    assert model.training
    dic.seed_noise(5)
    dic.seed_timesteps(77)
    model.eval()
    acc = [0.0, 0.0, 0.0]
    with torch.no_grad():
        for xb in loader:
            _, a, b, c = dic.train_func(model, None, xb, train=False)
            acc = [acc[0] + a.clone(), acc[1] + b.clone(), acc[2] + c.clone()]
    model.train()
    assert [f(vt), f(v1), f(vp)] == [f(acc[0] / 3), f(acc[1] / 3), f(acc[2] / 3)]
    # ... and a different timestep seed gives different draws (the seeds are what the streams hang on)
    dic.seed_noise(5)
    dic.seed_timesteps(78)
    wt, _, _ = dic.validate(model, loader)
    assert f(wt) != f(vt)
    # the oracle on the same injected draws agrees to the parity tolerance (one batch)
    t, noises, u = draws(m, 123)
    with torch.no_grad():
        model.eval()
        got = [f(v) for v in dic.train_func(model, None, x, train=False, t=t, noises=noises, cfg_uniform=u)]
        model.train()
    np.testing.assert_allclose(got, z["eval_losses"], rtol=1e-4)


def test_bench_shape_eval_losses_match_the_oracle_fp32_and_bf16():
    """BASELINE config 2's own size -- B=512, S=1, L=16, 12 layers, linear T=100 -- against the CPU oracle (oracle/ref_model.py, forward only:
    ~1-2 minutes on the GPU box's host cores): the fp32 engine, the DEFAULT bf16 engine (mean-row lo-weight correction + centred residual
    stream: the benchmarked mode) and its exact form "bf16w" (hi + lo bf16 weights as two K-loop passes) within the north-star tolerance 1e-4;
    the raw engine "bf16r" within 3e-3 with its deltas printed (what separates it from fp32 is the rounding of the WEIGHTS, one perturbation
    shared by all samples that a batch mean does not average out: profiles/r04_weight_rounding_probe.txt)."""
    B, S, L, V, nl = 512, 1, 16, 30522, 12
    dic.cfg.update(BATCH_SIZE=B, SAMPLE_SIZE=S, MAX_LENGTH=L, STEP_TOT=100, COSIN_SCHEDULE=False, ROUNDING_WEIGHT=0.5, VOCAB_SIZE=V,
                   LOSS_FUNC="series_sum_sample_mean", CLIP_ADDING_METHOD="concat", CLASSIFIER_FREE_WEIGHT=0.0, X_0_PREDICTION=True)
    dic.set_alpha_cumprod(None)
    E = synth.vocab_embedding(V, 768, 0)
    state = synth.denoiser_state(nl, 0)
    xb = synth.batch(B, L, V, 1)
    t = torch.from_numpy(synth.timesteps(S, 100, 0))
    nz = [torch.from_numpy(synth.noise((B, L, 768), 3, f"eps{i}")) for i in range(2)]
    rcfg = R.Config(BATCH_SIZE=B, SAMPLE_SIZE=S, MAX_LENGTH=L, STEP_TOT=100, COSIN_SCHEDULE=False, n_layers=nl, vocab=V)
    om = R.build(rcfg, state, E, requires_grad=False)
    with torch.no_grad():
        ref = np.array([float(v) for v in R.train_func(om, None, {k: torch.from_numpy(v) for k, v in xb.items()}, train=False, t=t, noises=nz)])
    del om
    x = {k: torch.from_numpy(v).cuda() for k, v in xb.items()}
    for dtype, tol in (("fp32", 1e-4), ("bf16w", 1e-4), ("bf16", 1e-4), ("bf16r", 3e-3)):
        model = dic.DistilBertModel(E, E, config=dict(n_layers=nl, dropout=0.0, attention_dropout=0.0), dtype=dtype)
        model.load_state(state)
        model.eval()
        with torch.no_grad():
            got = np.array([f(v) for v in dic.train_func(model, None, x, train=False, t=t, noises=nz)])
        rel = np.abs(got - ref) / np.abs(ref)
        print(f"bench shape, {dtype} vs oracle: losses {got} ref {ref} rel {rel}")
        assert rel.max() < tol, (dtype, rel)
        del model
        torch.cuda.empty_cache()


def test_bf16_engines_stay_near_fp32_along_a_training_run():
    """north_star's loss tolerance ALONG a run, not only at the initial weights (scripts/experiments/split_alloc_probe.py --trajectory and
    collapse_probe.py, profiles/r04_split_alloc_trajectory*.txt, r04_collapse_probe.txt): the split-weight engine trains at the bench shape on 8
    cycled batches; at 0 / 5 / 20 / 40 steps the eval losses of the bf16w and the plain bf16 engine on two held-out batches x three noise /
    timestep draws are compared with the fp32 HIP engine on the same weights (that engine is pinned to the CPU oracle at this shape by
    test_bench_shape_eval_losses_match_the_oracle_fp32_and_bf16: the oracle itself costs minutes per evaluation here).
    While the denoiser's output is (half-)collapsed onto one row -- the first hundreds of steps -- every bf16 rounding of a row-common quantity
    is the same for all tokens and does not average out of the batch-mean L1 terms.  Hence: the MLM-head pre-activation in fp32 (DIC_U_F32)
    and the mean-centred rounding-head input in every bf16 engine; the residual stream centred on predicted mean rows (dic_ln_fwd_cen) in the
    default engine, in fp32 in its exact form.  bf16w (the lo weight halves as a second K-loop pass) and the default bf16 (only their row-common
    part, as a bias: dic_lin_prep): inside 1e-4 at every state; the raw engine bf16r: 2.7e-4 at the initial weights (the weights' rounding),
    0.2-3e-4 along the run (reported, bound 5e-4)."""
    B, S, L, V, nl = 512, 1, 16, 30522, 12
    dic.cfg.update(BATCH_SIZE=B, SAMPLE_SIZE=S, MAX_LENGTH=L, STEP_TOT=100, COSIN_SCHEDULE=False, ROUNDING_WEIGHT=0.5, VOCAB_SIZE=V,
                   LOSS_FUNC="series_sum_sample_mean", CLIP_ADDING_METHOD="concat", CLASSIFIER_FREE_WEIGHT=0.0, X_0_PREDICTION=True)
    dic.set_alpha_cumprod(None)
    E = synth.vocab_embedding(V, 768, 0)
    kw = dict(config=dict(n_layers=nl, dropout=0.0, attention_dropout=0.0))
    f32, bw, bm, b16 = (dic.DistilBertModel(E, E, dtype=d, **kw) for d in ("fp32", "bf16w", "bf16", "bf16r"))
    assert bw.uvt32 and bw.head_centered and bw.res32 and b16.uvt32 and b16.head_centered and not b16.res32 and not b16.split_w and not b16.cen
    assert bm.lo_mode == "mean" and bm.cen and not bm.res32 and bw.lo_mode == "pass2" and not bw.cen
    held = [{k: torch.from_numpy(v).cuda() for k, v in synth.batch(B, L, V, 1 + 7 * i).items()} for i in range(2)]
    train = [{k: torch.from_numpy(v).cuda() for k, v in synth.batch(B, L, V, 100 + i).items()} for i in range(8)]
    draws = [(torch.from_numpy(synth.timesteps(S, 100, i)), [torch.from_numpy(synth.noise((B, L, 768), 3 + i, f"eps{j}")) for j in range(2)]) for i in range(3)]

    def evals(m):
        m.eval()
        out = []
        with torch.no_grad():
            for x in held:
                for t, nz in draws:
                    out.append([f(v) for v in dic.train_func(m, None, x, train=False, t=t, noises=nz)])
        return np.array(out)

    trainer = dic.AdamW(bw.parameters(), lr=1e-4)
    dic.seed_noise(1234)
    dic.diffusion.seed_timesteps(4321)
    done = 0
    for upto in (0, 5, 20, 40):
        bw.train()
        while done < upto:
            dic.train_func(bw, trainer, train[done % 8])
            done += 1
        state = bw.state_dict()
        f32.load_state_dict(state)
        b16.load_state_dict(state)
        bm.load_state_dict(state)
        ref = evals(f32)
        rel_w = (np.abs(evals(bw) - ref) / np.abs(ref)).max(0)
        rel_m = (np.abs(evals(bm) - ref) / np.abs(ref)).max(0)
        rel_b = (np.abs(evals(b16) - ref) / np.abs(ref)).max(0)
        print(f"after {done} steps: fp32 losses {ref[0]}; worst rel (total, x_t, x_1, prob) bf16w {rel_w}  bf16 {rel_m}  bf16r {rel_b}")
        assert rel_w.max() < 1e-4, (done, rel_w)
        assert rel_m.max() < 1e-4, (done, rel_m)
        assert rel_b.max() < 5e-4, (done, rel_b)                  # (reported, not claimed: 2.7e-4 at the initial weights, 0.2-2.5e-4 along the run)


def test_gradients_and_adamw_step_match_the_oracle_at_12_layers():
    """The benchmarked DEPTH against the CPU oracle, backward included (round-3 review: gradients were pinned to the reference at <= 6 layers
    only): B=16, S=1, L=16, 12 layers, full vocabulary, dropout off.  fp32 engine: losses 1e-4, every tensor's gradient norm 2e-3 and its
    first elements, parameters after the AdamW step 2e-6 -- the bars of the golden-fixture tests; the bf16 / split-weight engines: gradient
    norms within 2e-2 of the oracle's, tensor by tensor (what the data-parallel exchange and AdamW consume)."""
    B, S, L, V, nl = 16, 1, 16, 30522, 12
    dic.cfg.update(BATCH_SIZE=B, SAMPLE_SIZE=S, MAX_LENGTH=L, STEP_TOT=100, COSIN_SCHEDULE=False, ROUNDING_WEIGHT=0.5, VOCAB_SIZE=V,
                   LOSS_FUNC="series_sum_sample_mean", CLIP_ADDING_METHOD="concat", CLASSIFIER_FREE_WEIGHT=0.0, X_0_PREDICTION=True)
    dic.set_alpha_cumprod(None)
    E = synth.vocab_embedding(V, 768, 0)
    state = synth.denoiser_state(nl, 0)
    xb = synth.batch(B, L, V, 1)
    t = torch.from_numpy(synth.timesteps(S, 100, 5))
    nz = [torch.from_numpy(synth.noise((B, L, 768), 5, f"eps{i}")) for i in range(2)]
    rcfg = R.Config(BATCH_SIZE=B, SAMPLE_SIZE=S, MAX_LENGTH=L, STEP_TOT=100, COSIN_SCHEDULE=False, n_layers=nl, vocab=V)
    om = R.build(rcfg, state, E)
    otr = R.AdamW(om.parameters(), lr=1e-4)
    ol = np.array([float(v) for v in R.train_func(om, otr, {k: torch.from_numpy(v) for k, v in xb.items()}, t=t, noises=nz)])
    ograd = {n: p.grad.detach().clone() for n, p in om.p.items()}
    oparam = {n: p.detach().clone() for n, p in om.p.items()}
    x = {k: torch.from_numpy(v).cuda() for k, v in xb.items()}
    for dtype in ("fp32", "bf16w", "bf16", "bf16r"):
        model = dic.DistilBertModel(E, E, config=dict(n_layers=nl, dropout=0.0, attention_dropout=0.0), dtype=dtype)
        model.load_state(state)
        trainer = dic.AdamW(model.parameters(), lr=1e-4)
        got = np.array([f(v) for v in dic.train_func(model, trainer, x, t=t, noises=nz)])
        names = [n for n, _ in model.named_parameters()]
        keep = [n for n in names if not n.endswith("k_lin.bias")]           # analytically zero gradient (softmax shift invariance)
        grads = dict(zip(names, (p.grad for p in model.parameters())))
        rel_l = np.abs(got - ol) / np.abs(ol)
        gn = np.array([float(grads[n].double().norm()) for n in keep])
        on = np.array([float(ograd[n].double().norm()) for n in keep])
        rel_g = np.abs(gn - on) / (on + 1e-12)
        cos = min(float(torch.nn.functional.cosine_similarity(grads[n].flatten().double().cpu(), ograd[n].flatten().double(), dim=0)) for n in keep if float(ograd[n].norm()) > 1e-7)
        print(f"12 layers, {dtype}: loss rel {rel_l}, worst grad-norm rel {rel_g.max():.2e} ({keep[int(rel_g.argmax())]}), worst cosine {cos:.6f}")
        if dtype == "fp32":
            assert rel_l.max() < 1e-4
            np.testing.assert_allclose(gn, on, rtol=2e-3, atol=1e-6)
            for n in keep:
                a, b = grads[n].flatten()[:8].cpu().numpy(), ograd[n].flatten()[:8].numpy()
                assert np.abs(a - b).max() <= 2e-2 * (np.abs(b).max() + 1e-8), n
            params = dict(model.named_parameters())
            pn = np.array([float(params[n].double().norm()) for n in keep])
            opn = np.array([float(oparam[n].double().norm()) for n in keep])
            np.testing.assert_allclose(pn, opn, rtol=2e-6)
            assert cos > 0.9999
        else:
            # (the default bf16 engine at this small batch: the token-specific part of the weights' rounding, which its mean-row correction leaves
            # in, averages over 544 tokens instead of 17 408 -- measured 2-6e-5, held to the same 1e-4 as at the bench shape)
            assert rel_l.max() < {"bf16w": 1e-4, "bf16": 1e-4}.get(dtype, 3e-3), (dtype, rel_l)
            assert rel_g.max() < 2e-2 and cos > 0.995, (dtype, rel_g.max(), cos)
        del model, trainer
        torch.cuda.empty_cache()


def test_trained_collapsed_state_in_front_of_the_oracle():
    """Round-4 review, missing #3: the along-a-run parity claim compared the bf16 engines with the HIP fp32 engine, which itself had met the CPU
    oracle only at synthetic-init weights and two AdamW steps.  Here the fp32 HIP engine TRAINS 40 steps (B=16, 12 layers, lr 1e-4, dropout off,
    4 cycled batches) -- into the regime where the denoiser's rows have collapsed onto one vector (|mean row| >> rms distance from it), the one
    that broke the bf16 engines in round 4 -- and its state goes into oracle/ref_model.py: eval losses on a held-out batch (1e-4), the
    per-tensor gradient norms of step 41 (2e-3), and the token ids of a 5-pass sampling loop (identical beyond the decision bound).  The
    exact form bf16w evaluates the same state against the ORACLE within 1e-4, the default bf16 engine within 3e-4 (this batch is 32x smaller than
    the bench's: see the comment at the assertion)."""
    B, S, L, V, nl = 16, 1, 16, 30522, 12
    dic.cfg.update(BATCH_SIZE=B, SAMPLE_SIZE=S, MAX_LENGTH=L, STEP_TOT=100, COSIN_SCHEDULE=False, ROUNDING_WEIGHT=0.5, VOCAB_SIZE=V,
                   LOSS_FUNC="series_sum_sample_mean", CLIP_ADDING_METHOD="concat", CLASSIFIER_FREE_WEIGHT=0.0, X_0_PREDICTION=True)
    dic.set_alpha_cumprod(None)
    E = synth.vocab_embedding(V, 768, 0)
    kw = dict(config=dict(n_layers=nl, dropout=0.0, attention_dropout=0.0))
    m32 = dic.DistilBertModel(E, E, dtype="fp32", **kw)
    m32.load_state(synth.denoiser_state(nl, 0))
    trainer = dic.AdamW(m32.parameters(), lr=1e-4)
    batches = [synth.batch(B, L, V, 300 + i) for i in range(4)]
    dic.seed_noise(99)
    dic.diffusion.seed_timesteps(77)
    for i in range(40):
        dic.train_func(m32, trainer, {k: torch.from_numpy(v).cuda() for k, v in batches[i % 4].items()})
    state = {k: v.detach().cpu().numpy().copy() for k, v in m32.state_dict().items()}
    held = synth.batch(B, L, V, 9)
    t = torch.from_numpy(synth.timesteps(S, 100, 5))
    nz = [torch.from_numpy(synth.noise((B, L, 768), 21, f"eps{i}")) for i in range(2)]
    rcfg = R.Config(BATCH_SIZE=B, SAMPLE_SIZE=S, MAX_LENGTH=L, STEP_TOT=100, COSIN_SCHEDULE=False, n_layers=nl, vocab=V)
    om = R.build(rcfg, state, E)
    otr = R.AdamW(om.parameters(), lr=1e-4)
    xh = {k: torch.from_numpy(v) for k, v in held.items()}
    # the collapse is real: the oracle's own encoder output on the held-out batch
    with torch.no_grad():
        oe = np.array([float(v) for v in R.train_func(om, None, xh, train=False, t=t, noises=nz)])
    ol = np.array([float(v.detach()) for v in R.train_func(om, otr, xh, t=t, noises=nz)])           # step 41 on the oracle (same batch, t, noise)
    ograd = {n: p.grad.detach().clone() for n, p in om.p.items()}
    np.testing.assert_allclose(ol, oe, rtol=1e-6)
    x = {k: torch.from_numpy(v).cuda() for k, v in held.items()}
    m32.eval()
    with torch.no_grad():
        ge = np.array([f(v) for v in dic.train_func(m32, None, x, train=False, t=t, noises=nz)])
    rel = np.abs(ge - oe) / np.abs(oe)
    print(f"trained 40 steps: oracle eval losses {oe}, fp32 engine rel {rel}")
    assert oe[1] < 4.0, "the run must have left the initial regime (x_t loss 12 at the synthetic init)"
    assert rel.max() < 1e-4, rel
    m32.train()
    got = np.array([f(v) for v in dic.train_func(m32, _NoStep(m32), x, t=t, noises=nz)])
    assert (np.abs(got - ol) / np.abs(ol)).max() < 1e-4
    names = [n for n, _ in m32.named_parameters()]
    keep = [n for n in names if not n.endswith("k_lin.bias")]
    grads = dict(zip(names, (p.grad for p in m32.parameters())))
    gn = np.array([float(grads[n].double().norm()) for n in keep])
    on = np.array([float(ograd[n].double().norm()) for n in keep])
    big = on > 1e-5                                # (the query / key gradients of a collapsed denoiser are ~1e-8: attention no longer depends on them)
    worst = np.abs(gn - on)[big] / on[big]
    print(f"step-41 gradient norms ({int(big.sum())} tensors above 1e-5): worst rel {worst.max():.2e}")
    np.testing.assert_allclose(gn, on, rtol=2e-3, atol=1e-7)
    # sampling from the trained state: ids against the oracle's loop
    img = torch.from_numpy(held["image_clip"])
    start = torch.from_numpy(synth.noise((B, L + 2, 768), 23, "restored"))
    om2 = R.build(R.Config(BATCH_SIZE=B, MAX_LENGTH=L, n_layers=nl, vocab=V), state, E, requires_grad=False)
    oids, ohid = R.sample(om2, img, steps=5, start=start)
    m32.eval()
    ids, hid = dic.sample(m32, img, steps=5, start=start, return_hidden=True)
    c = ohid[:, :L].reshape(-1, 768).double()
    print(f"collapse of the sampled hidden state: |mean row| {float(c.mean(0).norm()):.3f}, rms |row - mean row| {float((c - c.mean(0)).norm(dim=1).pow(2).mean().sqrt()):.4f}")
    Et = torch.from_numpy(E).double()
    top2 = (ohid[:, :L].double() @ Et.t()).topk(2, -1).values
    margin = (top2[..., 0] - top2[..., 1]).numpy()
    drift = float((hid.cpu()[:, :L].double() - ohid[:, :L].double()).norm(dim=-1).max())
    bound = 2.0 * drift * float(Et.norm(dim=-1).max())
    same = ids.cpu().numpy() == oids.numpy()
    print(f"sampled ids vs oracle: agreement {float(same.mean()):.4f}, hidden drift {drift:.2e}, decision bound {bound:.2e}, tokens above it {int((margin > bound).sum())} / {margin.size}")
    assert bool(same[margin > bound].all()) and drift < 5e-3
    # the bf16 engines on the same trained state, against the ORACLE
    for dtype in ("bf16", "bf16w", "bf16r"):
        mb = dic.DistilBertModel(E, E, dtype=dtype, **kw)
        mb.load_state(state)
        mb.eval()
        with torch.no_grad():
            gb = np.array([f(v) for v in dic.train_func(mb, None, x, train=False, t=t, noises=nz)])
        relb = np.abs(gb - oe) / np.abs(oe)
        print(f"trained state, {dtype} vs oracle: rel {relb}")
        # (the default engine at SIXTEEN captions: its corrections remove what is common to all rows, the rest averages over 544 token rows instead of
        # 17 408 -- over batch sizes 16 / 64 / 512 x 5 / 20 / 40 training steps the L1 terms land between 1e-5 and 2e-4 at 16 captions, at or below
        # 1.3e-4 from 64 up, profiles/r05_collapsed_state_matrix.txt; the bench-shape tests hold it to 1e-4)
        assert relb.max() < {"bf16w": 1e-4, "bf16": 3e-4}.get(dtype, 3e-3), (dtype, relb)
        del mb
        torch.cuda.empty_cache()


def test_sampling_matches_the_oracle_at_12_layers_20_passes():
    """Config 4's loop at the benchmarked depth against the ORACLE's loop (round-3 review: the fp32 engine's sampling was pinned to the reference
    at 2 layers x 3 passes only): 32 images, 12 layers, 20 feedback passes from the same start noise.  fp32 engine: the hidden state after 20
    passes within 2e-3 and every token id whose oracle top-2 logit margin exceeds what that drift can flip identical (all of them in
    practice); bf16 / bf16w: agreement reported per margin bucket, ids beyond the decision bound identical."""
    B, L, V, nl, steps = 32, 16, 30522, 12, 20
    dic.cfg.update(MAX_LENGTH=L, CLASSIFIER_FREE_WEIGHT=0.0, CLIP_ADDING_METHOD="concat", VOCAB_SIZE=V)
    dic.set_alpha_cumprod(None)
    E = synth.vocab_embedding(V, 768, 0)
    state = synth.denoiser_state(nl, 0)
    img = torch.from_numpy(synth.batch(B, L, V, 2)["image_clip"])
    start = torch.from_numpy(synth.noise((B, L + 2, 768), 17, "restored"))
    rcfg = R.Config(BATCH_SIZE=B, MAX_LENGTH=L, n_layers=nl, vocab=V)
    om = R.build(rcfg, state, E, requires_grad=False)
    oids, ohid = R.sample(om, img, steps=steps, start=start)
    Et = torch.from_numpy(E).double()
    lg = ohid[:, :L].double() @ Et.t()
    top2 = lg.topk(2, -1).values
    margin = (top2[..., 0] - top2[..., 1]).numpy()
    wmax = float(Et.norm(dim=-1).max())
    for dtype in ("fp32", "bf16w", "bf16"):
        model = dic.DistilBertModel(E, E, config=dict(n_layers=nl), dtype=dtype)
        model.load_state(state)
        model.eval()
        ids, hid = dic.sample(model, img, steps=steps, start=start, return_hidden=True)
        same = ids.cpu().numpy() == oids.numpy()
        drift = float((hid.cpu()[:, :L].double() - ohid[:, :L].double()).norm(dim=-1).max())
        bound = 2.0 * drift * wmax
        for lo, hi in ((0, 0.01), (0.01, 0.03), (0.03, 0.1), (0.1, 1e9)):
            sel = (margin >= lo) & (margin < hi)
            if sel.any():
                print(f"{dtype} sampling vs oracle, margin [{lo}, {hi}): {int(sel.sum())} tokens, agreement {float(same[sel].mean()):.3f}")
        print(f"{dtype}: max row drift of the hidden state {drift:.3e}, id agreement {float(same.mean()):.4f}, decision bound {bound:.3e}")
        assert bool(same[margin > bound].all())
        if dtype == "fp32":
            assert float((hid.cpu() - ohid).abs().max()) < 2e-3 and float(same.mean()) > 0.995
        else:
            assert float(same.mean()) > 0.5
        del model
        torch.cuda.empty_cache()


# ------------------------------------------------------------------------------------------------ BASELINE full sizes: properties
class _NoStep:
    """Trainer stand-in: keeps train_func's backward but leaves the parameters alone."""
    def __init__(self, model): self.model = model
    def zero_grad(self): self.model.params.G.zero_()
    def step(self): pass


def _grads_for(model, x, B, t, noises):
    dic.cfg.update(BATCH_SIZE=B)
    l, *_ = dic.train_func(model, _NoStep(model), x, t=t, noises=noises)
    return model.params.G.clone(), f(l)


@pytest.mark.parametrize("dtype,n_layers,tol", [("bf16", 12, 3e-2), ("fp32", 2, 2e-4)])
def test_full_size_gradient_is_mean_of_shard_gradients(dtype, n_layers, tol):
    """Config 2/3 shape (512 captions x 16 tokens): the gradient of the global batch equals the mean of the gradients of its two
    halves (same t, per-item noise) -- the property the single RCCL all-reduce of data parallelism relies on."""
    B, L, V = 512, 16, 30522
    dic.cfg.update(BATCH_SIZE=B, SAMPLE_SIZE=1, MAX_LENGTH=L, STEP_TOT=100, COSIN_SCHEDULE=False, ROUNDING_WEIGHT=0.5,
                   LOSS_FUNC="series_sum_sample_mean", CLIP_ADDING_METHOD="concat", CLASSIFIER_FREE_WEIGHT=0.0, X_0_PREDICTION=True, VOCAB_SIZE=V)
    dic.set_alpha_cumprod(None)
    E = synth.vocab_embedding(V, 768, 0)
    model = dic.DistilBertModel(E, E, config=dict(n_layers=n_layers, dropout=0.0, attention_dropout=0.0), dtype=dtype)
    model.load_state(synth.denoiser_state(n_layers, 0))
    x = {k: torch.from_numpy(v).cuda() for k, v in synth.batch(B, L, V, 3).items()}
    t = torch.tensor([[[37]]])
    nz = [torch.from_numpy(synth.noise((B, L, 768), 11, f"eps{i}")) for i in range(2)]
    g_full, l_full = _grads_for(model, x, B, t, nz)
    h = B // 2
    halves = []
    for s in (slice(0, h), slice(h, B)):
        xs = {k: v[s] for k, v in x.items()}
        halves.append(_grads_for(model, xs, h, t, [n[s] for n in nz]))
    g_mean = 0.5 * (halves[0][0] + halves[1][0])
    l_mean = 0.5 * (halves[0][1] + halves[1][1])
    assert abs(l_full - l_mean) < 1e-3 * abs(l_full) if dtype == "bf16" else abs(l_full - l_mean) < 1e-5 * abs(l_full)
    err = float((g_full - g_mean).norm() / g_full.norm())
    print(dtype, "full-vs-shard-mean gradient rel err", err, "loss", l_full)
    assert err < tol and np.isfinite(l_full)


def test_rounding_loss_training_form_equals_the_recompute_path_at_the_bench_shape(monkeypatch):
    """Config 2 (B=512, 12 layers, bf16): the step whose rounding loss keeps exp(logit - c) in the forward (the default) against
    the step that recomputes the logits in the backward (DIC_CE_FUSED=0) on the same weights / batch / noise / timesteps: the same losses and,
    tensor by tensor, the same gradients up to the bf16 rounding both forms apply to their 16 384 x 30 522 gradient operand."""
    engine = importlib.import_module("diffusion-image-captioning_amd.engine")
    B, L, V, NL = 512, 16, 30522, 12
    dic.cfg.update(BATCH_SIZE=B, SAMPLE_SIZE=1, MAX_LENGTH=L, STEP_TOT=100, COSIN_SCHEDULE=False, ROUNDING_WEIGHT=0.5,
                   LOSS_FUNC="series_sum_sample_mean", CLIP_ADDING_METHOD="concat", CLASSIFIER_FREE_WEIGHT=0.0, X_0_PREDICTION=True, VOCAB_SIZE=V)
    dic.set_alpha_cumprod(None)
    E = synth.vocab_embedding(V, 768, 0)
    x = {k: torch.from_numpy(v).cuda() for k, v in synth.batch(B, L, V, 3).items()}
    t = torch.tensor([[[37]]])
    nz = [torch.from_numpy(synth.noise((B, L, 768), 11, f"eps{i}")) for i in range(2)]
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(engine.OPT, "ce_fused", fused)
        model = dic.DistilBertModel(E, E, config=dict(n_layers=NL, dropout=0.0, attention_dropout=0.0), dtype="bf16")
        model.load_state(synth.denoiser_state(NL, 0))
        assert model.ce_fused == fused
        out = dic.train_func(model, _NoStep(model), x, t=t, noises=nz)
        res[fused] = ([f(v) for v in out], model.params.G.clone(), {n: p.grad.float().norm().item() for n, p in zip(model.params.names, model.parameters())})
        del model
    (la, ga, na), (lb, gb, nb) = res[True], res[False]
    for a, b in zip(la, lb):
        assert abs(a - b) <= 2e-6 * abs(b), (la, lb)
    err = float((ga - gb).norm() / gb.norm())
    worst = max(abs(na[k] - nb[k]) / (nb[k] + 1e-12) for k in nb if nb[k] > 1e-6 and "k_lin.bias" not in k)
    print("fused vs recompute: losses", la, "gradient rel err", err, "worst per-tensor norm rel diff", worst)
    assert err < 2e-3 and worst < 2e-3


def test_full_size_training_is_deterministic_and_descends():
    """Config 2: B=512, 12 layers, bf16, dropout 0.1: same seeds -> bit-identical losses; loss decreases over steps."""
    B, L, V = 512, 16, 30522
    dic.cfg.update(BATCH_SIZE=B, SAMPLE_SIZE=1, MAX_LENGTH=L, STEP_TOT=100, COSIN_SCHEDULE=False, ROUNDING_WEIGHT=0.5,
                   LOSS_FUNC="series_sum_sample_mean", CLIP_ADDING_METHOD="concat", CLASSIFIER_FREE_WEIGHT=0.0, X_0_PREDICTION=True, VOCAB_SIZE=V)
    dic.set_alpha_cumprod(None)
    E = synth.vocab_embedding(V, 768, 0)
    x = {k: torch.from_numpy(v).cuda() for k, v in synth.batch(B, L, V, 1).items()}
    runs = []
    for rep in range(2):
        model = dic.DistilBertModel(E, E, config=dict(n_layers=12, dropout=0.1, attention_dropout=0.1), dtype="bf16", seed=1)
        trainer = dic.AdamW(model.parameters(), lr=1e-4)
        dic.seed_noise(7)
        ls = []
        for step in range(4):
            t = torch.from_numpy(synth.timesteps(1, 100, step))
            ls.append(f(dic.train_func(model, trainer, x, t=t)[0]))
        runs.append(ls)
    assert runs[0] == runs[1], runs
    assert all(np.isfinite(runs[0])) and runs[0][-1] < runs[0][0]


def test_config5_full_size_shard_mean_gradient_and_determinism():
    """BASELINE config 5 at ITS size -- B=512, seq_len 32 (+2 CLIP rows), classifier-free guidance p=0.2 w=0.3, 12 layers, bf16 -- through the
    size-independent properties config 2 is held to: the gradient of the global batch is the mean of the gradients of its two halves (same t,
    per-item noise and guidance draws; what the data-parallel all-reduce relies on), and two runs from the same seeds with dropout take
    bit-identical steps whose loss descends."""
    B, L, V, nl = 512, 32, 30522, 12
    dic.cfg.update(BATCH_SIZE=B, SAMPLE_SIZE=1, MAX_LENGTH=L, STEP_TOT=100, COSIN_SCHEDULE=False, ROUNDING_WEIGHT=0.5, CLASSIFIER_FREE_PROB=0.2,
                   LOSS_FUNC="series_sum_sample_mean", CLIP_ADDING_METHOD="concat", CLASSIFIER_FREE_WEIGHT=0.3, X_0_PREDICTION=True, VOCAB_SIZE=V)
    dic.set_alpha_cumprod(None)
    E = synth.vocab_embedding(V, 768, 0)
    try:
        model = dic.DistilBertModel(E, E, config=dict(n_layers=nl, dropout=0.0, attention_dropout=0.0), dtype="bf16")
        model.load_state(synth.denoiser_state(nl, 0))
        model.rank_rows_forced = False          # (the forced guidance rows 0/1 of ref :408-409 belong to a batch, not to a shard: off for the property)
        x = {k: torch.from_numpy(v).cuda() for k, v in synth.batch(B, L, V, 3).items()}
        t = torch.tensor([[[37]]])
        nz = [torch.from_numpy(synth.noise((B, L, 768), 11, f"eps{i}")) for i in range(2)]
        u = torch.from_numpy(synth.uniform(synth.stream_id("cfg", 11), (B, 1)))

        def grads(sl, b):
            dic.cfg.update(BATCH_SIZE=b)
            l, *_ = dic.train_func(model, _NoStep(model), {k: v[sl] for k, v in x.items()}, t=t, noises=[n[sl] for n in nz], cfg_uniform=u[sl])
            return model.params.G.clone(), f(l)
        g_full, l_full = grads(slice(0, B), B)
        h = B // 2
        (ga, la), (gb, lb) = grads(slice(0, h), h), grads(slice(h, B), h)
        err = float((g_full - 0.5 * (ga + gb)).norm() / g_full.norm())
        print("config 5 full size: full-vs-shard-mean gradient rel err", err, "loss", l_full, 0.5 * (la + lb))
        assert abs(l_full - 0.5 * (la + lb)) < 1e-3 * abs(l_full) and err < 3e-2 and np.isfinite(l_full)
        del model
        torch.cuda.empty_cache()
        dic.cfg.update(BATCH_SIZE=B)
        runs = []
        for rep in range(2):
            model = dic.DistilBertModel(E, E, config=dict(n_layers=nl, dropout=0.1, attention_dropout=0.1), dtype="bf16", seed=1)
            trainer = dic.AdamW(model.parameters(), lr=1e-4)
            dic.seed_noise(7)
            dic.seed_guidance(9)
            ls = []
            for step in range(3):
                tt = torch.from_numpy(synth.timesteps(1, 100, step))
                ls.append(f(dic.train_func(model, trainer, x, t=tt)[0]))
            runs.append(ls)
            del model, trainer
            torch.cuda.empty_cache()
        assert runs[0] == runs[1], runs
        assert all(np.isfinite(runs[0])) and runs[0][-1] < runs[0][0]
    finally:
        dic.cfg.update(CLASSIFIER_FREE_WEIGHT=0.0, MAX_LENGTH=16)


def test_graphed_and_eager_steps_interleave_in_any_order():
    """graph.GraphedTrainStep next to eager calls (advisor finding, round 3): an eager train_func / an eval-mode validate-style call between
    replays moves the host-side seed / optimizer / result-slot counters, so the next graphed call must RE-CAPTURE instead of replaying stale
    seeds; other batch shapes in between must not evict the workspaces the graph points into.  The mixed sequence must equal the all-eager one
    bit for bit (losses and parameters), and losses returned earlier must keep their values."""
    B, L, V, nl = 64, 16, 3000, 2
    dic.cfg.update(BATCH_SIZE=B, SAMPLE_SIZE=1, MAX_LENGTH=L, STEP_TOT=100, COSIN_SCHEDULE=False, ROUNDING_WEIGHT=0.5, VOCAB_SIZE=V,
                   LOSS_FUNC="series_sum_sample_mean", CLIP_ADDING_METHOD="concat", CLASSIFIER_FREE_WEIGHT=0.0, X_0_PREDICTION=True)
    dic.set_alpha_cumprod(None)
    E = synth.vocab_embedding(V, 768, 0)
    x = {k: torch.from_numpy(v).cuda() for k, v in synth.batch(B, L, V, 1).items()}
    xs_other = [{k: torch.from_numpy(v).cuda() for k, v in synth.batch(b, L, V, 2).items()} for b in (8, 24, 40)]
    plan = "GGEGGVGEEGOGG"          # G graphed step, E eager step, V eval-mode forward (validate-style), O eager eval calls on three OTHER batch shapes
    results = {}
    for mode in ("eager", "mixed"):
        model = dic.DistilBertModel(E, E, config=dict(n_layers=nl, dropout=0.1, attention_dropout=0.1), dtype="bf16", seed=3)
        trainer = dic.AdamW(model.parameters(), lr=1e-4)
        dic.seed_all(77)
        step = dic.GraphedTrainStep(model, trainer, x, warmup=1) if mode == "mixed" else None
        if mode == "eager":                                   # the graphed object's constructor ran its warm-up step eagerly: mirror it
            dic.train_func(model, trainer, x)
        losses, kept = [], []
        for c in plan:
            if c == "G" and step is not None:
                out = step()
            elif c in "GE":
                out = dic.train_func(model, trainer, x)
            elif c == "V":
                model.eval()
                with torch.no_grad():
                    out = dic.train_func(model, None, x, train=False)
                model.train()
            else:
                model.eval()
                for xo in xs_other:
                    dic.cfg.update(BATCH_SIZE=xo["input_ids"].shape[0])
                    with torch.no_grad():
                        out = dic.train_func(model, None, xo, train=False)
                dic.cfg.update(BATCH_SIZE=B)
                model.train()
            kept.append(out[0])
            losses.append(f(out[0]))
        torch.cuda.synchronize()
        assert [f(v) for v in kept] == losses, "a loss returned earlier changed its value"
        results[mode] = (losses, model.params.P.clone(), step.captures if step is not None else 0)
        if step is not None:
            step.release()
        del model, trainer
    (le, pe, _), (lm, pm, ncap) = results["eager"], results["mixed"]
    print("interleaved graph / eager steps:", lm, "captures", ncap)
    assert le == lm, (le, lm)
    assert torch.equal(pe, pm)
    assert ncap >= 4            # the initial capture + one after each block of foreign calls


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_graphed_training_step_is_bit_identical_to_eager(dtype):
    """The step captured once as a hipGraph and replayed (graph.GraphedTrainStep) against plain eager calls: dropout on, same seeds -- every loss
    of 14 steps and every parameter afterwards bit-identical; eager steps continue seamlessly after graphed ones (host counters in step)."""
    B, S, L, V, nl = 8, 2, 16, 1500, 2
    dic.cfg.update(BATCH_SIZE=B, SAMPLE_SIZE=S, MAX_LENGTH=L, STEP_TOT=100, COSIN_SCHEDULE=False, ROUNDING_WEIGHT=0.5, VOCAB_SIZE=V,
                   LOSS_FUNC="series_sum_sample_mean", CLIP_ADDING_METHOD="concat", CLASSIFIER_FREE_WEIGHT=0.0, X_0_PREDICTION=True)
    dic.set_alpha_cumprod(None)
    E = synth.vocab_embedding(V, 768, 0)
    x = {k: torch.from_numpy(v).cuda() for k, v in synth.batch(B, L, V, 1).items()}
    x2 = {k: torch.from_numpy(v).cuda() for k, v in synth.batch(B, L, V, 2).items()}
    def fresh():
        model = dic.DistilBertModel(E, E, config=dict(n_layers=nl, dropout=0.1, attention_dropout=0.1), dtype=dtype, seed=3)
        trainer = dic.AdamW(model.parameters(), lr=1e-3)
        dic.seed_noise(11)
        dic.seed_timesteps(500)
        return model, trainer
    batch = lambda i: x if (i < 2 or i % 2 == 0) else x2          # (the graphed run's two warm-up steps use its construction batch)
    model, trainer = fresh()
    step = dic.GraphedTrainStep(model, trainer, x, warmup=2)       # steps 0, 1 eagerly
    got = [[f(v) for v in step(batch(i))] for i in range(2, 12)]   # steps 2..11 replayed, batches alternating through the static input buffers
    got += [[f(v) for v in dic.train_func(model, trainer, batch(i))] for i in range(12, 14)]      # and two eager steps behind them
    assert step.captures == 1 and step.replays == 10
    P_graph = model.params.P.clone()
    model, trainer = fresh()
    ref = [[f(v) for v in dic.train_func(model, trainer, batch(i))] for i in range(14)]
    assert ref[2:] == got, "graphed losses differ from the eager ones"
    assert torch.equal(model.params.P, P_graph)
    assert len({tuple(r) for r in ref}) == 14                      # (every step drew fresh noise / masks / timesteps)
    # the workspaces a captured graph points into are pinned against the engine's eviction only while a GraphedTrainStep needs them (round-4
    # advisor: dropping the object used to leave multi-GB workspaces pinned for good)
    import gc
    model, trainer = fresh()
    s1 = dic.GraphedTrainStep(model, trainer, x, warmup=1)
    pinned = lambda: sum(1 for c_ in (model._ws, model._ce_ws) for w_ in c_.values() if w_.get("pinned"))
    assert pinned() == 2
    s2 = dic.GraphedTrainStep(model, trainer, x, warmup=0)         # a second graphed step on the same workspaces: pins are counted
    del s1
    gc.collect()
    assert pinned() == 2
    s2.release()
    assert pinned() == 0
    s3 = dic.GraphedTrainStep(model, trainer, x, warmup=0)
    assert pinned() == 2
    del s3
    gc.collect()
    assert pinned() == 0


def test_config5_seq32_guidance_bf16_matches_fp32():
    """Config 5 shape: seq_len 32 (+2 CLIP rows = 34 tokens, beyond one MFMA tile), classifier-free guidance p=0.2 w=0.3."""
    B, S, L, V = 8, 2, 32, 5000
    dic.cfg.update(BATCH_SIZE=B, SAMPLE_SIZE=S, MAX_LENGTH=L, STEP_TOT=100, COSIN_SCHEDULE=False, ROUNDING_WEIGHT=0.5, CLASSIFIER_FREE_PROB=0.2,
                   LOSS_FUNC="series_sum_sample_mean", CLIP_ADDING_METHOD="concat", CLASSIFIER_FREE_WEIGHT=0.3, X_0_PREDICTION=True, VOCAB_SIZE=V)
    dic.set_alpha_cumprod(None)
    E = synth.vocab_embedding(V, 768, 0)
    x = {k: torch.from_numpy(v).cuda() for k, v in synth.batch(B, L, V, 5).items()}
    t = torch.from_numpy(synth.timesteps(S, 100, 2))
    nz = [torch.from_numpy(synth.noise((B, L, 768), 4, f"eps{i}")) for i in range(2)]
    u = torch.from_numpy(synth.uniform(synth.stream_id("cfg", 4), (S * B, 1)))
    out = {}
    for dtype in ("fp32", "bf16r", "bf16", "bf16w"):
        model = dic.DistilBertModel(E, E, config=dict(n_layers=2, dropout=0.0, attention_dropout=0.0), dtype=dtype)
        model.load_state(synth.denoiser_state(2, 0))
        trainer = dic.AdamW(model.parameters(), lr=1e-4)
        out[dtype] = [np.array([f(v) for v in dic.train_func(model, trainer, x, t=t, noises=nz, cfg_uniform=u)]) for _ in range(2)]
    # fp32 against the CPU oracle
    rcfg = R.Config(BATCH_SIZE=B, SAMPLE_SIZE=S, MAX_LENGTH=L, STEP_TOT=100, COSIN_SCHEDULE=False, n_layers=2, vocab=V, CLASSIFIER_FREE_WEIGHT=0.3)
    om = R.build(rcfg, synth.denoiser_state(2, 0), E)
    otr = R.AdamW(om.parameters(), lr=1e-4)
    xo = {k: v.cpu() for k, v in x.items()}
    ref = [np.array([float(v) for v in R.train_func(om, otr, xo, t=t, noises=nz, cfg_uniform=u)]) for _ in range(2)]
    np.testing.assert_allclose(out["fp32"], ref, rtol=1e-4)
    np.testing.assert_allclose(out["bf16r"], ref, rtol=5e-3)
    for dtype in ("bf16", "bf16w"):                # the parity modes on the guided, 34-token path (16 captions: the small-batch bound)
        print(dtype, "config-5 shape, two steps, rel:", np.abs(np.array(out[dtype]) - np.array(ref)) / np.abs(np.array(ref)))
        np.testing.assert_allclose(out[dtype], ref, rtol=1e-3)


def test_sampling_is_batch_permutation_equivariant_at_config4_size():
    """Config 4 shape (batch 2048 images): every image is refined independently, so permuting the batch permutes the ids
    bit-for-bit (no kernel depends on a sequence's position in the batch)."""
    Bn, L, V = 2048, 16, 30522
    dic.cfg.update(MAX_LENGTH=L, CLASSIFIER_FREE_WEIGHT=0.0, CLIP_ADDING_METHOD="concat", VOCAB_SIZE=V)
    E = synth.vocab_embedding(V, 768, 0)
    model = dic.DistilBertModel(E, E, config=dict(n_layers=6), dtype="fp32")
    model.load_state(synth.denoiser_state(6, 0))
    model.eval()
    img = torch.from_numpy(synth.batch(Bn, L, V, 9)["image_clip"]).cuda()
    start = torch.from_numpy(synth.noise((Bn, L + 2, 768), 21, "restored")).cuda()
    perm = torch.from_numpy(np.random.RandomState(0).permutation(Bn)).cuda()
    ids = dic.sample(model, img, steps=4, start=start)
    ids_p = dic.sample(model, img[perm], steps=4, start=start[perm])
    assert torch.equal(ids[perm], ids_p)
    assert ids.shape == (Bn, L) and int(ids.min()) >= 0 and int(ids.max()) < V


# ------------------------------------------------------------------------------------------------ edge cases
def _tiny_cfg(B, S, L, V, **kw):
    base = dict(BATCH_SIZE=B, SAMPLE_SIZE=S, MAX_LENGTH=L, STEP_TOT=100, COSIN_SCHEDULE=False, ROUNDING_WEIGHT=0.5, LOSS_FUNC="series_sum_sample_mean",
                CLIP_ADDING_METHOD="concat", CLASSIFIER_FREE_WEIGHT=0.0, X_0_PREDICTION=True, VOCAB_SIZE=V, USE_X_T_LOSS=True, USE_X_1_LOSS=True,
                USE_PROB_LOSS=True)
    base.update(kw)
    dic.cfg.update(**base)
    dic.set_alpha_cumprod(None)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_minimum_batch_all_padding_and_vocab_edges(dtype):
    """B=1, S=1 (two sequences in the whole step: every GEMM is a single ragged tile), a caption that is ALL padding (only the CLIP image key
    is attendable), ids at both ends of the vocabulary."""
    B, S, L, V = 1, 1, 16, 1003
    _tiny_cfg(B, S, L, V)
    E = synth.vocab_embedding(V, 768, 0)
    x = {k: torch.from_numpy(v).cuda() for k, v in synth.batch(B, L, V, 4).items()}
    x["attention_mask"][:] = 0
    x["input_ids"][0, 0], x["input_ids"][0, 1] = 0, V - 1
    t = torch.tensor([[[99]]])
    nz = [torch.from_numpy(synth.noise((B, L, 768), 3, f"eps{i}")) for i in range(2)]
    model = dic.DistilBertModel(E, E, config=dict(n_layers=2, dropout=0.0, attention_dropout=0.0), dtype=dtype)
    model.load_state(synth.denoiser_state(2, 0))
    got = np.array([f(v) for v in dic.train_func(model, dic.AdamW(model.parameters(), lr=1e-4), x, t=t, noises=nz)])
    rcfg = R.Config(BATCH_SIZE=B, SAMPLE_SIZE=S, MAX_LENGTH=L, STEP_TOT=100, COSIN_SCHEDULE=False, n_layers=2, vocab=V)
    om = R.build(rcfg, synth.denoiser_state(2, 0), E)
    ref = np.array([float(v) for v in R.train_func(om, R.AdamW(om.parameters(), lr=1e-4), {k: v.cpu() for k, v in x.items()}, t=t, noises=nz)])
    np.testing.assert_allclose(got, ref, rtol=1e-4 if dtype == "fp32" else 5e-3)


def test_loss_switches_and_direct_loss_call_signature():
    """`loss(model, x_t, x_1, x_tgt, x_0, image_clip, text_clip, mask, idx, loss_func)` called the way the reference's train_func calls it,
    with the USE_* switches (ref :112-114, 416-443) and a loss function passed as a callable named like the reference's."""
    B, S, L, V = 3, 2, 16, 800
    _tiny_cfg(B, S, L, V)
    E = synth.vocab_embedding(V, 768, 0)
    model = dic.DistilBertModel(E, E, config=dict(n_layers=1, dropout=0.0, attention_dropout=0.0), dtype="fp32")
    model.load_state(synth.denoiser_state(1, 0))
    x = {k: torch.from_numpy(v).cuda() for k, v in synth.batch(B, L, V, 6).items()}
    t = torch.from_numpy(synth.timesteps(S, 100, 1)).cuda()
    x_0 = model.embedding(x["input_ids"])
    nz = [torch.from_numpy(synth.noise((B, L, 768), 8, f"eps{i}")) for i in range(2)]
    x_t, x_tgt = dic.generate_diffuse_pair(x_0, t, noises=(nz[0], None))
    assert x_tgt is x_0 and x_t.shape == (S * B, L, 768)
    x_1 = dic.diffuse_t(x_0, torch.ones(1, dtype=torch.int64), noise=nz[1])

    def series_sum_sample_mean(a, b):          # same __name__ as the reference's LOSS_FUNC
        raise AssertionError("never called: the name selects the HIP loss kernel")
    with torch.no_grad():
        full = [f(v) for v in dic.loss(model, x_t, x_1, None, x_0, x["image_clip"], x["text_clip"], x["attention_mask"], x["input_ids"], series_sum_sample_mean)]
        dic.cfg.update(USE_X_T_LOSS=False, USE_PROB_LOSS=False)
        try:
            part = [f(v) for v in dic.loss(model, x_t, x_1, None, x_0, x["image_clip"], x["text_clip"], x["attention_mask"], x["input_ids"], "series_sum_sample_mean")]
        finally:
            dic.cfg.update(USE_X_T_LOSS=True, USE_PROB_LOSS=True)
    assert part[0] == 0.0 and part[2] == 0.0 and part[1] == pytest.approx(full[1], rel=1e-6)
    rcfg = R.Config(BATCH_SIZE=B, SAMPLE_SIZE=S, MAX_LENGTH=L, STEP_TOT=100, COSIN_SCHEDULE=False, n_layers=1, vocab=V)
    om = R.build(rcfg, synth.denoiser_state(1, 0), E, requires_grad=False)
    ref = R.loss(om, x_t.cpu(), x_1.cpu(), None, x_0.cpu(), x["image_clip"].cpu(), x["text_clip"].cpu(), x["attention_mask"].cpu(), x["input_ids"].cpu())
    np.testing.assert_allclose(full, [float(v) for v in ref], rtol=1e-4)
    # shape errors surface as AssertionError, as in the reference (ref :396-400)
    with pytest.raises(AssertionError):
        dic.loss(model, x_t[:-1], x_1, None, x_0, x["image_clip"], x["text_clip"], x["attention_mask"], x["input_ids"], "series_sum_sample_mean")
    with pytest.raises(NotImplementedError):
        dic.loss(model, x_t, x_1, None, x_0, x["image_clip"], x["text_clip"], x["attention_mask"], x["input_ids"], "huber")
    # API parity of the small pieces
    logits = model.lm_head(x_0)
    assert logits.shape == (B, L, V) and torch.equal(logits.argmax(-1).cpu(), x["input_ids"].cpu())     # tied head: E[id] . E^T peaks at id


def test_maximum_sequence_length_64_tokens():
    """L = 62 (+2 CLIP rows = 64 tokens): the attention kernels' upper bound."""
    B, S, L, V = 2, 1, 62, 600
    _tiny_cfg(B, S, L, V)
    E = synth.vocab_embedding(V, 768, 0)
    x = {k: torch.from_numpy(v).cuda() for k, v in synth.batch(B, L, V, 7).items()}
    t = torch.tensor([[[10]]])
    nz = [torch.from_numpy(synth.noise((B, L, 768), 2, f"eps{i}")) for i in range(2)]
    rcfg = R.Config(BATCH_SIZE=B, SAMPLE_SIZE=S, MAX_LENGTH=L, STEP_TOT=100, COSIN_SCHEDULE=False, n_layers=1, vocab=V, CLASSIFIER_FREE_WEIGHT=0.5)
    dic.cfg.update(CLASSIFIER_FREE_WEIGHT=0.5)
    u = torch.tensor([[0.9], [0.9]])
    om = R.build(rcfg, synth.denoiser_state(1, 0), E)
    ref = np.array([float(v) for v in R.train_func(om, R.AdamW(om.parameters(), lr=1e-4), {k: v.cpu() for k, v in x.items()}, t=t, noises=nz, cfg_uniform=u)])
    for dtype, tol in (("fp32", 1e-4), ("bf16", 5e-3)):
        model = dic.DistilBertModel(E, E, config=dict(n_layers=1, dropout=0.0, attention_dropout=0.0), dtype=dtype)
        model.load_state(synth.denoiser_state(1, 0))
        got = np.array([f(v) for v in dic.train_func(model, dic.AdamW(model.parameters(), lr=1e-4), x, t=t, noises=nz, cfg_uniform=u)])
        np.testing.assert_allclose(got, ref, rtol=tol)
    dic.cfg.update(CLASSIFIER_FREE_WEIGHT=0.0)


def test_optional_timestep_embedding_end_to_end_fp32():
    """cfg.TIMESTEP_EMBEDDING (off in every parity configuration: the reference's denoiser takes no t, ref :271).  With a zero table the
    step is bit-identical to the model without it; with a random table the gradient the kernels return for it matches a central finite
    difference of the loss along a random direction, rows of unused timesteps get exactly zero, and AdamW moves the used rows."""
    m = dict(B=4, S=3, L=16, step_tot=50, cosine=False, rounding_weight=0.5, loss="series_sum_sample_mean", fusion="concat", cfg_w=0.0,
             cfg_prob=0.2, x0_pred=True, x_t_step_interval=100, vocab=1200, n_layers=1, wseed=0, dseed=1)
    t, noises, u = draws(m, 5)
    try:
        model0, x = build_model(m, "fp32")
        model0.eval()
        with torch.no_grad():
            ref = [f(v) for v in dic.train_func(model0, None, x, train=False, t=t, noises=noises)]
        dic.cfg.update(TIMESTEP_EMBEDDING=True)
        model, x = build_model(m, "fp32")
        assert [n for n, _ in model.named_parameters()][-1] == "timestep_embedding.weight"
        table = dict(model.named_parameters())["timestep_embedding.weight"]
        assert table.shape == (m["step_tot"], 768)
        saved = table.clone()
        table.zero_()
        model.eval()
        with torch.no_grad():
            got = [f(v) for v in dic.train_func(model, None, x, train=False, t=t, noises=noises)]
        assert got == ref
        table.copy_(saved * 10)
        model.train()
        trainer = dic.AdamW(model.parameters(), lr=1e-3)
        before = table.clone()
        dic.train_func(model, trainer, x, train=True, t=t, noises=noises)
        g = table.grad.clone()
        used = sorted(set(t.flatten().tolist()) | {1})
        unused = [i for i in range(m["step_tot"]) if i not in used]
        assert bool((g[unused] == 0).all()) and all(float(g[i].abs().max()) > 0 for i in used)
        assert bool((table[used] != before[used]).any())
        # directional finite difference at the pre-step parameters
        model.load_state({n: (before if n == "timestep_embedding.weight" else p) for n, p in synth.denoiser_state(m["n_layers"], m["wseed"]).items()} |
                         {"timestep_embedding.weight": before})
        d = torch.randn(table.shape, generator=torch.Generator().manual_seed(0)).cuda()
        eps = 2e-2
        vals = []
        model.eval()
        for sgn in (+1, -1):
            table.copy_(before + sgn * eps * d)
            with torch.no_grad():
                vals.append(f(dic.train_func(model, None, x, train=False, t=t, noises=noises)[0]))
        fd = (vals[0] - vals[1]) / (2 * eps)
        an = float((g * d).sum())
        assert abs(fd - an) < 2e-2 * max(abs(an), 1e-3), (fd, an)
        # inference honours the table too: sample() takes row STEP_TOT-1 on its first pass and row 1 on the later ones; forward() insists on t
        table.copy_(before)
        start = torch.from_numpy(synth.noise((m["B"], m["L"] + 2, 768), 9, "restored"))
        _, h0 = dic.sample(model, x["image_clip"], steps=2, start=start, return_hidden=True)
        bump = torch.randn(768, generator=torch.Generator().manual_seed(3)).cuda()       # (a constant shift of a row would vanish in the LayerNorm)
        table[1].add_(bump)
        _, h1 = dic.sample(model, x["image_clip"], steps=2, start=start, return_hidden=True)
        assert float((h0 - h1).abs().max()) > 1e-3
        table[m["step_tot"] - 1].add_(bump)
        _, h2 = dic.sample(model, x["image_clip"], steps=2, start=start, return_hidden=True)
        assert float((h2 - h1).abs().max()) > 1e-3
        table[7].add_(bump)                                  # a row no pass reads
        _, h3 = dic.sample(model, x["image_clip"], steps=2, start=start, return_hidden=True)
        assert torch.equal(h2, h3)
        n = m["B"]
        args = (start[:n, :m["L"]].cuda(), x["image_clip"].reshape(n, 1, 512), x["text_clip"].reshape(n, 1, 512), x["attention_mask"],
                torch.tensor([[1, 0]] * n).cuda())
        with pytest.raises(ValueError):
            model(*args)
        _, xa = model(*args, with_logits=False, t=torch.full((n,), 3))
        table[3].add_(bump)
        _, xb = model(*args, with_logits=False, t=torch.full((n,), 3))
        assert float((xa - xb).abs().max()) > 1e-3
    finally:
        dic.cfg.update(TIMESTEP_EMBEDDING=False)


@pytest.mark.parametrize("world,dtype,layers,single,cfg_w", [(2, "fp32", 4, "0", "0"), (2, "bf16", 4, "1", "0"), (4, "bf16", 12, "0", "0"),
                                                             (4, "bf16", 4, "0", "0.3"), (2, "fp32", 4, "0", "0.3"), (8, "bf16", 12, "0", "0"),
                                                             (8, "bf16w", 12, "1", "0"), (2, "bf16r", 4, "0", "0"), (8, "bf16", 12, "1", "0")])
def test_data_parallel_step_of_the_real_engine_on_ranks_sharing_one_gpu(world, dtype, layers, single, cfg_w):
    """SURVEY section 8e with the HIP engine instead of the oracle: `world` processes (gloo; RCCL refuses two ranks on one device) take
    the step on their shards through parallel.GradReducer -- slices issued from the backward + streamed AdamW, or the one-collective exchange --
    and must reproduce the single-process full-batch loss and gradient; after a second step every rank holds bit-identical parameters
    (scripts/dist_check.py does the comparisons and exits non-zero on a mismatch).  cfg_w > 0: with classifier-free guidance (injected
    draws; the forced rows 0/1 of the global batch live on rank 0)."""
    import socket
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ, DIC_DIST_SHARE_GPU="1", DIC_DIST_BACKEND="gloo", DTYPE=dtype, LAYERS=str(layers), DIC_DP_SINGLE=single, CFG=cfg_w,
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "scripts", "dist_check.py")], env=env, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in (r.stdout + r.stderr).splitlines() if ln.startswith("rank ")]
    assert r.returncode == 0 and len(lines) >= 1 and all("-> OK" in ln for ln in lines), (r.stdout + r.stderr)[-3000:]
