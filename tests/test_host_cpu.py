"""CPU-only checks: host logic, the flat parameter store, the C-ABI library's exports (no compute calls without a GPU),
the loud-failure rule, and the data-parallel path on 2 gloo ranks with the oracle as the compute stand-in."""
import importlib
import os
import re
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dic = importlib.import_module("diffusion-image-captioning_amd")


def test_library_builds_loads_and_exports_every_declared_symbol():
    dic.build()
    L = dic.lib()
    header = open(os.path.join(ROOT, "include", "dic_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:int|size_t|double|const char\*)\s+(dic_\w+)\s*\(", header, flags=re.M))
    assert len(declared) >= 25, declared
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    assert set(dic._lib.EXPORTS) == declared
    assert L.dic_version() == dic._lib.ABI_VERSION == 19


def test_gemm_params_ctypes_mirror_matches_the_header_struct():
    """The Python side passes DicGemmParams by pointer: field order and C types must follow include/dic_hip.h exactly
    (and so must the binding shown in INTEGRATION.md)."""
    import ctypes as C
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "include", "dic_hip.h")).read()
    body = re.search(r"typedef struct DicGemmParams \{(.*?)\} DicGemmParams;", src, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        head, *rest = [d.strip() for d in decl.split(",")]
        ctype = head.rsplit(None, 1)[0].replace("const ", "").strip() if "*" not in head else "ptr"
        name = head.replace("*", " ").split()[-1]
        fields.append((name, ctype))
        for r in rest:
            fields.append((r.replace("*", " ").split()[-1], "ptr" if "*" in r else ctype))
    kind = {"ptr": C.c_void_p, "int": C.c_int, "float": C.c_float, "uint64_t": C.c_uint64, "int64_t": C.c_int64}
    mirror = dic._lib.GemmParams._fields_
    assert [n for n, _ in mirror] == [n for n, _ in fields]
    for (n, t), (_, ct) in zip(mirror, fields):
        assert t is kind[ct], f"{n}: {t} vs {ct}"
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    doc_fields = re.findall(r'\("(\w+)", ', doc[doc.index("class GemmParams"):doc.index("L.dic_gemm.argtypes")])
    assert doc_fields == [n for n, _ in fields]


def test_workspace_size_queries():
    """Host-only entry points of the C-ABI: callers size every workspace from these (kernels never allocate)."""
    L = dic.lib()
    assert L.dic_gemm_split_ws_bytes(768, 768, 1, 0) == 0
    assert L.dic_gemm_split_ws_bytes(3072, 768, 7, 1) == 7 * (3072 * 768 + 3072) * 4
    assert L.dic_ce_n_partials(30522, 128) == 2 * 239 and L.dic_ce_n_partials(30522, 256) == 4 * 120
    assert L.dic_ce_partial_bytes(16384, 30522, 256) == 16384 * 480 * 16
    assert L.dic_colsum_ws_bytes(0, 512, 2304) == 0                      # fp32, <= 1024 rows: single launch, no workspace
    assert L.dic_colsum_ws_bytes(1, 18432, 768) == 64 * 768 * 4
    assert L.dic_ln_partial_bytes(512, 3, 768) == 512 * 3 * 768 * 4


def test_data_parallel_exchange_groups_cover_every_layer_once():
    for n_layers in (1, 2, 6, 12, 13):
        for group in (1, 2, 3, 4):
            seen = []
            for i in reversed(range(n_layers)):                 # the backward finishes layers in descending order
                r = dic.parallel.exchange_group(i, n_layers, group)
                if r is not None:
                    assert r[0] == i
                    seen += list(range(*r))
            assert sorted(seen) == list(range(n_layers)), (n_layers, group, seen)
            # a group is only exchanged once all of its layers are done: every layer of it is >= the trigger layer
    assert dic.parallel.exchange_group(9, 12, 3) == (9, 12) and dic.parallel.exchange_group(2, 12, 3) == (2, 3)
    assert dic.parallel.exchange_group(10, 12, 3) is None


def test_product_path_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        dic.DistilBertModel(None, None, config=dict(n_layers=1))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        dic.diffuse_t(torch.zeros(1, 16, 768), torch.zeros(1, dtype=torch.int64))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "diffusion-image-captioning_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), fn


def test_synth_is_deterministic_and_sane():
    a = dic.synth.normal(dic.synth.stream_id("x", 3), (1000, 64))
    b = dic.synth.normal(dic.synth.stream_id("x", 3), (1000, 64))
    assert np.array_equal(a, b) and a.dtype == np.float32
    assert abs(a.mean()) < 0.02 and abs(a.std() - 1) < 0.02
    assert float(a.flat[0]) == pytest.approx(float(dic.synth.normal(dic.synth.stream_id("x", 3), (5,))[0]))
    bt = dic.synth.batch(8, 16, 30522, 1)
    assert bt["input_ids"].min() >= 0 and bt["input_ids"].max() < 30522
    assert np.allclose(np.linalg.norm(bt["image_clip"], axis=1), 1, atol=1e-6)
    lens = bt["attention_mask"].sum(1)
    assert lens.min() >= 6 and lens.max() <= 16 and (np.diff(bt["attention_mask"], axis=1) <= 0).all()


def test_param_store_layout_matches_reference_parameter_list():
    from importlib import import_module
    ParamStore = import_module("diffusion-image-captioning_amd.params").ParamStore
    st = ParamStore(6, "cpu", concat=True, bf16_shadow=False)
    ps = st.parameters()
    assert len(ps) == 108
    assert sum(p.numel() for p in ps) == 44_303_616
    specs = dic.synth.denoiser_param_specs(6)
    assert [tuple(p.shape) for p in ps] == [s[1] for s in specs]
    # views alias the flat buffers; grads alias G; q/k/v are slices of one stacked matrix
    st.P.fill_(1.0)
    assert all(float(p.min()) == 1.0 for p in ps)
    st.G.fill_(2.0)
    assert all(float(p.grad.min()) == 2.0 for p in ps)
    names = [n for n, _ in st.named_parameters()]
    q = ps[names.index("model.distilbert.transformer.layer.2.attention.q_lin.weight")]
    k = ps[names.index("model.distilbert.transformer.layer.2.attention.k_lin.weight")]
    assert k.data_ptr() - q.data_ptr() == 768 * 768 * 4
    # [LayerNorm.weight | LayerNorm.bias | bias of the Linear feeding it] contiguous -> one column-sum launch
    assert st.off("L0.ln1b") - st.off("L0.ln1g") == 768 and st.off("L0.bo") - st.off("L0.ln1b") == 768
    assert st.off("L3.ln2b") - st.off("L3.ln2g") == 768 and st.off("L3.b2") - st.off("L3.ln2b") == 768
    assert st.off("bvt") - st.off("vln_g") == 2 * 768
    # load/state round trip
    state = dic.synth.denoiser_state(6, 0)
    st.load_state(state)
    for n, p in st.named_parameters():
        assert np.array_equal(p.numpy(), state[n]), n
    st12 = ParamStore(12, "cpu", concat=False, bf16_shadow=False)
    assert len(st12.parameters()) == 3 + 12 * 16 + 4 + 4


def test_config_mirrors_reference_globals():
    c = dic.Config()
    assert (c.BATCH_SIZE, c.MAX_LENGTH, c.SAMPLE_SIZE, c.STEP_TOT, c.COSIN_SCHEDULE) == (8, 16, 100, 1000, True)
    assert (c.ROUNDING_WEIGHT, c.CLASSIFIER_FREE_PROB, c.X_T_STEP_INTERVAL, c.LOSS_FUNC) == (0.5, 0.2, 100, "series_sum_sample_mean")
    with pytest.raises(AttributeError):
        c.update(NOT_A_KNOB=1)


def test_bench_flop_model_matches_survey():
    sys.path.insert(0, ROOT)
    import bench
    assert abs(bench.gflop_per_seq(16, 6) - 6.173) / 6.173 < 0.02
    assert abs(bench.gflop_per_seq(16, 12) - 10.777) / 10.777 < 0.02


# ------------------------------------------------------------------ data parallel on 2 CPU ranks (gloo)
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dp_worker(rank, world, port, out, nl=1, cfg_w=0.0):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    sys.path.insert(0, ROOT)
    d = importlib.import_module("diffusion-image-captioning_amd")
    from oracle import ref_model as R
    ParamStore = importlib.import_module("diffusion-image-captioning_amd.params").ParamStore
    r, w, _ = d.parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    B, S, L, V = (4, 2, 16, 300) if cfg_w <= 0 else (2 * world, 1, 16, 300)
    E = d.synth.vocab_embedding(V, 768, 0)
    full = {k: torch.from_numpy(v) for k, v in d.synth.batch(B, L, V, 1).items()}
    mine = d.parallel.shard(full)
    assert len(mine["input_ids"]) == B // world
    # the reference shares ONE t-vector across the whole batch (ref :461): the t draw is a Philox kernel keyed by a per-process step counter
    # (diffusion._next_t_seed); the ranks must continue from rank 0's counter even when their torch seeds differ
    torch.manual_seed(100 + rank)
    t = torch.from_numpy(d.synth.timesteps(S, 100, 3))
    # ... while everything drawn PER ITEM (noise eps, dropout masks, guidance uniforms) must differ between the ranks: the kernels
    # key their Philox / hash streams by (seed, local element index), so ranks sharing a seed would repeat each other's draws
    diffusion = importlib.import_module("diffusion-image-captioning_amd.diffusion")

    class SeedModel:                              # exactly what configure_model_for_rank touches on a Denoiser
        dropout_seed_base = 0x5EED0000
        device = torch.device("cpu")

        def set_dropout_seed(self, s):
            self.seed = s
    sm = d.parallel.configure_model_for_rank(SeedModel())
    t_seeds = [diffusion._next_t_seed() for _ in range(5)]          # what the next five train_func / validate calls would key their t draw with
    d.parallel.assert_shared_timestep_seed()
    mine_draws = dict(t=t_seeds, dropout=sm.seed, noise=diffusion._state["noise_seed"],
                      guidance=diffusion._guidance_uniform(16, "cpu").flatten().tolist(), forced=sm.rank_rows_forced)
    draws = [None] * world
    torch.distributed.all_gather_object(draws, mine_draws)
    assert all(dr["t"] == draws[0]["t"] for dr in draws) and len(set(draws[0]["t"])) == 5
    for k in ("dropout", "noise", "guidance"):
        assert len({str(dr[k]) for dr in draws}) == world, k
    assert [x["forced"] for x in draws] == [True] + [False] * (world - 1)
    if rank == 1:                                   # a rank that ran one step more (rank-0-only validation, uneven shards) is caught
        diffusion._next_t_seed()
    try:
        d.parallel.assert_shared_timestep_seed()
        raised = False
    except RuntimeError:
        raised = True
    assert raised
    if rank == 1:
        diffusion._state["t_seed"] -= 1
    assert draws[0]["dropout"] == SeedModel.dropout_seed_base and draws[0]["noise"] == diffusion.NOISE_SEED_BASE   # rank 0 = the single-GPU streams
    noise_full = [torch.from_numpy(d.synth.noise((B, L, 768), 5, f"eps{i}")) for i in range(2)]
    noise_mine = [n[rank * (B // world):(rank + 1) * (B // world)] for n in noise_full]
    cfg = R.Config(BATCH_SIZE=B // world, SAMPLE_SIZE=S, MAX_LENGTH=L, STEP_TOT=100, COSIN_SCHEDULE=False, n_layers=nl, vocab=V,
                   CLASSIFIER_FREE_WEIGHT=cfg_w, CLASSIFIER_FREE_PROB=0.2)
    model = R.build(cfg, d.synth.denoiser_state(nl, 0), E)
    u_mine = None
    if cfg_w > 0:
        # classifier-free guidance: the forced unguided / guided rows 0 / 1 (ref :408-409) exist once per GLOBAL batch = on rank 0.  The oracle
        # forces them in every local batch, so the draws are chosen such that forcing changes nothing off rank 0 (row 0 unguided, row 1 guided)
        u_full = _dp_guidance_draws(B)
        u_mine = u_full[rank * (B // world):(rank + 1) * (B // world)]
    l, *_ = R.train_func(model, R.AdamW(model.parameters()), mine, train=False, t=t, noises=noise_mine, cfg_uniform=u_mine)
    l.backward()
    store = ParamStore(nl, "cpu", bf16_shadow=False)
    for (n, p), g in zip(store.named_parameters(), model.parameters()):
        p.grad.copy_(g.grad)

    class M:            # what the reducer needs from a model
        params = store

    class T:
        grad_scale = 1.0
    tr = T()
    red = d.parallel.GradReducer(M)            # layer slices go out asynchronously as the backward produces them ...
    assert red.active
    for i in reversed(range(nl)):
        red.layer_done(i)
    red.finish(tr)                             # ... then the small tail; together exactly one pass over the flat buffer
    assert tr.grad_scale == 1.0 / world
    expect = sum(1 for i in range(nl) if d.parallel.exchange_group(i, nl, red.group) is not None) + 1
    assert red.n_collectives == expect and (nl != 12 or expect == 7)          # 12 layers: 9-11, 6-8, 3-5, 2, 1, 0, tail
    # north_star's literal variant: ONE all-reduce of the whole buffer after the backward gives the same sums
    g_sliced = store.G.clone()
    for (n, p), g in zip(store.named_parameters(), model.parameters()):
        p.grad.copy_(g.grad)
    opts = importlib.import_module("diffusion-image-captioning_amd.options")
    opts.OPT.dp_single = True
    red1 = d.parallel.GradReducer(M)
    for i in reversed(range(nl)):
        red1.layer_done(i)
    red1.finish(T())
    opts.OPT.dp_single = False
    assert red1.n_collectives == 1
    assert float((store.G - g_sliced).abs().max()) <= 1e-6 * float(g_sliced.abs().max())      # (the reduction order inside a collective depends on its size)
    (lm,) = d.parallel.allreduce_scalars(l)
    if rank == 0:
        torch.save((t.clone(), store.G.clone() * tr.grad_scale, float(lm)), out)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def _dp_guidance_draws(B):
    u = torch.from_numpy(dic.synth.uniform(dic.synth.stream_id("cfg", 9), (B, 1))).clone()
    u[0::2] = 0.05 + 0.1 * u[0::2]          # even rows: <= 0.2 -> unguided
    u[1::2] = 0.3 + 0.6 * u[1::2]           # odd rows: > 0.2 -> guided
    return u


@pytest.mark.parametrize("world,nl,cfg_w", [(2, 1, 0.0), (4, 12, 0.3)])
def test_data_parallel_gradients_equal_full_batch_gloo(world, nl, cfg_w):
    from oracle import ref_model as R
    ParamStore = importlib.import_module("diffusion-image-captioning_amd.params").ParamStore
    import tempfile
    ctx = mp.get_context("spawn")
    out = os.path.join(tempfile.mkdtemp(), "dp_rank0.pt")
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, out, nl, cfg_w)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    t, g_dp, l_dp = torch.load(out)
    B, S, L, V = (4, 2, 16, 300) if cfg_w <= 0 else (2 * world, 1, 16, 300)
    E = dic.synth.vocab_embedding(V, 768, 0)
    full = {k: torch.from_numpy(v) for k, v in dic.synth.batch(B, L, V, 1).items()}
    noise_full = [torch.from_numpy(dic.synth.noise((B, L, 768), 5, f"eps{i}")) for i in range(2)]
    cfg = R.Config(BATCH_SIZE=B, SAMPLE_SIZE=S, MAX_LENGTH=L, STEP_TOT=100, COSIN_SCHEDULE=False, n_layers=nl, vocab=V,
                   CLASSIFIER_FREE_WEIGHT=cfg_w, CLASSIFIER_FREE_PROB=0.2)
    model = R.build(cfg, dic.synth.denoiser_state(nl, 0), E)
    l, *_ = R.train_func(model, R.AdamW(model.parameters()), full, train=False, t=t, noises=noise_full,
                         cfg_uniform=_dp_guidance_draws(B) if cfg_w > 0 else None)
    l.backward()
    store = ParamStore(nl, "cpu", bf16_shadow=False)
    for (n, p), g in zip(store.named_parameters(), model.parameters()):
        p.grad.copy_(g.grad)
    assert abs(l_dp - float(l)) < 1e-4 * abs(float(l))
    err = float((g_dp - store.G).abs().max() / store.G.abs().max())
    assert err < 1e-5, err


# ------------------------------------------------------------------ the PRODUCT's schedule tables (ref :337-346)
def test_product_alpha_cumprod_tables_match_the_reference_host():
    """diffusion.alpha_cumprod_table() itself (not the oracle's, not an injected table) against the tables the reference produced
    (tests/golden: cosine T=1000 and linear T=100).  torch.cos is not correctly rounded on every CPU, so the cosine table may differ
    from another host's by an ulp -- no more; the linear one is exact.  The sqrt tables q_sample uses must be the correctly rounded
    square roots of whatever table is in force."""
    diffusion = importlib.import_module("diffusion-image-captioning_amd.diffusion")
    saved = {k: getattr(dic.cfg, k) for k in ("COSIN_SCHEDULE", "STEP_TOT", "BETA_MIN", "BETA_MAX")}
    try:
        for name, cosine, T in (("base_b4s3l16", True, 1000), ("deep6_b2s2l16", False, 100)):
            z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
            dic.cfg.update(COSIN_SCHEDULE=cosine, STEP_TOT=T, BETA_MIN=0.0001, BETA_MAX=0.02)
            diffusion.set_alpha_cumprod(None)
            ac = diffusion.alpha_cumprod_table("cpu").numpy()
            ref = z["alpha_cumprod"]
            assert ac.dtype == np.float32 and ac.shape == ref.shape == (T,)
            ulp = np.abs(ac.view(np.int32).astype(np.int64) - ref.view(np.int32).astype(np.int64))
            assert ulp.max() <= (1 if cosine else 0), f"{name}: alpha_cumprod differs by {ulp.max()} ulp"
            assert ac[0] == 1.0
            np.testing.assert_array_equal(diffusion._state["sqrt_ac"].numpy(), np.sqrt(ac.astype(np.float64)).astype(np.float32))
            np.testing.assert_array_equal(diffusion._state["sqrt_1mac"].numpy(), np.sqrt((1 - ac).astype(np.float64)).astype(np.float32))
            # and with the reference host's table injected (what the golden tests do) the coefficient tables follow it exactly
            diffusion.set_alpha_cumprod(torch.from_numpy(ref))
            diffusion.alpha_cumprod_table("cpu")
            np.testing.assert_array_equal(diffusion._state["sqrt_ac"].numpy(), np.sqrt(ref.astype(np.float64)).astype(np.float32))
    finally:
        dic.cfg.update(**saved)
        diffusion.set_alpha_cumprod(None)


def test_checkpointed_rng_streams_are_rederived_per_rank(monkeypatch):
    """Advisor finding (round 3): harness.load_checkpoint restored the SAVING rank's noise / dropout / guidance streams on every rank.  A state
    saved by rank 0 and loaded on rank 2 must keep the shared timestep counter but shift the per-item streams by the rank mix -- exactly the
    separation parallel.configure_model_for_rank sets up at start -- and loading on the saving rank must restore everything as saved,
    guidance generator included (also from a pre-round-4 state that filed it under 'cuda:0' and carries no rank)."""
    import torch
    diffusion = importlib.import_module("diffusion-image-captioning_amd.diffusion")
    parallel = dic.parallel

    class M:
        _seed = 12345
    diffusion.seed_all(5)
    diffusion._state["t_seed"] = 777
    u0 = diffusion._guidance_uniform(7, "cpu")          # advance the guidance generator a little
    st = diffusion.rng_state(M())
    assert st["rank"] == 0 and st["guidance_draws"] == 7
    nxt = diffusion._guidance_uniform(5, "cpu")
    # same rank: exact continuation
    diffusion.seed_all(99)
    m = M(); m._seed = 1
    diffusion.set_rng_state(st, m)
    assert diffusion._state["t_seed"] == 777 and diffusion._state["noise_seed"] == st["noise_seed"] and m._seed == 12345
    assert torch.equal(diffusion._guidance_uniform(5, "cpu"), nxt)
    # legacy layout: generator under a device key, no rank
    legacy = {"t_seed": 777, "noise_seed": st["noise_seed"], "guidance": {"cuda:0": st["guidance"]["cpu"]}, "dropout_seed": 12345}
    diffusion.seed_all(99)
    diffusion.set_rng_state(legacy, m)
    assert torch.equal(diffusion._guidance_uniform(5, "cpu"), nxt)
    # another rank: shared t, shifted per-item streams, its own guidance draws
    monkeypatch.setattr(parallel, "rank", lambda: 2)
    m2 = M(); m2._seed = 1
    diffusion.set_rng_state(st, m2)
    assert diffusion._state["t_seed"] == 777
    assert diffusion._state["noise_seed"] == (st["noise_seed"] + 2 * parallel._RANK_MIX) & 0x7FFFFFFFFFFFFFFF
    assert m2._seed == (12345 + 2 * parallel._RANK_MIX) & 0x7FFFFFFFFFFFFFFF and m2._seed != 12345
    assert not torch.equal(diffusion._guidance_uniform(5, "cpu"), nxt)
    # ... and saved by rank 2, loaded by rank 2: as saved
    st2 = diffusion.rng_state(m2)
    assert st2["rank"] == 2
    nxt2 = diffusion._guidance_uniform(3, "cpu")
    diffusion.set_rng_state(st2, m2)
    assert torch.equal(diffusion._guidance_uniform(3, "cpu"), nxt2)


def test_restored_rank_continues_the_stream_an_uninterrupted_run_of_that_rank_would_draw(monkeypatch):
    """Round-4 advisor finding: a rank that loads rank 0's checkpoint must draw the SAME noise / dropout seeds as that rank would have drawn
    had it run from `configure_model_for_rank` without interruption (the noise counter once differed in bit 63: 64- vs 63-bit wrap)."""
    diffusion = importlib.import_module("diffusion-image-captioning_amd.diffusion")
    parallel = dic.parallel
    class M:
        device = "cpu"
        dropout_seed_base = 0x5EED0000
        rank_rows_forced = True
        def set_dropout_seed(self, s):
            self._seed = int(s) & 0x7FFFFFFFFFFFFFFF
    for r in (1, 2, 5, 7):
        # uninterrupted rank r: configure, then k steps
        monkeypatch.setattr(parallel, "rank", lambda r=r: r)
        diffusion.seed_all(2024)
        mu = parallel.configure_model_for_rank(M(), r)
        want = []
        for k in range(14):
            s = diffusion._next_seed()
            mu._seed += 64
            if k >= 10:
                want.append((s, mu._seed))
        # rank 0 runs 10 steps, saves; rank r loads and continues
        monkeypatch.setattr(parallel, "rank", lambda: 0)
        diffusion.seed_all(2024)
        m0 = parallel.configure_model_for_rank(M(), 0)
        for _ in range(10):
            diffusion._next_seed()
            m0._seed += 64
        st = diffusion.rng_state(m0)
        monkeypatch.setattr(parallel, "rank", lambda r=r: r)
        mr = M(); mr._seed = 0
        diffusion.set_rng_state(st, mr)
        got = []
        for _ in range(4):
            s = diffusion._next_seed()
            mr._seed += 64
            got.append((s, mr._seed))
        assert got == want, (r, [hex(a) for a, _ in got], [hex(a) for a, _ in want])


def test_generated_asm_of_the_four_wave_gemm_is_what_its_generator_emits(tmp_path):
    """csrc/gemm_w4a_asm.inc is generated (scripts/gen_w4a.py places every instruction of the four-wave GEMM's main loops and derives every counted
    wait from its own issue order): the committed file must be exactly what the committed generator writes, and the schedule's invariants hold --
    512 MFMAs per K-step pair and variant body in the steady-state loop, every lgkmcnt within the 4-bit counter."""
    import re
    import subprocess
    import sys
    out = tmp_path / "w4a.inc"
    subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gen_w4a.py"), str(out)], check=True, capture_output=True)
    want = open(os.path.join(ROOT, "diffusion-image-captioning_amd", "csrc", "gemm_w4a_asm.inc")).read()
    got = open(out).read()
    assert got == want, "regenerate: python scripts/gen_w4a.py diffusion-image-captioning_amd/csrc/gemm_w4a_asm.inc"
    bodies = re.findall(r"#define (W4A_BODY7?_\w+) \\\n((?:    \".*\n?)+)", got)
    assert len(bodies) == 18                                   # two tile heights x (k-contiguous B: 6 epilogues, k-major B: 3)
    for name, text in bodies:
        per_pair = 2 * 2 * (7 if name.startswith("W4A_BODY7") else 8) * 8          # MFMAs of two K-steps
        assert text.count("v_mfma_f32_16x16x32_bf16") % per_pair == 0 and text.count("v_mfma_f32_16x16x32_bf16") >= 3 * per_pair, name      # whole K-steps only
        assert all(int(n) <= 15 for n in re.findall(r"lgkmcnt\((\d+)\)", text)), name
        assert all(int(n) <= 63 for n in re.findall(r"vmcnt\((\d+)\)", text)), name
        assert text.count("s_barrier") >= 2 * 6, name


def test_every_generated_asm_gemm_body_reproduces_numpy_in_emulation():
    """scripts/w4a_emulate.py interprets the generated gfx950 text on the CPU (one workgroup: 4 waves x 64 lanes, LDS, a flat global memory; the C++
    prologue of csrc/gemm_w4a.h restated next to it) and compares C -- and GELU' of the two-output form -- with numpy on the same bf16 operands: all 18
    bodies (two tile heights x layouts x epilogues incl. the dropout mask and GELU), a problem with two row tiles of which the last is ragged, two column
    tiles and three K-step pairs; guard bytes behind C's last valid row and column must survive.  The emulator is itself checked by mutation: one
    fragment-read offset, one DPP bank mask and one LDS-DMA piece changed must each change the result."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    try:
        import gen_w4a as G
        import w4a_emulate as W
    finally:
        sys.path.pop(0)
    for ni, bkm, epi in W.bodies():
        worst, guards = W.run_case(ni, bkm, epi, 32 * ni + 80, 512, 384)
        assert worst <= 1.0 and guards, (ni, bkm, epi, worst, guards)
    # mutations of one body: the emulator must notice
    real = G.Gen.body

    def mutated(pred, mut):
        def body(self):
            lines = real(self)
            k = next(i for i, l in enumerate(lines) if pred(l))
            lines[k] = mut(lines[k])
            return lines
        G.Gen.body = body
        try:
            return W.run_case(8, False, "resid", 336, 256, 256)
        finally:
            G.Gen.body = real
    assert mutated(lambda l: False or l.startswith("ds_read_b128") and "offset:2048" in l, lambda l: l.replace("offset:2048", "offset:4096"))[0] > 1.0
    assert mutated(lambda l: "bank_mask:0xc" in l, lambda l: l.replace("bank_mask:0xc", "bank_mask:0x3"))[0] > 1.0
    assert mutated(lambda l: l.startswith("s_add_u32 m0") and l.endswith(", 4096"), lambda l: l[:-4] + "8192")[0] > 1.0


def test_driver_build_entry_point_runs_on_the_shipped_library(monkeypatch):
    """__graft_entry__.build() is the driver's "does it build" check; with DIC_BUILD_REUSE=1 it skips the compile and must still load the library,
    find every symbol and agree with its ABI version (a hard-coded version number in the entry point once outlived two version bumps)."""
    import subprocess
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build(); print('BUILD OK')"], cwd=ROOT, env=dict(os.environ, DIC_BUILD_REUSE="1"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "BUILD OK" in r.stdout, (r.stdout + r.stderr)[-1500:]


def test_measurement_options_of_the_asm_generator_are_correct_on_paper():
    """scripts/gen_w4a.py OUT block_waits early_side defer_stores writes a VARIANT file for a measurement build (scripts/experiments/w4a_variant_ab.sh; the
    shipped .inc is generated without options): per-block side-input waits in the epilogue, the side-input blocks that live in registers the K loop never
    uses requested one K-step earlier, and (plain bodies) half of a tile's stores issued from the next tile's first K-step.  All must pass the race checker
    and reproduce numpy in emulation before they cost a GPU minute."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    try:
        import w4a_emulate as W
        import w4a_hazard_check as H
    finally:
        sys.path.pop(0)
    opts = ("block_waits", "early_side", "defer_stores")
    assert H.check_all(shapes=((256, 2), (384, 2)), opts=opts) == 18 * 2
    for ni, bkm, epi in ((8, False, "dropres"), (7, True, "mulaux"), (7, False, "resid"), (8, True, "plain"), (7, False, "plain")):
        worst, guards = W.run_case(ni, bkm, epi, 32 * ni + 80, 512, 384, opts=opts)
        assert worst <= 1.0 and guards, (ni, bkm, epi, worst, guards)


def test_asm_gemm_tile_height_plan_fills_the_rounds_of_resident_workgroups():
    """Host logic of the two-height asm GEMM (dic_gemm_w4a_rows_plan, a pure function: no device): 224-row tiles where they turn a partly filled last
    round into a fuller one of shorter tiles -- the step's 17 408 tokens (78 x 224: 234 / 702 / 936 tiles = 1 / 3 / 4 rounds on 256 CUs instead of 204 / 612 /
    816) and the sampling pass's N = 768 shapes -- 256-row tiles where the shorter ones would add a round (34 816 x 2304: 5 rounds against 6) or where
    nothing is gained; the option forces either height."""
    L = dic.lib()
    plan = lambda M, N, K, cus=256: L.dic_gemm_w4a_rows_plan(M, N, K, cus)
    assert plan(17408, 2304, 768) == 256 and L.dic_set_option(b"gemm_w4a_rows", 0) == 0        # (the shipped setting forces 256; 0 = the per-launch plan tested here)
    assert [plan(17408, n, 768) for n in (768, 2304, 3072)] == [224, 224, 224] and plan(17408, 768, 3072) == 224
    assert plan(34816, 768, 768) == 224 and plan(34816, 768, 3072) == 224
    assert plan(34816, 2304, 768) == 256 and plan(34816, 3072, 768) == 256
    assert plan(256, 256, 256) == 256 and plan(65536, 4096, 4096) == 256           # tiny: one tile; huge: rounds dominate either way, ties keep 256
    assert plan(17408, 768, 768, cus=64) in (224, 256) and plan(0, 1, 1, 1) == -1
    try:
        assert L.dic_set_option(b"gemm_w4a_rows", 256) == 0 and plan(17408, 768, 768) == 256
        assert L.dic_set_option(b"gemm_w4a_rows", 224) == 0 and plan(34816, 2304, 768) == 224 and plan(128, 256, 256) == 256
        assert L.dic_set_option(b"gemm_w4a_rows", 192) != 0
    finally:
        importlib.import_module("diffusion-image-captioning_amd.options").push_to_library(L)


def test_counted_waits_and_barriers_of_the_four_wave_gemm_are_proven_by_symbolic_execution():
    """scripts/w4a_hazard_check.py executes every generated body through its real control flow (several K / tile counts) and checks, under the
    weakest completion assumption the ISA allows, that no register is read before its load was waited for, that an LDS stage is read only
    after publication (vmcnt wait + barrier) and refilled only after its reads returned (lgkmcnt wait + barrier), and that nothing is left
    outstanding -- and the fixed-latency hazards the hardware does not interlock (MFMA result -> v_accvgpr_read, VALU -> DPP, transcendental -> VALU,
    m0 -> LDS-DMA).  The checker itself is checked by mutation: every `vmcnt` weakened by one, a weakened `lgkmcnt` in front of an MFMA group
    and a dropped barrier must each be reported."""
    import re
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    try:
        import gen_w4a as G
        import w4a_hazard_check as H
    finally:
        sys.path.pop(0)
    assert H.check_all() == 18 * 4
    for ni, bkm, epi in ((8, False, "plain"), (7, True, "resid"), (7, False, "gelud")):
        lines = G.Gen(bkm, epi, ni).body()
        H.Sim(lines, ni, 384, 2).run()

        def mutated(pred, mut, nth):
            idx = [i for i, l in enumerate(lines) if pred(l)]
            out = list(lines)
            new = mut(out[idx[nth]])
            if new is None:
                del out[idx[nth]]
            else:
                out[idx[nth]] = new
            try:
                H.Sim(out, ni, 384, 2).run()
            except H.Violation:
                return True
            return False
        is_vm = lambda l: bool(re.search(r"vmcnt\(\d+\)", l)) and "lgkmcnt" not in l
        n_vm = sum(map(is_vm, lines))
        assert n_vm >= 7 and all(mutated(is_vm, lambda l: re.sub(r"vmcnt\((\d+)\)", lambda m: f"vmcnt({int(m.group(1)) + 1})", l), k) for k in range(n_vm)), (ni, bkm, epi)
        is_lg = lambda l: bool(re.fullmatch(r"s_waitcnt lgkmcnt\(\d+\)", l))
        caught = sum(mutated(is_lg, lambda l: re.sub(r"\((\d+)\)", lambda m: f"({int(m.group(1)) + 1})", l), k) for k in range(sum(map(is_lg, lines))))
        assert caught >= 0.75 * sum(map(is_lg, lines)), (ni, bkm, epi, caught)        # (a few waits are stricter than their consumer needs)
        assert mutated(lambda l: l == "s_barrier", lambda l: None, 1) and mutated(lambda l: l == "s_barrier", lambda l: None, 4), (ni, bkm, epi)
    # the fixed-latency rules (R6-R9) on minimal sequences: each must be reported without, and accepted with, the wait states the hardware needs
    def runs(lines):
        sim = H.Sim(lines, 8, 256, 1)
        try:
            for k in range(len(lines)):
                sim.step(k)
        except H.Violation:
            return False
        return True
    mf = "v_mfma_f32_16x16x32_bf16 a[0:3], v[0:3], v[64:67], 0"
    assert not runs([mf, "s_nop 10", "v_accvgpr_read_b32 v136, a0"]) and runs([mf, "s_nop 11", "v_accvgpr_read_b32 v136, a0"])      # (12 states: the guide's number for this MFMA)
    wide_store = "buffer_store_dwordx4 v[128:131], v174, s[68:71], s98 offen"
    assert not runs([wide_store, "v_mov_b32 v129, v0"]) and runs([wide_store, "s_nop 1", "v_mov_b32 v129, v0"])                       # R13
    dpp = "v_mov_b32_dpp v3, v1 row_ror:8 row_mask:0xf bank_mask:0xc"
    assert not runs(["v_mov_b32 v1, v2", "s_nop 0", dpp]) and runs(["v_mov_b32 v1, v2", "s_nop 1", dpp])
    assert not runs(["v_rcp_f32 v1, v1", "v_mul_f32 v2, v1, v1"]) and runs(["v_rcp_f32 v1, v1", "s_nop 0", "v_mul_f32 v2, v1, v1"])
    dma = "buffer_load_dwordx4 v152, s[52:55], s88 offen lds"
    assert not runs([f"s_add_u32 m0, s{G.S_M0A}, 0", dma]) and runs([f"s_add_u32 m0, s{G.S_M0A}, 0", "s_nop 0", dma])


def test_narrow_tile_asm_gemm_is_generated_emulated_and_race_checked_on_the_cpu(tmp_path):
    """Round 6: the narrow-tile bodies (scripts/gen_w4n.py -> csrc/gemm_w4n_asm.inc; 256 x 128 tiles, three LDS stages, the finished tile's epilogue drained as
    filler instructions under the next tile's K loop; every body in a loop form for K = 192 n and a loop-free form for K = 768).  (1) the committed file is what
    the committed generator writes; 12 K-steps of 64 MFMAs and 13 barriers (one per K-step + the prologue's) of text per body.  (2) functional emulation (scripts/w4n_emulate.py): all nine bodies reproduce numpy on a problem with
    a ragged second row tile and two column tiles, with zero and with one pass of the middle loop, guard bytes intact; a single-tile launch (prologue + post-loop
    epilogue only) too; mutations -- one accumulator move dropped, one fragment-read offset, one B piece's LDS target -- must change the result.  (3) symbolic
    execution (scripts/w4n_hazard_check.py): every counted wait, the barrier of every K-step and the accumulator hand-over (each of a[128:255] filled once and
    read once per tile; no write into a wide store's data registers right behind it; 12 states between an MFMA and another reader of its result) hold for K = 576 ... 2304 and 1-3 tiles; mutations -- every barrier dropped, lgkmcnt(0) removed from every barrier's wait, the barrier
    waits' vmcnt weakened -- must be reported."""
    import re
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    try:
        import gen_w4n as G
        import w4n_emulate as W
        import w4n_hazard_check as H
    finally:
        sys.path.pop(0)
    out = tmp_path / "w4n.inc"
    subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gen_w4n.py"), str(out)], check=True, capture_output=True)
    want = open(os.path.join(ROOT, "diffusion-image-captioning_amd", "csrc", "gemm_w4n_asm.inc")).read()
    assert open(out).read() == want, "regenerate: python scripts/gen_w4n.py diffusion-image-captioning_amd/csrc/gemm_w4n_asm.inc"
    bodies = re.findall(r"#define (W4N_BODY(?:12)?_\w+) \\\n((?:    \".*\n?)+)", want)
    assert len(bodies) == 20 and sum(n.startswith("W4N_BODY12_") for n, _ in bodies) == 10         # the loop form and the loop-free K = 768 form of every body
    for name, text in bodies:
        assert text.count("v_mfma_f32_16x16x32_bf16") == 12 * 64 and text.count("s_barrier") == 13, name
        assert text.count("v_accvgpr_mov_b32") == 2 * 128, name                         # the first K-step's hand-over + the one in front of the last tile's epilogue
        assert all(int(n) <= 15 for n in re.findall(r"lgkmcnt\((\d+)\)", text)) and all(int(n) <= 63 for n in re.findall(r"vmcnt\((\d+)\)", text)), name
    # (2) emulation
    for bkm, epi in G.BODIES:
        for K in (576, 768):
            # (CE_EXP, the rounding-head forward: a ragged last column tile, out-of-range targets, the slab sums and the target logits checked too)
            worst, guards = W.run_case_ce(336, 300, K) if epi == "ceexp" else W.run_case(bkm, epi, 336, 256, K)
            assert worst <= 1.0 and guards, (bkm, epi, K, worst, guards)
    worst, guards = W.run_case(False, "gelud", 200, 128, 576)
    assert worst <= 1.0 and guards
    for bkm, epi in G.BODIES:                                 # the loop-free form (what K = 768 launches take): several tiles per workgroup, and a single one
        for M, N in ((336, 256), (200, 128)):
            worst, guards = W.run_case_ce(M, N + 44, 768, opts=("flat=12",)) if epi == "ceexp" else W.run_case(bkm, epi, M, N, 768, opts=("flat=12",))
            assert worst <= 1.0 and guards, (bkm, epi, M, N, worst, guards)
    lines = W.body_lines(False, "resid")

    def emu_mutated(pred, mut):
        idx = [i for i, l in enumerate(lines) if pred(l)]
        out_ = list(lines)
        new = mut(out_[idx[len(idx) // 2]])
        if new is None:
            del out_[idx[len(idx) // 2]]
        else:
            out_[idx[len(idx) // 2]] = new
        try:
            worst_, guards_ = W.run_case(False, "resid", 336, 256, 576, lines=out_)
        except (AssertionError, IndexError):
            return True                                       # (an address outside the emulated memory / LDS is a detection too)
        return worst_ > 1.0 or not guards_
    assert emu_mutated(lambda l: l.startswith("v_accvgpr_mov_b32 a200,"), lambda l: None)
    assert emu_mutated(lambda l: l.startswith("ds_read_b128 v[40:43]"), lambda l: re.sub(r"offset:(\d+)", lambda m: f"offset:{int(m.group(1)) + 2048}", l))
    assert emu_mutated(lambda l: l.startswith(f"s_add_u32 m0, s{G.S_M0B}, {16384 + 2048}"), lambda l: l.replace(str(16384 + 2048), str(16384 + 3072)))
    # (3) symbolic execution
    st = "buffer_store_dwordx4 v[80:83], v122, s[68:71], s97 offen"              # R13 on a minimal sequence: reported without, accepted with, the pad

    def runs(seq):
        sim = H.Sim(seq, 768, 1)
        try:
            for k in range(len(seq)):
                sim.step(k)
        except H.Violation:
            return False
        return True
    assert not runs([st, "v_mov_b32 v81, v0"]) and runs([st, "s_nop 1", "v_mov_b32 v81, v0"])
    assert H.check_all() == 10 * 5
    assert H.check_all(opts=("flat=12",), shapes=((768, 1), (768, 2), (768, 4))) == 10 * 3
    for bkm, epi, fopts in ((False, "plain", ()), (True, "mulaux", ()), (False, "gelud", ()), (False, "gelud", ("flat=12",)), (True, "plain", ("flat=12",)), (False, "ceexp", ("flat=12",))):
        lines = G.generate(bkm, epi, fopts)[0]

        def reported(pred, mut, nth):
            idx = [i for i, l in enumerate(lines) if pred(l)]
            out_ = list(lines)
            new = mut(out_[idx[nth]])
            if new is None:
                del out_[idx[nth]]
            else:
                out_[idx[nth]] = new
            for K in ((768,) if fopts else (576, 768)):
                try:
                    H.Sim(out_, K, 3).run()
                except H.Violation:
                    return True
            return False
        is_bar = lambda l: l == "s_barrier"
        assert all(reported(is_bar, lambda l: None, k) for k in range(13)), (bkm, epi)
        is_c = lambda l: "lgkmcnt(0)" in l and "vmcnt(" in l and "vmcnt(0)" not in l
        n_c = sum(map(is_c, lines))
        assert n_c == 12 and all(reported(is_c, lambda l: l.replace(" lgkmcnt(0)", ""), k) for k in range(n_c)), (bkm, epi)
        weaker = lambda l: re.sub(r"vmcnt\((\d+)\)", lambda m: f"vmcnt({int(m.group(1)) + 1})", l)
        caught = sum(reported(is_c, weaker, k) for k in range(n_c))
        assert caught >= 9, (bkm, epi, caught)                # (a barrier's wait that a stricter, older wait of the epilogue queue already covers is not reportable)


def test_packed_fp32_instructions_of_the_narrow_queue_unpack_into_their_scalar_halves():
    """scripts/gen_w4n.py unpack_pk: the narrow bodies drain their epilogue BETWEEN MFMAs, where packed fp32 VALU operations are a measured anti-lever, so every
    v_pk_fma / v_pk_mul / v_pk_add_f32 of the queue is emitted as two scalar instructions.  The rewrite must keep the operand selection: op_sel_hi = 0 on an inline constant feeds
    both halves, neg_lo / neg_hi become the scalar negation prefix, a constant first source of a VOP2 form stays first, anything else is left alone -- and the generated default
    text holds no packed instruction at all."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    try:
        import gen_w4n as G
    finally:
        sys.path.pop(0)
    u = G.unpack_pk
    assert u("v_pk_fma_f32 v[96:97], v[96:97], v[192:193], 1.0 op_sel_hi:[1,1,0]") == ["v_fma_f32 v96, v96, v192, 1.0", "v_fma_f32 v97, v97, v193, 1.0"]
    assert u("v_pk_mul_f32 v[92:93], v[92:93], v[96:97] neg_lo:[0,1] neg_hi:[0,1]") == ["v_mul_f32 v92, v92, -v96", "v_mul_f32 v93, v93, -v97"]
    assert u("v_pk_fma_f32 v[88:89], v[88:89], v[168:169], v[170:171] neg_lo:[0,0,1] neg_hi:[0,0,1]") == ["v_fma_f32 v88, v88, v168, -v170", "v_fma_f32 v89, v89, v169, -v171"]
    assert u("v_pk_fma_f32 v[92:93], v[92:93], 0.5, 0.5 op_sel_hi:[1,0,0]") == ["v_fma_f32 v92, v92, 0.5, 0.5", "v_fma_f32 v93, v93, 0.5, 0.5"]
    assert u("v_pk_add_f32 v[88:89], v[88:89], v[64:65]") == ["v_add_f32 v88, v88, v64", "v_add_f32 v89, v89, v65"]
    assert u("v_mov_b32 v1, v2") is None and u("v_cvt_pk_bf16_f32 v80, v88, v89") is None
    with pytest.raises(AssertionError):
        u("v_pk_mul_f32 v[92:93], v[92:93], v[96:97] op_sel_hi:[1,0]")          # a register pair read low-low would make the halves order-dependent: refused
    text = open(os.path.join(ROOT, "diffusion-image-captioning_amd", "csrc", "gemm_w4n_asm.inc")).read()
    assert "v_pk_" not in text
    assert any("v_pk_fma_f32" in ln for ln in G.generate(False, "gelu", ("flat=12", "pk=1"))[0])          # (the A/B variant keeps them)


def _run_bench(argv, env_extra=None, timeout=600):
    import subprocess
    env = dict(os.environ, **(env_extra or {}))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("world,opts,n_coll", [(8, "", 7), (2, "dp_single=1", 1)])
def test_bench_dry_run_rehearses_the_multi_gpu_plumbing(world, opts, n_coll):
    """Round-4 review item 6: no multi-GPU node has ever run this code, so the first real `bench.py --gpus 8` must not be able to fail on
    plumbing.  `--dry-run` goes through everything but the kernels on CPU tensors over gloo: the self-relaunch under torch.distributed.run,
    the rendezvous, per-rank seeds + shared timestep stream, the real GradReducer over the 12-layer flat gradient buffer in both schedules
    (7 sliced collectives / options.dp_single: exactly one), max-over-ranks timing and the JSON line with its `data_parallel` block."""
    import json
    r = _run_bench(["--gpus", str(world), "--dry-run", "--steps", "2", "--warmup", "1"], {"DIC_OPTIONS": opts})
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                       # rank 0 prints ONE line
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["dry_run"] is True and d["n_gpus"] == world and d["scaling"] == "weak" and d["config"]["parallelism"] == f"dp{world}"
    dp = d["data_parallel"]
    assert dp["rccl_ranks"] == world and dp["collectives_per_step"] == n_coll and len(dp["per_rank_captions_per_s"]) == world
    assert dp["gradient_bytes"] == 86830848 * 4
    assert d["options_non_default"] == ({"dp_single": True} if opts else {})


def _dp_dying_rank_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), DIC_OPTIONS="dp_timeout_s=30")
    sys.path.insert(0, ROOT)
    d = importlib.import_module("diffusion-image-captioning_amd")
    ParamStore = importlib.import_module("diffusion-image-captioning_amd.params").ParamStore
    d.parallel.init_from_env(backend="gloo")
    store = ParamStore(1, "cpu", bf16_shadow=False)

    class M:
        params = store
    red = d.parallel.GradReducer(M)
    red.layer_done(0)
    red.finish(type("T", (), {"grad_scale": 1.0})())              # one healthy step
    torch.distributed.barrier()
    if rank == 1:
        os._exit(0)                                               # dies between two steps, without a goodbye
    msg = "no error"
    try:
        red = d.parallel.GradReducer(M)
        red.layer_done(0)
        red.finish(type("T", (), {"grad_scale": 1.0})())
    except RuntimeError as e:
        msg = str(e)
    open(out, "w").write(msg)
    os._exit(0)


def test_data_parallel_failures_surface_as_one_clear_error_not_a_hang(tmp_path):
    """A peer that dies mid-run and a peer that never shows up both end in ONE RuntimeError that names the rank, the world size and what to check
    (parallel.init_from_env / GradReducer), within the process group's timeout (options.dp_timeout_s) -- not in a hang."""
    ctx = mp.get_context("spawn")
    out = str(tmp_path / "rank0.txt")
    port = _free_port()
    procs = [ctx.Process(target=_dp_dying_rank_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    msg = open(out).read()
    assert "data-parallel gradient exchange" in msg and "rank 0 of 2" in msg and "peer rank" in msg, msg
    # a rank that waits for a peer which never starts
    import subprocess
    code = ("import importlib, os, sys; sys.path.insert(0, %r); os.environ.update(RANK='0', WORLD_SIZE='2', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT='%d', "
            "DIC_OPTIONS='dp_timeout_s=5'); d = importlib.import_module('diffusion-image-captioning_amd'); d.parallel.init_from_env(backend='gloo')" % (ROOT, _free_port()))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "data-parallel start-up failed on rank 0 of 2" in r.stderr, r.stderr[-1500:]


def test_options_record_is_the_only_switchboard_and_pins_the_shipped_configuration():
    """Round-4 review: engine behaviour hung on ~40 DIC_* environment reads.  One record now (options.py): its defaults ARE the shipped
    configuration (pinned here -- a changed default must change this test), the package's hot-path modules and the C sources read no
    environment variable, DIC_OPTIONS / the legacy names parse into it, unknown names are refused, and the library takes its four
    process-global switches through dic_set_option."""
    import dataclasses
    opts = importlib.import_module("diffusion-image-captioning_amd.options")
    shipped = dict(wgrad_stream=True, wgrad_group="pair", bwd_sets=0, wgrad_cu_cap=0, ln_npart=512, gemm_tile="auto", gemm_v1=False, gelu_d=True, ce_fused=True,
                   head_center="1", uvt32=True, split_set="auto", lo_row_stride=16, qkv_pred=True, cen=True, cen_operand=True, res32="auto", sample_raw=True, streamed_adamw=True,
                   sample_graph=True, sample_w4a=True, sample_two_heights=True, gemm_w4a=True, gemm_w4a_mask=0x73, gemm_w4a_rows=256, gemm_w4n=False, gemm_w4n_mask=0x740, gemm_w4n_kmax=1024, gemm_w4n_flat=True, gemm_two_heights=False, dp_group=3,
                   dp_single=False, dp_cu_cap=0, dp_timing=False, force_reducer=False, dp_timeout_s=600)
    assert dataclasses.asdict(opts.Options()) == shipped
    assert opts.Options().n_bwd_sets == 4 and opts.from_env({"DIC_OPTIONS": "wgrad_group=1"}).n_bwd_sets == 2
    o = opts.from_env({"DIC_OPTIONS": "cen=0, wgrad_group=1,dp_group=4,gemm_w4a_mask=0xff", "DIC_WGRAD_STREAM": "0", "DIC_GEMM_W4A": "0"})
    assert o.non_default() == {"cen": False, "wgrad_group": "1", "dp_group": 4, "wgrad_stream": False, "sample_w4a": False, "gemm_w4a": False, "gemm_w4a_mask": 255}
    with pytest.raises(ValueError):
        opts.from_env({"DIC_OPTIONS": "no_such_switch=1"})
    with pytest.raises(ValueError):
        opts.from_env({"DIC_OPTIONS": "cen=maybe"})
    assert set(opts.LEGACY_ENV.values()) <= set(shipped)
    pkg = os.path.join(ROOT, "diffusion-image-captioning_amd")
    for mod in ("engine.py", "diffusion.py", "graph.py", "harness.py", "params.py", "train_embedding.py"):
        assert "environ" not in open(os.path.join(pkg, mod)).read(), mod
    par = open(os.path.join(pkg, "parallel.py")).read()
    assert set(re.findall(r'environ(?:\.get|\.setdefault)?\(?\[?"(\w+)"', par)) <= {"WORLD_SIZE", "LOCAL_RANK", "RANK", "MASTER_ADDR", "HSA_ENABLE_IPC_MODE_LEGACY",
                                                                                   "DIC_DIST_SHARE_GPU", "DIC_DIST_BACKEND"}
    for src in ("gemm.hip", "gemm_w4a.h", "attn.hip", "misc.hip", "common.h"):
        assert "getenv" not in open(os.path.join(pkg, "csrc", src)).read(), src
    L = dic.lib()
    assert L.dic_set_option(b"no_such_option", 1) != 0
    # a switch deleted with its code path is refused, not ignored; run-time changes of the library's switches go through set_option (record and library together)
    with pytest.raises(ValueError):
        opts.from_env({"DIC_LO_MODE": "pass2"})
    try:
        opts.set_option("gemm_w4a", False)
        assert opts.OPT.non_default().get("gemm_w4a") is False and L.dic_gemm_set_w4a(0) == 0
        with pytest.raises(RuntimeError):
            opts.set_option("gemm_w4a_rows", 192)
        with pytest.raises(ValueError):
            opts.set_option("no_such_switch", 1)
    finally:
        opts.set_option("gemm_w4a", True)
    assert L.dic_gemm_set_w4a(1) == 1 and "gemm_w4a" not in opts.OPT.non_default()


def test_unknown_precision_mode_is_refused_before_anything_else():
    """dtype is one of fp32 / bf16 (= bf16m) / bf16r / bf16w; a typo used to fall through to the fp32 engine silently."""
    with pytest.raises(ValueError):
        dic.DistilBertModel(None, None, dtype="bf16x")
