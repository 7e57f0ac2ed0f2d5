"""CustomTransformer (reference dquartic/model/building_blocks.py).
CPU: the oracle restatement against fixtures captured from the reference module itself (oracle/make_golden_tfm.py ->
tests/golden/tfm_tiny.npz: forward output and autograd gradients), the drop-in module's state_dict layout and default
initialisation against the same fixtures.  GPU: the fp32 matrix-core GEMM against float64 products over layouts, ragged edges and
split-K; the HIP forward / backward through the C ABI against the fixtures and against the oracle at larger shapes."""
import numpy as np
import pytest
import torch

from oracle import dq_oracle_tfm as OT

TAGS = ("a", "b")


def _case(g, tag):
    cfg = [int(v) for v in g[f"{tag}/config"]]
    params = {k[len(tag) + 7:]: torch.from_numpy(np.array(g[k])) for k in g if k.startswith(f"{tag}/param/")}
    grads = {k[len(tag) + 6:]: torch.from_numpy(np.array(g[k])) for k in g if k.startswith(f"{tag}/grad/")}
    t = lambda n: torch.from_numpy(np.array(g[f"{tag}/{n}"]))
    return cfg, params, grads, t("x_t"), t("t"), t("x_cond"), t("probe"), t("out")


def _rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.mark.parametrize("tag", TAGS)
def test_oracle_matches_reference_forward_and_gradients(golden, tag):
    g = golden("tfm_tiny.npz")
    cfg, params, grads, x, t, c, probe, out = _case(g, tag)
    p = {k: v.clone().requires_grad_() for k, v in params.items()}
    x, c = x.clone().requires_grad_(), c.clone().requires_grad_()
    y = OT.forward(p, x, t, c, cfg[2])
    assert _rel(y.detach(), out) < 2e-6
    assert np.array_equal(g[f"{tag}/out"], g[f"{tag}/out_eval"]) or _rel(torch.from_numpy(g[f"{tag}/out_eval"]), out) < 2e-6
    (y * probe).sum().backward()
    for k, v in p.items():
        assert _rel(v.grad, grads[k]) < 5e-6, k
    assert _rel(x.grad, grads["x_t"]) < 5e-6 and _rel(c.grad, grads["x_cond"]) < 5e-6


def test_rope_and_time_tables_known_answers():
    sin, cos = OT.rope_tables(4, 8)
    assert sin.shape == (4, 4) and torch.all(sin[0] == 0) and torch.all(cos[0] == 1)
    assert abs(float(sin[1, 0]) - np.sin(1.0)) < 1e-7 and abs(float(sin[3, 2]) - np.sin(3 * 10000 ** -0.5)) < 1e-7
    f = OT.time_freqs(8)
    assert float(f[0]) == 1.0 and abs(float(f[3]) - 1e-4) < 1e-9
    x = torch.randn(2, 5, 8)
    y = OT.apply_rope(x)
    assert torch.allclose(y.pow(2).sum(-1), x.pow(2).sum(-1), atol=1e-5)  # rotations preserve the pair norms
    assert torch.equal(y[:, 0], x[:, 0])  # position 0 is not rotated


@pytest.mark.parametrize("tag", TAGS)
def test_module_layout_and_default_init_match_reference(golden, tag):
    from dquartic.model.building_blocks import CustomTransformer

    g = golden("tfm_tiny.npz")
    cfg, params, *_ = _case(g, tag)
    torch.manual_seed(0 if tag == "a" else 1)  # the seed oracle/make_golden_tfm.py constructed the reference module under
    net = CustomTransformer(input_dim=cfg[0], hidden_dim=cfg[1], num_heads=cfg[2], num_layers=cfg[3])
    sd = net.state_dict()
    assert list(sd.keys()) == [str(k) for k in g[f"{tag}/keys"]]
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(params[k].shape), k
        assert torch.equal(v, params[k]), k  # same initialisers, same RNG consumption order
    net.load_state_dict({k: v + 1 for k, v in params.items()})
    assert torch.equal(net.state_dict()["layers.0.norm1.bias"], params["layers.0.norm1.bias"] + 1)


def test_unsupported_configurations_fail_loudly():
    from dquartic.model.building_blocks import CustomTransformer

    with pytest.raises(NotImplementedError, match="input_dim % 4"):
        CustomTransformer(input_dim=30, hidden_dim=16, num_heads=2, num_layers=1)
    with pytest.raises(NotImplementedError):
        CustomTransformer(input_dim=32, hidden_dim=16, num_heads=3, num_layers=1)
    net = CustomTransformer(input_dim=24, hidden_dim=16, num_heads=2, num_layers=1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.zeros(1, 3, 24), torch.zeros(1, dtype=torch.long), torch.zeros(1, 2))


# ------------------------------------------------------------------------------------------------ GPU
def _gemm(A, B, M, N, K, a_k, b_k, bias=None, C0=None, splits=0):
    from dquartic import _native as N_

    lib = N_.lib()
    C = torch.zeros(M, N, device="cuda") if C0 is None else C0.clone()
    n_s = max(int(lib.dq_gemm_scratch_floats(M, N, K)), splits * (M + 256) * (N + 128) if splits else 4)  # forced split: tile-padded
    scratch = torch.empty(n_s, device="cuda")
    N_.check(lib.dq_gemm(N_.ptr(A), N_.ptr(B), N_.ptr(C), N_.ptr(bias), M, N, K, A.shape[1], B.shape[1], N, int(a_k), int(b_k),
                         0 if C0 is None else 1, splits, N_.ptr(scratch), scratch.numel(), N_.stream_ptr()), "dq_gemm")
    return C


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(5, 16, 24), (34, 1024, 4000), (130, 260, 68), (257, 129, 36), (64, 128, 32), (1, 4, 4), (300, 40, 7)])
@pytest.mark.parametrize("layout", ["kk", "kn", "mn"])
def test_gemm_against_float64(M, N, K, layout):
    torch.manual_seed(M * 7 + N)
    pad = lambda v: (v + 3) // 4 * 4
    a_k, b_k = layout[0] == "k", layout[1] == "k"
    A = torch.randn((M, pad(K)) if a_k else (K, pad(M)), device="cuda")
    B = torch.randn((N, pad(K)) if b_k else (K, pad(N)), device="cuda")
    Am = (A[:, :K] if a_k else A[:, :M].t()).double()
    Bm = (B[:, :K].t() if b_k else B[:, :N]).double()
    bias = torch.randn(N, device="cuda")
    ref = Am @ Bm + bias.double()
    got = _gemm(A, B, M, N, K, a_k, b_k, bias=bias)
    tol = 2e-6 * float(ref.abs().max()) * max(1.0, K ** 0.5 / 8)
    assert float((got.double() - ref).abs().max()) < tol
    # += into an existing C, and a forced split-K: same numbers up to fp32 summation order
    C0 = torch.randn(M, N, device="cuda")
    got2 = _gemm(A, B, M, N, K, a_k, b_k, C0=C0, splits=3 if K >= 64 else 0)
    assert float((got2.double() - (Am @ Bm + C0.double())).abs().max()) < tol
    assert torch.equal(_gemm(A, B, M, N, K, a_k, b_k, bias=bias), got)  # bitwise repeatable


def _module(cfg, params):
    from dquartic.model.building_blocks import CustomTransformer

    net = CustomTransformer(input_dim=cfg[0], hidden_dim=cfg[1], num_heads=cfg[2], num_layers=cfg[3])
    net.load_state_dict(params)
    return net.cuda()


@pytest.mark.gpu
@pytest.mark.parametrize("tag", TAGS)
def test_hip_forward_backward_match_reference_fixtures(golden, tag):
    g = golden("tfm_tiny.npz")
    cfg, params, grads, x, t, c, probe, out = _case(g, tag)
    net = _module(cfg, params)
    with torch.no_grad():
        y = net(x.cuda(), t.cuda(), c.cuda())
    assert _rel(y.cpu(), out) < 1e-5
    xg, cg = x.cuda().requires_grad_(), c.cuda().requires_grad_()
    y = net(xg, t.cuda(), cg)
    assert _rel(y.detach().cpu(), out) < 1e-5
    (y * probe.cuda()).sum().backward()
    for k, p in net.named_parameters():
        assert _rel(p.grad.cpu(), grads[k]) < 2e-5, k
    assert _rel(xg.grad.cpu(), grads["x_t"]) < 2e-5 and _rel(cg.grad.cpu(), grads["x_cond"]) < 2e-5


@pytest.mark.gpu
def test_two_forwards_before_their_backwards(golden):
    """two grad-enabled forwards in flight (micro-batches whose losses are summed), then the backwards, in either order: each
    forward owns its training workspace and the native handle remembers every workspace that holds saved activations, so both backwards
    run and the summed gradient equals the sum of the separate ones."""
    g = golden("tfm_tiny.npz")
    cfg, params, grads, x, t, c, probe, out = _case(g, TAGS[0])
    net = _module(cfg, params)
    x2, c2 = (x * 0.5 + 0.1).cuda(), (c * 0.7 - 0.2).cuda()

    def grads_of(fn):
        for p in net.parameters():
            p.grad = None
        fn()
        return torch.cat([p.grad.reshape(-1) for p in net.parameters()]).clone()

    f1 = lambda: (net(x.cuda(), t.cuda(), c.cuda()) * probe.cuda()).sum()
    f2 = lambda: (net(x2, t.cuda(), c2) ** 2).mean()
    g1 = grads_of(lambda: f1().backward())
    g2 = grads_of(lambda: f2().backward())
    both = grads_of(lambda: (f1() + f2()).backward())  # forward 1, forward 2, one backward through both (the later forward's first)

    def earlier_first():
        a, b = f1(), f2()
        a.backward()   # the EARLIER forward's backward while the later one is still pending
        b.backward()
    sep = grads_of(earlier_first)
    tol = 2e-5 * float((g1 + g2).abs().max())
    assert float((both - (g1 + g2)).abs().max()) < tol
    assert float((sep - (g1 + g2)).abs().max()) < tol


@pytest.mark.gpu
@pytest.mark.parametrize("D,H,heads,layers,B,S1,S2", [(200, 64, 4, 2, 3, 34, 20), (1000, 128, 1, 1, 2, 70, 9), (64, 256, 8, 3, 5, 17, 34),
                                                     (96, 512, 8, 1, 2, 34, 34), (72, 1008, 4, 1, 1, 9, 5), (48, 1032, 2, 1, 1, 5, 7)])
def test_hip_matches_oracle_at_larger_shapes(D, H, heads, layers, B, S1, S2):
    from dquartic.model.building_blocks import CustomTransformer

    params = OT.init_params(D, H, layers, seed=D + H)
    net = CustomTransformer(input_dim=D, hidden_dim=H, num_heads=heads, num_layers=layers)
    net.load_state_dict(params)
    net = net.cuda()
    g = torch.Generator().manual_seed(5)
    x, c = torch.randn(B, S1, D, generator=g), torch.randn(B, S2, generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    probe = torch.randn(B, S1, D, generator=g)
    p = {k: v.clone().requires_grad_() for k, v in params.items()}
    xr, cr = x.clone().requires_grad_(), c.clone().requires_grad_()
    ref = OT.forward(p, xr, t, cr, heads)
    (ref * probe).sum().backward()
    xg, cg = x.cuda().requires_grad_(), c.cuda().requires_grad_()
    y = net(xg, t.cuda(), cg)
    assert _rel(y.detach().cpu(), ref.detach()) < 2e-5
    (y * probe.cuda()).sum().backward()
    worst = max(_rel(q.grad.cpu(), p[k].grad) for k, q in net.named_parameters())
    assert worst < 1e-4, worst
    assert _rel(xg.grad.cpu(), xr.grad) < 1e-4 and _rel(cg.grad.cpu(), cr.grad) < 1e-4
    # inference-mode call (shared layer buffers) gives the training-mode numbers
    with torch.no_grad():
        assert torch.equal(net(x.cuda(), t.cuda(), c.cuda()), y.detach())


def test_adapter_state_dict_is_the_transformers():
    from dquartic.model.building_blocks import CustomTransformer, DDIMTransformerAdapter

    net = DDIMTransformerAdapter(CustomTransformer(input_dim=24, hidden_dim=16, num_heads=2, num_layers=1))
    sd = net.state_dict()
    assert list(sd)[0] == "input_projection.weight" and not any(k.startswith("transformer.") for k in sd)
    net.load_state_dict({k: torch.full_like(v, 0.5) for k, v in sd.items()})
    assert float(net.transformer.flat_params.min()) == 0.5 == float(net.flat_params.max())
    assert [n for n, _ in net.trainable_named()] == list(sd)


@pytest.mark.gpu
@pytest.mark.parametrize("pred_type", ["eps", "x0"])
def test_native_train_step_matches_oracle(pred_type):
    """_train_one_batch on the adapter = q_sample + forward + (weighted) MSE + backward + clip + AdamW, all native: loss, the
    pre-clip gradient norm and the updated parameters against the oracle driven by torch autograd / torch.optim.AdamW."""
    from dquartic.model.building_blocks import CustomTransformer, DDIMTransformerAdapter
    from dquartic.model.model import DDIMDiffusionModel
    from dquartic.model.model_interface import FlatAdamW

    D, H, heads, layers, B, RT = 48, 32, 4, 2, 3, 6
    params = OT.init_params(D, H, layers, seed=3)
    tf = CustomTransformer(input_dim=D, hidden_dim=H, num_heads=heads, num_layers=layers)
    tf.load_state_dict(params)
    net = DDIMTransformerAdapter(tf).cuda()
    dm = DDIMDiffusionModel(model_class=net, num_timesteps=1000, beta_schedule_type="cosine", pred_type=pred_type, auto_normalize=True,
                            ms1_loss_weight=0.0, device="cuda")
    dm._set_optimizer(1e-3)
    assert isinstance(dm.optimizer, FlatAdamW)
    g = torch.Generator().manual_seed(11)
    x0, c2, c1 = torch.rand(B, RT, D, generator=g), torch.rand(B, RT, D, generator=g), torch.rand(B, RT, generator=g)
    t = torch.tensor([999, 400, 3])
    noise = torch.randn(B, RT, D, generator=g)
    loss = dm.train_step_fused(x0.cuda(), c2.cuda(), c1.cuda(), t=t.cuda(), noise=noise.cuda())
    dm.optimizer.grad_scale = 1.0
    dm.optimizer.step()
    # oracle
    p = {k: v.clone().requires_grad_() for k, v in params.items()}
    ab = dm.alpha_bars.cpu()[t][:, None, None]
    xn = 2 * x0 - 1
    x_t = torch.sqrt(ab) * xn + torch.sqrt(1 - ab) * noise
    out = OT.forward(p, x_t, t, 2 * c1 - 1, heads)
    if pred_type == "eps":
        ref = torch.nn.functional.mse_loss(out, noise)
    else:
        ref = (((out - xn) ** 2).flatten(1).mean(1) * dm.loss_weight.cpu()[t]).mean()
    opt = torch.optim.AdamW(list(p.values()), lr=1e-3)
    ref.backward()
    gn = torch.nn.utils.clip_grad_norm_(list(p.values()), 10.0)
    opt.step()
    assert abs(float(loss) - float(ref)) < 2e-5 * max(1.0, abs(float(ref)))
    assert abs(float(dm.optimizer.last_grad_norm) - float(gn)) < 1e-4 * float(gn)
    for k, v in net.state_dict().items():
        d = (v.cpu() - p[k].detach()).abs()
        if k.endswith("attention.in_proj_bias"):
            # the key bias shifts every score of a row equally: its gradient is zero up to rounding, and Adam's g / (|g| + eps)
            # turns that rounding noise (1e-9 on both sides) into updates of the order of lr -- not comparable
            d[H:2 * H] = 0
        assert float(d.max()) < 2e-5, k


@pytest.mark.gpu
def test_ddim_adapter_trains_the_transformer():
    """DDIMDiffusionModel drives the transformer through the 4-argument adapter (generic autograd path)."""
    from dquartic.model.building_blocks import CustomTransformer, DDIMTransformerAdapter
    from dquartic.model.model import DDIMDiffusionModel

    torch.manual_seed(0)
    net = DDIMTransformerAdapter(CustomTransformer(input_dim=64, hidden_dim=32, num_heads=2, num_layers=1)).cuda()
    dm = DDIMDiffusionModel(model_class=net, num_timesteps=1000, beta_schedule_type="cosine", pred_type="eps", auto_normalize=True,
                            ms1_loss_weight=0.0, device="cuda")
    x0, c2, c1 = torch.rand(2, 10, 64, device="cuda"), torch.rand(2, 10, 64, device="cuda"), torch.rand(2, 10, device="cuda")
    loss = dm.train_step(x0, c2, c1)
    loss.backward()
    assert torch.isfinite(loss) and all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())
    xs, noise = dm.sample(torch.randn(2, 10, 64, device="cuda"), c2, c1, num_steps=3)
    assert xs.shape == (2, 10, 64) and torch.isfinite(xs).all()


@pytest.mark.gpu
def test_bucketed_backward_reports_every_slice_and_overlapped_allreduce_keeps_the_gradient(tmp_path):
    """dq_tfm_bwd_buckets: the callbacks arrive in the order of dq_tfm_bucket_info with its slices, the gradient is the plain
    backward's, and BucketedAllReduce (communication stream + events, RCCL with one rank: sum over one rank = identity) leaves it
    unchanged -- the stream choreography is what runs here; the two-rank arithmetic is tests/test_dp_gloo.py."""
    import torch.distributed as dist
    from dquartic.model.building_blocks import CustomTransformer
    from dquartic.model.model_interface import BucketedAllReduce

    D, H, heads, layers, B, S1, S2 = 64, 128, 4, 3, 2, 17, 9
    net = CustomTransformer(input_dim=D, hidden_dim=H, num_heads=heads, num_layers=layers)
    net.load_state_dict(OT.init_params(D, H, layers, seed=9))
    net = net.cuda()
    net._ensure_flat()
    g = torch.Generator().manual_seed(2)
    x, c = torch.randn(B, S1, D, generator=g).cuda(), torch.randn(B, S2, generator=g).cuda()
    t, gout = torch.tensor([5, 900]).cuda(), torch.randn(B, S1, D, generator=g).cuda()
    net._run_fwd(x, t, c, training=True)
    plain = torch.empty_like(net.flat_params)
    net._run_bwd(x, c, gout, plain, False, False, accumulate=False)
    seen = []
    hooked = torch.full_like(plain, float("nan"))
    net._run_bwd(x, c, gout, hooked, False, False, accumulate=False, on_bucket=lambda i, o, n: seen.append((i, o, n)))
    assert seen == [(i, o, n) for i, (o, n) in enumerate(net.grad_buckets())] and len(seen) == layers + 1
    assert torch.equal(hooked, plain)
    with pytest.raises(ValueError, match="boom"):  # an exception inside the callback surfaces after the native call returns
        net._run_bwd(x, c, gout, hooked, False, False, accumulate=False, on_bucket=lambda i, o, n: (_ for _ in ()).throw(ValueError("boom")))
    dist.init_process_group("nccl", init_method=f"file://{tmp_path}/rdzv", rank=0, world_size=1)
    try:
        reduced = torch.full_like(plain, float("nan"))
        red = BucketedAllReduce(reduced)
        net._run_bwd(x, c, gout, reduced, False, False, accumulate=False, on_bucket=red.on_bucket)
        red.finish()
        torch.cuda.synchronize()
        assert torch.equal(reduced, plain)
    finally:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------ data-parallel (CPU, gloo)
def _dp_worker(rank, world, port, q):
    import os
    import sys

    import torch.distributed as dist

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, repo)
    sys.path.insert(0, os.path.join(repo, "diffusion-deconvolution-dia-msms-data_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dquartic.model.building_blocks import CustomTransformer, DDIMTransformerAdapter
        from oracle import dq_oracle_tfm as O

        D, H, heads, layers = 24, 16, 2, 1
        params = O.init_params(D, H, layers, seed=1)
        net = DDIMTransformerAdapter(CustomTransformer(D, H, heads, layers))
        net.load_state_dict(params)
        g = torch.Generator().manual_seed(3)
        X, C, T = torch.randn(4, 5, D, generator=g), torch.randn(4, 3, generator=g), torch.tensor([1, 50, 700, 999])
        probe = torch.randn(4, 5, D, generator=g)

        def flat_grad(idx):  # gradient of the mean over `idx` of the per-sample objective, in the module's flat layout
            p = {k: v.clone().requires_grad_() for k, v in params.items()}
            y = O.forward(p, X[idx], T[idx], C[idx], heads)
            ((y * probe[idx]).flatten(1).sum(1)).mean().backward()
            return torch.cat([p[n].grad.reshape(-1) for n, _ in net.trainable_named()])

        mine = list(range(rank, 4, world))  # rank r owns windows r, r + world, ...
        buf = net.flat_grads(zero=True)
        buf.copy_(flat_grad(mine))
        dist.all_reduce(buf)  # the ONE exchange step of the train step (model_interface.py: flat RCCL all-reduce)
        buf.mul_(1.0 / world)  # FlatAdamW.grad_scale
        ref = flat_grad(list(range(4)))
        q.put((rank, float((buf - ref).abs().max() / ref.abs().max()), buf.numel()))
    finally:
        dist.destroy_process_group()


def test_transformer_data_parallel_exchange_gloo_world2():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29631
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    out = sorted(q.get(timeout=5) for _ in range(2))
    assert [r for r, _, _ in out] == [0, 1]
    for _, err, n in out:
        assert err < 1e-5 and n > 0
    assert all(p.exitcode == 0 for p in procs)


def test_checkpoint_roundtrip_with_the_adapter(tmp_path):
    """The harness's checkpoint (reference model_interface.py:561-626) with the transformer behind its adapter: model keys are
    the transformer's own (a reference CustomTransformer checkpoint loads), the optimizer state has torch AdamW's layout."""
    from dquartic.model.building_blocks import CustomTransformer, DDIMTransformerAdapter
    from dquartic.model.model import DDIMDiffusionModel

    def make(seed):
        torch.manual_seed(seed)
        return DDIMDiffusionModel(model_class=DDIMTransformerAdapter(CustomTransformer(24, 16, 2, 1)), device="cpu")

    dm = make(0)
    dm._set_optimizer(1e-5)
    dm.optimizer._buffers()
    dm.optimizer._m.uniform_(-1, 1); dm.optimizer._v.uniform_(0, 1); dm.optimizer._step = 5
    dm.optimizer._publish_state()
    sch = dm._get_lr_schedule_with_warmup(2, 10)
    path = str(tmp_path / "tfm.ckpt")
    dm.save_checkpoint(sch, 2, 0.5, path)
    ck = torch.load(path, weights_only=False)
    assert list(ck["model_state_dict"])[0] == "input_projection.weight"
    assert len(ck["optimizer_state_dict"]["state"]) == len(ck["model_state_dict"])
    dm2 = make(1)
    dm2._set_optimizer(1e-5)
    ep, best, _ = dm2.load_checkpoint(dm2._get_lr_schedule_with_warmup(2, 10), path, "cpu")
    assert (ep, best) == (2, 0.5)
    assert torch.equal(dm2.model.flat_params, dm.model.flat_params)
    assert torch.equal(dm2.optimizer._m, dm.optimizer._m) and torch.equal(dm2.optimizer._v, dm.optimizer._v) and dm2.optimizer._step == 5


# ------------------------------------------------------------------------------------------------ bf16x3 precision mode (GPU)
def _gemm_bf16x3(A, B, M, N, K, a_k, b_k, bias=None, C0=None, splits=0):
    from dquartic import _native as N_

    lib = N_.lib()
    C = torch.zeros(M, N, device="cuda") if C0 is None else C0.clone()
    n_s = max(int(lib.dq_gemm_scratch_floats(M, N, K)), splits * (M + 256) * (N + 128) if splits else 4)
    scratch = torch.empty(n_s, device="cuda")
    N_.check(lib.dq_gemm_bf16x3(N_.ptr(A), N_.ptr(B), N_.ptr(C), N_.ptr(bias), M, N, K, A.shape[1], B.shape[1], N, int(a_k), int(b_k),
                                0 if C0 is None else 1, splits, N_.ptr(scratch), scratch.numel(), N_.stream_ptr()), "dq_gemm_bf16x3")
    return C


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(5, 16, 24), (34, 1024, 4000), (130, 260, 68), (257, 129, 36), (64, 128, 32), (1, 4, 4), (300, 40, 7),
                                   (1088, 1024, 1024)])
@pytest.mark.parametrize("layout", ["kk", "kn", "mn"])
def test_gemm_bf16x3_against_float64(M, N, K, layout):
    """The split-bf16 three-pass product (DQ_PRECISION_BF16X3): ~16 mantissa bits per operand.  STATED TOLERANCE of this mode:
    |C - C_exact| <= 2e-5 * sqrt(K) * max|A| * max|B| per element (a random-sign sum of K terms, each off by <= ~2^-16 relative) --
    about 10x the exact-fp32 kernel's; bitwise repeatable; same tails / layouts / split-K / += handling as the fp32 kernel."""
    torch.manual_seed(M * 7 + N)
    pad = lambda v: (v + 3) // 4 * 4
    a_k, b_k = layout[0] == "k", layout[1] == "k"
    A = torch.randn((M, pad(K)) if a_k else (K, pad(M)), device="cuda")
    B = torch.randn((N, pad(K)) if b_k else (K, pad(N)), device="cuda")
    Am = (A[:, :K] if a_k else A[:, :M].t()).double()
    Bm = (B[:, :K].t() if b_k else B[:, :N]).double()
    bias = torch.randn(N, device="cuda")
    ref = Am @ Bm + bias.double()
    got = _gemm_bf16x3(A, B, M, N, K, a_k, b_k, bias=bias)
    tol = 2e-5 * K ** 0.5 * float(Am.abs().max() * Bm.abs().max())
    err = float((got.double() - ref).abs().max())
    assert err < tol, (err, tol)
    exact = _gemm(A, B, M, N, K, a_k, b_k, bias=bias)
    assert float((exact.double() - ref).abs().max()) <= err + 1e-6 * float(ref.abs().max())  # (the fp32 kernel is at least as close)
    C0 = torch.randn(M, N, device="cuda")
    got2 = _gemm_bf16x3(A, B, M, N, K, a_k, b_k, C0=C0, splits=3 if K >= 64 else 0)
    assert float((got2.double() - (Am @ Bm + C0.double())).abs().max()) < tol
    assert torch.equal(_gemm_bf16x3(A, B, M, N, K, a_k, b_k, bias=bias), got)


@pytest.mark.gpu
def test_transformer_bf16x3_mode_within_its_stated_tolerance():
    """CustomTransformer with set_precision("bf16x3"): output within 1e-4 and parameter / input gradients within 5e-4 of the oracle
    (relative to each tensor's largest entry; observed on MI355X: 1.2e-5 / 3.4e-5) -- the tolerance of this mode; the default fp32
    mode on the same inputs stays within the 2e-5 / 1e-4 of the fp32 tests (observed 1.1e-6 / 1.8e-6), and switching back restores
    it bit for bit."""
    from dquartic.model.building_blocks import CustomTransformer

    D, H, heads, layers, B, S1, S2 = 1000, 256, 8, 3, 3, 21, 17
    params = OT.init_params(D, H, layers, seed=D + H)
    net = CustomTransformer(input_dim=D, hidden_dim=H, num_heads=heads, num_layers=layers)
    net.load_state_dict(params)
    net = net.cuda()
    g = torch.Generator().manual_seed(5)
    x, c = torch.randn(B, S1, D, generator=g), torch.randn(B, S2, generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    probe = torch.randn(B, S1, D, generator=g)
    p = {k: v.clone().requires_grad_() for k, v in params.items()}
    xr, cr = x.clone().requires_grad_(), c.clone().requires_grad_()
    ref = OT.forward(p, xr, t, cr, heads)
    (ref * probe).sum().backward()

    def run():
        for q in net.parameters():
            q.grad = None
        xg, cg = x.cuda().requires_grad_(), c.cuda().requires_grad_()
        y = net(xg, t.cuda(), cg)
        (y * probe.cuda()).sum().backward()
        worst = max(_rel(q.grad.cpu(), p[k].grad) for k, q in net.named_parameters())
        return y.detach().clone(), _rel(y.detach().cpu(), ref.detach()), max(worst, _rel(xg.grad.cpu(), xr.grad), _rel(cg.grad.cpu(), cr.grad))

    y32, e32, g32 = run()
    assert e32 < 2e-5 and g32 < 1e-4
    net.set_precision("bf16x3")
    y16, e16, g16 = run()
    print("bf16x3 mode: output rel err", e16, "worst gradient rel err", g16, "(fp32 mode:", e32, g32, ")")
    assert e16 < 1e-4 and g16 < 5e-4
    assert not torch.equal(y16, y32)  # the mode really ran
    net.set_precision("fp32")
    assert torch.equal(run()[0], y32)
    with pytest.raises(ValueError):
        net.set_precision("fp8")
