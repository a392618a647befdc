"""HIP-graph replay of the train step (train.GraphedStep): replay == eager TO THE BIT over consecutive steps -- render,
losses, backward, fused Adam -- with the random jitter on, while the per-iteration scalars (learning rates, Adam's bias
corrections, cos_anneal_ratio, flip_saturation) change every step through device memory."""
import pytest
import torch

from neuraludf_amd import synth

pytestmark = pytest.mark.gpu

RCONF = dict(n_samples=32, n_importance=32, n_outside=0, up_sample_steps=2, perturb=1.0)
RCONF_BG = dict(n_samples=32, n_importance=16, n_outside=8, up_sample_steps=2, perturb=1.0)


def _schedule(i):
    return dict(cos_anneal_ratio=min(1.0, 0.3 + 0.1 * i), flip_saturation=(0.0 if i < 2 else 0.9)), 5e-4 * (1.0 - 0.05 * i)


def _run(rconf, graphed, n_steps, n_rays=192):
    from neuraludf_amd.train import Trainer, GraphedStep
    dev = torch.device("cuda:0")
    tr = Trainer(dev, rconf, seed=0, fused_adam=True)
    tr.renderer.diagnostics = False
    scene = synth.make_scene("tiny")
    stepper = GraphedStep(tr, eager_steps=2) if graphed else tr.step
    torch.manual_seed(1234)                      # the jitter draws of `render` come from the default CUDA generator
    hist = []
    for i in range(n_steps):
        rays = synth.make_rays(scene, i % 3, n_rays, seed=100 + i)
        batch = {k: v.to(dev) for k, v in rays.items()}
        kw, lr = _schedule(i)
        for gi, g in enumerate(tr.optimizer.param_groups):
            g["lr"] = lr * (0.2 if gi == 0 else 1.0)
        loss, out = stepper(batch, **kw)
        hist.append((loss.clone(), out["color"].detach().clone(), out["weight_sum"].detach().clone(),
                     [p.detach().clone() for g in tr.param_groups for p in g]))
    torch.cuda.synchronize()
    return tr, stepper, hist


@pytest.mark.parametrize("rconf", [RCONF, RCONF_BG], ids=["no_background", "background_nerf"])
def test_graph_replay_equals_eager_to_the_bit(rconf):
    n = 7                                            # 2 eager steps, capture + replay on the 3rd call, 4 more replays
    _, _, eager = _run(rconf, False, n)
    tr, gs, graph = _run(rconf, True, n)
    assert gs.enabled and gs.replays == n - 2 and len(gs.graphs) == 1
    for i, (a, b) in enumerate(zip(eager, graph)):
        assert torch.equal(a[0], b[0]), ("loss", i, float(a[0]), float(b[0]))
        assert torch.equal(a[1], b[1]), ("colour", i)
        assert torch.equal(a[2], b[2]), ("weight_sum", i)
        for j, (p, q) in enumerate(zip(a[3], b[3])):
            assert torch.equal(p, q), ("parameter", i, j)
    # the optimizer's host state followed the replays (checkpoints stay interchangeable)
    steps = {float(st["step"]) for st in tr.optimizer.state.values()}
    assert steps == {float(n)}, steps


def test_new_loss_weights_are_a_new_capture_and_dp_stays_eager():
    from neuraludf_amd.train import Trainer, GraphedStep
    dev = torch.device("cuda:0")
    tr = Trainer(dev, RCONF, seed=0, fused_adam=True)
    gs = GraphedStep(tr, eager_steps=1)
    scene = synth.make_scene("tiny")
    batch = {k: v.to(dev) for k, v in synth.make_rays(scene, 0, 128, seed=5).items()}
    for _ in range(3):
        gs(batch)
    assert gs.replays == 2 and len(gs.graphs) == 1
    tr.tc["igr_weight"] = 0.2                         # a by-value kernel / python scalar of the step: part of the key
    l0, _ = gs(batch)                                 # eager again under the new key
    assert len(gs.graphs) == 2 and gs.replays == 2
    l1, _ = gs(batch)
    assert gs.replays == 3 and torch.isfinite(l1)
    other = {k: v.to(dev) for k, v in synth.make_rays(scene, 1, 64, seed=6).items()}      # another batch shape
    gs(other)
    assert len(gs.graphs) == 3
    tr2 = Trainer(dev, RCONF, seed=0, fused_adam=False)           # torch.optim.Adam: no dynamic-scalar path -> eager
    assert not GraphedStep(tr2).enabled


def test_blending_steps_replay_bit_identically():
    """pixel blending over source views (colour_pixel term on) and the full config-3 loss (pixel + patch blending, SSIM
    patch loss with its top-30 % trim -- formed on the device, loss._trimmed_mean; the patch warp's camera constants are
    computed outside the capture) are captured and replay bit-identically."""
    from neuraludf_amd.train import Trainer, GraphedStep
    dev = torch.device("cuda:0")
    rconf = dict(n_samples=24, n_importance=12, n_outside=0, up_sample_steps=3, perturb=1.0, upsampling_type="mix",
                 use_norm_grad_for_cosine=True, h_patch_size=3)
    scene = synth.make_scene("tiny")
    src = {k: v.to(dev) for k, v in synth.make_source_views(scene, 0, 8).items()}
    batch = {k: v.to(dev) for k, v in synth.make_rays(scene, 0, 96, seed=9, margin=6).items()}

    def run(graphed, lconf, n=5):
        tr = Trainer(dev, rconf, color_loss_conf=lconf, seed=0, fused_adam=True)
        st = GraphedStep(tr, eager_steps=1) if graphed else tr.step
        torch.manual_seed(7)
        hist = []
        for i in range(n):
            loss, out = st(batch, cos_anneal_ratio=0.5 + 0.1 * i, flip_saturation=0.9, blend=src)
            hist.append((loss.clone(), out["color_pixel"].detach().clone()))
        torch.cuda.synchronize()
        return st, hist

    lconf = dict(color_pixel_weight=0.5, color_patch_weight=0.0)
    _, eager = run(False, lconf)
    gs, graph = run(True, lconf)
    assert gs.replays == 4
    for i, (a, b) in enumerate(zip(eager, graph)):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), i
    batch["gt_patch_colors"] = torch.rand(96, 49, 3, generator=torch.Generator().manual_seed(3)).to(dev)
    lconf = dict(color_pixel_weight=0.5, color_patch_weight=0.1)
    _, eager = run(False, lconf)
    gs2, graph = run(True, lconf)
    assert gs2.replays == 4
    for i, (a, b) in enumerate(zip(eager, graph)):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), i


def test_graph_replay_equals_eager_in_the_16_bit_mode():
    """the same bit-identity in BASELINE config 5's operand mode (16-bit MFMA operands, bf16 saved state): the mode that was
    launch-bound at the headline shape (3.66 ms eager, 2.67 ms replayed)."""
    from neuraludf_amd import mlp
    assert mlp.PRECISION == "fp32"
    try:
        mlp.set_precision("mixed16")
        n = 5
        _, _, eager = _run(RCONF, False, n)
        _, gs, graph = _run(RCONF, True, n)
    finally:
        mlp.set_precision("fp32")
    assert gs.replays == n - 2
    for i, (a, b) in enumerate(zip(eager, graph)):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), i
        for j, (p, q) in enumerate(zip(a[3], b[3])):
            assert torch.equal(p, q), (i, j)


def test_training_loop_iterations_with_graph_equal_eager():
    """`Trainer.iteration` -- schedules (warm-up + cosine learning rate, cos-anneal ramp), GPU-generated ray batches from a
    different image every iteration, render, loss, backward, fused Adam -- with `enable_graph()` against the plain loop:
    identical losses and parameters over 12 iterations (2 eager + capture + 9 replays)."""
    from neuraludf_amd.dataset import RayBatchSource
    from neuraludf_amd.schedules import Schedules
    from neuraludf_amd.train import Trainer
    dev = torch.device("cuda:0")
    scene = synth.make_scene("tiny")
    n_views = 5
    g = torch.Generator().manual_seed(0)
    imgs = torch.rand(n_views, scene.H, scene.W, 3, generator=g)
    rconf = dict(n_samples=32, n_importance=16, n_outside=0, up_sample_steps=2, perturb=1.0)

    def run(graph):
        src = RayBatchSource(imgs.clone(), torch.ones_like(imgs), scene.intrinsics[:n_views], scene.c2w[:n_views])
        tr = Trainer(dev, rconf, seed=0, fused_adam=True)
        if graph:
            tr.enable_graph(eager_steps=2)
        sched = Schedules(end_iter=200, learning_rate=1e-3, learning_rate_geo=2e-4, learning_rate_alpha=0.05, warm_up_end=5.0,
                          anneal_end=50.0, fix_geo_end=0, color_base_weight=0.01, color_weight=1.0)
        torch.manual_seed(99)
        losses = []
        for it in range(12):
            loss, _, _ = tr.iteration(src, it, sched, batch_size=128)
            losses.append(loss.clone())
        torch.cuda.synchronize()
        return tr, losses

    a, la = run(False)
    b, lb = run(True)
    assert b.graphed.replays == 10
    for i, (x, y) in enumerate(zip(la, lb)):
        assert torch.equal(x, y), (i, float(x), float(y))
    for (n, p), (_, q) in zip(a.udf.named_parameters(), b.udf.named_parameters()):
        assert torch.equal(p, q), n
