"""RCCL smoke on one GPU: the process group the multi-GPU bench uses (backend "nccl" == RCCL) initialises in this
image and the two collectives of the data-parallel path (packed loss sums, flat gradient bucket) run through it.
With one rank the values are trivially unchanged; what is checked is that the RCCL code path itself works."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_rccl_collectives_single_rank():
    from neuraludf_amd import dist as nd
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        x = torch.arange(5, dtype=torch.float32, device=dev).requires_grad_(True)
        y = nd._AllReduceSum.apply(x * 2.0)            # the autograd rule of the packed loss sums
        y.sum().backward()
        assert torch.equal(y.detach(), x.detach() * 2.0)
        assert torch.equal(x.grad, torch.full_like(x, 2.0))
        ps = [torch.randn(7, 3, device=dev, requires_grad=True), torch.randn(11, device=dev, requires_grad=True),
              torch.randn(2, device=dev, requires_grad=True)]
        ps[0].grad, ps[1].grad = torch.ones_like(ps[0]), torch.full_like(ps[1], 3.0)      # ps[2].grad stays None
        b = nd.GradBucket(ps)
        assert b.flat.numel() == 21 + 11 + 2 and b.flat.is_cuda
        # the bucket's collective itself (world_size() == 1 short-circuits it in production; issue its body directly)
        torch._foreach_copy_(b.views[:2], [ps[0].grad, ps[1].grad])
        dist.all_reduce(b.flat[:32], op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
        assert float(b.flat[:32].sum()) == 21.0 + 33.0
        assert torch.equal(b.views[1], torch.full_like(ps[1], 3.0))
    finally:
        dist.destroy_process_group()


def test_ray_sharded_step_with_its_rccl_collectives_captures_and_replays():
    """the data-parallel step -- packed loss-sum all-reduce + gradient-bucket all-reduce, both RCCL --
    captured in a HIP graph (GraphedStep(capture_collectives=True)) and replayed.  One rank: dist.FORCE_COLLECTIVES issues
    the two collectives although they are the identity, so the captured launch sequence is the multi-GPU one; replays must
    equal the eager ray-sharded step bit for bit.  (Eager stays the multi-GPU default until a 2-GPU run exists.)"""
    from neuraludf_amd import dist as nd, synth
    from neuraludf_amd.train import Trainer, GraphedStep
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    nd.FORCE_COLLECTIVES = True
    try:
        rconf = dict(n_samples=32, n_importance=16, n_outside=0, up_sample_steps=2, perturb=1.0)
        batch = {k: v.to(dev) for k, v in synth.make_rays(synth.make_scene("tiny"), 0, 128, seed=5).items()}

        def run(graphed):
            tr = Trainer(dev, rconf, seed=0, data_parallel=True, fused_adam=True)
            st = GraphedStep(tr, eager_steps=2, capture_collectives=True) if graphed else tr.step
            torch.manual_seed(21)
            nd.collective_counts(reset=True)
            out = []
            for i in range(6):
                loss, _ = st(batch, cos_anneal_ratio=0.2 * i, flip_saturation=0.9)
                out.append(loss.clone())
            torch.cuda.synchronize()
            return tr, st, out, nd.collective_counts()

        a, _, eager, ce = run(False)
        b, gs, graph, cg = run(True)
        assert ce == {"all_reduce": 12, "all_gather": 0}, ce                # two per step, six steps
        assert gs.enabled and gs.captures == 1 and gs.replays == 4
        assert cg["all_reduce"] == 2 * 3, cg                                  # 2 eager steps + the capture issue them from Python
        for i, (x, y) in enumerate(zip(eager, graph)):
            assert torch.equal(x, y), (i, float(x), float(y))
        for (n, p), (_, q) in zip(a.udf.named_parameters(), b.udf.named_parameters()):
            assert torch.equal(p, q), n
    finally:
        nd.FORCE_COLLECTIVES = False
        dist.destroy_process_group()


def _run(cmd, env_extra, timeout=600):
    import subprocess
    import sys
    env = dict(os.environ)
    env.update(env_extra)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + cmd,
                          env=env, capture_output=True, text=True, timeout=timeout)


def test_two_rank_step_equals_full_batch(tmp_path):
    """2 ranks (gloo, both on this GPU) each render half of the rays with the HIP path, exchange the packed loss sums
    (renderer + fused ColorLoss) and the gradient bucket: global loss and gradients equal the single-process step on
    the full batch."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    from dp_gpu_worker import build_and_grads
    out = str(tmp_path / "dp.pt")
    r = _run([os.path.join(here, "dp_gpu_worker.py"), out], {})
    assert r.returncode == 0, r.stderr[-3000:]
    got = torch.load(out)
    loss, flat, _ = build_and_grads(torch.device("cuda:0"), 1, 0, False)
    assert abs(got["loss"] - loss) < 1e-5 * max(1.0, abs(loss)), (got["loss"], loss)
    den = float(flat.abs().max())
    assert float((got["grads"] - flat).abs().max()) < 2e-4 * den
    # exactly two collectives per ray-sharded step: the packed partial sums and the gradient bucket
    assert got["collectives"] == {"all_reduce": 2, "all_gather": 0}, got["collectives"]


def test_bench_two_rank_flow():
    """the driver's multi-GPU launch line for bench.py (torch.distributed.run, 2 ranks), on one GPU via the gloo test
    backend: the run completes (no rank-0-only collective), prints one JSON line, counts both ranks' rays."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = _run([os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--rays-per-gpu", "64"],
             {"NUDF_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_rays"] == 128 and d["value"] > 0 and "roofline" in d
    assert d["collectives_per_step"] == {"all_reduce": 2.0, "all_gather": 0.0} and d["gradient_message_floats"] > 600000
    assert "cpu_baseline" not in d


def test_bench_gpus_2_without_a_launcher_starts_two_ranks():
    """plain `python bench.py --gpus 2` (no torch.distributed.run in front, WORLD_SIZE unset): bench.py starts the two ranks
    itself and relays rank 0's line -- it must not silently run one rank and print n_gpus 1.  And under a launcher whose
    WORLD_SIZE disagrees with --gpus it exits non-zero with a JSON error."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"NUDF_DIST_BACKEND": "gloo", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--rays-per-gpu", "64", "--windows", "2", "--no-power"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_rays"] == 128 and d["value"] > 0
    assert d["collectives_per_step"] == {"all_reduce": 2.0, "all_gather": 0.0}
    # a launcher that started another number of ranks than --gpus asks for: an error, not a mislabelled line
    r = _run([os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"], {"NUDF_DIST_BACKEND": "gloo"})
    assert r.returncode != 0
    err = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(err) == 1 and "error" in json.loads(err[0]), r.stdout[-1000:]


def test_bench_two_rank_strong_scaling_flow():
    """`bench.py --scaling strong` ("global N fixed per config, rays sharded"; BASELINE configs[3] is 4096 global
    rays): 2 ranks on this GPU over the gloo test backend share a FIXED global batch -- the line says so, counts the global
    rays once, and carries the windowed timing."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = _run([os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--windows", "3", "--scaling",
              "strong", "--global-rays", "128"], {"NUDF_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["scaling"] == "strong" and d["n_gpus"] == 2
    assert d["config"]["global_rays"] == 128 and d["config"]["rays_per_gpu"] == 64
    assert len(d["window_ms"]) == 3 and sorted(d["window_ms"])[1] == pytest.approx(d["ms_per_step"])
    assert d["value"] == pytest.approx(128 * d["config"]["samples_per_ray"] / (d["ms_per_step"] * 1e-3), rel=1e-6)
    assert d["collectives_per_step"] == {"all_reduce": 2.0, "all_gather": 0.0}
    assert "eager" in d["config"]["launch"]          # gloo stages through the host: the ray-sharded step is not captured


def test_two_rank_step_rccl(tmp_path):
    """the same 2-rank ray-sharded step over RCCL (backend "nccl"), one rank per GPU: needs two GPUs -- skipped on the
    one-GPU boxes, runs wherever the driver gives the tests a multi-GPU node.  Checks loss / gradients against the
    single-process full batch and that the step issued exactly two collectives (packed sums + gradient bucket)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    from dp_gpu_worker import build_and_grads
    out = str(tmp_path / "dp_rccl.pt")
    r = _run([os.path.join(here, "dp_gpu_worker.py"), out, "nccl"], {})
    assert r.returncode == 0, r.stderr[-3000:]
    got = torch.load(out)
    loss, flat = build_and_grads(torch.device("cuda:0"), 1, 0, False)[:2]
    assert abs(got["loss"] - loss) < 1e-5 * max(1.0, abs(loss)), (got["loss"], loss)
    assert float((got["grads"] - flat).abs().max()) < 2e-4 * float(flat.abs().max())
    assert got["collectives"] == {"all_reduce": 2, "all_gather": 0}, got["collectives"]


def test_bench_two_rank_flow_rccl():
    """the driver's 2-GPU launch line of bench.py over RCCL itself (backend "nccl"): needs two GPUs (skipped on one-GPU
    boxes).  The line must say that both ranks ran over RCCL -- `rccl_ranks == n_gpus` -- so that the first SCALE run
    checks itself, and hold exactly two collectives per step."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = _run([os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--no-cpu-baseline"], {})
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["dist_backend"] == "nccl", d
    assert d["config"]["global_rays"] == 1024 and d["scaling"] == "weak"
    assert d["collectives_per_step"] == {"all_reduce": 2.0, "all_gather": 0.0}
