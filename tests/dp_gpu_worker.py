"""worker of tests/test_gpu_dist.py::test_two_rank_step_equals_full_batch: one rank of a 2-rank ray-sharded step on a
ONE-GPU box (gloo backend, both ranks on cuda:0).  Launched with torch.distributed.run; rank 0 saves the global loss
and the all-reduced gradients."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build_and_grads(dev, world, rank, data_parallel):
    from neuraludf_amd import dist as nd, synth
    from neuraludf_amd.train import Trainer
    rconf = dict(n_samples=16, n_importance=8, n_outside=4, up_sample_steps=2, perturb=0.0)
    tr = Trainer(dev, rconf, seed=0, data_parallel=data_parallel, fused_adam=True,
                 train_conf=dict(igr_weight=0.1, igr_ns_weight=0.05, sparse_weight=0.01, mask_weight=0.0))
    rays = synth.make_rays(synth.make_scene("tiny"), 0, 48, seed=7)
    batch = {k: nd.shard(v, rank, world).contiguous().to(dev) for k, v in rays.items()}
    nd.collective_counts(reset=True)
    loss, _ = tr.loss(batch, cos_anneal_ratio=0.7, flip_saturation=0.5, perturb_overwrite=0)
    loss.backward()
    if data_parallel:
        tr.bucket.all_reduce()
    params = [p for g in tr.param_groups for p in g]
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
    return float(loss.detach()), flat.cpu(), nd.collective_counts()


def main():
    out = sys.argv[1]
    backend = sys.argv[2] if len(sys.argv) > 2 else "gloo"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev_index = int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0       # nccl (= RCCL): one rank per GPU
    torch.cuda.set_device(dev_index)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        loss, flat, coll = build_and_grads(torch.device("cuda", dev_index), world, rank, True)
        if rank == 0:
            torch.save({"loss": loss, "grads": flat, "collectives": coll}, out)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
