"""Worker of tests/test_helper_surface_gpu.py::test_data_parallel_train_step_two_ranks (not collected by pytest).

Started as `python -m torch.distributed.run --nproc-per-node 2 tests/dp_train_worker.py` with the two ranks sharing GPU 0 over
gloo (the one-GPU rehearsal of the RCCL path).  Each rank (a) takes ONE training step alone on the whole 512-ray batch --
before any process group exists --, then (b) joins the group, rebuilds the same network and takes one data-parallel step
on ITS half of the batch: `train.train_step` -> `parallel.average_gradients` (one bucketed all-reduce) -> Adam.  Rank 0
prints one JSON line: whether the replicas ended bit-identical, and how far the data-parallel gradients / parameters are
from the single-rank step on the concatenated batch (the mean of two half-batch MSE gradients IS the full-batch gradient).
Reference counterpart: nn.DataParallel's scatter / replicate / gather around Network.forward
(NeRFs/HeadNeRF/train/distribute_nerf.py:423,457-466).
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

N_RAYS = 512


def build(dev):
    from idealnerf_amd import synthetic, train as T_
    from idealnerf_amd.audio_exp_nerf import Network
    from idealnerf_amd.helper import RenderConfig
    torch.manual_seed(0)          # audio nets
    H = W = 64
    syn = synthetic.frame(H, W, seed=0)
    cfg = RenderConfig(perturb=0.0, chunk=8192, near=syn["near"], far=syn["far"])     # nothing random in a step
    net = Network(H, W, syn["focal"], syn["near"], syn["far"], 8192, None, 64, 128, args=cfg).to(dev).train()
    synthetic.xavier_state_dict(net.face_nerf_coarse, 2, 300.0, 0.3)
    synthetic.xavier_state_dict(net.face_nerf_fine, 3, 300.0, 0.3)
    latent_codes = torch.ones(8, 32, device=dev, requires_grad=True)
    return net, T_.make_optimizer(net, latent_codes), latent_codes, syn


def batch(syn, dev, lo, hi):
    from idealnerf_amd import ops
    H = W = 64
    rs = np.random.RandomState(7)
    sel = torch.from_numpy(rs.choice(H * W, N_RAYS, replace=False))[lo:hi]
    rec = ops.frame_rays(syn["c2w"], H, W, syn["focal"], syn["near"], syn["far"], device=dev)
    rays = torch.stack([rec[sel.to(dev), 0:3], rec[sel.to(dev), 3:6]], 0).contiguous()
    bg = syn["bc"].reshape(-1, 3)[sel].contiguous().to(dev)
    tgt = torch.from_numpy(rs.uniform(0, 1, size=(N_RAYS, 3)).astype(np.float32))[lo:hi].to(dev)
    auds = torch.from_numpy(rs.standard_normal((8, 16, 29)).astype(np.float32)).to(dev)
    pose = torch.cat([syn["c2w"], torch.tensor([[0.0, 0.0, 0.0, 1.0]])], 0).to(dev)
    return (rays[None], tgt, bg, auds[None], torch.zeros(1, H, W, 3), pose, syn["expr"][None].to(dev), torch.tensor([3]))


def one_step(dev, lo, hi):
    from idealnerf_amd import train as T_
    net, opt, lat, syn = build(dev)
    info = T_.train_step(net, opt, batch(syn, dev, lo, hi), lat, 0, 8)
    torch.cuda.synchronize()
    named = list(net.named_parameters()) + [("latent_codes", lat)]
    grads = {k: (torch.zeros_like(p) if p.grad is None else p.grad.detach().clone()) for k, p in named}
    params = {k: p.detach().clone() for k, p in named}
    return grads, params, float(info["loss"])


def main():
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    g1, p1, loss1 = one_step(dev, 0, N_RAYS)                      # (a) alone, whole batch: no process group yet
    assert not dist.is_initialized()
    dist.init_process_group("gloo")
    half = N_RAYS // world
    g2, p2, loss2 = one_step(dev, rank * half, (rank + 1) * half)  # (b) data-parallel, this rank's share
    flat = torch.cat([p2[k].reshape(-1) for k in p2])
    lo, hi = flat.clone(), flat.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    losses = torch.tensor([loss2], dtype=torch.float64, device=dev)
    dist.all_reduce(losses)
    if rank == 0:
        rel = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))
        worst_g = max(((rel(g2[k], g1[k]), k) for k in g1 if float(g1[k].abs().max()) > 0), default=(0.0, ""))
        worst_p = max(((float((p2[k].double() - p1[k].double()).abs().max()), k) for k in p1))
        moved = max(float((p1[k] - q).abs().max()) for (k, q) in build(dev)[0].named_parameters() if k in p1)
        print(json.dumps({"metric": "dp_train_step", "ranks": world, "replicas_bit_identical": bool(torch.equal(lo, hi)),
                          "grad_rel_err_max": worst_g[0], "grad_rel_err_where": worst_g[1], "param_abs_err_max": worst_p[0],
                          "param_abs_err_where": worst_p[1], "params_moved_by": moved, "n_tensors": len(g1),
                          "loss_single": loss1, "loss_mean_of_ranks": float(losses.item()) / world}), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
