"""Multi-GPU tiling of the per-ray path: one process per GPU, weights replicated, the
frame split into contiguous row bands, one all-gather of the rendered tiles per frame.

This replaces the reference's single-process ``nn.DataParallel`` ray scatter / output
gather (NeRFs/HeadNeRF/train/distribute_nerf.py:457-466, test/test_distribute_nerf.py:
378-387): rays are independent, so each rank derives its own rays from (row0, nrows, c2w)
-- no input scatter -- and only the outputs are exchanged (393 KB per rank for a 512^2
frame at 8 ranks; one RCCL all_gather over xGMI, latency-bound).
"""
from typing import List, Tuple

import torch
import torch.distributed as dist


def row_band(H: int, rank: int, world: int) -> Tuple[int, int]:
    """Rows [r0, r1) of an H-row frame owned by ``rank``: bands differ by at most one row
    and tile the frame exactly."""
    base, rem = divmod(H, world)
    r0 = rank * base + min(rank, rem)
    return r0, r0 + base + (1 if rank < rem else 0)


def all_bands(H: int, world: int) -> List[Tuple[int, int]]:
    return [row_band(H, r, world) for r in range(world)]


def gather_rows(tile: torch.Tensor, H: int, group=None, force: bool = False) -> torch.Tensor:
    """tile: this rank's [rows_r, W, C] band -> the full [H, W, C] frame on every rank.
    Bands may differ by one row, so tiles are padded to the widest band for the
    fixed-size all_gather and trimmed afterwards.  ``force``: issue the collective even on a
    one-rank communicator (how a one-GPU box exercises the RCCL call itself)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return tile
    world = dist.get_world_size(group)
    bands = all_bands(H, world)
    max_rows = max(b - a for a, b in bands)
    pad = tile
    if tile.shape[0] < max_rows:
        pad = torch.cat([tile, tile.new_zeros((max_rows - tile.shape[0],) + tuple(tile.shape[1:]))], 0)
    out = tile.new_empty((world * max_rows,) + tuple(tile.shape[1:]))
    dist.all_gather_into_tensor(out, pad.contiguous(), group=group)
    if world * max_rows == H:     # equal bands (512 rows over 8 ranks): the gathered buffer IS the frame, no copy
        return out
    parts = [out[r * max_rows: r * max_rows + (b - a)] for r, (a, b) in enumerate(bands)]
    return torch.cat(parts, 0)


def average_gradients(params, group=None) -> None:
    """Data-parallel training (beyond the reference's DataParallel): every rank renders its
    own ray batch and the gradients are averaged before the optimizer step.  Both FaceNeRFs, the audio
    nets and the latent codes are 5.8 MB of gradients: they are flattened into ONE bucket and reduced
    with a single all-reduce (xGMI rings are latency-bound at this size; per-tensor calls would pay
    that latency ~60 times), then scattered back in place.  Parameters without a gradient on this
    rank contribute zeros, so ranks may differ in which optional branches they touched."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    params = [p for p in params if p.requires_grad]
    if not params:
        return
    world = dist.get_world_size(group)
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).to(torch.float32) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.div_(world)
    off = 0
    for p in params:
        n = p.numel()
        g = flat[off:off + n].view_as(p).to(p.dtype)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += n
