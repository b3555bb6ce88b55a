"""The reference's training-loop body (NeRFs/HeadNeRF/train/audio_exp_nerf.py:529-558) as
a function: render the sampled rays with gradients, MSE(fine) + MSE(coarse) +
10 * lc_weight * ||latent||, Adam step, exponential learning-rate decay, PSNR.
"""
import torch

from .helper import img2mse, mse2psnr


def make_optimizer(network, latent_codes, lrate=8e-4):
    """Adam over the network and the per-frame latent codes (audio_exp_nerf.py:493).  torch's default (foreach) form, as
    upstream: `fused=True` was measured in round 4 -- the same 14.3 ms step (the update is 0.09 ms of it either way), and on
    this ROCm build its steps did not track the CPU oracle's Adam (test_train_loop_adam_steps_match_oracle: 2 % off after
    three steps; 300 steps diverged) -- so it is not used."""
    return torch.optim.Adam(list(network.parameters()) + [latent_codes], lr=lrate, betas=(0.9, 0.999))


def decayed_lr(lrate, lrate_decay, global_step, decay_rate=0.1):
    """new_lrate = lrate * 0.1 ** (global_step / (lrate_decay * 1500))  (audio_exp_nerf.py:554-556)."""
    return lrate * (decay_rate ** (global_step / (lrate_decay * 1500)))


def train_step(network, optimizer, data, latent_codes, global_step, dataset_size, lrate=8e-4, lrate_decay=500):
    """One iteration of the loop at audio_exp_nerf.py:530-558.  ``data`` is the reference's
    8-tuple (batch_rays, target_s, bg_img, auds, raw_img, pose, expr, index)."""
    batch_rays, target_s, bg_img, auds, raw_img, pose, expr, index = data
    latent_code = latent_codes[int(index)]
    rgb, _, _, _, extras = network([(batch_rays, target_s, bg_img, auds, raw_img, pose, expr, latent_code, index),
                                    global_step, dataset_size])
    target = target_s.reshape(-1, 3).to(rgb.device, torch.float32)
    optimizer.zero_grad()
    img_loss = img2mse(rgb, target)
    loss = img_loss
    psnr = mse2psnr(img_loss.detach())
    if 'rgb0' in extras:
        loss = loss + img2mse(extras['rgb0'], target)
    latent_code_loss = torch.norm(latent_code) * network.args.lc_weight
    loss = loss + latent_code_loss * 10
    loss.backward()
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        from .parallel import average_gradients   # one bucketed all-reduce; a no-op for a single rank
        average_gradients([p for g in optimizer.param_groups for p in g['params']])
    optimizer.step()
    new_lrate = decayed_lr(lrate, lrate_decay, global_step)
    for group in optimizer.param_groups:
        group['lr'] = new_lrate
    return dict(loss=loss.detach(), psnr=psnr, latent_code_loss=latent_code_loss.detach(), lr=new_lrate)
