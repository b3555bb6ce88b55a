"""Drop-in for the reference's renderer class
``NeRFs/HeadNeRF/train/audio_exp_nerf.py::Network`` (:198-439; the eval script carries an
identical copy, test/eval_aud_exp_nerf.py:191-432).

Same constructor (typo ``N_samlpes`` included), same submodule names (checkpoint prefix
contract), same method names and return structures.  The per-ray work of ``render_rays``
is one C call into libidealnerf.so (coarse depths -> fused PE+MLP -> compositing ->
inverse-CDF sampling -> merge -> fused PE+MLP -> compositing).
"""
import logging
import os

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .helper import RenderConfig, linspace01
from .models.audio_net import AudioAttNet, AudioNet, DeepSpeechAudNet
from .models.face_nerf import FaceNeRF

logger = logging.getLogger("adnerf")


def init_weights(m):
    """audio_exp_nerf.py:442-448."""
    if isinstance(m, (nn.Linear, nn.Conv1d)):
        torch.nn.init.xavier_uniform_(m.weight)
        m.bias.data.fill_(0.01)


class Network(nn.Module):
    # True: perturb > 0 renders without autograd (the reference's default flags, also in eval) draw their random offsets inside
    # the kernels instead of as torch.rand tensors (same distribution, this library's own numbers; see _render).  Opt-in:
    # IDN_IN_KERNEL_DRAWS=1 (read once) or `net.in_kernel_draws = True`.
    in_kernel_draws = os.environ.get("IDN_IN_KERNEL_DRAWS", "0") == "1"

    def __init__(self, H, W, focal, near, far, chunk, intrinsic, N_samlpes, N_importance, args: RenderConfig = None):
        super().__init__()
        # flags: an explicit RenderConfig, or -- the reference's unchanged constructor call -- the process's flags, which
        # upstream's class reads from its import-time `args` global (audio_exp_nerf.py:25-26,213-226)
        from . import config
        if args is None:
            args = config.default_render_config("head")
        else:
            config.check_against_current(args, "head")
        self.args = args
        self.H, self.W, self.focal, self.near, self.far = H, W, focal, near, far
        self.chunk, self.intrinsic = chunk, intrinsic
        self.N_samples, self.N_importance = N_samlpes, N_importance
        self.output_ch, self.skips = 4, [4]
        mk = lambda: FaceNeRF(D=args.netdepth, W=args.netwidth, input_ch=63, dim_aud=args.dim_aud, output_ch=4,
                              skips=self.skips, dim_latent=args.dim_latent, dim_expr=args.dim_expr,
                              input_ch_views=27, use_viewdirs=args.use_viewdirs)
        self.face_nerf_coarse = mk()
        self.face_nerf_fine = mk()
        self.aud_net = AudioNet(args.dim_aud, args.win_size)
        self.aud_att_net = AudioAttNet()
        self.ds_aud_net = DeepSpeechAudNet()

    # ---- forward: audio_exp_nerf.py:221-272 ----------------------------------------
    def forward(self, inputs):
        args = self.args
        x, global_step, dataset_size = inputs
        batch_rays, target_s, bg_img, auds, raw_img, pose, expr, latent_code, index = x
        dev = self.face_nerf_coarse.alpha_linear.weight.device
        sq = lambda t: torch.squeeze(t).to(device=dev, dtype=torch.float32)
        # Full-frame rendering reads the camera matrix on the host (it becomes kernel arguments): keep the loader's CPU copy
        # rather than bouncing it through the device.  Training never needs it -- and a pose that already lives on the device
        # must not be fetched there: the device-to-host copy would block the host until the GPU has drained its queue, once
        # per step, and the step's ~100 small launches would then run with the host never ahead (measured: 1.2 ms of idle GPU
        # per 16 ms step, tools/train_timeline.py / train_host_time.py).
        pose_host = None if self.training is True else torch.squeeze(pose).detach().to(device="cpu", dtype=torch.float32)
        batch_rays, bg_img, auds, pose, expr = sq(batch_rays), sq(bg_img), sq(auds), sq(pose), sq(expr)
        latent_code = torch.squeeze(latent_code).to(dev)
        index = int(index)

        aud_feature, expr_feature = auds[index], None
        if args.dim_aud > 29 and global_step >= args.nosmo_iters:
            half = int(args.smo_size / 2)
            left_i, right_i = index - half, index + half
            pad_left, pad_right = 0, 0
            if left_i < 0:
                pad_left, left_i = -left_i, 0
            if right_i > dataset_size:
                pad_right, right_i = right_i - dataset_size, dataset_size
            win = auds[left_i:right_i]
            if pad_left > 0:
                win = torch.cat((torch.zeros_like(win)[:pad_left], win), dim=0)
            if pad_right > 0:
                win = torch.cat((win, torch.zeros_like(win)[:pad_right]), dim=0)
            aud_feature = self.aud_att_net(self.aud_net(win))
        elif args.dim_aud > 29:
            aud_feature = self.aud_net(aud_feature.unsqueeze(0) if aud_feature.dim() == 2 else aud_feature)
        else:
            aud_feature = self.ds_aud_net(aud_feature.unsqueeze(0) if aud_feature.dim() == 2 else aud_feature)
        if args.dim_expr > 0:
            expr_feature = expr
        render_poses = None if pose_host is None else pose_host[:3, :4]
        return self.render_dynamic_face(H=raw_img.shape[1], W=raw_img.shape[1], focal=self.focal, expr=expr_feature,
                                        poses=pose, latent_code=latent_code, render_poses=render_poses,
                                        chunk=args.chunk, near=self.near, far=self.far, rays=batch_rays,
                                        bc_rgb=bg_img, aud_para=aud_feature, ndc=False)

    # ---- batchify_rays: audio_exp_nerf.py:274-288 -----------------------------------
    def batchify_rays(self, rays, bc_rgb, aud_para, poses, latent_code, expr, chunk=1024 * 32):
        """Same results as the reference's chunk loop.  The loop exists upstream to bound the activation
        memory of a chunk (25 KB per sample); here activations never leave registers and the C call walks
        the rays in its own 32 768-ray passes over a fixed workspace, so a deterministic render
        is ONE call whatever `chunk` is -- one conditioning fold and one set of output tensors per frame
        instead of one per chunk.  With perturb > 0 (the reference's DEFAULT, also in eval: helper.py:70) the
        stratified offsets and the importance draws of the whole frame are drawn at once, from the same generator and
        the same distribution as upstream's per-chunk `torch.rand` calls (which chunk a draw lands in is not a
        contract: upstream's CUDA generator gives other numbers than this device's anyway).  Only training keeps the
        chunk loop (an autograd graph per chunk, as upstream)."""
        return self._batchify(rays, bc_rgb, aud_para, latent_code, expr, self.face_nerf_coarse, self.face_nerf_fine,
                              False, chunk)

    def _batchify(self, rays, bc_rgb, aud_para, latent_code, expr, coarse, fine, with_fg, chunk, frame=None):
        training = torch.is_grad_enabled() and self.training
        if not training:
            # `frame`: a full frame's row band with the rays still to be derived -- on the device, inside the one C call
            return self._render(rays, bc_rgb, aud_para, latent_code, expr, coarse, fine, with_fg, frame=frame)
        if frame is not None:   # the chunk loop slices ray records: materialise them
            rays = ops.frame_rays(torch.tensor(list(frame.c2w)).reshape(3, 4), frame.H, frame.W, frame.focal, frame.near_, frame.far_,
                                  frame.row0, frame.nrows, None if frame.cx < 0 else frame.cx, None if frame.cy < 0 else frame.cy,
                                  device=bc_rgb.device)
        all_ret = {}
        for i in range(0, rays.shape[0], chunk):
            ret = self._render(rays[i:i + chunk], bc_rgb[i:i + chunk], aud_para, latent_code, expr, coarse, fine, with_fg)
            for k in ret:
                all_ret.setdefault(k, []).append(ret[k])
        return {k: (v[0] if len(v) == 1 else torch.cat(v, 0)) for k, v in all_ret.items()}

    # ---- render_rays: audio_exp_nerf.py:290-364 -------------------------------------
    def render_rays(self, rays, bc_rgb, aud_para, poses, latent_code, expr, retraw=False, lindisp=False,
                    perturb=None, white_bkgd=False, raw_noise_std=0., attention_embed_ln=0, pytest=False,
                    taps=False):
        return self._render(rays, bc_rgb, aud_para, latent_code, expr, self.face_nerf_coarse, self.face_nerf_fine,
                            False, retraw, lindisp, perturb, white_bkgd, raw_noise_std, pytest, taps)

    def _render(self, rays, bc_rgb, aud_para, latent_code, expr, coarse, fine, with_fg, retraw=False, lindisp=False,
                perturb=None, white_bkgd=False, raw_noise_std=0., pytest=False, taps=False, frame=None):
        """Shared body of render_rays for the head pair and the torso pair of networks
        (``with_fg`` adds the torso variant's rgb_map_fg / rgb_map_fg0 / last_weight0,
        NeRFs/TorsoNeRF/train_torso.py:326-345)."""
        args = self.args
        perturb = args.perturb if perturb is None else perturb
        if torch.is_grad_enabled() and self.training:
            if white_bkgd or raw_noise_std > 0.:
                raise NotImplementedError("white_bkgd / raw_noise_std are built for inference only: the reference's "
                                          "training loop never sets them (audio_exp_nerf.py:297-299,534)")
            from .autograd import render_rays_apply
            return render_rays_apply(self, coarse, fine, rays, bc_rgb, aud_para, latent_code, expr, perturb, pytest,
                                     with_fg, lindisp=lindisp)
        bc_rgb = bc_rgb.to(torch.float32).contiguous()
        if frame is None:
            rays = rays.to(torch.float32).contiguous()
            n, dev = rays.shape[0], rays.device
        else:
            n, dev = frame.nrows * frame.W, bc_rgb.device
        S, Ni = args.N_samples, args.N_importance
        draws = None
        if self.in_kernel_draws and perturb > 0. and not pytest:
            # the frame's stratified offsets and importance draws are made INSIDE the kernels (idn_render_args.rng_mode): one seed
            # per call from torch's CPU generator (so torch.manual_seed governs it, and equally seeded ranks rendering row bands of
            # one frame draw from one table: a ray's row in it is its pixel index); no [n, S] / [n, Ni] random tensors exist
            draws = (int(torch.randint(0, 2 ** 62, (1,)).item()), 0 if frame is None else frame.row0 * frame.W)
            t_rand, u = None, None
        else:
            t_rand, u = self.draw_randoms(n, S, Ni, perturb, pytest, dev)
        from .helper import draw_sigma_noise
        # raw2outputs draws its noise per call, the coarse one first (baseline.py:353-361; numpy re-seeded each time under pytest)
        noise_c = draw_sigma_noise((n, S), raw_noise_std, pytest, dev)
        noise_f = draw_sigma_noise((n, S + Ni), raw_noise_std, pytest, dev) if Ni > 0 else None
        with torch.no_grad():
            fc = coarse.folded_bias(aud_para, expr, latent_code)
            ff = fine.folded_bias(aud_para, expr, latent_code) if Ni > 0 else None
            # each network renders in its own arithmetic: coarse "f32" + fine "bf16x3" is the mixed mode (the
            # coarse output drives the importance sampling, which amplifies arithmetic noise; nothing is
            # sampled after the fine pass)
            out = ops.render_rays_fwd(rays, bc_rgb, coarse.packed_weights(), fc,
                                      fine.packed_weights() if Ni > 0 else None, ff,
                                      linspace01(S, dev), u, Ni, t_rand=t_rand, with_fg=with_fg, taps=taps or retraw,
                                      precision=coarse.prec_code, precision_fine=fine.prec_code if Ni > 0 else None,
                                      lindisp=lindisp, white_bkgd=white_bkgd, noise_coarse=noise_c, noise_fine=noise_f, frame=frame,
                                      draws=draws)
        ret = {'rgb_map': out['rgb_map'], 'disp_map': out['disp_map'], 'acc_map': out['acc_map']}
        if with_fg:
            ret['rgb_map_fg'] = out['rgb_fg']
        if retraw:
            ret['raw'] = out['tap_raw_fine'] if Ni > 0 else out['tap_raw_coarse']
        if Ni > 0:
            for k in ('rgb0', 'disp0', 'acc0', 'z_std', 'last_weight'):
                ret[k] = out[k]
            if with_fg:
                ret['last_weight0'] = out['last_weight0']
                ret['rgb_map_fg0'] = out['rgb_fg0']
        if taps:
            ret.update({k: v for k, v in out.items() if k.startswith('tap_')})
        return ret

    @staticmethod
    def draw_randoms(n, S, Ni, perturb, pytest, dev):
        """t_rand / u exactly as the reference draws them (audio_exp_nerf.py:314-326,
        helper.py:279-293): none and linspace when perturb == 0; numpy seed 0 under pytest."""
        t_rand = None
        if perturb > 0.:
            if pytest:
                np.random.seed(0)
                t_rand = torch.Tensor(np.random.rand(n, S)).to(dev)
            else:
                t_rand = torch.rand((n, S), device=dev)
        u = None
        if Ni > 0:
            if perturb == 0. and pytest:   # numpy's linspace under the test override (helper.py:286-290)
                u = torch.Tensor(np.linspace(0., 1., Ni)).to(dev)
            elif perturb == 0.:
                u = linspace01(Ni, dev)
            elif pytest:
                np.random.seed(0)
                u = torch.Tensor(np.random.rand(n, Ni)).to(dev)
            else:
                u = torch.rand((n, Ni), device=dev)
        return t_rand, u

    def raw2outputs(self, raw, z_vals, rays_d, bc_rgb, raw_noise_std=0., white_bkgd=False, pytest=False):
        from .helper import raw2outputs
        return raw2outputs(raw, z_vals, rays_d, bc_rgb, raw_noise_std, white_bkgd, pytest)

    # ---- run_network: audio_exp_nerf.py:369-387 -------------------------------------
    def run_network(self, inputs, expr, viewdirs, aud, nerf_model, latent_code, netchunk=1024 * 64):
        """pts [n, S, 3] + unit viewdirs [n, 3] -> raw [n, S, 4]; both encodings fused.  Inference only:
        with autograd on and anything upstream requiring grad this raises instead of silently returning a
        tensor without a graph (training goes through render_rays, whose backward is built)."""
        if torch.is_grad_enabled() and (inputs.requires_grad or any(p.requires_grad for p in nerf_model.parameters())
                                        or any(t is not None and t.requires_grad for t in (aud, latent_code))):
            raise NotImplementedError(
                "Network.run_network carries no gradient: train through Network.render_rays / forward "
                "(audio_exp_nerf.py:534), or call under torch.no_grad() for inference")
        with torch.no_grad():
            folded = nerf_model.folded_bias(aud, expr, latent_code)
            return ops.query_points_fwd(nerf_model.packed_weights(), folded, inputs.to(torch.float32).contiguous(),
                                        viewdirs.to(torch.float32).contiguous(), nerf_model.prec_code)

    # ---- render_dynamic_face: audio_exp_nerf.py:389-439 ------------------------------
    def render_dynamic_face(self, H, W, focal, expr, poses, latent_code, render_poses=None, chunk=1024 * 32,
                            near=0., far=1., rays=None, bc_rgb=None, aud_para=None, ndc=False, use_viewdirs=True,
                            rows=None):
        if ndc or not use_viewdirs:
            raise NotImplementedError("ndc=True / use_viewdirs=False are dead in the reference (SURVEY 8 a1)")
        dev = self.face_nerf_coarse.alpha_linear.weight.device
        if render_poses is not None:
            # get_rays + the [o, d, near, far, viewdir] records (:396-427) happen on the device inside the render call
            # (idealnerf_render_frame_fwd): the camera travels as 12 floats, no [H W, 11] tensor exists
            row0, nrows = (0, H) if rows is None else (rows[0], rows[1] - rows[0])
            frame = ops.make_frame(render_poses.detach().cpu(), H, W, focal, near, far, row0, nrows)
            bc = bc_rgb[row0:row0 + nrows].reshape(-1, 3)
            sh = (nrows, W, 3)
            all_ret = self._batchify(None, bc, aud_para, latent_code, expr, self.face_nerf_coarse, self.face_nerf_fine, False, chunk,
                                     frame=frame)
        else:
            rays_o, rays_d = rays
            sh = rays_d.shape
            rays_o = rays_o.reshape(-1, 3).float()
            rays_d = rays_d.reshape(-1, 3).float()
            viewdirs = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)
            rec = torch.cat([rays_o, rays_d, near * torch.ones_like(rays_d[..., :1]),
                             far * torch.ones_like(rays_d[..., :1]), viewdirs], -1)
            all_ret = self.batchify_rays(rec, bc_rgb, aud_para, poses=poses, latent_code=latent_code, expr=expr, chunk=chunk)
        for k in all_ret:
            all_ret[k] = torch.reshape(all_ret[k], list(sh[:-1]) + list(all_ret[k].shape[1:]))
        k_extract = ['rgb_map', 'disp_map', 'acc_map', 'last_weight']
        ret_list = [all_ret[k] for k in k_extract]  # KeyError('last_weight') when N_importance == 0, as upstream (:437)
        ret_dict = {k: all_ret[k] for k in all_ret if k not in k_extract}
        return ret_list + [ret_dict]
