"""Drop-in for the head + torso renderer ``NeRFs/TorsoNeRF/train_torso.py::Network``
(:198-271; BASELINE config 5): two pairs of FaceNeRFs rendered on their own ray sets and
composited as ``rgb_com = rgb_head * last_weight_torso + rgb_fg_torso`` (:269-270).

The head pair is conditioned on (audio, expression, latent); the torso pair on
``[aud[:dim_aud_body] | PE_3(euler) | PE_3(translation)]`` of the head pose (:238-240) with
no expression / latent code.  Both pairs run the same HIP kernels: the conditioning
widths only enter the folded biases.
"""
import torch
import torch.nn as nn

from .audio_exp_nerf import Network as HeadNetwork
from .helper import RenderConfig, get_embedder
from .models.audio_net import AudioAttNet, AudioNet
from .models.face_nerf import FaceNeRF


def config_parser():
    """NeRFs/TorsoNeRF/run_nerf_helpers.py:231-365 (`from run_nerf_helpers import *` puts it into train_torso's namespace)."""
    from .config import ConfigParser
    return ConfigParser("torso")


def pose_to_euler_trans(poses):
    """[b, >=3, 4] -> [b, 6] (euler angles, translation)  (run_nerf_helpers.py:26-47)."""
    R = poses[:, :3, :3]
    e = torch.stack([torch.atan2(R[:, 2, 2], R[:, 1, 2]), torch.asin(-R[:, 0, 2]),
                     torch.atan2(R[:, 0, 0], -R[:, 0, 1])], dim=1)
    return torch.cat((e, poses[:, :3, 3]), dim=1)


class Network(HeadNetwork):
    def __init__(self, H, W, focal, near, far, chunk, N_samlpes, N_importance, args: RenderConfig = None,
                 dim_aud_body=None, dim_expr_head=79):
        """The reference's positional arguments (train_torso.py:185: no `intrinsic`, unlike the head-only Network).
        Without `args=` the flags are the process's TorsoNeRF flags (run_nerf_helpers.py:231-365: `dim_aud`,
        `dim_aud_body`, ...), as upstream's class reads them from its `args` global (train_torso.py:200-221)."""
        nn.Module.__init__(self)
        from . import config
        if args is None:
            args = config.default_render_config("torso")
            args.dim_expr = dim_expr_head
            ns = config.current_args("torso")
            if dim_aud_body is None and ns is not None:
                dim_aud_body = ns.dim_aud_body
        else:
            config.check_against_current(args, "torso")
        dim_aud_body = 64 if dim_aud_body is None else dim_aud_body
        self.args = args
        self.H, self.W, self.focal, self.near, self.far = H, W, focal, near, far
        self.chunk, self.intrinsic = chunk, None
        self.N_samples, self.N_importance = N_samlpes, N_importance
        self.output_ch, self.skips = 4, [4]
        self.dim_aud_body = dim_aud_body
        self.embed_torso_aud_fn, ch = get_embedder(3, 0)  # 21 per 3-vector (:38)
        head = lambda: FaceNeRF(D=args.netdepth, W=args.netwidth, input_ch=63, dim_aud=args.dim_aud, skips=self.skips,
                                dim_latent=32, dim_expr=dim_expr_head, input_ch_views=27)
        torso = lambda: FaceNeRF(D=args.netdepth, W=args.netwidth, input_ch=63, dim_aud=dim_aud_body + 2 * ch,
                                 skips=self.skips, input_ch_views=27)
        self.face_nerf_coarse, self.face_nerf_fine = head(), head()
        self.aud_net = AudioNet(args.dim_aud, args.win_size)
        self.aud_att_net = AudioAttNet()
        self.torso_coarse_nerf, self.torso_fine_nerf = torso(), torso()

    def torso_signal(self, aud_feature, pose):
        et = pose_to_euler_trans(pose.unsqueeze(0))
        emb = torch.cat((self.embed_torso_aud_fn(et[:, :3]), self.embed_torso_aud_fn(et[:, 3:])), dim=1)
        return torch.cat((aud_feature[..., :self.dim_aud_body], torch.squeeze(emb)), dim=-1)

    def forward(self, inputs):
        """-> (rgb_com, rgb_com0)   (train_torso.py:223-271)."""
        x, global_step, dataset_size = inputs
        batch_rays, batch_rays_torso, target_s, bg_img, auds, raw_img, pose, expr, latent_code, index = x
        dev = self.face_nerf_coarse.alpha_linear.weight.device
        sq = lambda t: torch.squeeze(t).to(device=dev, dtype=torch.float32)
        batch_rays, batch_rays_torso, bg_img, auds, pose, expr = (sq(batch_rays), sq(batch_rays_torso), sq(bg_img),
                                                                    sq(auds), sq(pose), sq(expr))
        latent_code = torch.squeeze(latent_code).to(dev)
        aud_window = auds[int(index)]
        aud_feature = self.aud_net(aud_window.unsqueeze(0) if aud_window.dim() == 2 else aud_window)
        aud_torso = self.torso_signal(aud_feature, pose)
        render_poses = None if self.training is True else pose[:3, :4]
        kw = dict(H=self.H, W=self.W, focal=self.focal, render_poses=render_poses, chunk=self.args.chunk,
                  near=self.near, far=self.far, bc_rgb=bg_img)
        rgb, _, _, last_w, rgb_fg, extras = self.render_pair(
            expr=expr, latent_code=latent_code, rays=batch_rays, aud_para=aud_feature,
            network_nerf={'coarse': self.face_nerf_coarse, 'fine': self.face_nerf_fine}, **kw)
        _, _, _, last_w_t, rgb_fg_t, extras_t = self.render_pair(
            expr=None, latent_code=None, rays=batch_rays_torso, aud_para=aud_torso,
            network_nerf={'coarse': self.torso_coarse_nerf, 'fine': self.torso_fine_nerf}, **kw)
        rgb_com = rgb * last_w_t[..., None] + rgb_fg_t
        rgb_com0 = extras['rgb0'] * extras_t['last_weight0'][..., None] + extras_t['rgb_map_fg0']
        return rgb_com, rgb_com0

    def render_pair(self, H, W, focal, expr, latent_code, render_poses=None, chunk=1024 * 32, near=0., far=1.,
                    rays=None, bc_rgb=None, aud_para=None, network_nerf=None, rows=None):
        """train_torso.py::render_dynamic_face (:381-426): [rgb_map, disp_map, acc_map, last_weight,
        rgb_map_fg, {rest}] for one coarse/fine pair."""
        from . import ops
        dev = self.face_nerf_coarse.alpha_linear.weight.device
        coarse, fine = network_nerf['coarse'], network_nerf['fine']
        frame = None
        if render_poses is not None:   # full frame: the rays are derived on the device inside the render call
            row0, nrows = (0, H) if rows is None else (rows[0], rows[1] - rows[0])
            frame, rec = ops.make_frame(render_poses.detach().cpu(), H, W, focal, near, far, row0, nrows), None
            bc = bc_rgb[row0:row0 + nrows].reshape(-1, 3)
            sh = (nrows, W, 3)
        else:
            rays_o, rays_d = rays
            sh = rays_d.shape
            rays_o, rays_d = rays_o.reshape(-1, 3).float(), rays_d.reshape(-1, 3).float()
            viewdirs = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)
            rec = torch.cat([rays_o, rays_d, near * torch.ones_like(rays_d[..., :1]),
                             far * torch.ones_like(rays_d[..., :1]), viewdirs], -1)
            bc = bc_rgb
        all_ret = self._batchify(rec, bc, aud_para, latent_code, expr, coarse, fine, True, chunk, frame=frame)
        for k in all_ret:
            all_ret[k] = torch.reshape(all_ret[k], list(sh[:-1]) + list(all_ret[k].shape[1:]))
        k_extract = ['rgb_map', 'disp_map', 'acc_map', 'last_weight', 'rgb_map_fg']
        return [all_ret[k] for k in k_extract] + [{k: v for k, v in all_ret.items() if k not in k_extract}]
