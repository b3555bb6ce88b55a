"""Drop-in for the reference's ablation model ``models.face_nerf_agg.FaceNeRFAgg``
(models/face_nerf_agg.py:8-89): a two-layer linear fuses [aud | expr/3] into a 64-d feature
that conditions the trunk and the colour branch.

The per-point network is the FaceNeRF one, so it runs on the same fused HIP kernel: the
fused feature takes the place of ``aud`` in the folded biases of layers 0 and 5, and its
columns of ``views_linears.0`` (a per-frame constant) are folded into that layer's bias
on the host side.  Same constructor, ``forward`` and state_dict keys as the reference.
Inference only.
"""
import torch
import torch.nn as nn

from .. import ops
from .face_nerf import PRECISIONS, _default_precision


class FaceNeRFAgg(nn.Module):
    def __init__(self, D=8, W=256, input_ch=63, input_ch_views=27, dim_agg=64, dim_aud=64, dim_expr=0,
                 dim_latent=0, output_ch=4, skips=None, use_viewdirs=True):
        super().__init__()
        if skips is None:
            skips = [4]
        if (D, W, list(skips), input_ch, input_ch_views, bool(use_viewdirs)) != (8, 256, [4], 63, 27, True):
            raise NotImplementedError("libidealnerf is compiled for D=8, W=256, skips=[4], input_ch=63, "
                                      "input_ch_views=27, use_viewdirs=True")
        self.D, self.W, self.skips, self.use_viewdirs = D, W, skips, use_viewdirs
        self.input_xyz_ch, self.input_views_ch = input_ch, input_ch_views
        self.dim_latent, self.dim_agg, self.dim_aud, self.dim_expr = dim_latent, dim_agg, dim_aud, dim_expr
        self.agg_linears = nn.ModuleList([nn.Linear(dim_expr + dim_aud, dim_agg), nn.Linear(dim_agg, dim_agg)])
        c_all = input_ch + dim_agg + dim_latent
        self.pts_linears = nn.ModuleList(
            [nn.Linear(c_all, W)] + [nn.Linear(W, W) if i not in skips else nn.Linear(W + c_all, W) for i in range(D - 1)])
        self.views_linears = nn.ModuleList(
            [nn.Linear(input_ch_views + W + dim_agg, W // 2)] + [nn.Linear(W // 2, W // 2) for _ in range(D // 4)])
        self.feature_linear = nn.Linear(W, W)
        self.alpha_linear = nn.Linear(W, 1)
        self.rgb_linear = nn.Linear(W // 2, 3)
        self.precision = _default_precision[0]
        self._packed, self._packed_key, self._views0 = {}, {}, None

    def _param_key(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def _kernel_params(self, views0_bias=None):
        """idn_facenerf_params of the equivalent FaceNeRF(dim_aud=dim_agg, dim_expr=0, dim_latent)."""
        sd = {k: v for k, v in self.named_parameters() if not k.startswith("agg_linears")}
        n_h = self.W + self.input_views_ch
        if self._views0 is None or self._views0[0] != self._param_key():
            self._views0 = (self._param_key(), self.views_linears[0].weight.detach()[:, :n_h].contiguous())
        sd["views_linears.0.weight"] = self._views0[1]
        if views0_bias is not None:
            sd["views_linears.0.bias"] = views0_bias
        return ops.params_struct(sd, self.dim_agg, 0, self.dim_latent), sd

    def packed_weights(self, precision: str = None) -> torch.Tensor:
        precision = precision or self.precision
        key = self._param_key()
        if self._packed.get(precision) is None or self._packed_key.get(precision) != key:
            with torch.no_grad():
                ps, _keep = self._kernel_params()
                self._packed[precision] = ops.pack_weights(ps, self.alpha_linear.weight.device, PRECISIONS[precision])
            self._packed_key[precision] = key
        return self._packed[precision]

    @property
    def prec_code(self) -> int:
        return PRECISIONS[self.precision]

    def agg_feature(self, aud, expr=None) -> torch.Tensor:
        """face_nerf_agg.py:53-62 for the one row all points share."""
        if expr is not None:
            h = torch.cat([aud, expr * 1 / 3], dim=-1)
        elif self.dim_expr == 0:
            h = aud
        else:
            raise RuntimeError(f"FaceNeRFAgg: expr is None, the network was built for dim_expr={self.dim_expr}")
        for layer in self.agg_linears:
            h = layer(h)
        return h

    def folded_bias(self, aud, expr=None, latent_code=None) -> torch.Tensor:
        dev = self.alpha_linear.weight.device
        with torch.no_grad():
            agg = self.agg_feature(aud.to(dev, torch.float32), None if expr is None else expr.to(dev, torch.float32))
            n_h = self.W + self.input_views_ch
            v0 = self.views_linears[0]
            bias = (v0.bias + v0.weight[:, n_h:] @ agg).contiguous()
            ps, _keep = self._kernel_params(bias)
            lat = None if latent_code is None else latent_code.detach().to(dev, torch.float32).contiguous()
            return ops.fold_conditioning(ps, agg.contiguous(), None, lat, dev)

    def forward(self, x, aud, expr=None, latent_code=None):
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            raise NotImplementedError("FaceNeRFAgg runs on the HIP path for inference only; wrap the call in torch.no_grad()")
        with torch.no_grad():
            return ops.facenerf_fwd(self.packed_weights(), self.folded_bias(aud, expr, latent_code),
                                    x.detach().to(torch.float32).contiguous(), self.prec_code)
