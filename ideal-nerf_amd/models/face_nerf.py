"""Drop-in for the reference's ``models.face_nerf.FaceNeRF`` (models/face_nerf.py:8-80).

Same constructor, same ``forward(x, aud, expr=None, latent_code=None) -> [N, 4]``, same
``state_dict`` keys (including the never-applied ``feature_linear``), so checkpoints
load unchanged.  The arithmetic runs in libidealnerf.so: the parameters are re-laid into
the MFMA weight stream once per update, the per-frame conditioning vectors are folded
into biases once per call, and the per-point contraction is one fused HIP kernel.
"""
import os

import torch
import torch.nn as nn

from .. import ops
from .._lib import IDN_PREC_BF16, IDN_PREC_BF16X3, IDN_PREC_BF16X6, IDN_PREC_F32, IDN_PREC_FP16X3

PRECISIONS = {"f32": IDN_PREC_F32, "bf16x3": IDN_PREC_BF16X3, "bf16": IDN_PREC_BF16, "fp16x3": IDN_PREC_FP16X3,
              "bf16x6": IDN_PREC_BF16X6}
# The arithmetic of modules created without further ado (DESIGN.md section 9 states the criterion, tests/parity_proof.py:
# default_precision_criterion holds the shipped choice to it); IDN_DEFAULT_PRECISION / set_default_precision override it.
SHIPPED_DEFAULT_PRECISION = "bf16x6"
_default_precision = [os.environ.get("IDN_DEFAULT_PRECISION", SHIPPED_DEFAULT_PRECISION)]   # modules created from here on


def set_render_precision(network, mode: str):
    """Arithmetic of a Network's coarse / fine pair: "f32" (fp32 MFMA), "bf16x6" (six bf16 piece products per fp32
    product: the fp32 kernel's parity at 1.7x its speed), "fp16x3", "bf16x3", "bf16", or "mixed" = exact fp32 for
    the coarse network (its output drives the importance sampling) and bf16x3 for the fine one (3/4 of
    the samples): about twice the fp32 speed inside the 1e-4 RGB budget, also on sharp scenes; "mixed6" = the same
    split with the coarse network in "bf16x6" (fp32-grade on the bf16 pipe): 2.7x the fp32 speed, same budget."""
    pairs = [(getattr(network, c, None), getattr(network, f, None))
             for c, f in (("face_nerf_coarse", "face_nerf_fine"), ("torso_coarse_nerf", "torso_fine_nerf"))]
    for coarse, fine in pairs:
        if coarse is None:
            continue
        if mode == "mixed":
            coarse.precision, fine.precision = "f32", "bf16x3"
        elif mode == "mixed6":    # the same split with the coarse network on the bf16 pipe at fp32 grade
            coarse.precision, fine.precision = "bf16x6", "bf16x3"
        elif mode in PRECISIONS:
            coarse.precision = fine.precision = mode
        else:
            raise ValueError(f"precision must be one of {sorted(PRECISIONS) + ['mixed', 'mixed6']}")


def invalidate_packed(module):
    """Drop the cached weight streams of every FaceNeRF under `module` (needed only after writes through
    ``.data``, which PyTorch's version counters do not see)."""
    for m in module.modules():
        if isinstance(m, FaceNeRF):
            m.invalidate_packed()


def set_default_precision(name: str):
    """Arithmetic of the MLP contraction for modules created afterwards: "f32" (exact fp32
    MFMA), "bf16x6" (weights and activations as three bf16 pieces, six MFMAs per product: fp32-grade),
    "bf16x3" (three bf16 MFMAs per product, ~1.5e-5 relative on the output), "fp16x3" or "bf16"
    (plain bf16, ~1e-2: PSNR-judged rendering only, BASELINE config 5)."""
    if name not in PRECISIONS:
        raise ValueError(f"precision must be one of {sorted(PRECISIONS)}")
    _default_precision[0] = name


class FaceNeRF(nn.Module):
    def __init__(self, D=8, W=256, input_ch=63, input_ch_views=27, dim_aud=64, dim_latent=0, dim_expr=0,
                 output_ch=4, skips=None, use_viewdirs=True):
        super().__init__()
        if skips is None:
            skips = [4]
        self.D, self.W = D, W
        self.input_xyz_ch, self.input_views_ch = input_ch, input_ch_views
        self.dim_aud, self.dim_expr, self.dim_latent = dim_aud, dim_expr, dim_latent
        self.skips, self.use_viewdirs = skips, use_viewdirs
        if (D, W, list(skips), input_ch, input_ch_views, bool(use_viewdirs)) != (8, 256, [4], 63, 27, True):
            raise NotImplementedError(
                "libidealnerf is compiled for the reference's fixed architecture D=8, W=256, skips=[4], "
                "input_ch=63, input_ch_views=27, use_viewdirs=True (audio_exp_nerf.py:213-224)")
        c_all = input_ch + dim_aud + dim_expr + dim_latent
        self.pts_linears = nn.ModuleList(
            [nn.Linear(c_all, W)] + [nn.Linear(W, W) if i not in skips else nn.Linear(W + c_all, W) for i in range(D - 1)])
        self.views_linears = nn.ModuleList(
            [nn.Linear(input_ch_views + W + dim_expr, W // 2)] + [nn.Linear(W // 2, W // 2) for _ in range(D // 4)])
        self.feature_linear = nn.Linear(W, W)  # present for checkpoint compatibility; never applied upstream
        self.alpha_linear = nn.Linear(W, 1)
        self.rgb_linear = nn.Linear(W // 2, 3)
        self.precision = _default_precision[0]   # inference arithmetic; training always runs an fp32-grade one (autograd.py)
        self._packed = {}
        self._packed_key = {}

    # -- kernel-side views of the parameters ------------------------------------------
    def _param_key(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def invalidate_packed(self):
        """Drop the cached MFMA weight streams.  They are keyed on (storage, autograd version) of every
        parameter, which optimizers, ``copy_`` and ``load_state_dict`` bump -- but writes through ``.data``
        (``p.data.add_()``, EMA swaps, weight clipping, ``init_weights``' bias fill) do NOT.  After such a
        write call this (or ``idealnerf_amd.invalidate_packed(network)``); ``train()`` / ``eval()``,
        ``load_state_dict`` and ``.to()`` call it themselves.  The folded biases are recomputed per call
        and need nothing."""
        self._packed.clear()
        self._packed_key.clear()

    def train(self, mode: bool = True):
        self.invalidate_packed()
        return super().train(mode)

    def load_state_dict(self, *args, **kwargs):
        self.invalidate_packed()
        return super().load_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        self.invalidate_packed()
        return super()._apply(fn, *args, **kwargs)

    def kernel_params(self):
        sd = {k: v for k, v in self.named_parameters()}
        return ops.params_struct(sd, self.dim_aud, self.dim_expr, self.dim_latent)

    def packed_weights(self, precision: str = None) -> torch.Tensor:
        """MFMA-fragment weight stream for `precision`, rebuilt only when a parameter changed."""
        precision = precision or self.precision
        key = self._param_key()
        if self._packed.get(precision) is None or self._packed_key.get(precision) != key:
            dev = self.alpha_linear.weight.device
            with torch.no_grad():
                self._packed[precision] = ops.pack_weights(self.kernel_params(), dev, PRECISIONS[precision])
            self._packed_key[precision] = key
        return self._packed[precision]

    @property
    def prec_code(self) -> int:
        return PRECISIONS[self.precision]

    def folded_bias(self, aud, expr=None, latent_code=None) -> torch.Tensor:
        self._check_cond(aud, expr, latent_code)
        dev = self.alpha_linear.weight.device
        f = lambda t: None if t is None else t.detach().to(dtype=torch.float32).contiguous()
        return ops.fold_conditioning(self.kernel_params(), f(aud), f(expr), f(latent_code), dev)

    def _check_cond(self, aud, expr, latent_code):
        for name, t, d in (("aud", aud, self.dim_aud), ("expr", expr, self.dim_expr),
                           ("latent_code", latent_code, self.dim_latent)):
            have = 0 if t is None else t.numel()
            if have != d:
                # the reference fails inside addmm with a shape RuntimeError
                raise RuntimeError(f"FaceNeRF: {name} has {have} elements, the network was built for {d}")

    def forward(self, x, aud, expr=None, latent_code=None):
        self._check_cond(aud, expr, latent_code)
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            from ..autograd import facenerf_apply
            return facenerf_apply(self, x, aud, expr, latent_code)
        with torch.no_grad():
            folded = self.folded_bias(aud, expr, latent_code)
            return ops.facenerf_fwd(self.packed_weights(), folded, x.detach().to(torch.float32).contiguous(),
                                    self.prec_code)
