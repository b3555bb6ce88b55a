"""Per-frame audio latent producers, state_dict compatible with the reference's
``models/audio_net.py`` (AudioNet :43-69, AudioAttNet :8-36, DeepSpeechAudNet :72-87).

They run once per frame on a [<=8, 16, 29] window and cost nothing next to the per-ray path of a FRAME (SURVEY
section 2).  In a TRAINING STEP, though, AudioNet's ~50 eager launches (forward + backward) were a third of the step's
small launches, each a few microseconds behind a dispatch gap: on the GPU `AudioNet.forward` therefore runs as one HIP
kernel and its backward as one (csrc/audio.hip, `idealnerf_audio_net_fwd / _bwd`); on the CPU, for other window sizes and
for more than eight windows with gradients it is the plain PyTorch module below.  The attention smoother and the
DeepSpeech squeeze stay PyTorch-ROCm.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

FUSED_AUDIO_NET = os.environ.get("IDN_FUSED_AUDIO_NET", "1") != "0"   # read once: 0 keeps the eager module (the A/B arm)


class _AudioNetFn(torch.autograd.Function):
    """AudioNet on [n, 16, 29] windows: one kernel forward, one backward (parameter gradients; the windows are data)."""

    @staticmethod
    def forward(ctx, windows, dim_aud, *params):
        from .. import ops
        save = any(ctx.needs_input_grad[2:])     # (grad mode is off inside forward: ask autograd what it will want back)
        with torch.no_grad():
            out, saved = ops.audio_net_fwd([p.detach() for p in params], windows, dim_aud, save=save)
        ctx.dim_aud = dim_aud
        if save:
            ctx.save_for_backward(windows, saved, *params)
        return out

    @staticmethod
    def backward(ctx, g):
        from .. import ops
        windows, saved, *params = ctx.saved_tensors
        grads = ops.audio_net_bwd([p.detach() for p in params], windows, saved, g.contiguous().to(torch.float32), ctx.dim_aud)
        return (None, None, *grads)


def _conv1d_k3(x: torch.Tensor, conv: nn.Conv1d) -> torch.Tensor:
    """``conv(x)`` for the reference's kernel-3, padding-1 convolutions on [n, C_in, L <= 16] as a gather of
    the L_out windows and ONE matrix product with the [C_out, 3 C_in] weight.  Same parameters, same result to
    fp32 rounding -- but on ROCm ``nn.Conv1d`` at these sizes goes to MIOpen's ``naive_conv_*`` kernels and
    their layout transposes: ~50 launches and 0.6 ms per training step, against ~0.1 ms this way."""
    stride = conv.stride[0]
    cols = F.pad(x, (1, 1)).unfold(2, 3, stride)                # [n, C_in, L_out, 3] (a view)
    n, c_in, l_out, _ = cols.shape
    cols = cols.permute(0, 2, 1, 3).reshape(n * l_out, c_in * 3)
    y = F.linear(cols, conv.weight.reshape(conv.out_channels, c_in * 3), conv.bias)
    return y.reshape(n, l_out, conv.out_channels).permute(0, 2, 1)


def _run_convnet(seq: nn.Sequential, x: torch.Tensor) -> torch.Tensor:
    for layer in seq:
        if isinstance(layer, nn.Conv1d) and layer.kernel_size == (3,) and layer.padding == (1,) and x.is_cuda:
            x = _conv1d_k3(x, layer)
        elif isinstance(layer, nn.LeakyReLU):
            x = F.leaky_relu(x, layer.negative_slope)
        else:
            x = layer(x)
    return x


class AudioAttNet(nn.Module):
    def __init__(self, dim_aud=32, seq_len=8):
        super().__init__()
        self.seq_len, self.dim_aud = seq_len, dim_aud
        chans = [dim_aud, 16, 8, 4, 2, 1]
        layers = []
        for cin, cout in zip(chans[:-1], chans[1:]):
            layers += [nn.Conv1d(cin, cout, kernel_size=3, stride=1, padding=1, bias=True), nn.LeakyReLU(0.02, True)]
        self.attentionConvNet = nn.Sequential(*layers)
        self.attentionNet = nn.Sequential(nn.Linear(seq_len, seq_len, bias=True), nn.Softmax(dim=1))

    def forward(self, x):  # x: [seq_len, >=dim_aud]
        y = x[..., :self.dim_aud].permute(1, 0).unsqueeze(0)
        y = _run_convnet(self.attentionConvNet, y)
        y = self.attentionNet(y.reshape(1, self.seq_len)).view(self.seq_len, 1)
        return torch.sum(y * x, dim=0)


class AudioNet(nn.Module):
    def __init__(self, dim_aud=76, win_size=16):
        super().__init__()
        self.win_size, self.dim_aud = win_size, dim_aud
        chans = [29, 32, 32, 64, 64]
        layers = []
        for cin, cout in zip(chans[:-1], chans[1:]):
            layers += [nn.Conv1d(cin, cout, kernel_size=3, stride=2, padding=1, bias=True), nn.LeakyReLU(0.02, True)]
        self.encoder_conv = nn.Sequential(*layers)
        self.encoder_fc1 = nn.Sequential(nn.Linear(64, 64), nn.LeakyReLU(0.02, True), nn.Linear(64, dim_aud))

    def _fused_params(self):
        return [t for i in (0, 2, 4, 6) for t in (self.encoder_conv[i].weight, self.encoder_conv[i].bias)] + \
               [t for i in (0, 2) for t in (self.encoder_fc1[i].weight, self.encoder_fc1[i].bias)]

    def forward(self, x):  # x: [n, 16, 29]
        if (FUSED_AUDIO_NET and x.is_cuda and self.win_size == 16 and x.dim() == 3 and tuple(x.shape[1:]) == (16, 29)
                and x.dtype == torch.float32 and self.dim_aud <= 128 and not x.requires_grad):
            params = self._fused_params()
            needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
            from ..ops import AUDIO_NET_MAX_BWD_WINDOWS
            if not needs_grad or x.shape[0] <= AUDIO_NET_MAX_BWD_WINDOWS:
                return _AudioNetFn.apply(x.contiguous(), self.dim_aud, *params).squeeze()
        half_w = int(self.win_size / 2)
        x = x[:, 8 - half_w:8 + half_w, :].permute(0, 2, 1)
        x = _run_convnet(self.encoder_conv, x).squeeze(-1)
        return self.encoder_fc1(x).squeeze()


class DeepSpeechAudNet(nn.Module):
    def __init__(self, dim_aud=29, win_size=16):
        super().__init__()
        self.win_size, self.dim_aud = win_size, dim_aud
        self.encoder_fc = nn.Sequential(nn.Linear(16, 1), nn.LeakyReLU(0.02, True))

    def forward(self, x):  # x: [n, 16, 29]
        x = self.encoder_fc(x.permute(0, 2, 1)).squeeze(-1)
        return x.squeeze()


def clip_audio_features(aud_net: AudioNet, aud_att_net: AudioAttNet, auds: torch.Tensor, smo_size: int = 8) -> torch.Tensor:
    """Smoothed audio feature of EVERY frame of a clip in one batched pass: [F, 16, 29] -> [F, dim_aud].

    The reference walks the clip frame by frame (NeRFs/TorsoNeRF/test_torso.py:478-498; the same
    window logic per call in audio_exp_nerf.py:246-262): slice a window of `smo_size` frames,
    zero-pad the RAW DeepSpeech window at the clip ends, run AudioNet on the window and AudioAttNet
    on its output.  AudioNet treats frames independently, so it is run once over the clip (plus
    once on an all-zero frame, whose feature is what a padded slot becomes -- not zero: the
    convolutions have biases), windows are gathered from that table, and AudioAttNet's
    convolutions/linear run batched over the F windows.
    """
    F = auds.shape[0]
    half = int(smo_size / 2)
    if F < smo_size:  # the reference's zeros_like(win)[:pad] clips the padding to the window length
        raise ValueError(f"clip of {F} frames is shorter than the smoothing window ({smo_size})")
    feats = aud_net(auds).reshape(F, -1)
    pad = aud_net(torch.zeros_like(auds[:1])).reshape(1, -1)
    table = torch.cat([feats, pad], dim=0)
    idx = torch.arange(F, device=auds.device)[:, None] + torch.arange(-half, half, device=auds.device)[None, :]
    idx = torch.where((idx >= 0) & (idx < F), idx, torch.full_like(idx, F))
    win = table[idx]                                                  # [F, smo, dim_aud]
    y = win[..., :aud_att_net.dim_aud].permute(0, 2, 1)               # [F, dim_att, smo]
    y = _run_convnet(aud_att_net.attentionConvNet, y)                 # [F, 1, smo]
    y = aud_att_net.attentionNet(y.reshape(F, aud_att_net.seq_len))   # softmax over the window
    return torch.sum(y.unsqueeze(-1) * win, dim=1)
