"""Per-frame audio latent producers, state_dict compatible with the reference's
``models/audio_net.py`` (AudioNet :43-69, AudioAttNet :8-36, DeepSpeechAudNet :72-87).

They run once per frame on a [<=8, 16, 29] window and cost nothing next to the per-ray
path, so they stay plain PyTorch-ROCm modules: device plumbing around
the HIP kernels, not part of the measured hot loop.
"""
import torch
import torch.nn as nn


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
        y = self.attentionConvNet(y)
        y = self.attentionNet(y.view(1, self.seq_len)).view(self.seq_len, 1)
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

    def forward(self, x):  # x: [n, 16, 29]
        half_w = int(self.win_size / 2)
        x = x[:, 8 - half_w:8 + half_w, :].permute(0, 2, 1)
        x = self.encoder_conv(x).squeeze(-1)
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
    y = aud_att_net.attentionConvNet(y)                               # [F, 1, smo]
    y = aud_att_net.attentionNet(y.reshape(F, aud_att_net.seq_len))   # softmax over the window
    return torch.sum(y.unsqueeze(-1) * win, dim=1)
