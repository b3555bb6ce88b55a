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
