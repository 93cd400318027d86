"""Scalar / Fourier-feature encoders shared by the discriminator (patch-parameter conditioning) and the camera-conditioned mapping network.

Reference: `src/training/layers.py:251-358` (ScalarEncoder1d, FourierEncoder1d, construct_log_spaced_freqs).  Eager tensor ops, as there.
"""
import numpy as np
import torch


def construct_log_spaced_freqs(grid_res, skip_n_high_freqs=0, skip_n_low_freqs=0):
    """layers.py:346-358: the lowest frequency has the period of the grid resolution."""
    num_freqs = np.ceil(np.log2(grid_res)).astype(int)
    grid_res = 2 ** num_freqs
    coefs = torch.tensor([2.0]).repeat(num_freqs) ** torch.arange(num_freqs) / grid_res
    coefs = coefs.float() * np.pi
    return coefs[skip_n_low_freqs:len(coefs) - skip_n_high_freqs]


class FourierEncoder1d(torch.nn.Module):
    """layers.py:304-340 (log-spaced frequencies, sin and cos)."""

    def __init__(self, coord_dim, max_x_value=100.0, use_cos=True):
        super().__init__()
        self.coord_dim, self.use_cos = coord_dim, use_cos
        self.register_buffer('fourier_coefs', construct_log_spaced_freqs(max_x_value))
        self.fourier_dim = self.fourier_coefs.shape[0]

    def get_dim(self):
        return self.fourier_dim * (2 if self.use_cos else 1)

    def forward(self, x):
        raw = self.fourier_coefs.view(1, 1, self.fourier_dim) * x.float().unsqueeze(2)
        return torch.cat([raw.sin(), raw.cos()], dim=2) if self.use_cos else raw.sin()


class ScalarEncoder1d(torch.nn.Module):
    """layers.py:251-299: scalars in [0,1] -> Fourier features + a learned embedding of the rounded value."""

    def __init__(self, coord_dim, x_multiplier, const_emb_dim, use_raw=False):
        super().__init__()
        self.coord_dim, self.const_emb_dim, self.x_multiplier, self.use_raw = coord_dim, const_emb_dim, x_multiplier, use_raw
        self.const_embed = torch.nn.Embedding(int(np.ceil(x_multiplier)) + 1, const_emb_dim) if const_emb_dim > 0 and x_multiplier > 0 else None
        self.fourier_encoder = FourierEncoder1d(coord_dim, max_x_value=x_multiplier) if x_multiplier > 0 else None
        self.fourier_dim = self.fourier_encoder.get_dim() if self.fourier_encoder is not None else 0
        self.raw_dim = 1 if use_raw else 0

    def get_dim(self):
        return self.coord_dim * (self.const_emb_dim + self.fourier_dim + self.raw_dim)

    def forward(self, x):
        B = x.shape[0]
        out = torch.empty(B, self.coord_dim, 0, device=x.device, dtype=x.dtype)
        if self.use_raw:
            out = torch.cat([out, x.unsqueeze(2)], dim=2)
        if self.fourier_encoder is not None or self.const_embed is not None:
            x = x.float() * self.x_multiplier
        if self.fourier_encoder is not None:
            out = torch.cat([out, self.fourier_encoder(x)], dim=2)
        if self.const_embed is not None:
            out = torch.cat([out, self.const_embed(x.round().long())], dim=2)
        return out.view(B, self.coord_dim * (self.raw_dim + self.const_emb_dim + self.fourier_dim))
