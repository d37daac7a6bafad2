"""Linear rational spline transformer (Dolatabadi et al. 2020) -- ATen composite path.

Numerics follow the reference's ``spline/linear_rational.py`` (:9-182): per element ``4K``
parameters ``[u_x (K) | u_y (K) | u_lambda (K) | u_d (K-1) | u_w0]``; knots from a softmax with a
1e-2 floor per bin, output knots from ``u_x + u_y / 100``; interior derivatives
``softplus(c + u_d / 100) + 1e-5`` with ``c = log(exp(1 - 1e-5) - 1)``, boundary derivatives
exactly 1; weights ``w_j = softplus(u_w0) * sqrt(d_0 / d_j)``; every bin is split at
``lambda = sigmoid(u_lambda)`` into two linear rational pieces; ``eps = 5e-10`` inside the
log-det.  ``phi`` is NOT clipped (unlike the rational-quadratic spline).  The HIP kernel is
csrc/tfk_lrs.hip; this module runs under autograd, in fp64 and on host tensors.
"""
from __future__ import annotations

import math
from typing import Sequence

import torch
import torch.nn.functional as F

from torchflows_amd.bijections.finite.autoregressive.transformers.spline.base import MonotonicSpline


class LinearRational(MonotonicSpline):
    native_kind = "lrs"

    def __init__(self, event_shape: Sequence[int], boundary: float = 50.0, **kwargs):
        super().__init__(event_shape, min_input=-boundary, max_input=boundary,
                         min_output=-boundary, max_output=boundary, **kwargs)
        self.boundary = boundary
        self.min_bin_width = 1e-2
        self.min_bin_height = 1e-2
        self.min_d = 1e-5
        self.const = math.log(math.exp(1 - self.min_d) - 1)
        self.eps = 5e-10

    @property
    def parameter_shape_per_element(self) -> torch.Size:
        return torch.Size((4 * self.n_bins,))

    @property
    def default_parameters(self) -> torch.Tensor:
        return torch.zeros(self.parameter_shape)

    def compute_bins(self, u: torch.Tensor, minimum: float, maximum: float, min_size: float) -> torch.Tensor:
        share = min_size + (1 - min_size * self.n_bins) * torch.softmax(u, dim=-1)
        inner = (maximum - minimum) * torch.cumsum(share, dim=-1)[..., :-1] + minimum
        return torch.cat([torch.full_like(u[..., :1], minimum), inner, torch.full_like(u[..., :1], maximum)], dim=-1)

    def _pieces(self, v: torch.Tensor, h: torch.Tensor, search_outputs: bool):
        """Knots, weights and the split point of the bin each element falls into."""
        K = self.n_bins
        u_x, u_y, u_l = h[..., :K], h[..., K:2 * K], h[..., 2 * K:3 * K]
        u_d, u_w0 = h[..., 3 * K:4 * K - 1], h[..., 4 * K - 1]
        knots_x = self.compute_bins(u_x, self.min_input, self.max_input, self.min_bin_width)
        knots_y = self.compute_bins(u_x + u_y / 100, self.min_output, self.max_output, self.min_bin_height)
        knots_d = F.pad(F.softplus(self.const + u_d / 100) + self.min_d, pad=(1, 1), mode="constant", value=1.0)
        lam = torch.sigmoid(u_l)
        w = F.softplus(u_w0)[..., None] * torch.sqrt(knots_d[..., :1] / knots_d)
        knots = knots_y if search_outputs else knots_x
        k = ((knots < v[..., None]).sum(dim=-1, keepdim=True) - 1).clamp(0, K - 1)   # searchsorted left
        pick = lambda t, idx: torch.gather(t, -1, idx).squeeze(-1)
        lam_k = pick(lam, k)
        w_k, w_k1 = pick(w, k), pick(w, k + 1)
        x_k, x_k1 = pick(knots_x, k), pick(knots_x, k + 1)
        y_k, y_k1 = pick(knots_y, k), pick(knots_y, k + 1)
        d_k, d_k1 = pick(knots_d, k), pick(knots_d, k + 1)
        y_m = ((1 - lam_k) * w_k * y_k + lam_k * w_k1 * y_k1) / ((1 - lam_k) * w_k + lam_k * w_k1)
        w_m = (lam_k * w_k * d_k + (1 - lam_k) * w_k1 * d_k1) * ((x_k1 - x_k) / (y_k1 - y_k))
        return lam_k, w_k, w_m, w_k1, x_k, x_k1, y_k, y_m, y_k1

    def forward_1d(self, x: torch.Tensor, h: torch.Tensor):
        lam, w_k, w_m, w_k1, x_k, x_k1, y_k, y_m, y_k1 = self._pieces(x, h, search_outputs=False)
        phi = (x - x_k) / (x_k1 - x_k)
        den_lo = w_k * (lam - phi) + w_m * phi
        out_lo = (w_k * y_k * (lam - phi) + w_m * y_m * phi) / den_lo
        ld_lo = torch.log(lam * w_k * w_m * (y_m - y_k)) - torch.log(den_lo ** 2 + self.eps) - torch.log(x_k1 - x_k)
        den_hi = w_m * (1 - phi) + w_k1 * (phi - lam)
        out_hi = (w_m * y_m * (1 - phi) + w_k1 * y_k1 * (phi - lam)) / den_hi
        ld_hi = (torch.log((1 - lam) * w_m * w_k1 * (y_k1 - y_m)) - torch.log(den_hi ** 2 + self.eps)
                 - torch.log(x_k1 - x_k))
        upper = phi > lam
        return torch.where(upper, out_hi, out_lo), torch.where(upper, ld_hi, ld_lo)

    def inverse_1d(self, z: torch.Tensor, h: torch.Tensor):
        lam, w_k, w_m, w_k1, x_k, x_k1, y_k, y_m, y_k1 = self._pieces(z, h, search_outputs=True)
        den_lo = w_k * (y_k - z) + w_m * (z - y_m)
        out_lo = (lam * w_k * (y_k - z)) / den_lo * (x_k1 - x_k) + x_k
        ld_lo = torch.log(lam * w_k * w_m * (y_m - y_k)) - torch.log(den_lo ** 2 + self.eps) + torch.log(x_k1 - x_k)
        den_hi = w_k1 * (y_k1 - z) + w_m * (z - y_m)
        out_hi = (lam * w_k1 * (y_k1 - z) + w_m * (z - y_m)) / den_hi * (x_k1 - x_k) + x_k
        ld_hi = (torch.log((1 - lam) * w_m * w_k1 * (y_k1 - y_m)) - torch.log(den_hi ** 2 + self.eps)
                 + torch.log(x_k1 - x_k))
        upper = z > y_m
        return torch.where(upper, out_hi, out_lo), torch.where(upper, ld_hi, ld_lo)
