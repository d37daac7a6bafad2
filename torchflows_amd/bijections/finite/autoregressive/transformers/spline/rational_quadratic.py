"""Rational-quadratic spline transformer (Durkan et al. 2019) -- ATen composite path.

Numerics follow the reference's ``spline/rational_quadratic.py`` (:10-200) including its
quirks: bin heights are built from ``u_x + u_y / 1000``, derivative
logits are divided by 1000 and padded with ``c = log(expm1(1 - 1e-5))`` so the boundary
derivative is 1e-5 + softplus(1.001 c), ``min_bin_size = 1e-3``, ``min_delta = 1e-5``,
``boundary = 50`` and ``n_bins = 8`` by default, and an input equal to an interior knot
belongs to the bin on its left.  The HIP kernel is csrc/tfk_rqs.hip; this module runs
under autograd, in fp64 and on host tensors, without the reference's asserts (host syncs).
"""
from __future__ import annotations

import math
from typing import Sequence

import torch
import torch.nn.functional as F

from torchflows_amd.bijections.finite.autoregressive.transformers.spline.base import MonotonicSpline


class RationalQuadratic(MonotonicSpline):
    native_kind = "rqs"

    def __init__(self, event_shape: Sequence[int], boundary: float = 50.0, **kwargs):
        super().__init__(event_shape, min_input=-boundary, max_input=boundary,
                         min_output=-boundary, max_output=boundary, **kwargs)
        self.boundary = boundary
        self.min_bin_size = 1e-3
        self.min_delta = 1e-5
        self.boundary_u_delta = math.log(math.expm1(1 - self.min_delta))

    @property
    def parameter_shape_per_element(self) -> torch.Size:
        return torch.Size((3 * self.n_bins - 1,))

    @property
    def default_parameters(self) -> torch.Tensor:
        return torch.zeros(self.parameter_shape)

    # -- knots ------------------------------------------------------------------
    def compute_bins(self, u: torch.Tensor, minimum: float, maximum: float):
        """Knot positions ``(…, K+1)`` pinned at both ends, and the bin sizes ``(…, K)``."""
        share = self.min_bin_size + (1 - self.min_bin_size * self.n_bins) * torch.softmax(u, dim=-1)
        inner = (maximum - minimum) * torch.cumsum(share, dim=-1)[..., :-1] + minimum
        first = torch.full_like(u[..., :1], minimum)
        last = torch.full_like(u[..., :1], maximum)
        knots = torch.cat([first, inner, last], dim=-1)
        return knots, knots[..., 1:] - knots[..., :-1]

    def _split(self, h: torch.Tensor):
        K = self.n_bins
        u_d = F.pad(h[..., 2 * K:], pad=(1, 1), mode="constant", value=self.boundary_u_delta)
        return h[..., :K], h[..., K:2 * K], u_d

    def _bin(self, v: torch.Tensor, h: torch.Tensor, search_outputs: bool):
        """Everything that depends only on the bin an element falls into."""
        u_x, u_y, u_d = self._split(h)
        K = self.n_bins
        bin_x, widths = self.compute_bins(u_x, self.min_input, self.max_input)
        bin_y, heights = self.compute_bins(u_x + u_y / 1000, self.min_output, self.max_output)
        deltas = self.min_delta + F.softplus(self.boundary_u_delta + u_d / 1000)
        knots = bin_y if search_outputs else bin_x
        # number of knots strictly below v, minus one (searchsorted, right=False)
        k = ((knots < v[..., None]).sum(dim=-1, keepdim=True) - 1).clamp(0, K - 1)
        pick = lambda t, idx: torch.gather(t, -1, idx).squeeze(-1)
        y_k, x_k = pick(bin_y, k), pick(bin_x, k)
        h_k, w_k = pick(heights, k), pick(widths, k)
        d_k, d_k1 = pick(deltas, k), pick(deltas, k + 1)
        s_k = h_k / w_k
        return x_k, y_k, w_k, h_k, d_k, d_k1, s_k, d_k1 + d_k - 2 * s_k

    @staticmethod
    def log_det(s_k, deltas_k, deltas_kp1, xi, xi_1m_xi, term1):
        log_num = 2 * torch.log(s_k) + torch.log(
            deltas_kp1 * xi ** 2 + 2 * s_k * xi_1m_xi + deltas_k * (1 - xi) ** 2)
        log_den = 2 * torch.log(s_k + term1 * xi_1m_xi)
        return log_num - log_den

    # -- the two directions on flat inputs (n,), parameters (n, 3K-1) -------------
    def forward_1d(self, x: torch.Tensor, h: torch.Tensor):
        x_k, y_k, w_k, h_k, d_k, d_k1, s_k, term1 = self._bin(x, h, search_outputs=False)
        xi = torch.clip((x - x_k) / w_k, 0.0, 1.0)
        q = xi * (1 - xi)
        z = y_k + h_k * (s_k * xi ** 2 + d_k * q) / (s_k + term1 * q)
        return z, self.log_det(s_k, d_k, d_k1, xi, q, term1)

    def inverse_1d(self, z: torch.Tensor, h: torch.Tensor):
        x_k, y_k, w_k, h_k, d_k, d_k1, s_k, term1 = self._bin(z, h, search_outputs=True)
        t0 = z - y_k
        t2 = h_k * d_k
        a = (h_k * s_k - t2) + t0 * term1
        b = t2 - t0 * term1
        c = -s_k * t0
        root = torch.clip(torch.sqrt(b ** 2 - 4 * a * c), min=0.0)
        xi = torch.clip(2 * c / (-b - root), 0.0, 1.0)
        q = xi * (1 - xi)
        return xi * w_k + x_k, -self.log_det(s_k, d_k, d_k1, xi, q, term1)
