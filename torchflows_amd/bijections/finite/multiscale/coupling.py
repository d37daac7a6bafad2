"""Image coupling masks (reference ``multiscale/coupling.py`` :6-82).

Integer rules kept bit-for-bit (tests compare with the reference's masks):

* ``Checkerboard``: ``arange(h*w) % 2`` reshaped to ``(h, w)`` and repeated over channels --
  for even ``w`` that is a *column-parity* mask, not the usual ``(i+j) % 2`` board;
* ``ChannelWiseHalfSplit``: the first ``c // 2`` channels are the source;
* ``constant_shape`` / ``target_shape`` are the image shapes the conditioner / transformer
  see: ``(c, h//2, w)`` for the checkerboard, channel slices for the channel-wise split.
"""
from __future__ import annotations

import torch

from torchflows_amd.bijections.finite.autoregressive.conditioning.coupling_masks import Coupling


class Checkerboard(Coupling):
    def __init__(self, event_shape, invert: bool = False, **kwargs):
        channels, height, width = event_shape
        board = (torch.arange(height * width) % 2).view(height, width).bool()
        mask = board[None].repeat(channels, 1, 1)
        super().__init__(event_shape, ~mask if invert else mask)

    @property
    def constant_shape(self):
        c, h, w = self.event_shape
        return c, h // 2, w

    @property
    def target_shape(self):
        return self.constant_shape


class ChannelWiseHalfSplit(Coupling):
    def __init__(self, event_shape, invert: bool = False):
        c, h, w = event_shape
        if c <= 1:
            raise ValueError("Number of channels must be at least 2")
        mask = (torch.arange(c) < c // 2)[:, None, None].repeat(1, h, w)
        super().__init__(event_shape, ~mask if invert else mask)

    # note: with invert=True and an odd channel count these shapes do not match the mask
    # (reference quirk Q13); the product only builds non-inverted odd splits
    @property
    def constant_shape(self):
        c, h, w = self.event_shape
        return c // 2, h, w

    @property
    def target_shape(self):
        c, h, w = self.event_shape
        return c - c // 2, h, w


def make_image_coupling(event_shape, coupling_type: str, **kwargs):
    if coupling_type == "checkerboard":
        return Checkerboard(event_shape, invert=False, **kwargs)
    if coupling_type == "checkerboard_inverted":
        return Checkerboard(event_shape, invert=True, **kwargs)
    if coupling_type == "channel_wise":
        return ChannelWiseHalfSplit(event_shape, invert=False)
    if coupling_type == "channel_wise_inverted":
        return ChannelWiseHalfSplit(event_shape, invert=True)
    raise ValueError(f"unknown image coupling type {coupling_type!r}")
