"""Coupling partitions: which flat event positions feed the conditioner (source) and which
are transformed (target).

Integer rules follow the reference exactly (``conditioning/coupling_masks.py``:
``PartialCoupling`` :6-47, ``Coupling`` :50-60, ``GraphicalCoupling`` :63-75,
``HalfSplit`` :78-81, ``make_coupling`` :84-99) and are compared bit-for-bit with it in
tests/.  In addition to the boolean masks each partition carries the two gather lists
(ascending flat indices = what ``x[..., mask]`` selects) that the HIP kernels take, and
knows whether the target is the contiguous tail -- the layout the vectorised kernels use.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch

from torchflows_amd.utils import event_size


class PartialCoupling:
    def __init__(self, event_shape: Sequence[int], source_mask: torch.Tensor, target_mask: torch.Tensor):
        self.event_shape = event_shape
        self.source_mask = source_mask
        self.target_mask = target_mask
        self.event_size = event_size(event_shape)
        flat_s = source_mask.reshape(-1)
        flat_t = target_mask.reshape(-1)
        self.source_index = torch.nonzero(flat_s, as_tuple=False).reshape(-1).to(torch.int32)
        self.target_index = torch.nonzero(flat_t, as_tuple=False).reshape(-1).to(torch.int32)

    @property
    def source_event_size(self) -> int:
        return int(self.source_index.numel())

    @property
    def target_event_size(self) -> int:
        return int(self.target_index.numel())

    @property
    def ignored_event_size(self) -> int:
        return self.event_size - int((self.source_mask | self.target_mask).sum())

    @property
    def constant_shape(self) -> Tuple[int, ...]:
        return (self.source_event_size,)

    @property
    def target_shape(self) -> Tuple[int, ...]:
        return (self.target_event_size,)

    # -- layout facts the kernels exploit ------------------------------------
    @property
    def target_is_tail(self) -> bool:
        """target == flat positions [D - T, D)"""
        T, D = self.target_event_size, self.event_size
        return T > 0 and bool(torch.equal(self.target_index, torch.arange(D - T, D, dtype=torch.int32)))

    @property
    def source_is_head(self) -> bool:
        """source == flat positions [0, S)"""
        S = self.source_event_size
        return S > 0 and bool(torch.equal(self.source_index, torch.arange(S, dtype=torch.int32)))


class Coupling(PartialCoupling):
    """Every position is either source or target (reference :50-60)."""

    def __init__(self, event_shape: Sequence[int], mask: torch.Tensor):
        super().__init__(event_shape, source_mask=mask, target_mask=~mask)

    @property
    def ignored_event_size(self) -> int:
        return 0


class HalfSplit(Coupling):
    """First ``D // 2`` flat positions are the source, the rest the target (reference :78-81)."""

    def __init__(self, event_shape: Sequence[int]):
        D = event_size(event_shape)
        super().__init__(event_shape, mask=(torch.arange(D) < D // 2).view(*event_shape))


class GraphicalCoupling(PartialCoupling):
    """Sources / targets read off a directed edge list (reference :63-75; vectors only)."""

    def __init__(self, event_shape: Sequence[int], edge_list: List[Tuple[int, int]]):
        if len(event_shape) != 1:
            raise ValueError("GraphicalCoupling is currently only implemented for vector data")
        D = event_size(event_shape)
        positions = torch.arange(D)
        sources = torch.tensor(sorted({int(a) for a, _ in edge_list}), dtype=torch.long)
        targets = torch.tensor(sorted({int(b) for _, b in edge_list}), dtype=torch.long)
        super().__init__(event_shape, torch.isin(positions, sources), torch.isin(positions, targets))


def make_coupling(event_shape: Sequence[int], edge_list: Optional[List[Tuple[int, int]]] = None,
                  coupling_type: str = "half_split", **kwargs) -> PartialCoupling:
    """Reference :84-99."""
    if edge_list is not None:
        return GraphicalCoupling(event_shape, edge_list)
    if coupling_type == "half_split":
        return HalfSplit(event_shape)
    raise ValueError(f"unknown coupling_type {coupling_type!r}")
