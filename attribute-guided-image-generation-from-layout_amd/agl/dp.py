"""Data-parallel gradient exchange: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" in the CPU tests).  The reference has no distributed code; plain DP
semantics are used: replicas hold full copies of G and the three D nets, BatchNorm statistics stay
local, and the only exchange is a SUM all-reduce of each flat gradient arena (scaled by 1/world inside
the fused Adam kernel).  The all-reduce runs on a side stream so it overlaps independent compute.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, group: Optional[dist.ProcessGroup] = None):
        self.enabled = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.group = group
        self.world = dist.get_world_size(group) if self.enabled else 1
        self._stream = None

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world

    def side_stream(self, device):
        if self._stream is None:      # high priority: the exchange + Adam sit on the step's critical path, the chains beside them do not
            self._stream = torch.cuda.Stream(device=device, priority=-1)
        return self._stream

    def all_reduce_(self, flat: torch.Tensor):
        """In-place SUM all-reduce of a flat gradient arena on the current stream."""
        if self.enabled:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        return flat

    def broadcast_(self, flat: torch.Tensor, src: int = 0):
        if self.enabled:
            dist.broadcast(flat, src=src, group=self.group)
        return flat
