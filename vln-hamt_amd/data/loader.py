"""`move_to_cuda` / `PrefetchLoader` with the reference's names and iteration protocol (pretrain_src/data/loader.py:77-124),
for loaders whose collate_fn comes from data/collate.py: the next batch's single H2D copy and its unpack kernels run on
a copy stream while the current step computes."""
from __future__ import annotations

import torch

from .collate import PackedBatch


def move_to_cuda(batch, device, out=None):
    """loader.py:77-87, plus PackedBatch -> the collated dict on `device`"""
    if isinstance(batch, PackedBatch):
        return batch.to_device(device, out=out)
    if isinstance(batch, torch.Tensor):
        return batch.to(device, non_blocking=True)
    if isinstance(batch, list):
        return [move_to_cuda(t, device) for t in batch]
    if isinstance(batch, tuple):
        return tuple(move_to_cuda(t, device) for t in batch)
    if isinstance(batch, dict):
        return {n: move_to_cuda(t, device) for n, t in batch.items()}
    return batch


def _record(obj, stream):
    if isinstance(obj, torch.Tensor):
        obj.record_stream(stream)
    elif isinstance(obj, (list, tuple)):
        for t in obj:
            _record(t, stream)
    elif isinstance(obj, dict):
        for t in obj.values():
            _record(t, stream)


class PrefetchLoader:
    """overlap compute and host->device transfer (loader.py:90-124): same `__iter__` / `__len__` / attribute forwarding;
    the transfer of batch i+1 is issued on a copy stream as soon as batch i has been handed out."""

    def __init__(self, loader, device: torch.device):
        self.loader = loader
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None

    def __iter__(self):
        loader_it = iter(self.loader)
        self.preload(loader_it)
        batch = self.next(loader_it)
        while batch is not None:
            yield batch
            batch = self.next(loader_it)

    def __len__(self):
        return len(self.loader)

    def preload(self, it):
        try:
            self.batch = next(it)
        except StopIteration:
            self.batch = None
            return
        if self.stream is None:
            self.batch = move_to_cuda(self.batch, self.device)
            return
        with torch.cuda.stream(self.stream):
            self.batch = move_to_cuda(self.batch, self.device)

    def next(self, it):
        batch = self.batch
        if batch is not None and self.stream is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_stream(self.stream)
            _record(batch, cur)
        self.preload(it)
        return batch

    def __getattr__(self, name):
        return self.loader.__getattribute__(name)
