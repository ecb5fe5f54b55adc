"""`MetaLoader`, `move_to_cuda`, `PrefetchLoader`, `build_dataloader` with the reference's names and iteration protocol
(pretrain_src/data/loader.py:18-75, 77-124, 127-164), for loaders whose collate_fn comes from data/collate.py: the next batch's
single H2D copy and its unpack kernels run on a copy stream while the current step computes, and the per-step task choice of a
multi-GPU job needs no collective."""
from __future__ import annotations

from typing import Dict, Iterator, Tuple

import numpy as np
import torch
import torch.distributed as dist
from torch.utils.data import DataLoader, RandomSampler, SequentialSampler
from torch.utils.data.distributed import DistributedSampler

from .collate import PackedBatch


class MetaLoader:
    """Endless iterator over several task loaders: every `accum_steps` steps a task is drawn with probability proportional to
    its ratio, then that task's next batch is yielded as `(task, batch)`; an exhausted loader gets `pre_epoch(epoch)` (a
    DistributedSampler's `set_epoch`) and is restarted (loader.py:18-75).

    `loaders`: {name: DataLoader | (DataLoader, ratio, pre_epoch)} as the reference builds it (main_r2r.py:191-199).

    Task choice, `sampling=`:
      * "reference": `torch.multinomial(ratios, 1)` on `device` from torch's global stream, and in distributed mode a broadcast
        of the draw from rank 0 (loader.py:56-59) -- the reference's exact draws; costs a collective plus a device->host read
        per step;
      * "shared_seed" (the default in distributed mode): task(step) is a pure function of (`seed`, draw index) on the host
        (parallel.TaskSchedule: numpy PCG64), identical on every rank by construction -- same distribution, no collective, no
        synchronisation.  Non-distributed default: "reference" with the ratios kept on the host."""

    def __init__(self, loaders, accum_steps: int = 1, distributed: bool = False, device=None, sampling: str | None = None, seed: int = 0):
        assert isinstance(loaders, dict)
        self.name2loader, self.name2iter, self.name2pre_epoch = {}, {}, {}
        self.names, ratios = [], []
        for n, l in loaders.items():
            if isinstance(l, tuple):
                l, r, p = l
            elif isinstance(l, DataLoader):
                r, p = 1, (lambda e: None)
            else:
                raise ValueError()
            self.names.append(n)
            self.name2loader[n], self.name2iter[n], self.name2pre_epoch[n] = l, iter(l), p
            ratios.append(r)
        self.accum_steps, self.device, self.distributed = accum_steps, device, distributed
        self.sampling = sampling or ("shared_seed" if distributed else "reference")
        assert self.sampling in ("reference", "shared_seed"), self.sampling
        explicit = sampling == "reference"
        self.sampling_ratios = torch.tensor(ratios).float().to(device if explicit else None)
        self.step = 0
        if self.sampling == "shared_seed":
            from ..parallel import TaskSchedule
            self._sched = TaskSchedule(tasks=self.names, ratios=dict(zip(self.names, ratios)), seed=seed, cyclic=False)
            self._draws = 0

    def _draw(self) -> int:
        if self.sampling == "shared_seed":
            t = self.names.index(self._sched.task_at(self._draws))
            self._draws += 1
            return t
        task_id = torch.multinomial(self.sampling_ratios, 1)
        if self.distributed:
            dist.broadcast(task_id, 0)              # make sure every process trains the same task
        return int(task_id.cpu().item())

    def __iter__(self) -> Iterator[Tuple]:
        """runs indefinitely"""
        task_id, epoch_id = None, 0
        while True:
            if self.step % self.accum_steps == 0:
                task_id = self._draw()
            self.step += 1
            task = self.names[task_id]
            try:
                batch = next(self.name2iter[task])
            except StopIteration:
                epoch_id += 1
                self.name2pre_epoch[task](epoch_id)          # reshuffle: DistributedSampler.set_epoch before the new iterator
                self.name2iter[task] = iter(self.name2loader[task])
                batch = next(self.name2iter[task])
            yield task, batch


def move_to_cuda(batch, device, out=None, text_pack=False):
    """loader.py:77-87, plus PackedBatch -> the collated dict on `device` (text_pack: see PackedBatch.to_device)"""
    if isinstance(batch, PackedBatch):
        return batch.to_device(device, out=out, text_pack=text_pack)
    if isinstance(batch, torch.Tensor):
        return batch.to(device, non_blocking=True)
    if isinstance(batch, list):
        return [move_to_cuda(t, device, text_pack=text_pack) for t in batch]
    if isinstance(batch, tuple):
        return tuple(move_to_cuda(t, device, text_pack=text_pack) for t in batch)
    if isinstance(batch, dict):
        return {n: move_to_cuda(t, device, text_pack=text_pack) for n, t in batch.items()}
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

    def __init__(self, loader, device: torch.device, text_pack: bool = False):
        """text_pack: batches carry their text packing plan (three index tensors beyond the reference's keys, PackedBatch.to_device)"""
        self.loader = loader
        self.text_pack = text_pack
        self.device = torch.device(device)
        # (one HIP stream per role and process, distinct from the capture / side / update / exchange streams: streams.role_stream)
        self.stream = None
        if self.device.type == "cuda":
            from .. import streams
            self.stream = streams.role_stream(self.device, "h2d")

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
            self.batch = move_to_cuda(self.batch, self.device, text_pack=self.text_pack)
            return
        with torch.cuda.stream(self.stream):
            self.batch = move_to_cuda(self.batch, self.device, text_pack=self.text_pack)

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


def build_dataloader(task, dataset, collate_fn, is_train: bool, opts):
    """-> (DataLoader, pre_epoch) as loader.py:127-164: `opts.train_batch_size` / `val_batch_size` (halved for ITM, whose batch
    grows five-fold inside the model), a random / sequential sampler for one process -- batch scaled by the number of visible
    GPUs, the reference's DataParallel case -- or a DistributedSampler(shuffle=is_train) whose `set_epoch` is the pre-epoch hook."""
    batch_size = opts.train_batch_size if is_train else opts.val_batch_size
    if task == "itm":
        batch_size //= 2
    if opts.local_rank == -1:
        sampler = RandomSampler(dataset) if is_train else SequentialSampler(dataset)
        pre_epoch = lambda e: None
        n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 1
        if n_dev > 1:
            batch_size *= n_dev
    else:
        sampler = DistributedSampler(dataset, num_replicas=dist.get_world_size(), rank=dist.get_rank(), shuffle=is_train)
        pre_epoch = sampler.set_epoch
    loader = DataLoader(dataset, sampler=sampler, batch_size=batch_size, num_workers=opts.n_workers, pin_memory=opts.pin_mem,
                        collate_fn=collate_fn, drop_last=False)
    return loader, pre_epoch
