"""Multi-GPU sharding of the hot path: one process per GPU, ``torch.distributed`` (backend "nccl" =
RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference has no parallelism on this path (``main.py:259-288`` is a serial loop); tiles are
independent, so they shard embarrassingly:

* :func:`segment_page_sharded` -- ONE page, its tiles split into contiguous ranges of the
  reference's call order (x outer, y inner).  Each rank runs its range, then one **all-gather** of
  the u8 tile label maps puts every tile on every rank and each rank stitches the page mask locally
  (margin crop + last-writer-wins, closed form).  This is the collective ``north_star`` names.
* :func:`segment_pages_sharded` -- MANY pages, whole pages per rank (stitching stays local), one
  all-gather of the finished u8 page masks.

There is no other data-path collective.  The device work goes through a small backend object so
the same sharding/collective logic runs under gloo with a numpy backend in the CPU tests.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np


def shard_block(n_items: int, rank: int, world: int) -> Tuple[int, int, int]:
    """Contiguous equal blocks of size ceil(n/world): returns (first, count, block).
    Item t then sits at index t of the all-gathered [world*block] buffer -- no re-indexing."""
    block = -(-n_items // world) if n_items else 0
    first = min(rank * block, n_items)
    last = min(first + block, n_items)
    return first, last - first, block


class DeviceBackend:
    """libsbbseg on one GPU; tensors are torch CUDA tensors (device memory plumbing only)."""

    def __init__(self, model):
        import torch
        self.torch = torch
        self.model = model
        self.ctx = model.ctx
        self.H, self.W, self.classes, _ = self.ctx.model_info()
        self.device = torch.device("cuda", model.device)
        self.ctx.set_stream(torch.cuda.current_stream(self.device).cuda_stream)

    def empty(self, shape):
        return self.torch.empty(shape, dtype=self.torch.uint8, device=self.device)

    def to_device(self, page_u8: np.ndarray):
        return self.torch.from_numpy(np.ascontiguousarray(page_u8)).to(self.device)

    def tile_range(self, d_page, first: int, count: int, out_tiles) -> None:
        Hp, Wp = int(d_page.shape[0]), int(d_page.shape[1])
        if count:
            # the tile labels of a shard are only ever stitched (main.py:294-364 keeps each tile's owned region): let the decoder skip the
            # rest (sbbseg_set_owned_regions mode 2), unless the caller switched owned-region launches off altogether
            mode = self.ctx.owned_region_info()[0]
            if mode == 1:
                self.ctx.set_owned_regions(2)
            try:
                self.ctx.segment_tile_range_dev(d_page.data_ptr(), Hp, Wp, first, count, out_tiles.data_ptr())
            finally:
                if mode == 1:
                    self.ctx.set_owned_regions(1)

    def stitch(self, all_tiles, Hp: int, Wp: int, out_page) -> None:
        self.ctx.stitch_dev(all_tiles.data_ptr(), Hp, Wp, out_page.data_ptr())

    def whole_page(self, d_page, out_page) -> None:
        self.ctx.segment_page_dev(d_page.data_ptr(), int(d_page.shape[0]), int(d_page.shape[1]), out_page.data_ptr())

    def whole_pages(self, d_pages, out_pages) -> None:
        """Equally sized pages in one library call: their tiles are pooled into max_batch-sized chunks."""
        Hp, Wp = int(d_pages[0].shape[0]), int(d_pages[0].shape[1])
        self.ctx.segment_pages_dev([p.data_ptr() for p in d_pages], Hp, Wp, [o.data_ptr() for o in out_pages])


def _world(group):
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def segment_page_sharded(backend, d_page, n_tiles: int, group=None):
    """Label map [Hp, Wp] u8 of one page, tiles sharded over the ranks of ``group``; the result is
    complete on every rank.  ``n_tiles`` = nxf*nyf of the page (``_capi.tile_grid``)."""
    import torch.distributed as dist
    rank, world = _world(group)
    Hp, Wp = int(d_page.shape[0]), int(d_page.shape[1])
    first, count, block = shard_block(n_tiles, rank, world)
    mine = backend.empty((block, backend.H, backend.W))
    backend.tile_range(d_page, first, count, mine)
    if world > 1:
        all_tiles = backend.empty((world * block, backend.H, backend.W))
        dist.all_gather_into_tensor(all_tiles.view(-1), mine.view(-1), group=group)
    else:
        all_tiles = mine
    out = backend.empty((Hp, Wp))
    backend.stitch(all_tiles, Hp, Wp, out)
    return out


def segment_pages_sharded(backend, pages: Sequence[np.ndarray], group=None):
    """Label maps [n_pages, Hp, Wp] u8 for same-sized pages, whole pages per rank, one all-gather."""
    import torch.distributed as dist
    rank, world = _world(group)
    n = len(pages)
    Hp, Wp = pages[0].shape[:2]
    if any(p.shape[:2] != (Hp, Wp) for p in pages):
        raise ValueError("segment_pages_sharded needs equally sized pages")
    first, count, block = shard_block(n, rank, world)
    mine = backend.empty((block, Hp, Wp))
    if count and hasattr(backend, "whole_pages"):
        # groups of at most 8 pages on the device at once (the library pools the tiles of a group into full chunks): device
        # memory stays bounded by the group, not by the shard
        for g0 in range(0, count, 8):
            ks = range(g0, min(g0 + 8, count))
            backend.whole_pages([backend.to_device(pages[first + k]) for k in ks], [mine[k] for k in ks])
    else:
        for k in range(count):
            backend.whole_page(backend.to_device(pages[first + k]), mine[k])
    if world > 1:
        everything = backend.empty((world * block, Hp, Wp))
        dist.all_gather_into_tensor(everything.view(-1), mine.view(-1), group=group)
    else:
        everything = mine
    return everything[:n]
