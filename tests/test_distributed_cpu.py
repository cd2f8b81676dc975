"""Multi-rank path on CPU: world_size-2 gloo processes run distributed.segment_page_sharded /
segment_pages_sharded with a numpy backend that labels tiles like the fixture's FakeModel; the
stitched result must reproduce the CRC captured from the reference loop (tiling_golden.json)."""
import json
import os
import socket
import zlib

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import tiling
from sbb_textline_detection_amd import distributed as D

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "tiling_golden.json")))["cases"]


class NumpyBackend:
    """CPU stand-in for DeviceBackend: FakeModel's label rule per tile, oracle owner map for the stitch."""

    def __init__(self, H, W, classes=16, owned_only=False):
        self.H, self.W, self.classes = H, W, classes
        # owned_only: what sbbseg_segment_tile_range_dev returns under sbbseg_set_owned_regions(2) -- a tile's labels are defined on the
        # region the page stitch keeps of it (the library's closed form, csrc/region.h) and are junk (0xEE) everywhere else
        self.owned_only = owned_only

    def empty(self, shape):
        return torch.zeros(shape, dtype=torch.uint8)

    def to_device(self, page):
        return torch.from_numpy(np.ascontiguousarray(page))

    def tile_range(self, d_page, first, count, out_tiles):
        page = d_page.numpy().astype(np.int64)
        tiles, _, _ = tiling.tile_grid(page.shape[0], page.shape[1], self.H, self.W)
        yy, xx = np.mgrid[0:self.H, 0:self.W]
        for k in range(count):
            t = tiles[first + k]
            p = page[t["y0"]:t["y0"] + self.H, t["x0"]:t["x0"] + self.W]
            lab = (((first + k) * 5 + yy * 3 + xx * 7 + p[:, :, 0] + 2 * p[:, :, 1]) % self.classes).astype(np.uint8)
            if self.owned_only:
                from sbb_textline_detection_amd import _capi
                margin = int(0.1 * self.W)
                nx = max(tt["i"] for tt in tiles) + 1
                ny = max(tt["j"] for tt in tiles) + 1
                ylo, yhi = _capi.owned_range(page.shape[0], self.H, margin, ny, t["j"])
                xlo, xhi = _capi.owned_range(page.shape[1], self.W, margin, nx, t["i"])
                junk = np.full_like(lab, 0xEE)
                junk[ylo:yhi, xlo:xhi] = lab[ylo:yhi, xlo:xhi]
                lab = junk
            out_tiles[k] = torch.from_numpy(lab)

    def stitch(self, all_tiles, Hp, Wp, out_page):
        own = tiling.owner_map(Hp, Wp, self.H, self.W)
        tiles, _, _ = tiling.tile_grid(Hp, Wp, self.H, self.W)
        x0 = np.array([t["x0"] for t in tiles]); y0 = np.array([t["y0"] for t in tiles])
        yy, xx = np.mgrid[0:Hp, 0:Wp]
        out_page.copy_(torch.from_numpy(all_tiles.numpy()[own, yy - y0[own], xx - x0[own]]))

    def whole_page(self, d_page, out_page):
        n = len(tiling.tile_grid(d_page.shape[0], d_page.shape[1], self.H, self.W)[0])
        tiles = self.empty((n, self.H, self.W))
        self.tile_range(d_page, 0, n, tiles)
        self.stitch(tiles, d_page.shape[0], d_page.shape[1], out_page)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, case, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        be = NumpyBackend(case["model_h"], case["model_w"], case["classes"])
        page = tiling.coord_page(case["page_h"], case["page_w"])
        out = D.segment_page_sharded(be, be.to_device(page), case["n_calls"]).numpy()
        crc = zlib.crc32(np.ascontiguousarray(out).tobytes()) & 0xFFFFFFFF
        pages = [page, page[::-1].copy(), page[:, ::-1].copy()]
        maps = D.segment_pages_sharded(be, pages).numpy()
        ref1 = np.zeros(page.shape[:2], np.uint8); be1 = torch.zeros(page.shape[:2], dtype=torch.uint8)
        be.whole_page(be.to_device(pages[1]), be1)
        q.put((rank, crc, int(out.astype(np.int64).sum()), maps.shape, bool(np.array_equal(maps[0], out)),
               bool(np.array_equal(maps[1], be1.numpy()))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", [c for c in GOLD if (c["page_h"], c["page_w"]) in ((777, 1234), (449, 1000), (700, 900))],
                         ids=lambda c: f"{c['page_h']}x{c['page_w']}")
def test_sharded_page_matches_reference_fixture_world2(case):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, case, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, crc, total, shape, same_first, same_second in res:
        assert crc == case["out_crc32"] and total == case["out_sum"], f"rank {rank}"
        assert tuple(shape) == (3, case["page_h"], case["page_w"]) and same_first and same_second


def test_shard_block_covers_everything():
    for n in (0, 1, 7, 70, 108, 6912):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                first, count, block = D.shard_block(n, r, world)
                assert first == min(r * block, n) and count <= block
                seen += list(range(first, first + count))
            assert seen == list(range(n))


def _worker_pages(rank, world, port, n_pages, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        be = NumpyBackend(224, 224, 7)
        base = tiling.coord_page(500, 460)
        pages = [np.roll(base, 37 * k, axis=1).copy() for k in range(n_pages)]
        maps = D.segment_pages_sharded(be, pages).numpy()
        ok = maps.shape == (n_pages, 500, 460)
        for k, p in enumerate(pages):                      # every rank holds every page's map, equal to the single-process result
            one = torch.zeros(p.shape[:2], dtype=torch.uint8)
            be.whole_page(be.to_device(p), one)
            ok = ok and bool(np.array_equal(maps[k], one.numpy()))
        q.put((rank, ok, D.shard_block(n_pages, rank, world)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_pages", [5, 6])
def test_sharded_pages_world4_with_padding_rows(n_pages):
    """A page count the world does not divide: shard_block pads every rank's contribution to ceil(n/world) rows (rank 3 owns
    nothing at 5 or 6 pages over 4 ranks), the all-gather carries the padding, and the result is cut back to n pages."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_pages, args=(r, 4, port, n_pages, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    counts = {rank: blk[1] for rank, _, blk in res}
    assert sum(counts.values()) == n_pages and counts[3] == 0 and all(blk[2] == 2 for _, _, blk in res)


def _worker_w8(rank, world, port, case, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # every rank returns its tiles' OWNED regions only (sbbseg_set_owned_regions(2): what DeviceBackend.tile_range asks for)
        be = NumpyBackend(case["model_h"], case["model_w"], case["classes"], owned_only=True)
        page = tiling.coord_page(case["page_h"], case["page_w"])
        out = D.segment_page_sharded(be, be.to_device(page), case["n_calls"]).numpy()
        crc = zlib.crc32(np.ascontiguousarray(out).tobytes()) & 0xFFFFFFFF
        q.put((rank, crc, int(out.astype(np.int64).sum()), D.shard_block(case["n_calls"], rank, world)))
    finally:
        dist.destroy_process_group()


def test_sharded_3500x2500_page_world8_uneven_tail_owned_regions_only():
    """BASELINE configs[1]'s page over EIGHT ranks: 70 tiles -> 9 / 9 / ... / 9 / 7 (shard_block pads the last rank's contribution to the
    block of 9), every rank hands the all-gather only the OWNED region of each of its tiles (junk elsewhere), and the stitched mask on
    every rank still reproduces the CRC captured from the reference's own loop (tiling_golden.json)."""
    case = [c for c in GOLD if (c["page_h"], c["page_w"]) == (3500, 2500)][0]
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_w8, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    counts = {}
    for rank, crc, total, blk in res:
        assert crc == case["out_crc32"] and total == case["out_sum"], f"rank {rank}"
        counts[rank] = blk[1]
    assert [counts[r] for r in range(world)] == [9] * 7 + [7]


def test_sharded_64_pages_world8():
    """BASELINE configs[3]'s sharding: 64 pages over eight ranks = 8 whole pages per rank, one all-gather of the masks."""
    world, n_pages = 8, 64
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_pages, args=(r, world, port, n_pages, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert all(blk[1] == 8 and blk[2] == 8 for _, _, blk in res)
