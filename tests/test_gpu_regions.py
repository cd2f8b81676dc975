"""-m gpu: owned-region launches of the decoder (csrc/region.h, sbbseg_set_owned_regions) against the whole-tile launches.

do_prediction keeps only the margin-cropped, last-writer-wins part of every tile (main.py:276-281, 294-364).  The fused page paths launch
dec1 ... tail over that region (+ the halo the later levels read) only; everything they DO compute goes through the arithmetic of the
full launch, so the stitched label map must be the same byte for byte -- on every page size of the reference fixture, on the three
model sizes of the fixture (448 x 448, the non-square 320 x 480 whose margin comes from the width, 224 x 224), in both 16-bit modes,
under both lane settings, for pooled pages, and for the sharded tile ranges.  Activation buffers are filled with NaNs between the runs:
a level that read something an owned-region launch below it did not write would show."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from gpu_common import EXACT_MARGIN, make_model  # noqa: E402
from oracle import keras_forward as kf  # noqa: E402
from oracle import tiling  # noqa: E402
from sbb_textline_detection_amd import _capi  # noqa: E402
from sbb_textline_detection_amd.synthetic import synthetic_page  # noqa: E402

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "tiling_golden.json")))["cases"]


def _both(model, fn):
    """fn() with whole-tile launches, then -- buffers poisoned -- with owned-region launches."""
    model.ctx.set_owned_regions(0)
    full = fn()
    model.ctx.poison_activations(0xFF)
    model.ctx.set_owned_regions(1)
    model.ctx.profile_reset()
    owned = fn()
    return full, owned


@pytest.fixture(scope="module", params=["f16x3", "f16"])
def model448(request):
    cfg, w, g, model = make_model(2, 448, 448, seed=0, precision=request.param, max_batch=24)
    model.test_graph, model.test_weights, model.test_precision = g, w, request.param
    yield model
    model.release()


def test_plan_has_five_owned_region_levels(model448):
    assert model448.ctx.owned_region_levels() == 5          # tail, dec4 (224), dec3 (112), dec2 (56), dec1 (28)


@pytest.mark.parametrize("hw", [(c["page_h"], c["page_w"]) for c in GOLD if c["model_h"] == 448 and c["page_h"] * c["page_w"] <= 3500 * 2500],
                         ids=lambda s: f"{s[0]}x{s[1]}")
def test_owned_region_launches_equal_whole_tile_launches(hw, model448):
    m = model448
    page = synthetic_page(hw[0], hw[1], seed=hw[0] + hw[1])
    for dedupe in (True, False):
        m.ctx.set_dedupe(dedupe)
        for lanes in (2, 1):
            m.ctx.set_lanes(lanes)
            full, owned = _both(m, lambda: m.segment_page(page))
            assert 0.01 < float(full.mean()) < 0.99
            assert np.array_equal(full, owned), (hw, dedupe, lanes, int((full != owned).sum()))
    m.ctx.set_dedupe(True)
    m.ctx.set_lanes(2)
    # the work really shrank: every decoder level reports fewer executed patch-equivalents than patches launched (a 448 x 448 page is one
    # tile that owns everything: nothing to save there)
    prof = [o for o in m.ctx.profile() if o["exec_patches"] > 0]
    total = max(o["exec_patches"] for o in prof)
    saved = [o["name"] for o in prof if o["exec_patches"] < total - 1e-9]
    if hw != (448, 448):
        assert len(saved) >= 4, [(o["name"], o["exec_patches"]) for o in prof]


def test_owned_regions_on_the_other_entry_points(model448):
    """rescaled page, Otsu-binarised page, crop, pooled pages, the host pipeline: all go through the same tile ranges."""
    import torch
    m = model448
    page = synthetic_page(900, 700, seed=6)
    full, owned = _both(m, lambda: m.ctx.segment_page_scaled(page, 1200, 933))
    assert np.array_equal(full, owned)
    full, owned = _both(m, lambda: m.ctx.segment_page_otsu(page, 1080, 840)[0])
    assert np.array_equal(full, owned)
    full, owned = _both(m, lambda: m.ctx.segment_crop(page, 1080, 840, (37, 51, 760, 980), binarise=True)[0])
    assert np.array_equal(full, owned)
    pages = [synthetic_page(1000, 900, seed=s) for s in (1, 2, 3, 4, 5)]           # 9 tiles each, chunks of 24 span pages
    full, owned = _both(m, lambda: m.ctx.segment_pages(pages))
    for a, b in zip(full, owned):
        assert np.array_equal(a, b)
    d_pages = [torch.from_numpy(p).cuda() for p in pages]
    d_out = [torch.zeros((1000, 900), dtype=torch.uint8, device="cuda") for _ in pages]

    def pooled():
        m.ctx.segment_pages_dev([t.data_ptr() for t in d_pages], 1000, 900, [t.data_ptr() for t in d_out])
        m.ctx.synchronize()
        return [t.cpu().numpy().copy() for t in d_out]
    full2, owned2 = _both(m, pooled)
    for a, b, c in zip(full2, owned2, full):
        assert np.array_equal(a, b) and np.array_equal(a, c)


def test_sharded_tile_ranges_with_owned_regions(model448):
    """mode 2: sbbseg_segment_tile_range_dev computes owned regions too (a rank's tiles keep their owned regions whatever the range);
    ranges of uneven size, stitched, equal the one-call map.  Mode 1 leaves the tile-range entry point computing whole tiles."""
    import torch
    m = model448
    hp, wp = 1234, 1100
    page = synthetic_page(hp, wp, seed=12)
    m.ctx.set_owned_regions(0)
    ref = m.segment_page(page)
    xy, nxf, nyf = _capi.tile_grid(hp, wp, 448, 448)
    n = xy.shape[0]
    d_page = torch.from_numpy(page).cuda()
    for mode in (1, 2):
        m.ctx.set_owned_regions(mode)
        m.ctx.poison_activations(0xFF)
        d_tiles = torch.full((n, 448, 448), 7, dtype=torch.uint8, device="cuda")
        cuts = [0, 5, 6, n - 3, n]
        for a, b in zip(cuts[:-1], cuts[1:]):
            m.ctx.segment_tile_range_dev(d_page.data_ptr(), hp, wp, a, b - a, d_tiles[a:].data_ptr())
        d_out = torch.zeros((hp, wp), dtype=torch.uint8, device="cuda")
        m.ctx.stitch_dev(d_tiles.data_ptr(), hp, wp, d_out.data_ptr())
        m.ctx.synchronize()
        assert np.array_equal(d_out.cpu().numpy(), ref), mode
        untouched = int((d_tiles == 7).sum().item())
        assert (untouched == 0) if mode == 1 else (untouched > 0.2 * d_tiles.numel()), (mode, untouched)
    m.ctx.set_owned_regions(1)


def test_owned_region_page_matches_oracle_on_every_tile(model448):
    """One page, ALL of its tiles against the fp32 oracle (the map the owned-region launches produce, on each tile's owned pixels)."""
    m = model448
    if m.test_precision != "f16x3":
        pytest.skip("label-exact mode only")
    hp, wp = 1000, 1234                                     # 3 x 4 = 12 tiles: first / interior / penultimate / clamped last on both axes
    page = synthetic_page(hp, wp, seed=3)
    m.ctx.set_owned_regions(1)
    m.ctx.poison_activations(0xFF)
    got = m.segment_page(page)
    tiles, nxf, nyf = tiling.tile_grid(hp, wp, 448, 448)
    own = tiling.owner_map(hp, wp, 448, 448)
    differ = 0
    for k, t in enumerate(tiles):
        x = (page[t["y0"]:t["y0"] + 448, t["x0"]:t["x0"] + 448][None] / 255.0).astype(np.float32)
        ref = kf.forward(m.test_graph, m.test_weights, x)[0]
        sl = (slice(t["y0"] + t["ylo"], t["y0"] + t["yhi"]), slice(t["x0"] + t["xlo"], t["x0"] + t["xhi"]))
        r = ref[t["ylo"]:t["yhi"], t["xlo"]:t["xhi"]]
        srt = np.sort(r, axis=-1)
        mine = own[sl] == k
        mism = (got[sl] != r.argmax(-1)) & mine
        differ += int(mism.sum())
        assert not (mism & ((srt[..., -1] - srt[..., -2]) > EXACT_MARGIN)).any(), k
    print(f"[owned regions vs oracle, 12 tiles] {differ} labels differ, all inside the oracle's near-ties")


@pytest.mark.parametrize("mh,mw,hp,wp", [(320, 480, 700, 900), (224, 224, 512, 640), (320, 480, 1000, 1500), (224, 224, 900, 700)])
@pytest.mark.parametrize("precision", ["f16x3", "f16"])
def test_owned_regions_on_the_other_model_sizes(mh, mw, hp, wp, precision):
    """The fixture's non-square 320 x 480 model (margin = int(0.1 * 480) on BOTH axes, main.py:233) and its 224 x 224 model."""
    cfg, w, g, model = make_model(2, mh, mw, seed=4, precision=precision, max_batch=16, calib_hw=160)
    try:
        assert model.ctx.owned_region_levels() >= 3
        page = synthetic_page(hp, wp, seed=hp)
        full, owned = _both(model, lambda: model.segment_page(page))
        assert np.array_equal(full, owned), int((full != owned).sum())
    finally:
        model.release()


def test_generic_kernel_levels_and_tile_families(model448):
    """dec4 on conv_igemm_mfma (pixel table instead of the tile table), the 4-wave tile family, the plain gather: same map."""
    m = model448
    page = synthetic_page(1234, 777, seed=9)
    m.ctx.set_owned_regions(0)
    ref = m.segment_page(page)
    m.ctx.set_owned_regions(1)
    for variant in (1 << 23, 1, 1 << 17, (1 << 23) | (1 << 17)):
        m.ctx.set_conv_variant(variant)
        m.ctx.poison_activations(0xFF)
        try:
            got = m.segment_page(page)
        finally:
            m.ctx.set_conv_variant(0)
        if variant & 3:
            # another tile family rounds nothing differently (same K order per accumulator): still the same map
            pass
        assert np.array_equal(got, ref), hex(variant)


@pytest.mark.parametrize("precision", ["f16x3", "f16"])
def test_random_page_sizes_owned_equals_whole(precision):
    """Randomised geometry sweep on the 224 x 224 model (margin 22, mid 180): page extents that land in every regime of the clamp --
    exact multiples of mid, one pixel more, a repeated clamped tile (extent % 180 in (0, 44]: the dedupe paths), a last tile that owns a
    few columns only -- owned-region launches against whole-tile launches, byte for byte, with the reference's call list (dedupe off) too."""
    cfg, w, g, model = make_model(4, 224, 224, seed=9, precision=precision, max_batch=20, calib_hw=160)
    try:
        rng = np.random.RandomState(4321)
        sizes = [(224, 224), (225, 224 * 3), (180 * 3, 180 * 4), (180 * 3 + 1, 180 * 4 - 1), (180 * 2 + 44, 180 * 3 + 45), (404, 405), (583, 227)]
        sizes += [(int(rng.randint(224, 1000)), int(rng.randint(224, 1000))) for _ in range(6)]
        for hp, wp in sizes:
            page = synthetic_page(hp, wp, seed=hp * 7 + wp)
            for dedupe in (True, False):
                model.ctx.set_dedupe(dedupe)
                full, owned = _both(model, lambda: model.segment_page(page))
                assert np.array_equal(full, owned), (hp, wp, dedupe, int((full != owned).sum()))
    finally:
        model.release()
