// region.h -- owned-region launches of the decoder (round 6).
//
// do_prediction pastes only a part of every tile's label map into the page (main.py:294-364): the 10 % margin is cropped on every side
// that is not a page edge, and where the inward-clamped last tile of an axis overlaps its neighbour the later tile wins (main.py:276-281 +
// the paste order).  What survives of tile t along one axis -- its OWNED range -- is a closed form of the page geometry (region_own below:
// the per-axis rule `stitch_kernel`'s owner tables are built from).  A 3500 x 2500 page keeps 8.75 of the 14.05 Mpx its 70 tiles produce.
//
// The decoder is local: a level is a 3x3 conv over [nearest-x2 upsampling of the level below, skip from the encoder], followed by
// pointwise work.  So the rows a level must produce are the owned rows dilated by one pixel per 3x3 conv above it and halved per
// upsampling (region_down) -- the encoder, whose receptive field spans the tile, is computed whole.  The fused page paths launch the
// decoder convs and the tail over these regions only:
//   * conv_igemm_mfma (the parity-split decoder convs): a per-launch table maps the compact pixel index of a class grid to
//     (patch, y, x) -- kind 1 below;
//   * dec_halo_* and dec_tail_fused* (16 x 16 output tiles): a per-launch table of tile origins -- kind 0.
// Every computed pixel goes through exactly the arithmetic of the full launch (same K order, same epilogue), so the label map the stitch
// assembles is bit-identical; pixels outside a tile's region are simply never written (and never read: region_down covers every tap).
// The tables are built on the device per launch chunk (region.hip) from the closed forms below; the host evaluates the same closed forms
// for the entry counts (grid sizes, executed-work accounting).
#pragma once
#include "internal.h"

namespace sbbseg {

constexpr int kRegionMaxLevels = 8;

// one axis of the page's tile grid (main.py:246-281): n tiles of `tile` pixels, origin(t) = min(t * mid, extent - tile)
struct RegionAxis {
    int extent, tile, margin, mid, n;
};

struct RegionGeom {
    RegionAxis ax, ay;            // x axis (outer loop of the reference, index i), y axis (inner, index j)
    int tpp, ny;                  // tiles per page (ax.n * ay.n), tiles per column: grid index g -> i = g / ny, j = g % ny
    int n_levels;                 // level 0 = the network output (tail), level k = the decoder conv k steps below it
    int kind[kRegionMaxLevels];   // 0: table of 16 x 16 output tiles; 1: table of class-grid pixels (four output-parity classes)
    int Rh[kRegionMaxLevels], Rw[kRegionMaxLevels];      // output size of the level
    int align_x[kRegionMaxLevels];                        // kind 0: tile origins in x are multiples of this (16-byte label rows: 16; else 2)
};

SBBSEG_HD inline int region_origin(const RegionAxis& a, int t) { return t * a.mid + a.tile > a.extent ? a.extent - a.tile : t * a.mid; }

// [lo, hi) in TILE coordinates of what the stitch keeps of tile t (empty: lo == hi): the margin crop of main.py:294-364, minus
// everything a later tile of the axis writes -- later ranges chain up without a gap, so only the next tile's first pixel matters
SBBSEG_HD inline void region_own(const RegionAxis& a, int t, int& lo, int& hi)
{
    const int o = region_origin(a, t);
    lo = t == 0 ? 0 : a.margin;
    hi = t == a.n - 1 ? a.tile : a.tile - a.margin;
    if (t + 1 < a.n) {
        const int cut = region_origin(a, t + 1) + a.margin - o;
        if (cut < hi) hi = cut;
    }
    if (hi < lo) hi = lo;
}

// rows of the level below (half the resolution, R_below rows) that rows [lo, hi) of a level read: output row Y takes rows Y - 1 .. Y + 1 of
// the nearest-x2 upsampling = rows (Y - 1) >> 1 .. (Y + 1) >> 1 below (outside the tensor: zero padding)
SBBSEG_HD inline void region_down(int& lo, int& hi, int R_below)
{
    if (hi <= lo) { lo = hi = 0; return; }
    lo = lo > 0 ? (lo - 1) >> 1 : 0;
    hi = (hi >> 1) + 1;
    if (hi > R_below) hi = R_below;
}

// rows [lo, hi) / columns of level `level` that patch (i, j) of the grid must produce (before the level's own rounding below)
SBBSEG_HD inline void region_needed(const RegionGeom& g, int i, int j, int level, int& ylo, int& yhi, int& xlo, int& xhi)
{
    region_own(g.ay, j, ylo, yhi);
    region_own(g.ax, i, xlo, xhi);
    if (yhi <= ylo || xhi <= xlo) { ylo = yhi = xlo = xhi = 0; return; }
    for (int k = 1; k <= level; ++k) {
        region_down(ylo, yhi, g.Rh[k]);
        region_down(xlo, xhi, g.Rw[k]);
    }
}

// kind 1: the range grows to an even length, so that the two parities of an axis have the same number of class rows
SBBSEG_HD inline void region_even(int& lo, int& hi, int R)
{
    if ((hi - lo) & 1) {
        if (hi < R) ++hi; else --lo;
    }
}
// kind 0: 16-pixel tiles from `lo` rounded down to `align`: their number, and the origin of tile k (the last one is pulled inside the tensor)
SBBSEG_HD inline int region_tiles16(int lo, int hi, int align) { return hi > lo ? (hi - (lo & ~(align - 1)) + 15) >> 4 : 0; }
SBBSEG_HD inline int region_tile_origin(int lo, int align, int k, int R)
{
    const int o = (lo & ~(align - 1)) + 16 * k;
    return o + 16 > R ? R - 16 : o;
}

// table entries of patch (i, j) at `level`: kind 0 = tiles, kind 1 = pixels PER CLASS
SBBSEG_HD inline int region_entries(const RegionGeom& g, int i, int j, int level)
{
    int ylo, yhi, xlo, xhi;
    region_needed(g, i, j, level, ylo, yhi, xlo, xhi);
    if (yhi <= ylo) return 0;
    if (g.kind[level] == 0) return region_tiles16(ylo, yhi, 2) * region_tiles16(xlo, xhi, g.align_x[level]);
    region_even(ylo, yhi, g.Rh[level]);
    region_even(xlo, xhi, g.Rw[level]);
    return ((yhi - ylo) >> 1) * ((xhi - xlo) >> 1);
}

// table entry: patch (10 bits) | y (11 bits) | x (11 bits).  kind 0: tile origin / 2; kind 1: class-grid pixel
SBBSEG_HD inline uint32_t region_code(int n, int y, int x) { return ((uint32_t)n << 22) | ((uint32_t)y << 11) | (uint32_t)x; }
SBBSEG_HD inline int region_code_n(uint32_t c) { return (int)(c >> 22); }
SBBSEG_HD inline int region_code_y(uint32_t c) { return (int)((c >> 11) & 2047u); }
SBBSEG_HD inline int region_code_x(uint32_t c) { return (int)(c & 2047u); }
constexpr int kRegionMaxPatches = 1023, kRegionMaxCoord = 2047;

// device tables of one launch chunk: patch p of the chunk is tile g0 + p of the pooled tile list (page-major, grid index = tile % tpp)
struct RegionBuildParams {
    RegionGeom g;
    int g0, nb;
    uint32_t* out[kRegionMaxLevels];      // kind 0: [entries]; kind 1: [4 classes][entries] (class q = parity (q >> 1, q & 1))
    int total[kRegionMaxLevels];          // entries (per class) of the whole chunk, as the host counted them
};
hipError_t launch_region_build(const RegionBuildParams& p, hipStream_t s);

}  // namespace sbbseg
