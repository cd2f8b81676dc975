// region.hip -- builds the per-chunk tables of the owned-region decoder launches (region.h) on the device: one block per (patch, level).
// Integer work on a few hundred kilobytes per chunk; it runs on the lane's stream in front of the chunk's forward pass.
#include "region.h"

namespace sbbseg {

namespace {

__global__ __launch_bounds__(256) void region_build_kernel(const RegionBuildParams p)
{
    __shared__ int red[256];
    const int patch = blockIdx.x, level = blockIdx.y, tid = threadIdx.x;
    const RegionGeom& g = p.g;
    auto grid_ij = [&](int q, int& i, int& j) {
        const int local = (p.g0 + q) % g.tpp;
        i = local / g.ny;
        j = local - i * g.ny;
    };
    // entries of the patches in front of this one (a chunk has a few hundred patches: every block recounts them)
    int part = 0;
    for (int q = tid; q < patch; q += 256) {
        int i, j;
        grid_ij(q, i, j);
        part += region_entries(g, i, j, level);
    }
    red[tid] = part;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    const int base = red[0];
    int i, j, ylo, yhi, xlo, xhi;
    grid_ij(patch, i, j);
    region_needed(g, i, j, level, ylo, yhi, xlo, xhi);
    if (yhi <= ylo) return;
    uint32_t* out = p.out[level];
    if (g.kind[level] == 0) {
        const int ax = g.align_x[level];
        const int ty = region_tiles16(ylo, yhi, 2), tx = region_tiles16(xlo, xhi, ax);
        for (int e = tid; e < ty * tx; e += 256) {
            const int ky = e / tx, kx = e - ky * tx;
            out[base + e] = region_code(patch, region_tile_origin(ylo, 2, ky, g.Rh[level]) >> 1, region_tile_origin(xlo, ax, kx, g.Rw[level]) >> 1);
        }
    } else {
        region_even(ylo, yhi, g.Rh[level]);
        region_even(xlo, xhi, g.Rw[level]);
        const int ch = (yhi - ylo) >> 1, cw = (xhi - xlo) >> 1;
        const int total = p.total[level];
        for (int e = tid; e < 4 * ch * cw; e += 256) {
            const int q = e / (ch * cw), r = e - q * (ch * cw);
            const int py = q >> 1, px = q & 1;
            const int y = ((ylo + 1 - py) >> 1) + r / cw, x = ((xlo + 1 - px) >> 1) + r % cw;
            out[(size_t)q * total + base + r] = region_code(patch, y, x);
        }
    }
}

}  // namespace

hipError_t launch_region_build(const RegionBuildParams& p, hipStream_t s)
{
    if (p.nb <= 0 || p.g.n_levels <= 0) return hipSuccess;
    hipLaunchKernelGGL(region_build_kernel, dim3(p.nb, p.g.n_levels), dim3(256), 0, s, p);
    return hipGetLastError();
}

}  // namespace sbbseg
