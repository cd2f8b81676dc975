// block_probe.hip -- times block_x3<false, false> (the split-mode identity bottleneck block) on random data, standalone:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I sbb_textline_detection_amd/csrc [-DSBBSEG_BLOCK_ABL=mask] tools/probes/block_probe.hip -o block_probe
// Timing only (random operands): parity is what tests/ check.  ABL bits (probe builds only): 1 = no phase-A epilogue, 2 = no phase-B
// epilogue, 4 = no phase-C epilogue arithmetic, 8 = no MFMAs, 16 = no x fetches after the first tile, 32 = no output stores
#include "block_x3.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace sbbseg;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
static uint16_t h16(float f) { _Float16 h = (_Float16)f; return __builtin_bit_cast(uint16_t, h); }
int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 140, H = 111, W = 111;
    const size_t xb = kZeroHeaderBytes + (size_t)n * H * W * 256 * 4;
    std::vector<uint16_t> h(1 << 20);
    for (auto& v : h) v = h16((float)(rand() % 2001 - 1000) * 1e-3f);
    char *x, *out, *w1, *w2, *w3; float* cst;
    CK(hipMalloc(&x, xb)); CK(hipMalloc(&out, xb)); CK(hipMalloc(&w1, 1 << 20)); CK(hipMalloc(&w2, 1 << 20)); CK(hipMalloc(&w3, 1 << 20)); CK(hipMalloc(&cst, 8192));
    for (size_t o = 0; o < xb; o += h.size() * 2) CK(hipMemcpy(x + o, h.data(), std::min(h.size() * 2, xb - o), hipMemcpyHostToDevice));
    CK(hipMemset(x, 0, kZeroHeaderBytes));
    CK(hipMemcpy(w1, h.data(), 1 << 20, hipMemcpyHostToDevice)); CK(hipMemcpy(w2, h.data() + 1000, 1 << 20, hipMemcpyHostToDevice)); CK(hipMemcpy(w3, h.data() + 2000, 1 << 20, hipMemcpyHostToDevice));
    std::vector<float> c(2048);
    for (auto& v : c) v = (float)(rand() % 2001 - 1000) * 1e-4f;
    CK(hipMemcpy(cst, c.data(), 8192, hipMemcpyHostToDevice));
    BlockParams bp;
    bp.x = x; bp.n = n; bp.H = H; bp.W = W; bp.proj = 0; bp.pq = 1; bp.w1 = w1; bp.w2 = w2; bp.w3 = w3;
    bp.s1 = cst; bp.b1 = cst + 64; bp.s2 = cst + 128; bp.b2 = cst + 192; bp.s3 = cst + 256; bp.b3 = cst + 512; bp.out = out + kZeroHeaderBytes;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) CK(launch_block_x3(bp, 256, 0));
    CK(hipDeviceSynchronize());
    const int reps = 10;
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; ++i) CK(launch_block_x3(bp, 256, 0));
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
#ifndef SBBSEG_BLOCK_ABL
#define SBBSEG_BLOCK_ABL 0
#endif
    printf("ABL %d n %d: %.4f ms per launch\n", SBBSEG_BLOCK_ABL, n, ms / reps);
    return 0;
}
