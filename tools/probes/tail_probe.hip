// tail_probe.hip -- times launch_tail (split mode) on random data, standalone: hipcc --offload-arch=gfx950 -O3 -std=c++17
//   -I sbb_textline_detection_amd/csrc tools/probes/tail_probe.hip -o tail_probe
// usage: tail_probe [patches 140] [precision 3 = f16x3, 2 = f16] [probs 0|1].  Timing only (random operands): parity is what tests/ check.
#include "kernels.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace sbbseg;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 140, PH = 224, PW = 224;
    const int prec = argc > 2 ? atoi(argv[2]) : kF16X3;
    const int planes = prec == kF16X3 ? 2 : 1;
    const size_t src_bytes = kZeroHeaderBytes + (size_t)n * PH * PW * 64 * 2 * planes;
    const size_t img_bytes = kZeroHeaderBytes + (size_t)n * 4 * PH * PW * 8 * 2 * planes;
    const size_t w_bytes = prec == kF16X3 ? (size_t)4 * 2 * kT3HalfSteps * 2 * 64 * 16 : (size_t)4 * kTailKSteps * 2 * 2 * 64 * 16;
    std::vector<uint16_t> h(1 << 20);
    for (auto& v : h) v = f32_to_f16_rne((float)(rand() % 2001 - 1000) * 1e-3f);
    char *src, *img, *w; float *cst; uint8_t* labels;
    CK(hipMalloc(&src, src_bytes)); CK(hipMalloc(&img, img_bytes)); CK(hipMalloc(&w, w_bytes)); CK(hipMalloc(&cst, 4096)); CK(hipMalloc(&labels, (size_t)n * 4 * PH * PW));
    for (size_t o = 0; o < src_bytes; o += h.size() * 2) CK(hipMemcpy(src + o, h.data(), std::min(h.size() * 2, src_bytes - o), hipMemcpyHostToDevice));
    for (size_t o = 0; o < img_bytes; o += h.size() * 2) CK(hipMemcpy(img + o, h.data(), std::min(h.size() * 2, img_bytes - o), hipMemcpyHostToDevice));
    CK(hipMemset(src, 0, kZeroHeaderBytes)); CK(hipMemset(img, 0, kZeroHeaderBytes));      // (the buffers start with a zero header)
    CK(hipMemcpy(w, h.data(), w_bytes, hipMemcpyHostToDevice));
    std::vector<float> c(1024, 0.5f);
    for (auto& v : c) v = (float)(rand() % 2001 - 1000) * 1e-3f;
    for (int i = 0; i < 32; ++i) c[i] = 0.5f + 0.01f * i;            // BN scale
    CK(hipMemcpy(cst, c.data(), 4096, hipMemcpyHostToDevice));
    TailParams tp;
    tp.src0 = src; tp.img = img; tp.PH = PH; tp.PW = PW; tp.n = n; tp.wfrag = w; tp.scale = cst; tp.shift = cst + 32; tp.classes = 2;
    tp.head_w = cst + 64; tp.head_scale = cst + 256; tp.head_shift = cst + 260; tp.labels = labels; tp.probs = nullptr;
    float* probs = nullptr;
    const bool want_probs = argc > 3 && atoi(argv[3]);
    if (want_probs) { CK(hipMalloc(&probs, (size_t)n * 4 * PH * PW * 2 * sizeof(float))); tp.probs = probs; }
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) CK(launch_tail(tp, prec, 256, 0));
    CK(hipDeviceSynchronize());
    const int reps = 10;
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; ++i) CK(launch_tail(tp, prec, 256, 0));
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    {   // run-to-run determinism inside this process: the labels of two more launches must be identical
        std::vector<uint8_t> l0((size_t)n * 4 * PH * PW), l1(l0.size());
        CK(hipMemset(labels, 9, l0.size())); CK(launch_tail(tp, prec, 256, 0)); CK(hipDeviceSynchronize());
        CK(hipMemcpy(l0.data(), labels, l0.size(), hipMemcpyDeviceToHost));
        CK(hipMemset(labels, 7, l0.size())); CK(launch_tail(tp, prec, 256, 0)); CK(hipDeviceSynchronize());
        CK(hipMemcpy(l1.data(), labels, l1.size(), hipMemcpyDeviceToHost));
        size_t diff = 0, unwritten = 0;
        for (size_t i = 0; i < l0.size(); ++i) { diff += l0[i] != l1[i]; unwritten += l1[i] == 7; }
        printf("determinism: %zu of %zu labels differ between two launches, %zu never written\n", diff, l0.size(), unwritten);
    }
    std::vector<uint8_t> l((size_t)n * 4 * PH * PW);
    CK(hipMemcpy(l.data(), labels, l.size(), hipMemcpyDeviceToHost));
    unsigned long sum = 0; unsigned long long hsh = 1469598103934665603ull;
    for (auto v : l) { sum += v; hsh = (hsh ^ v) * 1099511628211ull; }
    unsigned long long ph = 1469598103934665603ull;
    if (want_probs) {
        std::vector<uint32_t> pr((size_t)n * 4 * PH * PW * 2);
        CK(hipMemcpy(pr.data(), probs, pr.size() * 4, hipMemcpyDeviceToHost));
        for (auto v : pr) ph = (ph ^ v) * 1099511628211ull;
    }
    printf("prec %d n %d: %.4f ms per launch (label sum %lu hash %016llx probs hash %016llx)\n", prec, n, ms / reps, sum, hsh, ph);
    return 0;
}
