/* ORACLE (test infrastructure, not product code).
 *
 * Plain-C fp32 restatement of the one arithmetic primitive the reference's forward pass is made
 * of: Keras/TF `Conv2D` on NHWC tensors with HWIO kernels (cross-correlation, explicit zero
 * padding).  The reference itself holds no arithmetic -- `model.predict` (main.py:287-288,
 * 373-374) runs keras==2.3.* / tensorflow-gpu==1.15.* (requirements.txt:5,10), which are not
 * vendored and not installable here, so this follows their documented inference semantics
 * [EXT] and is cross-checked against torch-CPU (tests/test_oracle_forward.py).
 * PARITY UNPINNED against the real Keras/TF path: the reference ships no golden vectors.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * Built by oracle/Makefile:  gcc -O3 -march=x86-64-v3 -fopenmp -shared -fPIC
 */
#include <stddef.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAX_COUT 4096

/* y[n,oy,ox,co] = bias[co] + sum_{ky,kx,ci} x[n, oy*sy+ky-pt, ox*sx+kx-pl, ci] * w[ky,kx,ci,co]
 * Accumulation order: ky, kx, ci ascending, fp32 (fused multiply-add where the compiler emits it). */
int oracle_conv2d_nhwc(const float* x, int N, int H, int W, int Cin,
                       const float* w, int KH, int KW, int Cout, const float* bias,
                       int sy, int sx, int pt, int pl, int Ho, int Wo, float* y)
{
    if (Cout > MAX_COUT) return -1;
    long rows = (long)N * Ho;
#pragma omp parallel for schedule(dynamic, 1)
    for (long r = 0; r < rows; ++r) {
        int n = (int)(r / Ho), oy = (int)(r % Ho);
        float acc[MAX_COUT];
        for (int ox = 0; ox < Wo; ++ox) {
            if (bias) memcpy(acc, bias, sizeof(float) * Cout);
            else memset(acc, 0, sizeof(float) * Cout);
            for (int ky = 0; ky < KH; ++ky) {
                int iy = oy * sy + ky - pt;
                if (iy < 0 || iy >= H) continue;
                for (int kx = 0; kx < KW; ++kx) {
                    int ix = ox * sx + kx - pl;
                    if (ix < 0 || ix >= W) continue;
                    const float* xp = x + (((size_t)n * H + iy) * W + ix) * Cin;
                    const float* wp = w + ((size_t)(ky * KW + kx) * Cin) * Cout;
                    for (int ci = 0; ci < Cin; ++ci) {
                        float xv = xp[ci];
                        const float* wr = wp + (size_t)ci * Cout;
                        for (int co = 0; co < Cout; ++co) acc[co] += xv * wr[co];
                    }
                }
            }
            memcpy(y + (((size_t)n * Ho + oy) * Wo + ox) * Cout, acc, sizeof(float) * Cout);
        }
    }
    return 0;
}

int oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void oracle_set_num_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}
