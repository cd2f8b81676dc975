// What does `buffer_load_dwordx4 ... offen lds` do with out-of-range lanes on gfx950?
// (the fast gather of conv_igemm_mfma relies on: OOB lane -> 16 zero bytes land in LDS)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/buffer_lds_oob_probe.hip -o /tmp/oob && /tmp/oob
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void k(const char* p, uint32_t nrec, const uint32_t* offs, uint32_t soff, uint32_t bias, uint32_t* out)
{
    __shared__ __attribute__((aligned(16))) uint32_t smem[64 * 4];
    for (int i = 0; i < 4; ++i) smem[threadIdx.x * 4 + i] = 0xABABABABu;
    __syncthreads();
    // SRD base sits `bias` bytes BEFORE the buffer; soffset carries +bias
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(p - bias), 0, nrec + bias, 0x00020000);
    const uint32_t vo = offs[threadIdx.x];
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)smem, 16, vo, soff + bias, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < 4; ++i) out[threadIdx.x * 4 + i] = smem[threadIdx.x * 4 + i];
}

int main()
{
    const uint32_t n = 1 << 20;
    std::vector<uint32_t> h(n / 4);
    for (uint32_t i = 0; i < n / 4; ++i) h[i] = i;           // word i holds i
    char* d; hipMalloc(&d, n); hipMemcpy(d, h.data(), n, hipMemcpyHostToDevice);
    std::vector<uint32_t> offs(64);
    for (int l = 0; l < 64; ++l) {
        uint32_t o = 4096u + l * 64u;
        if (l % 4 == 1) o |= 0x80000000u;                     // far out of range
        if (l % 4 == 2) o = n - 8;                            // straddles the end (partial)
        if (l % 4 == 3) o = n + 1024;                         // just past the end
        offs[l] = o;
    }
    uint32_t *doffs, *dout; hipMalloc(&doffs, 256); hipMalloc(&dout, 1024);
    hipMemcpy(doffs, offs.data(), 256, hipMemcpyHostToDevice);
    for (uint32_t bias : {0u, 8192u}) {
        for (uint32_t soff : {0u, 256u}) {
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, n, doffs, soff, bias, dout);
            std::vector<uint32_t> o(256);
            hipMemcpy(o.data(), dout, 1024, hipMemcpyDeviceToHost);
            printf("bias %u soffset %u\n", bias, soff);
            int bad = 0;
            for (int l = 0; l < 8; ++l) {
                const uint32_t want0 = (offs[l] + soff) / 4;
                printf("  lane %d voff 0x%08x -> %08x %08x %08x %08x (in-range word would be %08x)\n", l, offs[l], o[l * 4], o[l * 4 + 1], o[l * 4 + 2], o[l * 4 + 3], want0);
            }
            for (int l = 0; l < 64; ++l) {
                if (l % 4 == 0) { for (int i = 0; i < 4; ++i) bad += o[l * 4 + i] != (offs[l] + soff) / 4 + i; }
                if (l % 4 == 1 || l % 4 == 3) { for (int i = 0; i < 4; ++i) bad += o[l * 4 + i] != 0; }
            }
            printf("  %s\n", bad ? "MISMATCH" : "ok: valid lanes exact, out-of-range lanes zero-filled");
        }
    }
    return 0;
}
