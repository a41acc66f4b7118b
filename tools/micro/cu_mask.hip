// Does a CU-masked stream confine a kernel, and do two masked streams run side by side?  (development probe)
//   hipcc --offload-arch=gfx950 -O2 -o cu_mask cu_mask.hip && ./cu_mask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void where(unsigned *out, int spin)
{
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}

int main()
{
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("CUs %d\n", p.multiProcessorCount);
    const int words = (p.multiProcessorCount + 31) / 32;
    for (int variant = 0; variant < 3; ++variant) {
        std::vector<uint32_t> mask(words, 0);
        if (variant == 0) for (int i = 0; i < 64; ++i) mask[i / 32] |= 1u << (i % 32);              // first 64 bits
        if (variant == 1) for (int i = 64; i < p.multiProcessorCount; ++i) mask[i / 32] |= 1u << (i % 32);
        if (variant == 2) for (int i = 0; i < p.multiProcessorCount; i += 8) mask[i / 32] |= 1u << (i % 32);   // every 8th bit
        hipStream_t st; CK(hipExtStreamCreateWithCUMask(&st, words, mask.data()));
        const int grid = 512;
        unsigned *d; CK(hipMalloc(&d, grid * 8));
        hipLaunchKernelGGL(where, dim3(grid), dim3(1024), 0, st, d, 100000);   // 1 ms at 100 MHz
        CK(hipStreamSynchronize(st));
        std::vector<unsigned> h(2 * grid); CK(hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost));
        std::map<unsigned, int> perXcc; std::map<unsigned long long, int> perCu;
        for (int b = 0; b < grid; ++b) {
            const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
            const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
            perXcc[xcc]++; perCu[((unsigned long long)xcc << 16) | (se << 8) | (sh << 4) | cu]++;
        }
        printf("variant %d: distinct CUs used %zu; per XCC:", variant, perCu.size());
        for (auto &kv : perXcc) printf(" %u:%d", kv.first, kv.second);
        printf("\n");
        CK(hipFree(d)); CK(hipStreamDestroy(st));
    }
    // two masked streams at once: do they overlap in time?
    {
        std::vector<uint32_t> a(words, 0), b(words, 0);
        for (int i = 0; i < p.multiProcessorCount; ++i) ((i % 4 == 0) ? a : b)[i / 32] |= 1u << (i % 32);
        hipStream_t sa, sb; CK(hipExtStreamCreateWithCUMask(&sa, words, a.data())); CK(hipExtStreamCreateWithCUMask(&sb, words, b.data()));
        unsigned *d; CK(hipMalloc(&d, 4096 * 8));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int both = 0; both < 2; ++both) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, sa));
            hipLaunchKernelGGL(where, dim3(64), dim3(1024), 0, sa, d, 1000000);          // 10 ms, one WG per CU of the small partition
            if (both) hipLaunchKernelGGL(where, dim3(192 * 4), dim3(1024), 0, sb, d + 1024, 250000);   // 4 rounds x 2.5 ms on the large one
            CK(hipEventRecord(e1, sa));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("small-partition kernel %s: %.2f ms\n", both ? "with the large partition busy" : "alone", ms);
        }
    }
    return 0;
}
