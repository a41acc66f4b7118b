// Issue-rate probes for gfx950 (development aid): how many SALU / VALU / mixed instructions one CU retires per cycle as a
// function of the number of resident waves.  Build: hipcc --offload-arch=gfx950 -O3 -o issue_rates issue_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

template <int KIND>
__global__ void probe(int iters, float *out, long long *cyc)
{
    float a = threadIdx.x, b = 1.5f, c = 2.5f, d = 3.5f;
    int s0 = iters, s1 = 1, s2 = 2, s3 = 3;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == 0) {          // 64 independent-ish SALU (4 chains)
            asm volatile(REP16("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n")
                         : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        } else if constexpr (KIND == 1) {   // 64 VALU (4 chains)
            asm volatile(REP16("v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n v_add_f32 %3, %3, %3\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if constexpr (KIND == 2) {   // 32 VALU + 32 SALU interleaved
            asm volatile(REP16("v_add_f32 %0, %0, %0\n s_add_u32 %4, %4, 1\n v_add_f32 %1, %1, %1\n s_add_u32 %5, %5, 1\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+s"(s0), "+s"(s1) : : "scc");
        } else if constexpr (KIND == 3) {   // 64 v_cndmask with sgpr mask
            asm volatile(REP16("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");
        } else if constexpr (KIND == 4) {   // 64 v_pk_mul_f32
            asm volatile(REP16("v_pk_mul_f32 %0, %0, %0\n v_pk_mul_f32 %1, %1, %1\n v_pk_mul_f32 %0, %0, %0\n v_pk_mul_f32 %1, %1, %1\n")
                         : "+v"(*(double *)&a), "+v"(*(double *)&c));
        } else if constexpr (KIND == 5) {   // 64 v_readlane/writelane pairs (spill traffic)
            asm volatile(REP16("v_readlane_b32 %2, %0, 3\n v_writelane_b32 %1, %2, 5\n v_readlane_b32 %3, %0, 7\n v_writelane_b32 %1, %3, 9\n")
                         : "+v"(a), "+v"(b), "+s"(s0), "+s"(s1));
        } else if constexpr (KIND == 6) {   // 64 DPP moves
            asm volatile(REP16("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                               "v_mov_b32_dpp %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if constexpr (KIND == 7) {   // 16 barriers
            asm volatile(REP16("s_barrier\n"));
        } else if constexpr (KIND == 8) {   // v_cmp to sgpr pair + s_and (band-test shape): 32 + 32
            asm volatile(REP16("v_cmp_lt_f32 vcc, %0, %1\n s_and_b64 %4, vcc, exec\n v_cmp_lt_f32 vcc, %2, %3\n s_and_b64 %5, vcc, exec\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+s"(*(long long *)&s0), "+s"(*(long long *)&s2) : : "vcc", "scc");
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + (float)(s0 + s1 + s2 + s3);
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND>
void run(const char *name, int per_iter)
{
    float *out; long long *cyc;
    hipMalloc(&out, 4 << 20); hipMalloc(&cyc, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 5000;
    for (int waves : {1, 2, 4, 8, 16}) {
        // one workgroup per CU (256 CUs), `waves` waves in it
        probe<KIND><<<256, 64 * waves>>>(10, out, cyc);
        hipEventRecord(e0);
        probe<KIND><<<256, 64 * waves>>>(iters, out, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double instr = (double)iters * per_iter;
        printf("%-28s waves/CU %2d: %.3f ms  -> %.2f ns per wave-instruction-stream instr, %.2f instr/ns/CU, memtime ticks/instr %.3f\n", name, waves, ms,
               ms * 1e6 / instr, instr * waves / (ms * 1e6), (double)c / instr);
    }
    hipFree(out); hipFree(cyc);
}

int main()
{
    setvbuf(stdout, nullptr, _IOLBF, 0);
    printf("start\n");
    run<0>("salu s_add x64", 64);
    run<1>("valu v_add_f32 x64", 64);
    run<2>("valu+salu interleaved x64", 64);
    run<3>("v_cndmask vcc x64", 64);
    run<4>("v_pk_mul_f32 x64", 64);
    run<5>("readlane+writelane x64", 64);
    run<6>("v_mov dpp wave_shr x64", 64);
    run<7>("s_barrier x16", 16);
    run<8>("v_cmp+s_and x64", 64);
    return 0;
}
