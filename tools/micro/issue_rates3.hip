// Issue-rate probes, part 3 (round 6): what a SELECT costs on gfx950.  profiles/r02/issue_rates.log has v_cndmask_b32 ..., vcc at 0.42 instructions per ns and CU
// whatever the number of waves -- nine times slower than v_add_f32 and shared by the whole CU -- and the DP step is made of selects.  This probe measures the forms a
// select can take.  Build: hipcc --offload-arch=gfx950 -O3 -o issue_rates3 issue_rates3.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

template <int KIND>
__global__ void probe(int iters, float *out, long long *cyc)
{
    float a = threadIdx.x, b = 1.5f, c = 2.5f, d = 3.5f;
    unsigned m0 = threadIdx.x * 0x9E3779B9u, m1 = ~m0;
    long long s8 = 0x5555555555555555ll, s10 = 0x3333333333333333ll;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == 0) {          // v_cndmask e32 (implicit vcc), vcc never written
            asm volatile(REP16("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");
        } else if constexpr (KIND == 1) {   // v_cndmask e64, mask in an SGPR pair
            asm volatile(REP16("v_cndmask_b32_e64 %0, %0, %1, %4\n v_cndmask_b32_e64 %1, %1, %2, %5\n v_cndmask_b32_e64 %2, %2, %3, %4\n v_cndmask_b32_e64 %3, %3, %0, %5\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(s8), "s"(s10));
        } else if constexpr (KIND == 2) {   // the usual pair: v_cmp to vcc + v_cndmask on it (32 + 32)
            asm volatile(REP16("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_f32 vcc, %2, %3\n v_cndmask_b32 %0, %0, %1, vcc\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");
        } else if constexpr (KIND == 3) {   // select through EXEC: v_cmpx + v_mov under the new EXEC + EXEC restored (21 + 21 + 21 -> counted as 64)
            asm volatile(REP16("v_cmpx_lt_f32 %0, %1\n v_mov_b32 %2, %3\n s_mov_b64 exec, -1\n v_cmpx_lt_f32 %2, %3\n")
                         REP16("v_mov_b32 %0, %1\n s_mov_b64 exec, -1\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");
        } else if constexpr (KIND == 4) {   // bitwise select with the mask in a vector register
            asm volatile(REP16("v_bfi_b32 %0, %4, %0, %1\n v_bfi_b32 %1, %5, %1, %2\n v_bfi_b32 %2, %4, %2, %3\n v_bfi_b32 %3, %5, %3, %0\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m0), "v"(m1));
        } else if constexpr (KIND == 5) {   // v_max_f32
            asm volatile(REP16("v_max_f32 %0, %0, %1\n v_max_f32 %1, %1, %2\n v_max_f32 %2, %2, %3\n v_max_f32 %3, %3, %0\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if constexpr (KIND == 6) {   // v_cmp e32 to vcc only
            asm volatile(REP16("v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_f32 vcc, %3, %0\n")
                         : : "v"(a), "v"(b), "v"(c), "v"(d) : "vcc");
        } else if constexpr (KIND == 7) {   // v_cmp e64 to an SGPR pair
            asm volatile(REP16("v_cmp_lt_f32_e64 %4, %0, %1\n v_cmp_lt_f32_e64 %5, %1, %2\n v_cmp_lt_f32_e64 %4, %2, %3\n v_cmp_lt_f32_e64 %5, %3, %0\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+s"(s8), "+s"(s10));
        } else if constexpr (KIND == 8) {   // v_cndmask e32 with an inline constant as the false side
            asm volatile(REP16("v_cndmask_b32 %0, 0, %1, vcc\n v_cndmask_b32 %1, 0, %2, vcc\n v_cndmask_b32 %2, 0, %3, vcc\n v_cndmask_b32 %3, 0, %0, vcc\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");
        } else if constexpr (KIND == 9) {   // v_cndmask e32, independent destinations (no chain through the data)
            asm volatile(REP16("v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %4, %5, vcc\n v_cndmask_b32 %2, %4, %5, vcc\n v_cndmask_b32 %3, %4, %5, vcc\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m0), "v"(m1) : "vcc");
        } else if constexpr (KIND == 10) {  // v_cndmask through SDWA (a different encoding of the same operation)
            asm volatile(REP16("v_cndmask_b32_sdwa %0, %0, %1, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_cndmask_b32_sdwa %1, %1, %2, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n"
                               "v_cndmask_b32_sdwa %2, %2, %3, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_cndmask_b32_sdwa %3, %3, %0, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");
        } else if constexpr (KIND == 11) {  // v_add_co_u32 (writes vcc) + v_addc_co_u32 (reads and writes vcc)
            asm volatile(REP16("v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %1, vcc, %1, %2, vcc\n v_add_co_u32 %2, vcc, %2, %3\n v_addc_co_u32 %3, vcc, %3, %0, vcc\n")
                         : "+v"(m0), "+v"(m1), "+v"(a), "+v"(b) : : "vcc");
        } else if constexpr (KIND == 12) {  // v_med3_f32 (three-operand VOP3, no mask)
            asm volatile(REP16("v_med3_f32 %0, %0, %1, %2\n v_med3_f32 %1, %1, %2, %3\n v_med3_f32 %2, %2, %3, %0\n v_med3_f32 %3, %3, %0, %1\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if constexpr (KIND == 13) {  // v_cndmask e64 with the mask in an SGPR pair, independent destinations
            asm volatile(REP16("v_cndmask_b32_e64 %0, %4, %5, %6\n v_cndmask_b32_e64 %1, %4, %5, %7\n v_cndmask_b32_e64 %2, %4, %5, %6\n v_cndmask_b32_e64 %3, %4, %5, %7\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m0), "v"(m1), "s"(s8), "s"(s10));
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + (float)(m0 + m1) + (float)(s8 + s10);
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND>
void run(const char *name, int per_iter)
{
    float *out; long long *cyc;
    hipMalloc(&out, 4 << 20); hipMalloc(&cyc, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 3000;
    for (int waves : {1, 4, 8, 16}) {
        probe<KIND><<<256, 64 * waves>>>(10, out, cyc);
        hipEventRecord(e0);
        probe<KIND><<<256, 64 * waves>>>(iters, out, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double instr = (double)iters * per_iter;
        printf("%-44s waves/CU %2d: %.3f ms  -> %.2f ns per instr of a wave, %.2f instr/ns/CU, memtime ticks/instr %.3f\n", name, waves, ms,
               ms * 1e6 / instr, instr * waves / (ms * 1e6), (double)c / instr);
    }
    hipFree(out); hipFree(cyc);
}

int main()
{
    setvbuf(stdout, nullptr, _IOLBF, 0);
    run<0>("v_cndmask e32 vcc (chain)", 64);
    run<9>("v_cndmask e32 vcc (independent)", 64);
    run<8>("v_cndmask e32 vcc, 0 as false side", 64);
    run<1>("v_cndmask e64 sgpr mask (chain)", 64);
    run<13>("v_cndmask e64 sgpr mask (independent)", 64);
    run<10>("v_cndmask sdwa vcc", 64);
    run<2>("v_cmp e32 + v_cndmask e32", 64);
    run<3>("v_cmpx + v_mov + s_mov exec (x21 each)", 96);
    run<4>("v_bfi_b32 vgpr mask", 64);
    run<5>("v_max_f32", 64);
    run<12>("v_med3_f32", 64);
    run<6>("v_cmp e32 -> vcc", 64);
    run<7>("v_cmp e64 -> sgpr pair", 64);
    run<11>("v_add_co + v_addc_co", 64);
    return 0;
}
