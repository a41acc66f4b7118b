// Issue-rate probes, part 5 (round 6): the cost of a VALU instruction by its OPERANDS and ENCODING on gfx950, at the occupancy of the throughput kernels
// (5 workgroups x 4 waves per CU) and at 16 waves.  issue_rates4: v_add_f32 x, x, x 4.0 per ns and CU, v_cndmask_b32_e64 2.1-2.3.  Which is it that halves the
// rate -- two distinct vector sources, or the 64-bit encoding?  Build: hipcc --offload-arch=gfx950 -O3 -w -o issue_rates5 issue_rates5.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

template <int KIND>
__global__ void probe(int iters, float *out)
{
    float a = threadIdx.x, b = 1.5f, c = 2.5f, d = 3.5f, e = 4.5f, f = 5.5f, g = 6.5f, h = 7.5f;
    float s = 1.0000001f;
    long long m = 0x5555555555555555ll;
    asm volatile("" : "+s"(s), "+s"(m));
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == 0)  asm volatile(REP16("v_mul_f32_e32 %0, %4, %0\n v_mul_f32_e32 %1, %4, %1\n v_mul_f32_e32 %2, %4, %2\n v_mul_f32_e32 %3, %4, %3\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(s));                     // sgpr x vgpr, e32
        if constexpr (KIND == 1)  asm volatile(REP16("v_mul_f32_e32 %0, %0, %4\n v_mul_f32_e32 %1, %1, %5\n v_mul_f32_e32 %2, %2, %6\n v_mul_f32_e32 %3, %3, %7\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "v"(g), "v"(h));    // two distinct vgprs, e32
        if constexpr (KIND == 2)  asm volatile(REP16("v_mul_f32_e32 %0, 2.0, %0\n v_mul_f32_e32 %1, 2.0, %1\n v_mul_f32_e32 %2, 2.0, %2\n v_mul_f32_e32 %3, 2.0, %3\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));                              // inline constant x vgpr
        if constexpr (KIND == 3)  asm volatile(REP16("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %5, %6\n v_fma_f32 %2, %2, %6, %7\n v_fma_f32 %3, %3, %7, %4\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "v"(g), "v"(h));           // three vgprs, VOP3
        if constexpr (KIND == 4)  asm volatile(REP16("v_cndmask_b32_e64 %0, 0, %0, %4\n v_cndmask_b32_e64 %1, 0, %1, %4\n v_cndmask_b32_e64 %2, 0, %2, %4\n v_cndmask_b32_e64 %3, 0, %3, %4\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(m));   // VOP3, one vgpr source
        if constexpr (KIND == 5)  asm volatile(REP16("v_mov_b32_e32 %0, %4\n v_mov_b32_e32 %1, %5\n v_mov_b32_e32 %2, %6\n v_mov_b32_e32 %3, %7\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "v"(g), "v"(h));                         // move
        if constexpr (KIND == 6)  asm volatile(REP16("v_cmp_lt_f32_e32 vcc, %0, %1\n v_cmp_lt_f32_e32 vcc, %1, %2\n v_cmp_lt_f32_e32 vcc, %2, %3\n v_cmp_lt_f32_e32 vcc, %3, %0\n") : : "v"(a), "v"(b), "v"(c), "v"(d) : "vcc");                  // compare, two vgprs
        if constexpr (KIND == 7)  asm volatile(REP16("v_cmp_lt_f32_e32 vcc, %4, %0\n v_cmp_lt_f32_e32 vcc, %4, %1\n v_cmp_lt_f32_e32 vcc, %4, %2\n v_cmp_lt_f32_e32 vcc, %4, %3\n") : : "v"(a), "v"(b), "v"(c), "v"(d), "s"(s) : "vcc");          // compare, sgpr and vgpr
        if constexpr (KIND == 8)  asm volatile(REP16("v_add_f32_e64 %0, %0, %4\n v_add_f32_e64 %1, %1, %5\n v_add_f32_e64 %2, %2, %6\n v_add_f32_e64 %3, %3, %7\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "v"(g), "v"(h));    // the same add in the 64-bit encoding
        if constexpr (KIND == 9)  asm volatile(REP16("v_add_f32_e64 %0, %0, %0\n v_add_f32_e64 %1, %1, %1\n v_add_f32_e64 %2, %2, %2\n v_add_f32_e64 %3, %3, %3\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));                                      // x + x in the 64-bit encoding
        if constexpr (KIND == 10) asm volatile(REP16("v_add_f32_e32 %0, %0, %0\n v_add_f32_e32 %1, %1, %1\n v_add_f32_e32 %2, %2, %2\n v_add_f32_e32 %3, %3, %3\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));                                      // x + x, e32 (reference: 4.0)
        if constexpr (KIND == 11) asm volatile(REP16("v_pk_mul_f32 %0, %0, %2\n v_pk_mul_f32 %1, %1, %3\n v_pk_mul_f32 %0, %0, %3\n v_pk_mul_f32 %1, %1, %2\n") : "+v"(*(double *)&a), "+v"(*(double *)&c) : "v"(*(double *)&e), "v"(*(double *)&g));   // packed, two pairs
        if constexpr (KIND == 12) asm volatile(REP16("v_mov_b32_dpp %0, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %6 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %7 wave_shr:1 row_mask:0xf bank_mask:0xf\n")
                                               : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "v"(g), "v"(h));                                                                                                                   // DPP move
        if constexpr (KIND == 13) asm volatile(REP16("v_max_i32_e32 %0, %0, %4\n v_max_i32_e32 %1, %1, %5\n v_max_i32_e32 %2, %2, %6\n v_max_i32_e32 %3, %3, %7\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "v"(g), "v"(h));     // integer, two vgprs
        if constexpr (KIND == 14) asm volatile(REP16("v_add_u32_e32 %0, %4, %0\n v_add_u32_e32 %1, %4, %1\n v_add_u32_e32 %2, %4, %2\n v_add_u32_e32 %3, %4, %3\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(s));                       // integer, sgpr and vgpr
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h;
}

template <int KIND>
void run(const char *name)
{
    float *out;
    hipMalloc(&out, 8 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 3000;
    const int cfg[][2] = {{1, 16}, {5, 4}};
    for (auto &c : cfg) {
        const int wg = c[0], waves = c[1];
        probe<KIND><<<256 * wg, 64 * waves>>>(10, out);
        hipEventRecord(e0);
        probe<KIND><<<256 * wg, 64 * waves>>>(iters, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double instr = (double)iters * 64;
        printf("%-46s %d wg x %2d waves per CU: %.2f instr/ns/CU\n", name, wg, waves, instr * waves * wg / (ms * 1e6));
    }
    hipFree(out);
}

int main()
{
    setvbuf(stdout, nullptr, _IOLBF, 0);
    run<10>("v_add_f32 e32  x, x, x");
    run<9>("v_add_f32 e64  x, x, x");
    run<1>("v_mul_f32 e32  x, x, y (two vgprs)");
    run<8>("v_add_f32 e64  x, x, y (two vgprs)");
    run<0>("v_mul_f32 e32  x, s, x (sgpr, vgpr)");
    run<2>("v_mul_f32 e32  x, 2.0, x (constant, vgpr)");
    run<14>("v_add_u32 e32  x, s, x");
    run<13>("v_max_i32 e32  x, x, y");
    run<3>("v_fma_f32      x, x, y, z (VOP3, three vgprs)");
    run<4>("v_cndmask e64  x, 0, x, s[] (VOP3, one vgpr)");
    run<5>("v_mov_b32 e32  x, y");
    run<12>("v_mov_b32 dpp  x, y wave_shr:1");
    run<6>("v_cmp e32      vcc, x, y");
    run<7>("v_cmp e32      vcc, s, x");
    run<11>("v_pk_mul_f32   xy, xy, zw");
    return 0;
}
