// Second round of gfx950 issue-rate probes (development aid): select idioms, SGPR-operand VALU, LDS single-lane ops.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

template <int KIND>
__global__ void probe(int iters, float *out)
{
    __shared__ float4 lds[1024];
    float a = threadIdx.x, b = 1.5f, c = 2.5f, d = 3.5f, e = 0.f, f = 0.f, g = 0.f, h = 0.f;
    int s0 = iters, s1 = 1, s2 = 2, s3 = 3;
    long long m0 = 0x5555555555555555ll, m1 = 0x3333333333333333ll;
    unsigned addr = (threadIdx.x & 63) * 16;
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == 0) {          // independent v_cndmask vcc (sources fixed)
            asm volatile(REP16("v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %4, %5, vcc\n v_cndmask_b32 %2, %4, %5, vcc\n v_cndmask_b32 %3, %4, %5, vcc\n")
                         : "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(a), "v"(b) : "vcc");
        } else if constexpr (KIND == 1) {   // v_cndmask e64 with sgpr pair
            asm volatile(REP16("v_cndmask_b32_e64 %0, %4, %5, %6\n v_cndmask_b32_e64 %1, %4, %5, %7\n v_cndmask_b32_e64 %2, %4, %5, %6\n v_cndmask_b32_e64 %3, %4, %5, %7\n")
                         : "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(a), "v"(b), "s"(m0), "s"(m1));
        } else if constexpr (KIND == 2) {   // v_bfi with vgpr mask
            asm volatile(REP16("v_bfi_b32 %0, %6, %4, %5\n v_bfi_b32 %1, %6, %4, %5\n v_bfi_b32 %2, %6, %4, %5\n v_bfi_b32 %3, %6, %4, %5\n")
                         : "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(a), "v"(b), "v"(c));
        } else if constexpr (KIND == 3) {   // v_max_f32
            asm volatile(REP16("v_max_f32 %0, %4, %5\n v_max_f32 %1, %4, %5\n v_max_f32 %2, %4, %5\n v_max_f32 %3, %4, %5\n")
                         : "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(a), "v"(b));
        } else if constexpr (KIND == 4) {   // v_cmp VOPC -> vcc
            asm volatile(REP16("v_cmp_gt_f32 vcc, %0, %1\n v_cmp_gt_f32 vcc, %1, %2\n v_cmp_gt_f32 vcc, %2, %3\n v_cmp_gt_f32 vcc, %3, %0\n")
                         : : "v"(a), "v"(b), "v"(c), "v"(d) : "vcc");
        } else if constexpr (KIND == 5) {   // v_cmp e64 -> sgpr pair
            asm volatile(REP16("v_cmp_gt_f32_e64 %0, %2, %3\n v_cmp_gt_f32_e64 %1, %3, %4\n v_cmp_gt_f32_e64 %0, %4, %5\n v_cmp_gt_f32_e64 %1, %5, %2\n")
                         : "+s"(m0), "+s"(m1) : "v"(a), "v"(b), "v"(c), "v"(d));
        } else if constexpr (KIND == 6) {   // VALU with an SGPR source operand
            asm volatile(REP16("v_add_f32 %0, %4, %0\n v_add_f32 %1, %5, %1\n v_add_f32 %2, %4, %2\n v_add_f32 %3, %5, %3\n")
                         : "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "s"(s1), "s"(s2));
        } else if constexpr (KIND == 7) {   // v_cmp -> vcc then dependent v_cndmask (the usual pair), 4 independent pairs
            asm volatile(REP16("v_cmp_gt_f32 vcc, %4, %5\n v_cndmask_b32 %0, %4, %5, vcc\n v_cmp_gt_f32 vcc, %5, %4\n v_cndmask_b32 %1, %4, %5, vcc\n")
                         : "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(a), "v"(b) : "vcc");
        } else if constexpr (KIND == 8) {   // v_readfirstlane
            asm volatile(REP16("v_readfirstlane_b32 %0, %4\n v_readfirstlane_b32 %1, %5\n v_readfirstlane_b32 %2, %4\n v_readfirstlane_b32 %3, %5\n")
                         : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(a), "v"(b));
        } else if constexpr (KIND == 9) {   // ds_read_b128 all lanes, 4 in flight then wait
            float4 r0, r1, r2, r3;
            asm volatile(REP4("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072\n s_waitcnt lgkmcnt(0)\n")
                         : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(addr) : "memory");
            e += r0.x + r1.y + r2.z + r3.w;
        } else if constexpr (KIND == 10) {  // dependent ds_read_b32 chain (latency)
            unsigned p = addr;
            asm volatile(REP16("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n") : "+v"(p) : : "memory");
            e += (float)p;
        } else if constexpr (KIND == 11) {  // single-lane ds_max_u32 + s_waitcnt (atomic completion latency)
            if ((threadIdx.x & 63) == 63) {
                unsigned p = 0; unsigned v = it;
                asm volatile(REP16("ds_max_u32 %0, %1\n s_waitcnt lgkmcnt(0)\n") : : "v"(p), "v"(v) : "memory");
            }
        } else if constexpr (KIND == 12) {  // v_med3 / v_min3 style 3-operand
            asm volatile(REP16("v_med3_f32 %0, %4, %5, %6\n v_med3_f32 %1, %4, %5, %6\n v_max3_f32 %2, %4, %5, %6\n v_max3_f32 %3, %4, %5, %6\n")
                         : "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(a), "v"(b), "v"(c));
        } else if constexpr (KIND == 13) {  // v_mov dpp row_shr:1
            asm volatile(REP16("v_mov_b32_dpp %0, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                               "v_mov_b32_dpp %2, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n")
                         : "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(a), "v"(b));
        } else if constexpr (KIND == 14) {  // v_add with dpp wave_shr fused
            asm volatile(REP16("v_add_f32_dpp %0, %4, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %5, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                               "v_add_f32_dpp %2, %4, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %5, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n")
                         : "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(a), "v"(b));
        } else if constexpr (KIND == 15) {  // s_and_saveexec + restore pairs (exec juggling)
            asm volatile(REP16("s_and_saveexec_b64 %0, %1\n s_mov_b64 exec, %0\n s_and_saveexec_b64 %0, %1\n s_mov_b64 exec, %0\n")
                         : "+s"(m0) : "s"(m1) : "scc");
        } else if constexpr (KIND == 16) {  // v_cndmask with vcc written once per 16 (is it the vcc READ that costs?)
            asm volatile("v_cmp_gt_f32 vcc, %4, %5\n" REP16("v_cndmask_b32 %0, %4, %5, vcc\n v_add_f32 %1, %4, %5\n v_cndmask_b32 %2, %4, %5, vcc\n v_add_f32 %3, %4, %5\n")
                         : "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(a), "v"(b) : "vcc");
        }
    }
    lds[threadIdx.x & 1023] = make_float4(e, f, g, h);
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h + (float)(s0 + s1 + s2 + s3) + (float)(m0 + m1) + lds[(threadIdx.x * 7) & 1023].x;
}

template <int KIND>
void run(const char *name, int per_iter)
{
    float *out;
    (void)hipMalloc(&out, 4 << 20);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4000;
    for (int waves : {1, 4, 8, 16}) {
        probe<KIND><<<256, 64 * waves>>>(10, out);
        (void)hipEventRecord(e0);
        probe<KIND><<<256, 64 * waves>>>(iters, out);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double instr = (double)iters * per_iter;
        printf("%-34s waves/CU %2d: %.3f ms -> %.2f ns/instr/wave, %.2f instr/ns/CU\n", name, waves, ms, ms * 1e6 / instr, instr * waves / (ms * 1e6));
    }
    (void)hipFree(out);
}

int main()
{
    setvbuf(stdout, nullptr, _IOLBF, 0);
    run<0>("v_cndmask vcc independent", 64);
    run<1>("v_cndmask e64 sgpr-pair", 64);
    run<2>("v_bfi vgpr mask", 64);
    run<3>("v_max_f32", 64);
    run<4>("v_cmp VOPC->vcc", 64);
    run<5>("v_cmp e64->sgpr pair", 64);
    run<6>("v_add_f32 sgpr operand", 64);
    run<7>("v_cmp+v_cndmask pairs", 64);
    run<8>("v_readfirstlane", 64);
    run<9>("ds_read_b128 x4 + wait", 16);
    run<10>("ds_read_b32 dependent", 16);
    run<11>("ds_max_u32 1 lane + wait", 16);
    run<12>("v_med3/v_max3", 64);
    run<13>("v_mov dpp row_shr", 64);
    run<14>("v_add_f32 dpp wave_shr fused", 64);
    run<15>("saveexec+restore", 64);
    run<16>("v_cndmask/v_add alternating", 64);
    return 0;
}
