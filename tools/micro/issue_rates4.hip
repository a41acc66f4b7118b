// Issue-rate probes, part 4 (round 6): VALU throughput of a CU as a function of the DEPENDENCY DISTANCE inside a wave (how many independent chains its
// instruction stream interleaves) and of the number of resident waves.  issue_rates3 showed 2.0 instr/ns/CU for streams whose instructions depend on the
// one three places back, 3.9 for four fully independent chains.  Build: hipcc --offload-arch=gfx950 -O3 -w -o issue_rates4 issue_rates4.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

template <int CHAINS, int OP>
__global__ void probe(int iters, float *out)
{
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + 1.5f * i;
    for (int it = 0; it < iters; ++it) {
        if constexpr (OP == 0) {           // v_add_f32 x, x, x : 64 per iteration
            if constexpr (CHAINS == 1) asm volatile(REP16(REP4("v_add_f32 %0, %0, %0\n")) : "+v"(v[0]));
            if constexpr (CHAINS == 2) asm volatile(REP16("v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n") : "+v"(v[0]), "+v"(v[1]));
            if constexpr (CHAINS == 3) asm volatile(REP16("v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n v_add_f32 %0, %0, %0\n") REP16("v_add_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n") : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]));
            if constexpr (CHAINS == 4) asm volatile(REP16("v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n v_add_f32 %3, %3, %3\n") : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
            if constexpr (CHAINS == 8) asm volatile(REP4(REP4("v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n v_add_f32 %3, %3, %3\n") REP4("v_add_f32 %4, %4, %4\n v_add_f32 %5, %5, %5\n v_add_f32 %6, %6, %6\n v_add_f32 %7, %7, %7\n"))
                                                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
        } else {                           // v_cndmask_b32_e64 with an SGPR mask (the select of the DP step)
            long long m = 0x5555555555555555ll;
            if constexpr (CHAINS == 1) asm volatile(REP16(REP4("v_cndmask_b32_e64 %0, %0, %0, %1\n")) : "+v"(v[0]) : "s"(m));
            if constexpr (CHAINS == 2) asm volatile(REP16("v_cndmask_b32_e64 %0, %0, %0, %2\n v_cndmask_b32_e64 %1, %1, %1, %2\n v_cndmask_b32_e64 %0, %0, %0, %2\n v_cndmask_b32_e64 %1, %1, %1, %2\n") : "+v"(v[0]), "+v"(v[1]) : "s"(m));
            if constexpr (CHAINS == 4) asm volatile(REP16("v_cndmask_b32_e64 %0, %0, %0, %4\n v_cndmask_b32_e64 %1, %1, %1, %4\n v_cndmask_b32_e64 %2, %2, %2, %4\n v_cndmask_b32_e64 %3, %3, %3, %4\n") : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : "s"(m));
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CHAINS, int OP>
void run(const char *name)
{
    float *out;
    hipMalloc(&out, 8 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 3000;
    // wg x waves: workgroups per CU x waves per workgroup (256 CUs)
    const int cfg[][2] = {{1, 1}, {1, 4}, {1, 8}, {1, 16}, {4, 4}, {5, 4}, {2, 8}, {8, 4}};
    for (auto &c : cfg) {
        const int wg = c[0], waves = c[1];
        probe<CHAINS, OP><<<256 * wg, 64 * waves>>>(10, out);
        hipEventRecord(e0);
        probe<CHAINS, OP><<<256 * wg, 64 * waves>>>(iters, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double instr = (double)iters * 64;
        printf("%-34s chains %d  %d wg x %2d waves per CU: %.2f ns per instr of a wave, %.2f instr/ns/CU\n", name, CHAINS, wg, waves, ms * 1e6 / instr, instr * waves * wg / (ms * 1e6));
    }
    hipFree(out);
}

int main()
{
    setvbuf(stdout, nullptr, _IOLBF, 0);
    run<1, 0>("v_add_f32"); run<2, 0>("v_add_f32"); run<3, 0>("v_add_f32"); run<4, 0>("v_add_f32"); run<8, 0>("v_add_f32");
    run<1, 1>("v_cndmask e64"); run<2, 1>("v_cndmask e64"); run<4, 1>("v_cndmask e64");
    return 0;
}
