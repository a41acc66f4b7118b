// Measures pinned-allocation cost and H2D/D2H rates (development aid for the host staging design).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t B = 900ull << 20;
    double t = now(); hipFree(0); printf("runtime init %.1f ms\n", now() - t);
    void *d; t = now(); hipMalloc(&d, B); printf("hipMalloc %.1f ms\n", now() - t);
    void *h; t = now(); hipHostMalloc(&h, B, hipHostMallocDefault); printf("hipHostMalloc 900MB %.1f ms\n", now() - t);
    t = now(); memset(h, 1, B); printf("first memset pinned %.1f ms\n", now() - t);
    t = now(); memset(h, 2, B); printf("second memset pinned %.1f ms\n", now() - t);
    for (int r = 0; r < 2; ++r) { t = now(); hipMemcpy(d, h, B, hipMemcpyHostToDevice); printf("H2D pinned %.1f ms (%.1f GB/s)\n", now() - t, B / 1e6 / (now() - t)); }
    for (int r = 0; r < 2; ++r) { t = now(); hipMemcpy(h, d, B, hipMemcpyDeviceToHost); printf("D2H pinned %.1f ms (%.1f GB/s)\n", now() - t, B / 1e6 / (now() - t)); }
    t = now(); char *p = (char *)malloc(B); memset(p, 1, B); printf("malloc+first touch %.1f ms\n", now() - t);
    for (int r = 0; r < 2; ++r) { t = now(); hipMemcpy(d, p, B, hipMemcpyHostToDevice); printf("H2D pageable %.1f ms (%.1f GB/s)\n", now() - t, B / 1e6 / (now() - t)); }
    t = now(); hipHostRegister(p, B, hipHostRegisterDefault); printf("hipHostRegister %.1f ms\n", now() - t);
    t = now(); hipMemcpy(d, p, B, hipMemcpyHostToDevice); printf("H2D registered %.1f ms (%.1f GB/s)\n", now() - t, B / 1e6 / (now() - t));
    t = now(); free(p); printf("free %.1f ms\n", now() - t);
    t = now(); hipHostFree(h); printf("hipHostFree %.1f ms\n", now() - t);
    return 0;
}
