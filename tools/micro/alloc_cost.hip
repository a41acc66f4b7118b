// How long do hipMalloc / hipFree / a device-to-device copy of multi-GB buffers take?  (development probe for the row-plane growth policy)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    hipFree(0);
    for (double gb : {0.5, 2.0, 4.0, 8.0, 16.0}) {
        const size_t n = (size_t)(gb * (1 << 30));
        void *a = nullptr, *b = nullptr;
        double t0 = now();
        if (hipMalloc(&a, n) != hipSuccess) { printf("%.1f GB: hipMalloc failed\n", gb); continue; }
        double t1 = now();
        hipMalloc(&b, n);
        double t2 = now();
        hipMemcpy(b, a, n, hipMemcpyDeviceToDevice); hipDeviceSynchronize();
        double t3 = now();
        hipMemcpy(b, a, n, hipMemcpyDeviceToDevice); hipDeviceSynchronize();
        double t4 = now();
        hipFree(a);
        double t5 = now();
        hipFree(b);
        double t6 = now();
        printf("%.1f GB: malloc %.1f / %.1f ms, first copy %.1f ms, second copy %.1f ms, free %.1f / %.1f ms\n", gb, t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5);
    }
    return 0;
}
