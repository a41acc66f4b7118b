// How many 256-thread workgroups share a CU at a given static LDS size (hipOccupancyMaxActiveBlocksPerMultiprocessor): the allocation granule decides whether
// five workgroups of 32 KB fit the 160 KB of a gfx950 CU.   hipcc --offload-arch=gfx950 -O2 -o /tmp/lds_occ tools/micro/lds_occupancy.hip && /tmp/lds_occ
#include <hip/hip_runtime.h>
#include <cstdio>
template <int BYTES>
__global__ __launch_bounds__(256) void k(int *out)
{
    __shared__ int s[BYTES / 4];
    s[threadIdx.x] = threadIdx.x;
    __syncthreads();
    out[threadIdx.x] = s[(threadIdx.x * 7) % (BYTES / 4)];
}
template <int BYTES>
void probe()
{
    int n = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k<BYTES>, 256, 0);
    printf("%6d bytes: %d workgroups per CU\n", BYTES, n);
}
int main()
{
    probe<29696>(); probe<31744>(); probe<32000>(); probe<32256>(); probe<32512>(); probe<32736>(); probe<32768>(); probe<33024>(); probe<39936>(); probe<40960>(); probe<41216>();
    return 0;
}
