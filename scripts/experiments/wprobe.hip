// write-throughput probe: `grid` workgroups of `nth` threads; each workgroup writes `bytes_per_wg` contiguous bytes, 16 B per lane per store
#include <hip/hip_runtime.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ void wprobe_kernel(char* out, long long bytes_per_wg, int v) {
    char* base = out + (long long)blockIdx.x * bytes_per_wg;
    i32x4 x{v, v + 1, v + 2, v + 3};
    for (long long off = (long long)threadIdx.x * 16; off < bytes_per_wg; off += (long long)blockDim.x * 16) *(i32x4*)(base + off) = x;
}
extern "C" int wprobe(void* out, int grid, int nth, long long bytes_per_wg, void* stream) {
    hipLaunchKernelGGL(wprobe_kernel, dim3(grid), dim3(nth), 0, (hipStream_t)stream, (char*)out, bytes_per_wg, 1);
    return (int)hipGetLastError();
}
