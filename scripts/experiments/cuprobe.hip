// Per-CU streaming bandwidth probe: `grid` workgroups of 512 threads read two streams and write one (the LayerNorm-backward pattern: 2 reads + 1 write
// of n bytes each), VB bytes per lane per access (8 or 16), U accesses in flight per lane.  How many CUs does an HBM-bound kernel need?
#include <hip/hip_runtime.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
template <typename V, int U>
__global__ __launch_bounds__(512) void cuprobe_kernel(const V* a, const V* b, V* c, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride * U) {
        V x[U], y[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const long long j = i + u * stride < n ? i + u * stride : i; x[u] = a[j]; y[u] = b[j]; }
#pragma unroll
        for (int u = 0; u < U; ++u) if (i + u * stride < n) c[i + u * stride] = x[u] + y[u];
    }
}
extern "C" int cuprobe(const void* a, const void* b, void* c, long long bytes, int grid, int vb, int u, void* stream) {
    hipStream_t st = (hipStream_t)stream;
#define GO(V, U) hipLaunchKernelGGL((cuprobe_kernel<V, U>), dim3(grid), dim3(512), 0, st, (const V*)a, (const V*)b, (V*)c, bytes / (long long)sizeof(V))
    if (vb == 8) { if (u == 1) GO(i32x2, 1); else if (u == 4) GO(i32x2, 4); else GO(i32x2, 8); }
    else { if (u == 1) GO(i32x4, 1); else if (u == 4) GO(i32x4, 4); else GO(i32x4, 8); }
    return (int)hipGetLastError();
}
