// Store-pattern probe for the GEMM epilogue: one 512-thread workgroup per CU (128 KB LDS, the GEMM's residency) writes 256 x 256 bf16 tiles of a
// row-major [M][N] matrix, 16 B per lane per store, with the lane -> address maps the epilogue could use:
//   pattern 0: one instruction = 16 rows x 64 B  (register-direct epilogue today: lane (g,t) -> row t, 16-byte chunk g)
//   pattern 1: one instruction =  8 rows x 128 B (full cache lines)
//   pattern 2: one instruction =  4 rows x 256 B
//   hipcc --offload-arch=gfx950 -O3 -o ab/stprobe scripts/experiments/stprobe.hip && ab/stprobe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int i32x4 __attribute__((ext_vector_type(4)));
// PAT 3: pattern 1 + `s_waitcnt vmcnt(0); s_barrier` after every tile (what a persistent GEMM pays between two tiles);
// PAT 4: as 3, and 64 KB of LDS-DMA loads issued right behind the tile's stores (the next tile's first K-step): does the load wait behind the stores?
template <int PAT, bool NT>
__global__ __launch_bounds__(512, 2) void probe(char* C, int ldc_bytes, int tiles_n, int tiles_total, int passes, unsigned long long* out) {
    extern __shared__ char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 2, wn = wave & 3;            // 2 x 4 waves, wave tile 128 rows x 64 columns (128 B per row)
    i32x4 v{tid, 1, 2, 3};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int ps = 0; ps < passes; ++ps)
        for (int tile = blockIdx.x; tile < tiles_total; tile += gridDim.x) {
            const int bm = tile / tiles_n, bn = tile % tiles_n;
            char* base = C + ((size_t)bm * 256 + wm * 128) * ldc_bytes + (size_t)bn * 512 + wn * 128;
            // 128 rows x 128 B per wave = 16 instructions of 1 KB
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                int row, chunk;
                if (PAT == 0) { row = 16 * (s >> 1) + (lane & 15); chunk = (s & 1) * 4 + (lane >> 4); }
                else if (PAT == 1) { row = 8 * s + (lane >> 3); chunk = lane & 7; }
                else { row = 8 * s + (lane >> 3); chunk = lane & 7; }
                char* p = base + (size_t)row * ldc_bytes + chunk * 16;
                if (NT) __builtin_nontemporal_store(v, (i32x4*)p); else *(i32x4*)p = v;
            }
            if (PAT == 4) {
                const unsigned long long b = (unsigned long long)C;
                i32x4 r;
                r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b); r[1] = __builtin_amdgcn_readfirstlane((int)((b >> 32) & 0xffffu));
                r[2] = 0x7fffffff; r[3] = 0x00020000;
                const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (wave * 8 + j) * 1024);
                    const unsigned vo = ((unsigned)(blockIdx.x % 64) * 256u + (wave * 8 + j) * 8 + (lane >> 3)) * 6144u + (lane & 7) * 16;
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(vo), "s"(r), "s"(dst) : "memory");
                }
            }
            if (PAT >= 3) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = r1 - r0; }
    if (smem[tid] == 0x7f && out[0] == 7) out[1] = 2;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <int PAT, bool NT>
void run(const char* name, char* C, int M, int N, int grid, unsigned long long* out) {
    const int tiles_n = N / 256, tiles = (M / 256) * tiles_n, passes = 4;
    auto kern = probe<PAT, NT>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 131072, 0, C, N * 2, tiles_n, tiles, passes, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 131072, 0, C, N * 2, tiles_n, tiles, passes, out);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    static unsigned long long h[2048];
    CK(hipMemcpy(h, out, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost));
    double cyc = 0; for (int i = 0; i < grid; ++i) cyc += h[2 * i]; cyc /= grid;
    const double bytes = (double)tiles * 131072 * passes;
    printf("%-44s N=%5d grid %3d: %7.1f us  %5.2f TB/s  %5.1f B/clk/CU  %6.0f cycles per 128 KB tile\n", name, N, grid, ms * 1e3, bytes / (ms * 1e-3) / 1e12,
           bytes / grid / cyc, cyc / (tiles * passes / (double)grid));
}
int main() {
    const int M = 17408;
    char* C; unsigned long long* out;
    CK(hipMalloc(&C, (size_t)M * 3072 * 2)); CK(hipMalloc(&out, 2048 * 16));
    for (int N : {768, 3072})
        for (int grid : {256, 64}) {
            run<0, false>("16 rows x 64 B per instruction", C, M, N, grid, out);
            run<1, false>("8 rows x 128 B per instruction", C, M, N, grid, out);
            run<0, true>("16 rows x 64 B per instruction, nt", C, M, N, grid, out);
            run<1, true>("8 rows x 128 B per instruction, nt", C, M, N, grid, out);
            run<3, false>("8 rows x 128 B, wait + barrier per tile", C, M, N, grid, out);
            run<4, false>("same + 64 KB LDS-DMA behind the stores", C, M, N, grid, out);
        }
    return 0;
}
