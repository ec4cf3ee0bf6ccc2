// L2 -> LDS staging probe: what one workgroup per CU (512 threads, 128 KB LDS: the GEMM's residency) gets out of the memory path when
// it stages 64 KB "K-steps" shaped like the GEMM's operand tiles (256 rows x 128 B of A from its own row panel + 256 x 128 B of a shared B),
// with nothing else running.  Modes: LDS-DMA (`buffer_load_dwordx4 ... lds`) or plain loads into VGPRs; 1 or 2 stages in flight.
//   hipcc --offload-arch=gfx950 -O3 -o ab/dmaprobe scripts/experiments/dmaprobe.hip && ab/dmaprobe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512, 2) void probe(const char* A, const char* B, int ldb_bytes, int ksteps, int passes, int rows_a, int rows_b, unsigned long long* out) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int pa = blockIdx.x % (rows_a / 256), pb = blockIdx.x % (rows_b / 256);
    const char* a0 = A + (size_t)pa * 256 * ldb_bytes;
    const char* b0 = B + (size_t)pb * 256 * ldb_bytes;
    auto rsrc = [](const char* base, long long bytes) {
        const unsigned long long b = (unsigned long long)base;
        i32x4 r;
        r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
        r[1] = __builtin_amdgcn_readfirstlane((int)((b >> 32) & 0xffffu));
        r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
        r[3] = 0x00020000;
        return r;
    };
    const i32x4 rsA = rsrc(a0, 256ll * ldb_bytes), rsB = rsrc(b0, 256ll * ldb_bytes);
    unsigned vo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int row = 8 * (wave * 4 + j) + (lane >> 3); vo[j] = (unsigned)row * (unsigned)ldb_bytes + (lane & 7) * 16; }
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    i32x4 sink{0, 0, 0, 0};
    i32x4 regs[DEPTH][8];
    auto issue = [&](int stage, unsigned koff) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (MODE == 0) {
                const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + stage * 65536 + (wave * 4 + j) * 1024);
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(vo[j] + koff), "s"(rsA), "s"(dst) : "memory");
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(vo[j] + koff), "s"(rsB), "s"(dst + 32768) : "memory");
            } else {
                asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(regs[stage][2 * j]) : "v"(vo[j] + koff), "s"(rsA) : "memory");
                asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(regs[stage][2 * j + 1]) : "v"(vo[j] + koff), "s"(rsB) : "memory");
            }
        }
    };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int ps = 0; ps < passes; ++ps) {
        if (DEPTH == 2) issue(0, 0);
        for (int k = 0; k < ksteps; ++k) {
            if (DEPTH == 1) {
                issue(0, (unsigned)k * 128u);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                if (k + 1 < ksteps) { issue((k + 1) & 1, (unsigned)(k + 1) * 128u); asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (MODE == 1) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { asm volatile("" : "+v"(regs[DEPTH == 2 ? (k & 1) : 0][j])); }
                if (k == ksteps - 1 && ps == passes - 1) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) sink += regs[DEPTH == 2 ? (k & 1) : 0][j];
                }
            }
            asm volatile("s_barrier" ::: "memory");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = r1 - r0; }
    if (sink[0] == 0x12345678) out[0] = 1;
    if (MODE == 0 && smem[tid] == 0x7f && out[0] == 7) out[1] = 2;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE, int DEPTH>
void run(const char* name, const char* A, const char* B, int K, int rows_a, int rows_b, int grid, unsigned long long* out) {
    const int ksteps = K / 64, passes = 20, ld = K * 2;
    auto kern = probe<MODE, DEPTH>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 131072, 0, A, B, ld, ksteps, passes, rows_a, rows_b, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 131072, 0, A, B, ld, ksteps, passes, rows_a, rows_b, out);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    static unsigned long long h[2048];
    CK(hipMemcpy(h, out, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost));
    double cyc = 0, rt = 0;
    for (int i = 0; i < grid; ++i) { cyc += h[2 * i]; rt += h[2 * i + 1]; }
    cyc /= grid; rt /= grid;
    const double steps = (double)ksteps * passes;
    printf("%-34s K=%5d grid %3d: %7.3f us per 64 KB step  %6.0f shader cycles  %5.1f B/clk/CU  %6.2f TB/s chip  (clock %.2f GHz)\n", name, K, grid,
           ms * 1e3 / steps, cyc / steps, 65536.0 / (cyc / steps), 65536.0 * grid / (ms * 1e-3 / steps) / 1e12, cyc / rt * 0.1);
}

// K32 ring: steps of 32 k (64-byte rows: one DMA instruction covers 16 rows), DEPTH steps in flight, ring of DEPTH+1 slots of 32 KB
template <int DEPTH>
__global__ __launch_bounds__(512, 2) void probe32(const char* A, const char* B, int ldb_bytes, int ksteps, int passes, int rows_a, int rows_b, unsigned long long* out) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int pa = blockIdx.x % (rows_a / 256), pb = blockIdx.x % (rows_b / 256);
    const char* a0 = A + (size_t)pa * 256 * ldb_bytes;
    const char* b0 = B + (size_t)pb * 256 * ldb_bytes;
    auto rsrc = [](const char* base, long long bytes) {
        const unsigned long long b = (unsigned long long)base;
        i32x4 r;
        r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
        r[1] = __builtin_amdgcn_readfirstlane((int)((b >> 32) & 0xffffu));
        r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
        r[3] = 0x00020000;
        return r;
    };
    const i32x4 rsA = rsrc(a0, 256ll * ldb_bytes), rsB = rsrc(b0, 256ll * ldb_bytes);
    unsigned vo[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) { const int row = 16 * (wave * 2 + j) + (lane >> 2); vo[j] = (unsigned)row * (unsigned)ldb_bytes + (lane & 3) * 16; }
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    constexpr int SLOTS = DEPTH + 1;
    auto issue = [&](int slot, unsigned koff) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + slot * 32768 + (wave * 2 + j) * 1024);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(vo[j] + koff), "s"(rsA), "s"(dst) : "memory");
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(vo[j] + koff), "s"(rsB), "s"(dst + 16384) : "memory");
        }
    };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int ps = 0; ps < passes; ++ps) {
        int slot = 0;
        for (int k = 0; k < DEPTH - 1 && k < ksteps; ++k) { issue(slot, (unsigned)k * 64u); slot = slot + 1 == SLOTS ? 0 : slot + 1; }
        for (int k = 0; k < ksteps; ++k) {
            if (k + DEPTH - 1 < ksteps) {
                issue(slot, (unsigned)(k + DEPTH - 1) * 64u); slot = slot + 1 == SLOTS ? 0 : slot + 1;
                if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                if (DEPTH == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                if (DEPTH == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = r1 - r0; }
    if (smem[tid] == 0x7f && out[0] == 7) out[1] = 2;
}
template <int DEPTH>
void run32(const char* name, const char* A, const char* B, int K, int rows_a, int rows_b, int grid, unsigned long long* out) {
    const int ksteps = K / 32, passes = 20, ld = K * 2;
    auto kern = probe32<DEPTH>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 163840, 0, A, B, ld, ksteps, passes, rows_a, rows_b, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 163840, 0, A, B, ld, ksteps, passes, rows_a, rows_b, out);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    static unsigned long long h[2048];
    CK(hipMemcpy(h, out, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost));
    double cyc = 0, rt = 0;
    for (int i = 0; i < grid; ++i) { cyc += h[2 * i]; rt += h[2 * i + 1]; }
    cyc /= grid; rt /= grid;
    const double steps = (double)ksteps * passes / 2;          // per 64 KB, comparable with the lines above
    printf("%-34s K=%5d grid %3d: %7.3f us per 64 KB (2 steps)  %6.0f shader cycles  %5.1f B/clk/CU  %6.2f TB/s chip  (clock %.2f GHz)\n", name, K, grid,
           ms * 1e3 / steps, cyc / steps, 65536.0 / (cyc / steps), 65536.0 * grid / (ms * 1e-3 / steps) / 1e12, cyc / rt * 0.1);
}

int main() {
    const int rows_a = 17408, rows_b = 768, KMAX = 3072;
    char *A, *B;
    unsigned long long* out;
    CK(hipMalloc(&A, (size_t)rows_a * KMAX * 2)); CK(hipMalloc(&B, (size_t)rows_b * KMAX * 2)); CK(hipMalloc(&out, 2048 * 16));
    CK(hipMemset(A, 1, (size_t)rows_a * KMAX * 2)); CK(hipMemset(B, 1, (size_t)rows_b * KMAX * 2));
    for (int K : {768, 3072}) {
        for (int grid : {256, 64}) {
            run<0, 1>("LDS-DMA, 1 stage in flight", A, B, K, rows_a, rows_b, grid, out);
            run<0, 2>("LDS-DMA, 2 stages in flight", A, B, K, rows_a, rows_b, grid, out);
            run<1, 1>("VGPR loads, 1 stage in flight", A, B, K, rows_a, rows_b, grid, out);
            run32<1>("K32 ring, 1 step (32 KB) in flight", A, B, K, rows_a, rows_b, grid, out);
            run32<2>("K32 ring, 2 steps in flight", A, B, K, rows_a, rows_b, grid, out);
            run32<3>("K32 ring, 3 steps in flight", A, B, K, rows_a, rows_b, grid, out);
            run32<4>("K32 ring, 4 steps in flight", A, B, K, rows_a, rows_b, grid, out);
        }
    }
    return 0;
}
