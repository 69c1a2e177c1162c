// mfma_probe.hip — what limits an fp32-MFMA loop on gfx950: registers only vs LDS operand reads at the rate of
// the conv kernels (tuning aid; build: hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ unsigned long long g_clk[2];   // block 0: shader-clock ticks (s_memtime) and 100 MHz ticks (s_memrealtime) across the loop

// NT accumulator tiles per wave; per group of 4 k-steps: RA ds_read_b128 for A, RB for B (0 = operands stay in registers)
template <int NT, int RA, int RB, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void probe(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) { unsigned u = (i + 1) * 2654435761u; u ^= u >> 15; u *= 0x85ebca6bu; u ^= u >> 13; lds[i] = (float)(int)u * (1.0f / 2147483648.0f); }   // random in [-1, 1): operand bits toggle like real data (DVFS)
    __syncthreads();
    f32x16 acc[NT];
    for (int t = 0; t < NT; ++t) for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
    f32x4 a[NT], b[2];
    for (int t = 0; t < NT; ++t) a[t] = (f32x4){1.0f + lane, 0.5f, 0.25f, 2.f};
    b[0] = b[1] = (f32x4){0.001f * lane, 0.002f, 0.003f, 0.004f};
    const float* base = lds + (lane & 31) * 36 + 4 * (lane >> 5);
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        const float* p = base + (it & 7) * 8;
#pragma unroll
        for (int t = 0; t < NT; ++t) if (t < RA) a[t] = *(const f32x4*)(p + t * 1152);
        if (RB) b[0] = *(const f32x4*)(p + 6000);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][j], b[0][j], acc[t], 0, 0, 0);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { g_clk[0] = __builtin_readcyclecounter() - c0; g_clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
    float s = 0.f;
    for (int t = 0; t < NT; ++t) for (int j = 0; j < 16; ++j) s += acc[t][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the same work per iteration on v_mfma_f32_16x16x4_f32: every 32x32 accumulator tile becomes four 16x16 tiles, every
// 32x32x2 instruction two 16x16x4 instructions (same FLOPs, same operand registers, same LDS reads)
template <int NT, int RA, int RB, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void probe16(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) { unsigned u = (i + 1) * 2654435761u; u ^= u >> 15; u *= 0x85ebca6bu; u ^= u >> 13; lds[i] = (float)(int)u * (1.0f / 2147483648.0f); }   // random in [-1, 1): operand bits toggle like real data (DVFS)
    __syncthreads();
    f32x4 acc[NT * 4];
    for (int t = 0; t < NT * 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 a[NT], b[2];
    for (int t = 0; t < NT; ++t) a[t] = (f32x4){1.0f + lane, 0.5f, 0.25f, 2.f};
    b[0] = b[1] = (f32x4){0.001f * lane, 0.002f, 0.003f, 0.004f};
    const float* base = lds + (lane & 31) * 36 + 4 * (lane >> 5);
    for (int it = 0; it < iters; ++it) {
        const float* p = base + (it & 7) * 8;
#pragma unroll
        for (int t = 0; t < NT; ++t) if (t < RA) a[t] = *(const f32x4*)(p + t * 1152);
        if (RB) b[0] = *(const f32x4*)(p + 6000);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[t * 4 + (j & 1) * 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][j], b[0][j], acc[t * 4 + (j & 1) * 2], 0, 0, 0);
                acc[t * 4 + (j & 1) * 2 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][j], b[0][(j + 1) & 3], acc[t * 4 + (j & 1) * 2 + 1], 0, 0, 0);
            }
    }
    float s = 0.f;
    for (int t = 0; t < NT * 4; ++t) for (int j = 0; j < 4; ++j) s += acc[t][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NT, int RA, int RB, int WAVES, bool S16 = false>
void run(const char* name, int blocks) {
    float* out; hipMalloc(&out, (size_t)blocks * 64 * WAVES * 4);
    int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto k = S16 ? probe16<NT, RA, RB, WAVES> : probe<NT, RA, RB, WAVES>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 40000);
    for (int w = 0; w < 3; ++w) k<<<blocks, 64 * WAVES, 40000>>>(out, iters);      // warm the clocks into their loaded state
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<<<blocks, 64 * WAVES, 40000>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)blocks * WAVES * iters * NT * 4 * 4096.0;
    unsigned long long clk[2] = {0, 0};
    if (!S16) hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_clk), sizeof(clk));
    printf("%-58s %8.3f ms  %7.1f TFLOP/s   shader clock %.0f MHz\n", name, ms, fl / ms / 1e9, clk[1] ? clk[0] / (clk[1] * 1e-2) : 0.0);
    hipFree(out);
}

int main() {
    run<5, 0, 0, 4>("5 acc tiles, registers only, 1 wave/SIMD (256 WG x 4 waves)", 256);
    run<5, 0, 0, 4>("5 acc tiles, registers only, 2 waves/SIMD (512 WG)", 512);
    run<5, 5, 0, 4>("5 acc, 5 LDS b128 / 20 MFMA (conv fwd v2 rate), 2 waves/SIMD", 512);
    run<5, 5, 1, 4>("5 acc, 6 LDS b128 / 20 MFMA, 2 waves/SIMD", 512);
    run<5, 2, 0, 4>("5 acc, 2 LDS b128 / 20 MFMA, 2 waves/SIMD", 512);
    run<5, 5, 0, 4>("5 acc, 5 LDS b128 / 20 MFMA, 1 wave/SIMD", 256);
    run<10, 5, 0, 4>("10 acc, 5 LDS b128 / 40 MFMA, 1 wave/SIMD", 256);
    run<10, 10, 0, 4>("10 acc, 10 LDS b128 / 40 MFMA, 1 wave/SIMD", 256);
    run<5, 0, 0, 4, true>("16x16x4: 5 tiles, registers only, 1 wave/SIMD", 256);
    run<5, 0, 0, 4, true>("16x16x4: 5 tiles, registers only, 2 waves/SIMD", 512);
    run<5, 5, 0, 4, true>("16x16x4: 5 tiles, 5 LDS b128 / 40 MFMA, 2 waves/SIMD", 512);
    run<5, 5, 1, 4, true>("16x16x4: 5 tiles, 6 LDS b128 / 40 MFMA, 2 waves/SIMD", 512);
    run<5, 5, 0, 4>("32x32x2 again: 5 acc, 5 LDS b128 / 20 MFMA, 2 waves/SIMD", 512);
    return 0;
}
