// Microbenchmark: issue-to-issue cycles of back-to-back int8 MFMAs, one and two waves per SIMD,
// with NACC independent accumulators, on one workgroup or on every CU.
// Build: hipcc --offload-arch=gfx950 -O2 tools/mfma_rate.hip -o tools/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int KIND, int NACC>
__global__ __launch_bounds__(512) void k(long long *out, int *sink, int iters)
{
    v4i A = {1 + (int)threadIdx.x, 2, 3, 4}, B = {5, 6, 7, 8 + (int)threadIdx.x};
    long long t0, t1;
    int s = 0;
    if constexpr (KIND == 0) {
        v4i c[NACC];
        for (int i = 0; i < NACC; ++i) c[i] = v4i{0, 0, 0, 0};
        __syncthreads();
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int r = 0; r < 16 / NACC; ++r)
#pragma unroll
                for (int j = 0; j < NACC; ++j) c[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B, c[j], 0, 0, 0);
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < NACC; ++i) s += c[i].x;
    } else if constexpr (KIND == 1) {
        v16i c[NACC];
        for (int i = 0; i < NACC; ++i)
            for (int j = 0; j < 16; ++j) c[i][j] = 0;
        __syncthreads();
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int r = 0; r < 16 / NACC; ++r)
#pragma unroll
                for (int j = 0; j < NACC; ++j) c[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B, c[j], 0, 0, 0);
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < NACC; ++i) s += c[i][0];
    } else {
        v8bf a8, b8;
        for (int i = 0; i < 8; ++i) a8[i] = (__bf16)1.0f, b8[i] = (__bf16)2.0f;
        v4f c[NACC];
        for (int i = 0; i < NACC; ++i) c[i] = v4f{0, 0, 0, 0};
        __syncthreads();
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int r = 0; r < 16 / NACC; ++r)
#pragma unroll
                for (int j = 0; j < NACC; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, c[j], 0, 0, 0);
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < NACC; ++i) s += (int)c[i].x;
    }
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[threadIdx.x >> 6] = t1 - t0;
    sink[threadIdx.x] = s;
}

template <int KIND, int NACC>
static void run(const char *name, int blocks, int threads)
{
    long long *out;
    int *sink;
    hipMalloc(&out, 64);
    hipMalloc(&sink, 2048);
    const int iters = 1000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, NACC>), dim3(blocks), dim3(threads), 0, 0, out, sink, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<KIND, NACC>), dim3(blocks), dim3(threads), 0, 0, out, sink, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    long long h[8];
    hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
    const double n = iters * 16.0;
    printf("%-10s acc %2d  blocks %3d x %3d threads: wave0 %.2f ticks/MFMA", name, NACC, blocks, threads, h[0] / n);
    if (threads > 256) printf("  wave4 %.2f", h[4] / n);
    printf("   (kernel %.1f us -> %.2f ns per MFMA per wave)\n", ms * 1e3, ms * 1e6 / n);
    hipFree(out);
    hipFree(sink);
}

int main()
{
    run<0, 8>("i8 16x64", 1, 64);
    run<0, 8>("i8 16x64", 1, 256);
    run<0, 8>("i8 16x64", 256, 256);
    run<0, 8>("i8 16x64", 256, 512);
    run<0, 1>("i8 16x64", 256, 256);
    run<0, 2>("i8 16x64", 256, 256);
    run<0, 4>("i8 16x64", 256, 256);
    run<0, 16>("i8 16x64", 256, 256);
    run<1, 4>("i8 32x32", 1, 64);
    run<1, 4>("i8 32x32", 256, 256);
    run<1, 4>("i8 32x32", 256, 512);
    run<1, 1>("i8 32x32", 256, 256);
    run<2, 8>("bf16 16x32", 1, 64);
    run<2, 8>("bf16 16x32", 256, 256);
    run<2, 8>("bf16 16x32", 256, 512);
    return 0;
}
