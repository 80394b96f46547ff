// Microbenchmark: cycles per MFMA (16x16x64 int8) of ONE wave that threads K independent VALU
// instructions after every MFMA, alone on its SIMD (256 threads = one wave per SIMD).
// Build: hipcc --offload-arch=gfx950 -O2 tools/mfma_mix.hip -o tools/mfma_mix
#include <hip/hip_runtime.h>
#include <cstdio>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int K>
__global__ __launch_bounds__(256) void k(long long *out, int *sink, int iters)
{
    v4i A = {1, 2, 3, 4}, B = {5, 6, 7, 8};
    v4i c[8];
    for (int i = 0; i < 8; ++i) c[i] = v4i{0, 0, 0, 0};
    int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            c[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B, c[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (K >= 1) asm volatile("v_med3_i32 %0, %0, 0, %1" : "+v"(a0) : "v"(1000));
            if constexpr (K >= 2) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a1) : "v"(3));
            if constexpr (K >= 3) asm volatile("v_med3_i32 %0, %0, 0, %1" : "+v"(a2) : "v"(1000));
            if constexpr (K >= 4) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a3) : "v"(3));
            if constexpr (K >= 5) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a0) : "v"(3));
            if constexpr (K >= 6) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a1) : "v"(3));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    int s = a0 + a1 + a2 + a3;
    for (int i = 0; i < 8; ++i) s += c[i].x;
    sink[threadIdx.x] = s;
}

template <int K>
__global__ __launch_bounds__(256) void k32(long long *out, int *sink, int iters)
{
    v4i A = {1, 2, 3, 4}, B = {5, 6, 7, 8};
    v16i c[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) c[i][j] = 0;
    int a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            c[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B, c[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int v = 0; v < K; ++v) {
                if (v & 1) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a[v & 7]) : "v"(3));
                else asm volatile("v_med3_i32 %0, %0, 0, %1" : "+v"(a[v & 7]) : "v"(1000));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    int s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    for (int i = 0; i < 4; ++i) s += c[i][0];
    sink[threadIdx.x] = s;
}

template <int K>
void run32(long long *d_out, int *d_sink)
{
    const int iters = 2000;
    for (int r = 0; r < 2; ++r) {
        hipLaunchKernelGGL(k32<K>, dim3(256), dim3(256), 0, 0, d_out, d_sink, iters);
        (void)hipDeviceSynchronize();
    }
    long long h[4];
    (void)hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
    printf("32x32x32: K=%d VALU per MFMA: %.1f ticks per MFMA\n", K, (double)h[0] / (iters * 4.0));
}

template <int K>
void run(long long *d_out, int *d_sink)
{
    const int iters = 2000;
    hipLaunchKernelGGL(k<K>, dim3(256), dim3(256), 0, 0, d_out, d_sink, iters);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k<K>, dim3(256), dim3(256), 0, 0, d_out, d_sink, iters);
    hipDeviceSynchronize();
    long long h[4];
    hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
    printf("K=%d VALU per MFMA: %.1f memtime ticks per MFMA (x21 = %.1f core cycles at 2.1 GHz)\n", K,
           (double)h[0] / (iters * 8.0), (double)h[0] / (iters * 8.0) * 21.0);
}

int main()
{
    long long *d_out;
    int *d_sink;
    hipMalloc(&d_out, 256 * 4 * sizeof(long long));
    hipMalloc(&d_sink, 256 * sizeof(int));
    run<0>(d_out, d_sink);
    run<1>(d_out, d_sink);
    run<2>(d_out, d_sink);
    run<3>(d_out, d_sink);
    run<4>(d_out, d_sink);
    run<6>(d_out, d_sink);
    run32<0>(d_out, d_sink);
    run32<2>(d_out, d_sink);
    run32<4>(d_out, d_sink);
    run32<6>(d_out, d_sink);
    run32<8>(d_out, d_sink);
    return 0;
}
