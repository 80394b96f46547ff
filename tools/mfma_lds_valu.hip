// Microbenchmark: an MFMA stream whose B operands arrive from LDS (ds_read_b128 issued AHEAD steps
// early, one read per MPB MFMAs), with K vector instructions threaded after every MFMA.
// One wave per SIMD.  Build: hipcc --offload-arch=gfx950 -O2 tools/mfma_lds_valu.hip -o tools/mfma_lds_valu
#include <hip/hip_runtime.h>
#include <cstdio>

typedef int v4i __attribute__((ext_vector_type(4)));

template <int K, int MPB, bool LDS>
__global__ __launch_bounds__(256) void k(long long *out, int *sink, int iters)
{
    __shared__ v4i buf[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) buf[i] = v4i{i, 1, 2, 3};
    v4i A = {1 + (int)threadIdx.x, 2, 3, 4};
    v4i c[8];
    for (int i = 0; i < 8; ++i) c[i] = v4i{0, 0, 0, 0};
    int a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
    const v4i *src = buf + (threadIdx.x & 63);
    constexpr int STEPS = 16, AHEAD = 4;
    v4i Bq[STEPS];
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int I = 0; I < AHEAD; ++I) Bq[I] = LDS ? src[64 * I] : A;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int I = 0; I < STEPS; ++I) {
            if (I + AHEAD < STEPS) Bq[I + AHEAD] = LDS ? src[64 * (I + AHEAD)] : A;
#pragma unroll
            for (int m = 0; m < MPB; ++m) {
                c[(I + m) & 7] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, Bq[I], c[(I + m) & 7], 0, 0, 0);
                if constexpr (K > 0) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int v = 0; v < K; ++v) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a[(m + v) & 7]) : "v"(3));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[threadIdx.x >> 6] = t1 - t0;
    int s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + c[i].x;
    sink[threadIdx.x] = s;
}

template <int K, int MPB, bool LDS>
static void run(long long *out, int *sink)
{
    const int iters = 500;
    hipLaunchKernelGGL((k<K, MPB, LDS>), dim3(256), dim3(256), 0, 0, out, sink, iters);
    hipLaunchKernelGGL((k<K, MPB, LDS>), dim3(256), dim3(256), 0, 0, out, sink, iters);
    long long h[4];
    hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
    printf("B from %s, %d MFMA per B, %d VALU per MFMA: %.2f cycles per MFMA\n", LDS ? "LDS " : "regs", MPB, K,
           h[0] / (iters * 16.0 * MPB));
}

int main()
{
    long long *out;
    int *sink;
    hipMalloc(&out, 64);
    hipMalloc(&sink, 2048);
    run<0, 1, false>(out, sink);
    run<2, 1, false>(out, sink);
    run<0, 1, true>(out, sink);
    run<1, 1, true>(out, sink);
    run<2, 1, true>(out, sink);
    run<0, 3, true>(out, sink);
    run<1, 3, true>(out, sink);
    run<2, 3, true>(out, sink);
    run<0, 5, true>(out, sink);
    run<2, 5, true>(out, sink);
    return 0;
}
