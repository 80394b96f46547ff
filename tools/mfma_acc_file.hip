// Microbenchmark: does it matter for the vector instructions threaded between int8 MFMAs whether the
// MFMA accumulators live in the VGPR half or the AGPR half of the register file?
// One wave per SIMD; K independent v_mad_u32_u24 after every MFMA.
// Build: hipcc --offload-arch=gfx950 -O2 tools/mfma_acc_file.hip -o tools/mfma_acc_file
#include <hip/hip_runtime.h>
#include <cstdio>

typedef int v4i __attribute__((ext_vector_type(4)));

template <bool AGPR, int K>
__global__ __launch_bounds__(256) void k(long long *out, int *sink, int iters)
{
    v4i A = {1 + (int)threadIdx.x, 2, 3, 4}, B = {5, 6, 7, 8 + (int)threadIdx.x};
    v4i c[8];
    for (int i = 0; i < 8; ++i) c[i] = v4i{0, 0, 0, 0};
    int a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if constexpr (AGPR) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(c[j]) : "v"(A), "v"(B));
                else asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(c[j]) : "v"(A), "v"(B));
#pragma unroll
                for (int v = 0; v < K; ++v) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a[(j + v) & 7]) : "v"(3));
            }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[threadIdx.x >> 6] = t1 - t0;
    int s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + c[i].x;
    sink[threadIdx.x] = s;
}

// the pieces the C2 stage threads between its MFMAs: gap j even: two saturating multiply-adds;
// gap j odd: the two sub-dword max that consume them
template <int MODE, int NACC = 8>
__global__ __launch_bounds__(256) void kq(long long *out, int *sink, int iters)
{
    v4i A = {1 + (int)threadIdx.x, 2, 3, 4}, B = {5, 6, 7, 8 + (int)threadIdx.x};
    v4i c[8];
    for (int i = 0; i < 8; ++i) c[i] = v4i{0, 0, 0, 0};
    int a0 = threadIdx.x, a1 = a0 + 7, t0r = 0, t1r = 0, m = 723 << 8, cm = 45 * 723 << 8;
    unsigned p0 = 0, p1 = 0;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(c[j % NACC]) : "v"(A), "v"(B));
                if constexpr (MODE == 0) {
                    if (j & 1) {
                        asm volatile("v_max_i32_sdwa %0, sext(%1), 0 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD" : "+v"(p0) : "v"(t0r));
                        asm volatile("v_max_i32_sdwa %0, sext(%1), 0 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD" : "+v"(p1) : "v"(t1r));
                    } else {
                        asm volatile("v_mad_i32_i24 %0, %1, %2, %3 clamp" : "=v"(t0r) : "v"(a0), "v"(m), "v"(cm));
                        asm volatile("v_mad_i32_i24 %0, %1, %2, %3 clamp" : "=v"(t1r) : "v"(a1), "v"(m), "v"(cm));
                    }
                } else if constexpr (MODE == 1) {  // multiplies only
                    asm volatile("v_mad_i32_i24 %0, %1, %2, %3 clamp" : "=v"(t0r) : "v"(a0), "v"(m), "v"(cm));
                    asm volatile("v_mad_i32_i24 %0, %1, %2, %3 clamp" : "=v"(t1r) : "v"(a1), "v"(m), "v"(cm));
                } else if constexpr (MODE == 2) {  // sub-dword max only
                    asm volatile("v_max_i32_sdwa %0, sext(%1), 0 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD" : "+v"(p0) : "v"(a0));
                    asm volatile("v_max_i32_sdwa %0, sext(%1), 0 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD" : "+v"(p1) : "v"(a1));
                } else if constexpr (MODE == 4) {
                } else {  // multiplies without the saturation
                    asm volatile("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(t0r) : "v"(a0), "v"(m), "v"(cm));
                    asm volatile("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(t1r) : "v"(a1), "v"(m), "v"(cm));
                }
            }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[threadIdx.x >> 6] = t1 - t0;
    int s = p0 + p1 + t0r + t1r;
    for (int i = 0; i < 8; ++i) s += c[i].x;
    sink[threadIdx.x] = s;
}
template <int MODE, int NACC = 8>
static void runq(long long *out, int *sink, const char *what)
{
    const int iters = 1000;
    hipLaunchKernelGGL((kq<MODE, NACC>), dim3(256), dim3(256), 0, 0, out, sink, iters);
    hipLaunchKernelGGL((kq<MODE, NACC>), dim3(256), dim3(256), 0, 0, out, sink, iters);
    long long h[4];
    hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
    printf("%s: %.2f cycles per MFMA\n", what, h[0] / (iters * 16.0));
}

// accumulate chains whose MFMAs write a different register tuple than the one they read as C
// (ping-pong), NACC chains interleaved
template <int NACC, int K>
__global__ __launch_bounds__(256) void kr(long long *out, int *sink, int iters)
{
    v4i A = {1 + (int)threadIdx.x, 2, 3, 4}, B = {5, 6, 7, 8 + (int)threadIdx.x};
    v4i c[NACC], d[NACC];
    for (int i = 0; i < NACC; ++i) c[i] = v4i{0, 0, 0, 0}, d[i] = v4i{0, 0, 0, 0};
    int a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 12 / NACC; ++r) {
#pragma unroll
            for (int j = 0; j < NACC; ++j) {
                if (r & 1) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %3" : "=&v"(c[j]) : "v"(A), "v"(B), "v"(d[j]));
                else asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %3" : "=&v"(d[j]) : "v"(A), "v"(B), "v"(c[j]));
#pragma unroll
                for (int v = 0; v < K; ++v) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a[(j + v) & 7]) : "v"(3));
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[threadIdx.x >> 6] = t1 - t0;
    int s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    for (int i = 0; i < NACC; ++i) s += c[i].x + d[i].x;
    sink[threadIdx.x] = s;
}
template <int NACC, int K>
static void runr(long long *out, int *sink)
{
    const int iters = 1000;
    hipLaunchKernelGGL((kr<NACC, K>), dim3(256), dim3(256), 0, 0, out, sink, iters);
    hipLaunchKernelGGL((kr<NACC, K>), dim3(256), dim3(256), 0, 0, out, sink, iters);
    long long h[4];
    hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
    printf("C read from another tuple than D, %d chains interleaved, %d VALU per MFMA: %.2f cycles per MFMA\n", NACC, K,
           h[0] / (iters * 12.0));
}

template <bool AGPR, int K>
static void run(long long *out, int *sink)
{
    const int iters = 1000;
    hipLaunchKernelGGL((k<AGPR, K>), dim3(256), dim3(256), 0, 0, out, sink, iters);
    hipLaunchKernelGGL((k<AGPR, K>), dim3(256), dim3(256), 0, 0, out, sink, iters);
    long long h[4];
    hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
    printf("accumulators in %s, %d VALU per MFMA: %.2f cycles per MFMA\n", AGPR ? "AGPRs" : "VGPRs", K, h[0] / (iters * 16.0));
}

int main()
{
    long long *out;
    int *sink;
    hipMalloc(&out, 64);
    hipMalloc(&sink, 2048);
    runq<0>(out, sink, "2 requant pieces per MFMA (mad, mad | max, max alternating)");
    runq<0, 4>(out, sink, "the same, accumulator reused every 4th MFMA");
    runq<0, 3>(out, sink, "the same, accumulator reused every 3rd MFMA");
    runq<0, 2>(out, sink, "the same, accumulator reused every 2nd MFMA");
    runq<0, 1>(out, sink, "the same, one accumulator");
    runq<4, 8>(out, sink, "no vector instructions, 8 accumulators");
    runq<4, 3>(out, sink, "no vector instructions, accumulator reused every 3rd MFMA");
    runq<4, 2>(out, sink, "no vector instructions, accumulator reused every 2nd MFMA");
    runq<4, 1>(out, sink, "no vector instructions, one accumulator");
    runq<1>(out, sink, "2 saturating v_mad_i32_i24 per MFMA");
    runq<2>(out, sink, "2 v_max_i32_sdwa per MFMA");
    runq<3>(out, sink, "2 plain v_mad_i32_i24 per MFMA");
    runr<1, 0>(out, sink);
    runr<2, 0>(out, sink);
    runr<3, 0>(out, sink);
    runr<4, 0>(out, sink);
    runr<6, 0>(out, sink);
    runr<2, 2>(out, sink);
    runr<3, 2>(out, sink);
    runr<4, 2>(out, sink);
    runr<6, 2>(out, sink);
    run<true, 0>(out, sink);
    run<true, 1>(out, sink);
    run<true, 2>(out, sink);
    run<true, 3>(out, sink);
    run<true, 4>(out, sink);
    run<false, 0>(out, sink);
    run<false, 1>(out, sink);
    run<false, 2>(out, sink);
    run<false, 3>(out, sink);
    run<false, 4>(out, sink);
    return 0;
}
