// Microbenchmark: cycles per instruction of the library's interleaved requant blocks (requant12_int,
// requant16_int), one wave per SIMD, straight-line code over 96 live accumulators like the C2 epilogue.
// Build: hipcc --offload-arch=gfx950 -O2 -Iinclude -Iqcnn_gpu_amd/csrc tools/requant_rate.hip -o tools/requant_rate
#include "fused_common.h"
#include <cstdio>
using namespace qcnn;

template <int MODE>
__global__ __launch_bounds__(512) void k(long long *out, unsigned *sink, Requant q1, Requant q2, int iters)
{
    v4i a[8], b[8], c[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = v4i{(int)threadIdx.x * 3 + i, i * 1000, -i * 77, 5000 + i};
        b[i] = a[i] + 17;
        c[i] = a[i] - 4000;
    }
    unsigned acc = 0;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int y = 0; y < 8; ++y) {
            unsigned d0, d1, d2;
            if constexpr (MODE == 0) {
                requant12_int(a[y], b[y], c[y], q1, q2, d0, d1, d2);
            } else {
                d0 = requant4_int(a[y].x, a[y].y, a[y].z, a[y].w, q1);
                d1 = requant4_int(b[y].x, b[y].y, b[y].z, b[y].w, q1);
                d2 = requant4_int(c[y].x, c[y].y, c[y].z, c[y].w, q2);
            }
            acc ^= d0 + d1 + d2;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int y = 0; y < 8; ++y) a[y].x += it;
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[threadIdx.x >> 6] = t1 - t0;
    sink[threadIdx.x] = acc;
}

int main()
{
    long long *out;
    unsigned *sink;
    hipMalloc(&out, 64);
    hipMalloc(&sink, 2048);
    Requant q1 = make_requant(11512, 723, 16), q2 = make_requant(10182, 205, 14);
    for (int threads : {256, 512}) {
        long long h[8];
        hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, out, sink, q1, q2, 1000);
        hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
        printf("%d threads: requant12_int x8: %.1f cycles per 8 rows (192 VALU) wave0", threads, h[0] / 1000.0);
        if (threads > 256) printf(", wave4 %.1f", h[4] / 1000.0);
        printf("\n");
        hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, out, sink, q1, q2, 1000);
        hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
        printf("%d threads: 3 x requant4_int x8: %.1f cycles per 8 rows wave0", threads, h[0] / 1000.0);
        if (threads > 256) printf(", wave4 %.1f", h[4] / 1000.0);
        printf("\n");
    }
    return 0;
}
