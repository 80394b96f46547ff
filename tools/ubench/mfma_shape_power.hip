// Microbenchmark: sustained int8 matrix rate of the two MFMA shapes at the power cap, on operands with the
// statistics of the network's tensors (activations 0..127 with about half of them zero, weights of small magnitude)
// and on uniformly random bytes.  Every wave cycles through eight A and eight B fragments; twelve waves per CU.
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_shape_power.hip -o tools/ubench/mfma_shape_power
#include <hip/hip_runtime.h>
#include <cstdio>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__device__ unsigned lcg(unsigned &h)
{
    h = h * 1664525u + 1013904223u;
    return h >> 8;
}
// kind 0: all zero, 1: uniformly random bytes, 2: network-like (A = weights in -24..24, B = activations 0..127, half zero)
__device__ void fill(v4i (&A)[8], v4i (&B)[8], int kind, unsigned h)
{
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 4; ++j) {
            unsigned a = 0, b = 0;
            for (int k = 0; k < 4; ++k) {
                int wa = 0, wb = 0;
                if (kind == 1) wa = (int)(lcg(h) & 255) - 128, wb = (int)(lcg(h) & 255) - 128;
                if (kind == 2) {
                    wa = (int)(lcg(h) % 49) - 24;
                    const unsigned r = lcg(h);
                    wb = (r & 1) ? (int)((r >> 1) & 127) : 0;
                }
                a |= (unsigned)(wa & 255) << (8 * k);
                b |= (unsigned)(wb & 255) << (8 * k);
            }
            A[i][j] = (int)a;
            B[i][j] = (int)b;
        }
}

template <int SHAPE>
__global__ __launch_bounds__(768) void k(int iters, int kind, long long *out, int *sink)
{
    v4i A[8], B[8];
    fill(A, B, kind, (threadIdx.x + 1) * 2654435761u ^ (blockIdx.x * 40503u));
    int s = 0;
    const long long t0 = clock64();
    if constexpr (SHAPE == 16) {
        v4i acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = v4i{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                acc[i & 7] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[i & 7], B[(i + (i >> 3)) & 7], acc[i & 7], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    } else {
        v16i acc[4];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 16; ++j) acc[i][j] = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i)  // the same number of operations as sixteen 16x16x64
                acc[i & 3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[i & 7], B[(i + (i >> 2)) & 7], acc[i & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
    }
    const long long t1 = clock64();
    if (s == 0x12345678) *sink = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[threadIdx.x >> 6] = t1 - t0;
}

int main()
{
    long long *d;
    int *sink;
    hipMalloc(&d, 12 * 8);
    hipMalloc(&sink, 4);
    const int iters = 40000;
    const char *names[3] = {"all-zero", "random bytes", "network-like"};
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int kind = 0; kind < 3; ++kind)
        for (int shape : {16, 32, 16, 32}) {
            float last = 0;
            long long h[12];
            for (int rep = 0; rep < 40; ++rep) {  // more than half a second of back-to-back launches: the clock settles
                hipEventRecord(e0, 0);
                if (shape == 16) hipLaunchKernelGGL(k<16>, dim3(256), dim3(768), 0, 0, iters, kind, d, sink);
                else hipLaunchKernelGGL(k<32>, dim3(256), dim3(768), 0, 0, iters, kind, d, sink);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&last, e0, e1);
            }
            hipMemcpy(h, d, 12 * 8, hipMemcpyDeviceToHost);
            const double ops = 256.0 * 12 * 16 * iters * 2 * 16 * 16 * 64;
            printf("%-13s %s: %.2f ms, %.2f POP/s, %.0f cycles -> %.2f GHz\n", names[kind], shape == 16 ? "16x16x64" : "32x32x32", last,
                   ops / last / 1e12, (double)h[8], h[8] / (last * 1e6));
        }
    return 0;
}
