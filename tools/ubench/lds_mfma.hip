// Diagnostic microbenchmark (never shipped): the rate of 16x16x64 int8 MFMAs whose operands come from LDS, as a
// function of the ds_read_b128 per MFMA, with two or three waves per SIMD on every CU.  One workgroup per CU; each
// wave reads fragments (64 lanes x 16 bytes, conflict-free: lane * 16 + a rotating kilobyte offset) from a region
// of random bytes and feeds them to M MFMAs per iteration, R reads per iteration issued one iteration ahead.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));

// Vector stream of the requantiser (saturating multiply-add, then sub-dword max into a byte of another register), V
// instructions per iteration.
template <int V>
__device__ __forceinline__ long long vec_stream(int iters, int seed, int *sink)
{
    int x[8], t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = seed + i, t[i] = seed * 3 + i;
    const int mul = seed | 1;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < V / 2; ++i) {
            asm volatile("v_mad_i32_i24 %0, %1, %2, 0 clamp" : "=v"(t[i & 7]) : "v"(x[(i + 3) & 7]), "s"(mul));
            asm volatile("v_max_i32_sdwa %0, sext(%1), 0 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD"
                         : "+v"(x[i & 7]) : "v"(t[(i + 5) & 7]));
        }
    }
    const long long t1 = clock64();
    int s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i] + t[i];
    if (s == 0x12345678) *sink = s;
    return t1 - t0;
}

// MIX = 1: the second wave of every SIMD (waves 4..7 of an 8-wave workgroup) runs the vector stream instead
template <int R, int M, int MIX = 0>
__global__ void k(int iters, long long *out, int *sink, const int *init)
{
    __shared__ __attribute__((aligned(16))) int lds[40 * 256];  // 40 KiB
    for (int i = threadIdx.x; i < 40 * 256; i += blockDim.x) lds[i] = init[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (MIX && w >= 4) {
        const long long c = vec_stream<32>(iters, threadIdx.x, sink);
        if (lane == 0 && blockIdx.x == 0) out[w] = c;
        return;
    }
    const v4i *base = reinterpret_cast<const v4i *>(lds) + lane;
    v4i f[2][8], acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        f[0][i] = base[((w + i) % 40) * 64];
        f[1][i] = base[((w + i + 8) % 40) * 64];
        acc[i] = v4i{0, 0, 0, 0};
    }
    // LDS byte address of this lane's piece of fragment 0; the reads are inline assembly with hand-counted waits
    // (the compiler would wait for all of them right behind their issue)
    const unsigned a0 = (unsigned)(size_t)(lds) + lane * 16 + (w & 3) * 1024;
    const long long t0 = clock64();
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            // reads of the next iteration into the other set (its first R fragments are refreshed, the rest stay)
#pragma unroll
            for (int r = 0; r < R; ++r)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[half ^ 1][r]) : "v"(a0), "n"(r * 4096 + 2048 * 0) : "memory");
            // the reads issued one iteration ago (into this set) have landed when at most R are outstanding
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(R) : "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < M; ++m)
                acc[m & 7] = __builtin_amdgcn_mfma_i32_16x16x64_i8(f[half][m & 7], f[half][(m + 3) & 7], acc[m & 7], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const long long t1 = clock64();
    int s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][2];
    if (s == 0x12345678) *sink = s;
    if (lane == 0 && blockIdx.x == 0) out[w] = t1 - t0;
}

template <int R, int M>
void run(int waves, long long *d, int *sink, const int *init)
{
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 12; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<R, M>), dim3(256), dim3(256 * waves), 0, 0, iters, d, sink, init);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    long long h[12];
    hipMemcpy(h, d, 12 * 8, hipMemcpyDeviceToHost);
    long long worst = 0;
    for (int i = 0; i < 4 * waves; ++i) worst = h[i] > worst ? h[i] : worst;
    const double mf = (double)iters * M;
    printf("%d waves/SIMD, %d reads + %2d MFMA per iteration (%.2f per MFMA): %6.2f cycles per MFMA and SIMD, %.2f GHz, %.2f POP/s\n", waves, R, M,
           (double)R / M, (double)worst / (mf * waves), worst / (ms * 1e6), 256.0 * 4 * waves * mf * 32768 / ms / 1e9);
}

template <int R, int M>
void run_mix(long long *d, int *sink, const int *init)
{
    const int iters = 20000;
    for (int rep = 0; rep < 6; ++rep) {
        hipLaunchKernelGGL((k<R, M, 1>), dim3(256), dim3(512), 0, 0, iters, d, sink, init);
        hipDeviceSynchronize();
    }
    long long h[12];
    hipMemcpy(h, d, 12 * 8, hipMemcpyDeviceToHost);
    printf("LDS-fed matrix wave (%d reads + %d MFMA per iteration) next to a vector wave (32 per iteration): %6.2f cycles per MFMA, %6.2f cycles per vector instruction\n",
           R, M, (double)h[0] / ((double)iters * M), (double)h[4] / ((double)iters * 32));
}

int main()
{
    long long *d;
    int *sink, *init;
    hipMalloc(&d, 12 * 8);
    hipMalloc(&sink, 4);
    hipMalloc(&init, 40 * 1024);
    std::vector<int> h(40 * 256);
    unsigned x = 12345;
    for (auto &v : h) v = (int)(x = x * 1664525u + 1013904223u);
    hipMemcpy(init, h.data(), 40 * 1024, hipMemcpyHostToDevice);
    run_mix<0, 8>(d, sink, init);
    run_mix<2, 8>(d, sink, init);
    run_mix<4, 8>(d, sink, init);
    run_mix<8, 8>(d, sink, init);
    for (int waves : {1, 2, 3}) {
        run<0, 8>(waves, d, sink, init);
        run<2, 8>(waves, d, sink, init);
        run<4, 8>(waves, d, sink, init);
        run<5, 8>(waves, d, sink, init);
        run<6, 8>(waves, d, sink, init);
        run<8, 8>(waves, d, sink, init);
    }
    return 0;
}
