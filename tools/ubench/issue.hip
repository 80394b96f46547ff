// Diagnostic microbenchmark (never shipped): how MFMA and vector ALU instructions of the waves that
// share a SIMD overlap on gfx950.  One 512- or 768-thread workgroup per CU; wave w runs on SIMD w & 3.
// Each wave runs `iters` times a body of M independent 16x16x64 int8 MFMAs and V independent
// v_mad_i32_i24, chosen per wave "slot" (w >> 2), and reports its cycles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));

template <int M, int V, int KIND = 0>
__device__ __forceinline__ void body(v4i (&acc)[8], int (&x)[8], v4i A, v4i B, int mul, int add)
{
#pragma unroll
    for (int i = 0; i < (M > V ? M : V); ++i) {
        if (i < M) acc[i & 7] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B, acc[i & 7], 0, 0, 0);
        // V / M vector instructions after each MFMA (all of them when M == 0)
        constexpr int per = M ? (V + M - 1) / M : 1;
        if (M == 0 || i < M) {
#pragma unroll
            for (int j = 0; j < per; ++j) {
                const int idx = M ? i * per + j : i;
                if (idx < V) {
                    if constexpr (KIND == 0) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(x[idx & 7]) : "v"(mul), "v"(add));
                    else if constexpr (KIND == 1)  // the ReLU + pack instruction of the requantiser
                        asm volatile("v_max_i32_sdwa %0, sext(%1), 0 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD"
                                     : "+v"(x[idx & 7]) : "v"(mul));
                    else if constexpr (KIND == 2)  // the requantiser's pair: saturating multiply-add, then ReLU + pack
                        if (idx & 1) asm volatile("v_max_i32_sdwa %0, sext(%1), 0 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD"
                                     : "+v"(x[idx & 7]) : "v"(x[(idx - 1) & 7]));
                        else asm volatile("v_mad_i32_i24 %0, %1, %2, 0 clamp" : "=v"(x[idx & 7]) : "v"(add), "s"(mul));
                    else if constexpr (KIND == 3) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x[idx & 7]) : "v"(mul), "v"(add));
                    else if constexpr (KIND == 4) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(x[idx & 7]) : "v"(mul), "v"(add));
                    else asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(x[idx & 7]) : "v"(mul));
                }
            }
        }
    }
}

template <int M, int V, int KIND = 0>
__device__ __forceinline__ long long run(int iters, v4i A, v4i B, int seed, int *sink)
{
    v4i acc[8];
    int x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = v4i{seed, i, 0, 0}, x[i] = seed + i;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        body<M, V, KIND>(acc, x, A, B, seed | 1, 3);
        __builtin_amdgcn_sched_barrier(0);
    }
    const long long t1 = clock64();
    int s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3] + x[i];
    if (s == 0x12345678) *sink = s;
    return t1 - t0;
}

// Sustained matrix-pipe rate on operands that change from one instruction to the next (eight random A and B
// fragments per lane, used in turn): what the clock does under realistic operand switching.
__global__ void k_rand(int iters, long long *out, int *sink, unsigned seed)
{
    v4i A[8], B[8], acc[8];
    unsigned h = (threadIdx.x + 1) * 2654435761u ^ (blockIdx.x * 40503u) ^ seed;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            h = h * 1664525u + 1013904223u;
            A[i][j] = seed ? (int)h : 0;
            h = h * 1664525u + 1013904223u;
            B[i][j] = seed ? (int)h : 0;
        }
        acc[i] = v4i{0, 0, 0, 0};
    }
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i & 7] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[i & 7], B[(i + (i >> 3)) & 7], acc[i & 7], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    const long long t1 = clock64();
    int s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    if (s == 0x12345678) *sink = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[threadIdx.x >> 6] = t1 - t0;
}

// MFMA stream whose A and B operands change with every instruction (eight fragments each, random contents): the
// register-file traffic of a real stream, next to the vector stream of another wave.
__device__ __forceinline__ long long run_rot(int iters, int seed, int *sink)
{
    v4i A[8], B[8], acc[8];
    unsigned h = (threadIdx.x + 1) * 2654435761u ^ (unsigned)seed;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            h = h * 1664525u + 1013904223u;
            A[i][j] = (int)h;
            h = h * 1664525u + 1013904223u;
            B[i][j] = (int)h;
        }
        acc[i] = v4i{0, 0, 0, 0};
    }
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i & 7] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[i & 7], B[(i + (i >> 3)) & 7], acc[i & 7], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    const long long t1 = clock64();
    int s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    if (s == 0x12345678) *sink = s;
    return t1 - t0;
}

// role: 0 idle, 1 MFMA x16, 2 VALU x32, 3 MFMA 8 + VALU 8, 4 MFMA 8 + VALU 16, 5 MFMA 8 + VALU 24, 6 MFMA 8 + VALU 32, 7 VALU x16+MFMA x2
__global__ void k(int r0, int r1, int r2, int iters, long long *out, int *sink)
{
    const int w = threadIdx.x >> 6, slot = w >> 2;
    const int role = slot == 0 ? r0 : slot == 1 ? r1 : r2;
    const v4i A = {(int)threadIdx.x, 1, 2, 3}, B = {4, 5, (int)threadIdx.x, 7};
    __syncthreads();
    long long c = 0;
    switch (role) {
    case 1: c = run<16, 0>(iters, A, B, threadIdx.x, sink); break;
    case 2: c = run<0, 32>(iters, A, B, threadIdx.x, sink); break;
    case 3: c = run<8, 8>(iters, A, B, threadIdx.x, sink); break;
    case 4: c = run<8, 16>(iters, A, B, threadIdx.x, sink); break;
    case 5: c = run<8, 24>(iters, A, B, threadIdx.x, sink); break;
    case 6: c = run<8, 32>(iters, A, B, threadIdx.x, sink); break;
    case 7: c = run<2, 16>(iters, A, B, threadIdx.x, sink); break;
    case 8: __builtin_amdgcn_s_setprio(3); c = run<16, 0>(iters, A, B, threadIdx.x, sink); break;
    case 9: c = run<0, 64>(iters, A, B, threadIdx.x, sink); break;
    case 10: c = run<0, 32, 1>(iters, A, B, threadIdx.x, sink); break;
    case 11: c = run<0, 32, 2>(iters, A, B, threadIdx.x, sink); break;
    case 12: c = run<0, 32, 3>(iters, A, B, threadIdx.x, sink); break;
    case 13: c = run<0, 32, 4>(iters, A, B, threadIdx.x, sink); break;
    case 14: c = run<0, 32, 5>(iters, A, B, threadIdx.x, sink); break;
    case 15: c = run<8, 16, 2>(iters, A, B, threadIdx.x, sink); break;
    case 16: c = run_rot(iters, threadIdx.x, sink); break;
    default: break;
    }
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[w] = c;
}

int main(int argc, char **argv)
{
    long long *d;
    int *sink;
    hipMalloc(&d, 12 * 8);
    hipMalloc(&sink, 4);
    const int iters = 2000;
    const int cases[][4] = {
        // waves per SIMD, role of slot 0, 1, 2
        {1, 1, 0, 0}, {1, 2, 0, 0}, {1, 3, 0, 0}, {1, 4, 0, 0}, {1, 5, 0, 0}, {1, 6, 0, 0}, {1, 7, 0, 0},
        {2, 1, 2, 0}, {2, 1, 1, 0}, {2, 2, 2, 0}, {2, 3, 3, 0}, {2, 4, 4, 0}, {2, 5, 5, 0}, {2, 1, 7, 0},
        {2, 2, 1, 0}, {2, 9, 1, 0}, {2, 9, 8, 0}, {2, 7, 1, 0}, {2, 7, 8, 0}, {2, 1, 9, 0},
        {1, 10, 0, 0}, {1, 11, 0, 0}, {1, 12, 0, 0}, {1, 13, 0, 0}, {1, 14, 0, 0}, {1, 15, 0, 0},
        {2, 1, 10, 0}, {2, 1, 11, 0}, {2, 1, 12, 0}, {2, 1, 13, 0}, {2, 1, 14, 0}, {2, 10, 1, 0}, {2, 11, 1, 0},
        {3, 1, 11, 11}, {3, 1, 1, 11}, {3, 11, 1, 1}, {3, 11, 11, 1}, {3, 2, 1, 1},
        {1, 16, 0, 0}, {2, 16, 9, 0}, {2, 9, 16, 0}, {2, 16, 11, 0}, {3, 16, 11, 11}, {3, 11, 16, 16}, {3, 16, 16, 11}, {3, 11, 11, 16},
        {3, 1, 2, 2}, {3, 1, 1, 2}, {3, 3, 3, 3}, {3, 4, 4, 4}, {3, 5, 5, 5}, {3, 1, 1, 1}, {3, 2, 2, 2}, {3, 1, 7, 7},
    };
    const char *names[] = {"idle", "M16", "V32", "M8V8", "M8V16", "M8V24", "M8V32", "M2V16", "M16hi", "V64",
                           "sdwa32", "req32", "perm32", "med32", "pkmax32", "M8req16", "M16rot"};
    const int nm[] = {0, 16, 0, 8, 8, 8, 8, 2, 16, 0, 0, 0, 0, 0, 0, 8, 16}, nv[] = {0, 0, 32, 8, 16, 24, 32, 16, 0, 64, 32, 32, 32, 32, 32, 16, 0};
    for (auto &cs : cases) {
        const int waves = cs[0];
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(d, 0, 12 * 8);
            hipLaunchKernelGGL(k, dim3(256), dim3(256 * waves), 0, 0, cs[1], cs[2], cs[3], iters, d, sink);
            hipDeviceSynchronize();
        }
        long long h[12];
        hipMemcpy(h, d, 12 * 8, hipMemcpyDeviceToHost);
        printf("%d waves/SIMD:", waves);
        for (int s = 0; s < waves; ++s) {
            const int r = cs[1 + s];
            printf("  %-6s %7.1f cyc/iter (%d MFMA, %d VALU)", names[r], (double)h[4 * s] / iters, nm[r], nv[r]);
        }
        printf("\n");
    }
    // sustained rate of the matrix pipe alone: three MFMA waves per SIMD on every CU for about ten milliseconds
    {
        const int it2 = 40000;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k, dim3(256), dim3(768), 0, 0, 1, 1, 1, it2, d, sink);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            long long h[12];
            hipMemcpy(h, d, 12 * 8, hipMemcpyDeviceToHost);
            const double mfmas = 256.0 * 12 * 16 * it2, ops = mfmas * 2 * 16 * 16 * 64;
            printf("MFMA only, 12 waves per CU: %.2f ms, %.2f POP/s int8, slowest wave %.0f cycles -> %.2f GHz shader clock\n", ms,
                   ops / ms / 1e12, (double)h[8], h[8] / (ms * 1e6));
        }
    }
    for (unsigned seed : {0u, 12345u}) {
        const int it2 = 40000;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        for (int rep = 0; rep < 40; ++rep) {  // about half a second of back-to-back launches
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k_rand, dim3(256), dim3(768), 0, 0, it2, d, sink, seed);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            if (rep % 13 != 0) continue;
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            long long h[12];
            hipMemcpy(h, d, 12 * 8, hipMemcpyDeviceToHost);
            const double mfmas = 256.0 * 12 * 16 * it2, ops = mfmas * 2 * 16 * 16 * 64;
            printf("MFMA only, %s operands, launch %d: %.2f ms, %.2f POP/s int8, slowest wave %.0f cycles -> %.2f GHz shader clock\n",
                   seed ? "random" : "all-zero", rep, ms, ops / ms / 1e12, (double)h[8], h[8] / (ms * 1e6));
        }
    }
    return 0;
}
