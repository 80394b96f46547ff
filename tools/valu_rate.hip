// Microbenchmark: issue cost (cycles per wave-instruction) of the VALU instructions the requant
// epilogues use, alone and next to a wave that issues int8 MFMAs back to back on the same SIMD.
// Build: hipcc --offload-arch=gfx950 -O2 tools/valu_rate.hip -o tools/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef int v4i __attribute__((ext_vector_type(4)));

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int OP>
__device__ __forceinline__ void body(int &a0, int &a1, int &a2, int &a3, int &a4, int &a5, int &a6, int &a7, int k)
{
    if constexpr (OP == 0) {  // v_mad_u32_u24
        REP8(asm volatile("v_mad_u32_u24 %0, %0, %8, %9\n v_mad_u32_u24 %1, %1, %8, %9\n v_mad_u32_u24 %2, %2, %8, %9\n v_mad_u32_u24 %3, %3, %8, %9\n"
                          "v_mad_u32_u24 %4, %4, %8, %9\n v_mad_u32_u24 %5, %5, %8, %9\n v_mad_u32_u24 %6, %6, %8, %9\n v_mad_u32_u24 %7, %7, %8, %9\n"
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k), "v"(k));)
    } else if constexpr (OP == 1) {  // v_perm_b32
        REP8(asm volatile("v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n"
                          "v_perm_b32 %4, %4, %8, %9\n v_perm_b32 %5, %5, %8, %9\n v_perm_b32 %6, %6, %8, %9\n v_perm_b32 %7, %7, %8, %9\n"
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k), "v"(k));)
    } else if constexpr (OP == 2) {  // v_med3_i32
        REP8(asm volatile("v_med3_i32 %0, %0, 0, %8\n v_med3_i32 %1, %1, 0, %8\n v_med3_i32 %2, %2, 0, %8\n v_med3_i32 %3, %3, 0, %8\n"
                          "v_med3_i32 %4, %4, 0, %8\n v_med3_i32 %5, %5, 0, %8\n v_med3_i32 %6, %6, 0, %8\n v_med3_i32 %7, %7, 0, %8\n"
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));)
    } else if constexpr (OP == 3) {  // v_add_u32
        REP8(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                          "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));)
    } else if constexpr (OP == 4) {  // v_max_i32 + v_min_i32
        REP8(asm volatile("v_max_i32 %0, %0, %8\n v_min_i32 %1, %1, %8\n v_max_i32 %2, %2, %8\n v_min_i32 %3, %3, %8\n"
                          "v_max_i32 %4, %4, %8\n v_min_i32 %5, %5, %8\n v_max_i32 %6, %6, %8\n v_min_i32 %7, %7, %8\n"
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));)
    } else if constexpr (OP == 5) {  // v_mul_lo_u32
        REP8(asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                          "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n"
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));)
    } else if constexpr (OP == 6) {  // v_ashrrev_i32
        REP8(asm volatile("v_ashrrev_i32 %0, 1, %0\n v_ashrrev_i32 %1, 1, %1\n v_ashrrev_i32 %2, 1, %2\n v_ashrrev_i32 %3, 1, %3\n"
                          "v_ashrrev_i32 %4, 1, %4\n v_ashrrev_i32 %5, 1, %5\n v_ashrrev_i32 %6, 1, %6\n v_ashrrev_i32 %7, 1, %7\n"
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));)
    } else if constexpr (OP == 7) {  // v_mad_i32_i24
        REP8(asm volatile("v_mad_i32_i24 %0, %0, %8, %9\n v_mad_i32_i24 %1, %1, %8, %9\n v_mad_i32_i24 %2, %2, %8, %9\n v_mad_i32_i24 %3, %3, %8, %9\n"
                          "v_mad_i32_i24 %4, %4, %8, %9\n v_mad_i32_i24 %5, %5, %8, %9\n v_mad_i32_i24 %6, %6, %8, %9\n v_mad_i32_i24 %7, %7, %8, %9\n"
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k), "v"(k));)
    }
}

// mode 0: only the VALU waves run; mode 1: waves 4-7 issue MFMAs back to back meanwhile
// mode 2: waves 4-7 run the same VALU body (two VALU waves per SIMD)
template <int OP>
__global__ __launch_bounds__(512) void k(long long *out, int *sink, int mode, int iters)
{
    const int wv = threadIdx.x >> 6;
    int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    long long t0 = 0, t1 = 0;
    if (wv < 4 || mode == 2) {
        __syncthreads();
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i) body<OP>(a0, a1, a2, a3, a4, a5, a6, a7, 3);
        t1 = __builtin_amdgcn_s_memtime();
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wv] = t1 - t0;
        sink[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    } else {
        v4i A = {1, 2, 3, 4}, B = {5, 6, 7, 8};
        v4i c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        __syncthreads();
        t0 = __builtin_amdgcn_s_memtime();
        if (mode == 1) {
            for (int i = 0; i < iters * 4; ++i) {
                c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B, c3, 0, 0, 0);
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wv] = t1 - t0;
        sink[threadIdx.x] = c0.x + c1.y + c2.z + c3.w;
    }
}

template <int OP>
static void run(const char *name, int grid)
{
    long long *out;
    int *sink;
    hipMalloc(&out, (size_t)grid * 8 * sizeof(long long));
    hipMalloc(&sink, 512 * sizeof(int));
    const int iters = 64;  // 64 * 64 = 4096 VALU instructions per wave
    for (int mode = 0; mode < 3; ++mode) {
        hipMemset(out, 0, (size_t)grid * 8 * sizeof(long long));
        k<OP><<<grid, 512>>>(out, sink, mode, iters);
        k<OP><<<grid, 512>>>(out, sink, mode, iters);
        hipDeviceSynchronize();
        long long h[8];
        hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
        printf("%-14s mode %d: VALU wave %.2f cyc/instr", name, mode, (double)h[0] / (iters * 64));
        if (mode == 1) printf("   (MFMA wave: %.2f cyc/mfma over %d)", (double)h[4] / (iters * 16), iters * 16);
        if (mode == 2) printf("   (second VALU wave %.2f)", (double)h[4] / (iters * 64));
        printf("\n");
    }
    hipFree(out);
    hipFree(sink);
}

int main(int argc, char **argv)
{
    const int grid = argc > 1 ? atoi(argv[1]) : 1;
    printf("grid %d workgroups of 512 threads\n", grid);
    run<3>("v_add_u32", grid);
    run<0>("v_mad_u32_u24", grid);
    run<7>("v_mad_i32_i24", grid);
    run<1>("v_perm_b32", grid);
    run<2>("v_med3_i32", grid);
    run<4>("v_max/min_i32", grid);
    run<6>("v_ashrrev_i32", grid);
    run<5>("v_mul_lo_u32", grid);
    return 0;
}
