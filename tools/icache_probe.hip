// Microbenchmark: cycles per instruction of straight-line 8-byte vector instructions as a function of
// the code footprint of the loop body (every CU runs the same code, one wave per SIMD).
// Build: hipcc --offload-arch=gfx950 -O2 tools/icache_probe.hip -o tools/icache_probe
#include <hip/hip_runtime.h>
#include <cstdio>

#define BODY(N)                                                                                        \
    asm volatile(".rept " #N "\n v_mad_u32_u24 %0, %0, %4, %4\n v_mad_u32_u24 %1, %1, %4, %4\n"        \
                 " v_mad_u32_u24 %2, %2, %4, %4\n v_mad_u32_u24 %3, %3, %4, %4\n .endr"                \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k))

template <int KB>
__global__ __launch_bounds__(512) void kern(long long *out, int *sink, int iters)
{
    int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, k = 3;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if constexpr (KB == 4) BODY(128);
        if constexpr (KB == 16) BODY(512);
        if constexpr (KB == 32) BODY(1024);
        if constexpr (KB == 48) BODY(1536);
        if constexpr (KB == 64) BODY(2048);
        if constexpr (KB == 96) BODY(3072);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[threadIdx.x >> 6] = t1 - t0;
    sink[threadIdx.x] = a0 + a1 + a2 + a3;
}

template <int KB>
static void run(int grid, int threads)
{
    long long *out;
    int *sink;
    hipMalloc(&out, 64);
    hipMalloc(&sink, 2048);
    const int iters = 16;
    hipLaunchKernelGGL(kern<KB>, dim3(grid), dim3(threads), 0, 0, out, sink, iters);
    hipLaunchKernelGGL(kern<KB>, dim3(grid), dim3(threads), 0, 0, out, sink, iters);
    long long h[8];
    hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
    printf("body %3d KB, %3d workgroups x %d threads: %.2f cycles per instruction (wave 0)", KB, grid, threads,
           (double)h[0] / (iters * KB * 128.0));
    if (threads > 256) printf(", wave 4: %.2f", (double)h[4] / (iters * KB * 128.0));
    printf("\n");
    hipFree(out);
    hipFree(sink);
}

int main()
{
    for (int grid : {1, 256}) {
        run<4>(grid, 256);
        run<16>(grid, 256);
        run<32>(grid, 256);
        run<48>(grid, 256);
        run<64>(grid, 256);
        run<96>(grid, 256);
    }
    run<4>(256, 512);
    run<32>(256, 512);
    run<48>(256, 512);
    run<64>(256, 512);
    return 0;
}
