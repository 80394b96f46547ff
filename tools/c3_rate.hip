// Microbenchmark: cycles per MFMA of the C3 stage's loop shape (kernels_fused_b.hip) run by ONE wave per SIMD:
// 21 B fragments read from LDS with a look-ahead of P fragments, 11 A fragments in registers, 71 MFMAs into 15
// accumulators per pass, accumulators started from a bias.  Variants: look-ahead depth, scheduling fences
// between the steps on or off, one or two waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I qcnn_gpu_amd/csrc -I include tools/c3_rate.hip -o tools/c3_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <utility>

typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4i mfma16(v4i a, v4i b, v4i c) { return __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0); }
template <typename F, int... I>
__device__ __forceinline__ void sf_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void sf(F &&f) { sf_impl(static_cast<F &&>(f), std::make_integer_sequence<int, N>{}); }

constexpr int ROWS = 5, WIN = ROWS + 2, STEPS = 3 * WIN, INC = 34, PX = 48;

template <int P, bool FENCE>
__global__ __launch_bounds__(768) void k(long long *out, int *sink, int iters, int waves)
{
    __shared__ __attribute__((aligned(16))) unsigned char sIn[22 * INC * PX + 64];
    for (int i = threadIdx.x; i < (int)sizeof(sIn) / 4; i += blockDim.x) reinterpret_cast<int *>(sIn)[i] = i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (wv >= waves) return;
    const int g = lane >> 4, c = lane & 15;
    v4i f31[3][3], f32[2];
    for (int s = 0; s < 3; ++s)
        for (int r = 0; r < 3; ++r) f31[s][r] = v4i{lane + s, r, 3, 4};
    f32[0] = v4i{1, lane, 3, 4};
    f32[1] = v4i{1, 2, lane, 4};
    const v4i b0 = {1, 2, 3, 4}, b1 = {5, 6, 7, 8}, b2 = {9, 10, 11, 12};
    const unsigned char *rowbase = sIn + ((wv & 3) * 5 * INC + c) * PX + g * 16;
    const unsigned char *colbase[3] = {rowbase, rowbase + PX, rowbase + 2 * PX};
    int s_acc = 0;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        v4i acc31[ROWS], acc32[2][ROWS];
#pragma unroll
        for (int y = 0; y < ROWS; ++y) acc31[y] = b0, acc32[0][y] = b1, acc32[1][y] = b2;
        v4i Bq[STEPS];
#pragma unroll
        for (int I = 0; I < P; ++I) Bq[I] = *reinterpret_cast<const v4i *>(colbase[I / WIN] + (I % WIN) * (INC * PX));
        __builtin_amdgcn_sched_barrier(0);
        sf<STEPS>([&](auto ic) {
            constexpr int I = decltype(ic)::value;
            constexpr int s = I / WIN, yy = I % WIN;
            constexpr int J = I + P;
            if constexpr (J < STEPS) Bq[J] = *reinterpret_cast<const v4i *>(colbase[J / WIN] + (J % WIN) * (INC * PX));
            const v4i B = Bq[I];
            constexpr int lo = yy - 2 > 0 ? yy - 2 : 0, hi = yy < ROWS - 1 ? yy : ROWS - 1;
#pragma unroll
            for (int y = lo; y <= hi; ++y) acc31[y] = mfma16(f31[s][yy - y], B, acc31[y]);
            if constexpr (s == 1 && yy >= 1 && yy <= ROWS) {
                acc32[0][yy - 1] = mfma16(f32[0], B, acc32[0][yy - 1]);
                acc32[1][yy - 1] = mfma16(f32[1], B, acc32[1][yy - 1]);
            }
            if constexpr (FENCE) __builtin_amdgcn_sched_barrier(0);
        });
#pragma unroll
        for (int y = 0; y < ROWS; ++y) s_acc += acc31[y].x + acc32[0][y].y + acc32[1][y].z;
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[wv] = t1 - t0;
    if (s_acc == 0x12345678) *sink = s_acc;
}

template <int P, bool FENCE>
static void run(const char *name, int waves)
{
    long long *out, h[12];
    int *sink;
    (void)hipMalloc(&out, sizeof h);
    (void)hipMalloc(&sink, 4);
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<P, FENCE>), dim3(1), dim3(768), 0, 0, out, sink, iters, waves);
        (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    // s_memtime ticks at 100 MHz; scale with the shader clock printed by the caller is not needed for ratios:
    printf("%-28s waves %2d: %7.1f ticks per pass of 71 MFMAs (wave 0), %7.1f (last wave)\n", name, waves, (double)h[0] / iters,
           (double)h[waves - 1] / iters);
    (void)hipFree(out);
    (void)hipFree(sink);
}

int main()
{
    run<4, true>("look-ahead 4, fences", 4);
    run<4, true>("look-ahead 4, fences", 8);
    run<8, true>("look-ahead 8, fences", 4);
    run<8, true>("look-ahead 8, fences", 8);
    run<4, false>("look-ahead 4, no fences", 4);
    run<4, false>("look-ahead 4, no fences", 8);
    run<21, false>("all 21 reads first", 4);
    run<21, false>("all 21 reads first", 8);
    return 0;
}
