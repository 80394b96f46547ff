// Probe: global_load_lds_dwordx4 on gfx950 -- lane -> LDS mapping, LDS offsets above 64 KiB,
// instruction offset, masked-off lanes.
// Build: hipcc --offload-arch=gfx950 -O2 tools/lds_direct_probe.hip -o tools/lds_direct_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k(const v4i *src, v4i *dst, int base_piece)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    v4i *lds = reinterpret_cast<v4i *>(smem);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 10240; i += 256) lds[i] = v4i{-1, -1, -1, -1};
    __syncthreads();
    // wave wv copies 128 pieces: two instructions, the second with an instruction offset; lanes >= 48
    // of the second one are masked off
    const v4i *g = src + wv * 128 + lane;
    v4i *l = lds + base_piece + wv * 128;
    __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
    if (lane < 48) __builtin_amdgcn_global_load_lds(g, l, 16, 1024, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = threadIdx.x; i < 10240; i += 256) dst[i] = lds[i];
}

int main()
{
    const int N = 10240;  // 160 KiB of 16-byte pieces
    std::vector<v4i> h(N), out(N);
    for (int i = 0; i < N; ++i) h[i] = v4i{i, i * 2, i * 3, i * 5};
    v4i *d_src, *d_dst;
    (void)hipMalloc(&d_src, N * sizeof(v4i));
    (void)hipMalloc(&d_dst, N * sizeof(v4i));
    (void)hipMemcpy(d_src, h.data(), N * sizeof(v4i), hipMemcpyHostToDevice);
    (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int bases[3] = {0, 4096 - 512, 10240 - 512};  // byte offsets 0, 56 KiB (crosses 64 KiB), 152 KiB
    for (int b = 0; b < 3; ++b) {
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 160 * 1024, 0, d_src, d_dst, bases[b]);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
        (void)hipMemcpy(out.data(), d_dst, N * sizeof(v4i), hipMemcpyDeviceToHost);
        int bad = 0, untouched_bad = 0;
        for (int i = 0; i < N; ++i) {
            const int rel = i - bases[b];
            bool expect_data = false;
            int srcidx = 0;
            if (rel >= 0 && rel < 512) {
                const int wv = rel / 128, r = rel % 128;
                if (r < 64) { expect_data = true; srcidx = wv * 128 + r; }
                else if (r < 64 + 48) { expect_data = true; srcidx = wv * 128 + 64 + (r - 64); }
            }
            if (expect_data) { if (out[i].x != srcidx || out[i].w != srcidx * 5) ++bad; }
            else if (out[i].x != -1) ++untouched_bad;
        }
        printf("base piece %5d (byte %6d): %d wrong copies, %d stray writes\n", bases[b], bases[b] * 16, bad, untouched_bad);
        if (bad) for (int i = bases[b]; i < bases[b] + 4; ++i) printf("   lds[%d].x = %d\n", i, out[i].x);
    }
    return 0;
}
