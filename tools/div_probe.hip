// Diagnostic: checks that the uniform-divisor division of the convolution epilogues (div_recip / div_rn_by in
// qcnn_gpu_amd/csrc/kernels_conv_mfma.hip, restated here) gives the bits of the compiler's IEEE float32
// division for every integer dividend in [-2^25, 2^25] and a set of integer divisors, and for random
// float dividends below 2^32.   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/div_probe.hip -o tools/div_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ float div_recip(float d)
{
    const float y0 = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, y0, 1.f);
    return __builtin_fmaf(e, y0, y0);
}
__device__ __forceinline__ float div_rn_by(float x, float d, float y)
{
    const float q0 = x * y;
    const float r0 = __builtin_fmaf(-d, q0, x);
    const float q1 = __builtin_fmaf(r0, y, q0);
    const float r1 = __builtin_fmaf(-d, q1, x);
    return __builtin_fmaf(r1, y, q1);
}

__global__ void k_ints(const int *divs, int ndiv, unsigned long long *bad)
{
    const long long n = (1ll << 26) + 1;
    for (int j = 0; j < ndiv; ++j) {
        const float d = (float)divs[j], y = div_recip(d);
        for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
            const float x = (float)(i - (1ll << 25));
            if (__float_as_uint(div_rn_by(x, d, y)) != __float_as_uint(__fdiv_rn(x, d))) atomicAdd(bad, 1ull);
        }
    }
}

__global__ void k_rand(const int *divs, int ndiv, unsigned long long seed, unsigned long long *bad)
{
    unsigned long long s = seed + (blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull;
    for (int it = 0; it < 4096; ++it) {
        s ^= s << 13, s ^= s >> 7, s ^= s << 17;
        const float x = (float)(long long)(s >> 31) - 4294967296.f + (float)(s & 1023) / 1024.f;  // |x| < 2^32, fractional
        const float d = (float)divs[(s >> 20) % ndiv], y = div_recip(d);
        if (__float_as_uint(div_rn_by(x, d, y)) != __float_as_uint(__fdiv_rn(x, d))) atomicAdd(bad, 1ull);
    }
}

int main()
{
    std::vector<int> divs;
    for (int d = 1; d <= 40; ++d) divs.push_back(d);
    for (int d : {47, 63, 64, 65, 100, 127, 128, 129, 181, 255, 256, 257, 1000, 4095, 4096, 4097, 65535, 65536, 65537, 131071,
                  1000003, 16777215, 16777216, 2147483520})
        divs.push_back(d);
    int *dd;
    unsigned long long *bad, h[2] = {0, 0};
    hipMalloc(&dd, divs.size() * sizeof(int));
    hipMalloc(&bad, 2 * sizeof(unsigned long long));
    hipMemcpy(dd, divs.data(), divs.size() * sizeof(int), hipMemcpyHostToDevice);
    hipMemset(bad, 0, 2 * sizeof(unsigned long long));
    hipLaunchKernelGGL(k_ints, dim3(4096), dim3(256), 0, 0, dd, (int)divs.size(), bad);
    hipLaunchKernelGGL(k_rand, dim3(4096), dim3(256), 0, 0, dd, (int)divs.size(), 12345ull, bad + 1);
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    hipMemcpy(h, bad, sizeof h, hipMemcpyDeviceToHost);
    printf("divisors %zu: integer dividends [-2^25, 2^25]: %llu mismatches; random dividends: %llu mismatches\n", divs.size(), h[0], h[1]);
    return (h[0] || h[1]) ? 1 : 0;
}
