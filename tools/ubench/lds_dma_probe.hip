#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned char *src, int *dst, int n, int mode)
{
    __shared__ __attribute__((aligned(16))) int sm[1024];
    const int lane = threadIdx.x;
    for (int i = lane; i < 1024; i += 64) sm[i] = -7;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(src), 0, n, 0x00020000);
    const unsigned ldsbase = (unsigned)(size_t)(sm + 64);
    if (mode == 0) {
        const int voff = lane < 14 ? lane : 0x7f000000;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_ubyte %1, %2, %3 offen lds" ::"s"(ldsbase), "v"(voff), "s"(r), "s"(3) : "memory");
    } else if (mode == 1) {
        const int voff = lane < 4 ? lane * 4 : 0x7f000000;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" ::"s"(ldsbase), "v"(voff), "s"(r), "s"(3) : "memory");
    } else {
        // exec-masked: only lanes < 14 take part
        const int voff = lane;
        asm volatile("s_mov_b32 m0, %0\n\ts_mov_b64 s[10:11], exec\n\ts_mov_b64 exec, 0x3fff\n\tbuffer_load_ubyte %1, %2, %3 offen lds\n\ts_mov_b64 exec, s[10:11]" ::"s"(ldsbase), "v"(voff), "s"(r), "s"(3) : "memory", "s10", "s11");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 1024; i += 64) dst[i] = sm[i];
}
int main()
{
    const int n = 4096;
    std::vector<unsigned char> h(n);
    for (int i = 0; i < n; ++i) h[i] = (unsigned char)(i + 100);
    unsigned char *s;
    int *d;
    hipMalloc(&s, n);
    hipMalloc(&d, 1024 * 4);
    hipMemcpy(s, h.data(), n, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, s, d, n, mode);
        std::vector<int> o(1024);
        hipMemcpy(o.data(), d, 1024 * 4, hipMemcpyDeviceToHost);
        printf("mode %d:\n", mode);
        for (int i = 60; i < 60 + 80; ++i) {
            printf(" %08x", o[i]);
            if ((i - 60) % 8 == 7) printf("\n");
        }
        printf("\n");
    }
    return 0;
}
