// Checks, with exact integer data, the lane<->element maps the fused kernels rely on for the
// gfx950 int8 MFMA instructions:
//   16x16x64: lane l holds A[row l&15][k = 16*(l>>4)+j], B[k = 16*(l>>4)+j][col l&15], j = 0..15;
//             D reg i of lane l = D[row 4*(l>>4)+i][col l&15]
//   32x32x32: lane l holds A[row l&31][k = 16*(l>>5)+j], B[k = 16*(l>>5)+j][col l&31];
//             D reg i of lane l = D[row (i&3)+8*(i>>2)+4*(l>>5)][col l&31]
// (The kernels only need that A's (k-group, byte) pairs with B's (k-group, byte); the check below
//  is stricter.)  Build: hipcc --offload-arch=gfx950 -O2 tools/mfma_probe.hip -o tools/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ void probe16(const v4i *a, const v4i *b, v4i *d)
{
    const int l = threadIdx.x;
    v4i c = {1000, 2000, 3000, 4000};
    d[l] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[l], b[l], c, 0, 0, 0);
}
__global__ void probe32(const v4i *a, const v4i *b, v16i *d)
{
    const int l = threadIdx.x;
    v16i c = {};
    d[l] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[l], b[l], c, 0, 0, 0);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)

int main()
{
    srand(7);
    int fails = 0;
    {
        static int8_t A[16][64], B[64][16];
        for (auto &r : A) for (auto &v : r) v = (int8_t)(rand() % 256 - 128);
        for (auto &r : B) for (auto &v : r) v = (int8_t)(rand() % 256 - 128);
        int8_t ha[64][16], hb[64][16];
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 16; ++j) {
                ha[l][j] = A[l & 15][16 * (l >> 4) + j];
                hb[l][j] = B[16 * (l >> 4) + j][l & 15];
            }
        v4i *da, *db, *dd;
        CK(hipMalloc(&da, 1024)); CK(hipMalloc(&db, 1024)); CK(hipMalloc(&dd, 1024));
        CK(hipMemcpy(da, ha, 1024, hipMemcpyHostToDevice));
        CK(hipMemcpy(db, hb, 1024, hipMemcpyHostToDevice));
        probe16<<<1, 64>>>(da, db, dd);
        int hd[64][4];
        CK(hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost));
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 4; ++i) {
                const int row = 4 * (l >> 4) + i, col = l & 15;
                int ref = 1000 * (i + 1);
                for (int k = 0; k < 64; ++k) ref += (int)A[row][k] * (int)B[k][col];
                if (ref != hd[l][i]) { if (fails < 5) printf("16x16x64 mismatch lane %d reg %d: %d vs %d\n", l, i, hd[l][i], ref); ++fails; }
            }
        printf("mfma_i32_16x16x64_i8: %s\n", fails ? "FAIL" : "PASS");
    }
    int fails32 = 0;
    {
        static int8_t A[32][32], B[32][32];
        for (auto &r : A) for (auto &v : r) v = (int8_t)(rand() % 256 - 128);
        for (auto &r : B) for (auto &v : r) v = (int8_t)(rand() % 256 - 128);
        int8_t ha[64][16], hb[64][16];
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 16; ++j) {
                ha[l][j] = A[l & 31][16 * (l >> 5) + j];
                hb[l][j] = B[16 * (l >> 5) + j][l & 31];
            }
        v4i *da, *db; v16i *dd;
        CK(hipMalloc(&da, 1024)); CK(hipMalloc(&db, 1024)); CK(hipMalloc(&dd, 4096));
        CK(hipMemcpy(da, ha, 1024, hipMemcpyHostToDevice));
        CK(hipMemcpy(db, hb, 1024, hipMemcpyHostToDevice));
        probe32<<<1, 64>>>(da, db, dd);
        int hd[64][16];
        CK(hipMemcpy(hd, dd, 4096, hipMemcpyDeviceToHost));
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 16; ++i) {
                const int row = (i & 3) + 8 * (i >> 2) + 4 * (l >> 5), col = l & 31;
                int ref = 0;
                for (int k = 0; k < 32; ++k) ref += (int)A[row][k] * (int)B[k][col];
                if (ref != hd[l][i]) { if (fails32 < 5) printf("32x32x32 mismatch lane %d reg %d: %d vs %d\n", l, i, hd[l][i], ref); ++fails32; }
            }
        printf("mfma_i32_32x32x32_i8: %s\n", fails32 ? "FAIL" : "PASS");
    }
    return (fails || fails32) ? 1 : 0;
}
