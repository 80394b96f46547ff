// Kernel B of the fused forward pass: conc1 -> (C3_1 || C3_2) -> requant -> concat (LDS only) ->
// C4 -> applyRes_y residual add + clamp -> I1.x_rec (see fused_common.h for the overall design).
//
// Persistent, wave-specialised like kernel A: one 768-thread workgroup per CU (three waves per SIMD)
// walks 18 x 30 output tiles through a three-stage pipeline with one workgroup barrier per step.  In
// step n
//   waves 0-3  (movers)    load the conc1 tile n from HBM straight into LDS buffer n&1 (direct
//                          global -> LDS loads; frame-edge tiles through registers), then run C4 +
//                          residual on the conc2 tile n-2 (LDS);
//   waves 4-11 (C3 waves)  run C3_1 || C3_2 + requant on the conc1 tile n-1 (LDS buffer (n-1)&1),
//                          5 region rows x 16 columns each, and write the conc2 tile n-1 into LDS
//                          buffer (n-1)&1.
// Both LDS tiles keep dense 48-byte pixels: the conc1 tile is byte-for-byte a copy of 22 row segments
// of the HBM image, which is what lets the movers load it with direct global -> LDS instructions;
// bank conflicts of the B-fragment reads are avoided by the lane -> pixel and k-group -> chunk maps
// of qcnn_internal.h.
#define STAMP_ROW (wv + 8)  // diagnostic builds: rows 8..19 of the stamp buffer (kernel A owns 0..7)
#include "fused_common.h"

namespace qcnn {

constexpr int C3_WAVES = 8;                       // waves 4 .. 4 + C3_WAVES - 1 run the C3 stage
constexpr int C3_WROWS = kb::C3R / (C3_WAVES / 2);  // region rows per C3 wave (x 16 columns)
constexpr int C3_WIN = C3_WROWS + 2;              // input rows per C3 wave
constexpr int C3_STEPS = 3 * C3_WIN;
constexpr int KB_THREADS = 64 * (4 + C3_WAVES);
static_assert(C3_WROWS * (C3_WAVES / 2) == kb::C3R, "the C3 waves tile the region rows");
constexpr int TILE_ROT = 37;    // rotation of the workgroup -> tile map between rounds
#ifndef C3_PREFETCH_N
#define C3_PREFETCH_N 4
#endif
#ifndef C4_PREFETCH_N
#define C4_PREFETCH_N 4
#endif
constexpr int C3_PREFETCH = C3_PREFETCH_N;  // B fragments requested ahead of their MFMAs in the C3 stage
constexpr int C4_PREFETCH = C4_PREFETCH_N;  // the same in the C4 stage

__device__ __forceinline__ void step_barrier_b()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <bool EXACT>
__global__ __launch_bounds__(KB_THREADS) void k_c3c4(const FusedArgs a)
{
    using namespace kb;
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
    unsigned char *sInBase = smem;                   // [2][INR][INC][64] conc1 tiles
    unsigned char *sC3Base = smem + 2 * SIN_BYTES;   // [2][C3R][C3C][48] conc2 tiles (+ slack)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = a.H, W = a.W;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int ntiles = tiles_x * tiles_y * a.nframes;
    const int first = blockIdx.x, stride = gridDim.x;
    const int N = (ntiles + stride - 1) / stride;  // rounds: every workgroup takes at most one tile per round
    // The position inside a round rotates by TILE_ROT per round, so no workgroup is handed a
    // frame-edge column in every round.  p0 / p1 / p2 = the tiles of rounds n, n-1, n-2 in step n.
    TileWalker walk;
    walk.init(tiles_x, tiles_y, a.nframes, first, stride, TILE_ROT);
    TilePos p0 = {0, 0, 0, false}, p1 = p0, p2 = p0;
    const int g = lane >> 4;
    const int c = kb_pixel_of_lane(lane & 15);  // column of this lane inside a wave's 16-pixel tile
    const int gc = kb_chunk_of_kgroup(g);       // 16-byte channel chunk this lane's k-group reads
    STAMP_INIT();

    if (wv < 4) {
        // ================================ movers ================================
        const int p = wv;
        const int mtid = p * 64 + lane;  // 0..255
        const v4i *w4 = reinterpret_cast<const v4i *>(a.wpack + PackedOffsets::C4W) + lane;
        const v4i f4[3] = {w4[0], w4[64], w4[128]};
        const float b4 = a.bias_f[BIAS_OFF[5]];
        const int b4i = EXACT ? 0 : a.bias_i[BIAS_OFF[5]];  // integer path (a zero accumulator start costs the MFMA nothing)
        const int mul = a.rq[5].mul, shift = a.rq[5].shift;
        constexpr int PIECES_ROW = INC * 3;  // 16-byte pieces per tile row
        constexpr int PIECES = INR * PIECES_ROW;
        constexpr int ITER = (PIECES + 255) / 256;
        static_assert((ITER - 1) * 256 <= PIECES, "only the last piece of a lane may fall outside the tile");
        // The conc1 tile is dense in LDS ([INR][INC][48] bytes), and so is each of its rows in HBM:
        // piece pi = 256 it + 64 p + lane is LDS bytes 16 pi .., so one wave instruction moves 64
        // consecutive pieces and can be a direct global -> LDS load.  Per lane and `it`: the
        // tile-invariant global offset (relative to the tile's first halo pixel) and
        // row | column << 8 | chunk << 16 for frame-edge tiles.
        int goff[ITER], rpc[ITER];
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int pi = it * 256 + mtid;
            const int row = pi / PIECES_ROW, rem = pi - row * PIECES_ROW;
            const int px = rem / 3, ch = rem - px * 3;
            goff[it] = (row * conc_wp(W) + px) * 48 + ch * 16;
            rpc[it] = row | (px << 8) | (ch << 16);
        }
        const bool last_valid = (ITER - 1) * 256 + mtid < PIECES;
        const int cb = (p & 1) * 16, O0 = (p >> 1) * 9;  // C4: wave = 9 output rows x 16 columns
        unsigned char ynext[9] = {}, ynext_in[9] = {};
        for (int n = 0; n <= N + 1; ++n) {
            p2 = p1;
            p1 = p0;
            p0 = walk.pos;
            if (n >= N) p0.valid = false;
            walk.next();
            STAMP();
            // (1) the conc1 tile n: rows y0-2 .. y0+TH+1, cols x0-2 .. x0+TW+1, zero outside the frame.
            // A tile whose halo lies inside the frame is loaded straight into LDS buffer n&1; a
            // frame-edge tile is handled in (3).
            const bool tl = p0.valid;
            unsigned char *sIn = sInBase + (n & 1) * SIN_BYTES;
            bool direct = false;
            const int8_t *org = a.conc1;
            if (tl) {
                const int x0 = p0.tx * TW, y0 = p0.ty * TH;
                direct = y0 >= 2 && y0 - 2 + INR <= H && x0 >= 2 && x0 - 2 + INC <= W;
                org = a.conc1 + QCNN_CONC_OFF(a, p0.f) + ((ptrdiff_t)y0 * conc_wp(W) + x0) * 48;  // image (y0-2, x0-2)
            }
            auto request = [&](auto itc) {  // direct load of the 64 pieces this wave owns in group `it`
                constexpr int it = decltype(itc)::value;
                if (it + 1 < ITER || last_valid)
                    __builtin_amdgcn_global_load_lds(org + goff[it], sIn + (it * 256 + p * 64) * 16, 16, 0, 0);
            };
            // the Y pixels the residual add of the NEXT step needs (tile n-1), clamped, branch-free;
            // requested before the tile loads so that nothing younger has to land first
            {
                if (p1.valid) {
                    const int ty = p1.ty;
                    const uint8_t *yf = a.y + (size_t)p1.f * a.frame_stride;
                    const int gx = p1.tx * TW + cb + c;
                    const int gxc = gx < W ? gx : W - 1;
#pragma unroll
                    for (int y = 0; y < 9; ++y) {
                        const int gy = ty * TH + O0 + y;
                        ynext_in[y] = yf[(size_t)(gy < H ? gy : H - 1) * W + gxc];
                    }
                }
            }
            // the tile loads go out first: their issue stalls (about 100 cycles each) cost no issue
            // slot, and they push C4 behind the C3 waves' MFMA phase, next to their VALU epilogue
            if (direct) static_for<ITER>(request);
            STAMP();
            // (2) C4 (3x3, 48 -> 1) + applyRes_y on the conc2 tile n-2.  The three kernel rows r are
            // MFMA rows 0..2: D_yy[r][pixel] = sum_s W4[r][s] . X[yy][pixel+s], and out[y] = D_y[0] +
            // D_{y+1}[1] + D_{y+2}[2] stays inside one lane.
            if (p2.valid) {
                const int f = p2.f;
                const int x0 = p2.tx * TW, y0 = p2.ty * TH;
                uint8_t *recf = a.rec + (size_t)f * a.frame_stride;
                const unsigned char *sC3 = sC3Base + (n & 1) * SC3_BYTES;
                const int ox = cb + c;
                const int gx = x0 + ox;
                unsigned char ypix[9];  // residual input: requested one step ago, landed before that step's barrier
#pragma unroll
                for (int y = 0; y < 9; ++y) ypix[y] = ynext[y];
                const unsigned char *rowbase = sC3 + (O0 * C3C + cb + c) * PX + gc * 16;  // chunk 3: zero weights
                v4i D[11];
#pragma unroll
                for (int yy = 0; yy < 11; ++yy) D[yy] = v4i{0, 0, 0, 0};
                // step I = 11 s + yy; LDS reads run C4_PREFETCH steps ahead; every third step also
                // issues one group of the direct tile loads
                v4i Bq[33];
#pragma unroll
                for (int I = 0; I < C4_PREFETCH; ++I)
                    Bq[I] = *reinterpret_cast<const v4i *>(rowbase + ((I % 11) * C3C + I / 11) * PX);
                __builtin_amdgcn_sched_barrier(0);
                static_for<33>([&](auto ic) {
                    constexpr int I = decltype(ic)::value;
                    constexpr int J = I + C4_PREFETCH;
                    if constexpr (J < 33) Bq[J] = *reinterpret_cast<const v4i *>(rowbase + ((J % 11) * C3C + J / 11) * PX);
                    D[I % 11] = mfma16(f4[I / 11], Bq[I], D[I % 11]);
                    __builtin_amdgcn_sched_barrier(0);
                });
                STAMP();
                if (g == 0 && ox < TW && gx < W) {
                    uint8_t *dst = recf + (size_t)(y0 + O0) * W + gx;
                    // all nine results first, each in a register of its own, then the stores: a store's data
                    // register cannot be rewritten before the memory pipeline has read it
                    uint8_t out[9];
#pragma unroll
                    for (int y = 0; y < 9; ++y) {
                        const int acc = D[y][0] + D[y + 1][1] + D[y + 2][2] + b4i;
                        if constexpr (EXACT) {
                            const float u = (float)acc + b4;  // C4.u, inference/cnn.cu:148-155
                            int temp = (int)u;                // applyRes_GPU_y, inference/cnn.cu:515-520
                            temp = (int)((unsigned)temp * (unsigned)mul + (1u << (shift - 1))) >> shift;
                            const short rec = (short)((int)ypix[y] + temp);
                            out[y] = rec > 255 ? (uint8_t)255 : rec < 0 ? (uint8_t)0 : (uint8_t)rec;
                        } else {
                            // the same in integers: the host has proved that acc (bias included) and the
                            // multiplier fit 24 bits, that the product does not wrap and that the sum with
                            // the pixel fits a short
                            const int temp = (__mul24(acc, mul) + (1 << (shift - 1))) >> shift;
                            const int rec = (int)ypix[y] + temp;
                            out[y] = (uint8_t)min(max(rec, 0), 255);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const int nrows = H - (y0 + O0);
#pragma unroll
                    for (int y = 0; y < 9; ++y)
                        if (y < nrows) dst[(size_t)y * W] = out[y];
                }
            }
            STAMP();
            // (3) a frame-edge tile goes through registers: clamped addresses (branch-free), the pieces
            // outside the frame zeroed, then LDS buffer n&1.  Rare, so its load latency is not hidden.
            if (tl && !direct) {
                const int x0 = p0.tx * TW, y0 = p0.ty * TH;
                const int8_t *frame = a.conc1 + QCNN_CONC_OFF(a, p0.f);
                v4i v[ITER];
#pragma unroll
                for (int it = 0; it < ITER; ++it) {
                    const int gy = y0 - 2 + (rpc[it] & 0xff), gx = x0 - 2 + ((rpc[it] >> 8) & 0xff);
                    const int cy = gy < 0 ? 0 : gy >= H ? H - 1 : gy, cx = gx < 0 ? 0 : gx >= W ? W - 1 : gx;
                    v[it] = *reinterpret_cast<const v4i *>(frame + ((size_t)(cy + 2) * conc_wp(W) + cx + 2) * 48 + (rpc[it] >> 16) * 16);
                    if (cy != gy || cx != gx) v[it] = v4i{0, 0, 0, 0};  // zero padding of C3
                }
#pragma unroll
                for (int it = 0; it < ITER; ++it)
                    if (it + 1 < ITER || last_valid) *reinterpret_cast<v4i *>(sIn + (it * 256 + mtid) * 16) = v[it];
            }
            STAMP();
            // the direct loads write LDS asynchronously: they must have landed before the barrier
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
#pragma unroll
            for (int y = 0; y < 9; ++y) ynext[y] = ynext_in[y];
            step_barrier_b();
        }
    } else {
        // ================================ C3 waves ================================
        // C3_1 (3x3, 48 -> 16) and C3_2 (1x1, 48 -> 32) on the C3R x C3C region; wave = C3_WROWS region
        // rows x 16 columns x 48 channels; two C3 waves per SIMD share the VALU in their epilogues.  The 11 A fragments stay in registers for the whole kernel.
        const int cw = wv - 4;
        const int cb = (cw & 1) * 16, R0 = (cw >> 1) * C3_WROWS;
        v4i f31[3][3], f32[2];
        {
            const v4i *w31 = reinterpret_cast<const v4i *>(a.wpack + PackedOffsets::C3W31) + lane;
            const v4i *w32 = reinterpret_cast<const v4i *>(a.wpack + PackedOffsets::C3W32) + lane;
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int r = 0; r < 3; ++r) f31[s][r] = w31[(s * 3 + r) * 64];
            f32[0] = w32[0];
            f32[1] = w32[64];
        }
        const Requant q31 = a.rq[3], q32 = a.rq[4];
        float bf[12];
        if constexpr (EXACT) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                bf[i] = a.bias_f[BIAS_OFF[3] + 4 * g + i];
                bf[4 + i] = a.bias_f[BIAS_OFF[4] + 4 * g + i];
                bf[8 + i] = a.bias_f[BIAS_OFF[4] + 16 + 4 * g + i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 12; ++i) bf[i] = 0.f;
        }
        v4i b0 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0}, b2 = {0, 0, 0, 0};  // integer path: bias starts the accumulator
        if constexpr (!EXACT) {
            b0 = *reinterpret_cast<const v4i *>(a.bias_i + BIAS_OFF[3] + 4 * g);
            b1 = *reinterpret_cast<const v4i *>(a.bias_i + BIAS_OFF[4] + 4 * g);
            b2 = *reinterpret_cast<const v4i *>(a.bias_i + BIAS_OFF[4] + 16 + 4 * g);
        }
        for (int n = 0; n <= N + 1; ++n) {
            STAMP();
            p1 = p0;
            p0 = walk.pos;
            if (n >= N) p0.valid = false;
            walk.next();
            if (p1.valid) {
                const int x0 = p1.tx * TW, y0 = p1.ty * TH;
                const unsigned char *sIn = sInBase + ((n - 1) & 1) * SIN_BYTES;
                unsigned char *sC3 = sC3Base + ((n - 1) & 1) * SC3_BYTES;
                v4i acc31[C3_WROWS], acc32[2][C3_WROWS];
#pragma unroll
                for (int y = 0; y < C3_WROWS; ++y) {
                    acc31[y] = b0;
                    acc32[0][y] = b1;
                    acc32[1][y] = b2;
                }
                // region (i, jj) <-> image (y0-1+i, x0-1+jj); tap (r, s) reads tile (i+r, jj+s).
                // Step I = C3_WIN s + yy issues every MFMA that uses the B fragment of (input row yy,
                // column shift s) and requests the fragment of step I + C3_PREFETCH; the scheduling
                // fence after each step keeps that many LDS reads in flight ahead of their use.
                // chunk 3 (zero weights) is the first chunk of the next pixel
                const unsigned char *rowbase = sIn + (R0 * INC + cb + c) * PX + gc * 16;
                const unsigned char *colbase[3];
#pragma unroll
                for (int s = 0; s < 3; ++s) colbase[s] = rowbase + s * PX;
                v4i Bq[C3_STEPS];
#pragma unroll
                for (int I = 0; I < C3_PREFETCH; ++I)
                    Bq[I] = *reinterpret_cast<const v4i *>(colbase[I / C3_WIN] + (I % C3_WIN) * (INC * PX));
                __builtin_amdgcn_sched_barrier(0);
                static_for<C3_STEPS>([&](auto ic) {
                    constexpr int I = decltype(ic)::value;
                    constexpr int s = I / C3_WIN, yy = I % C3_WIN;
                    constexpr int J = I + C3_PREFETCH;
                    if constexpr (J < C3_STEPS) Bq[J] = *reinterpret_cast<const v4i *>(colbase[J / C3_WIN] + (J % C3_WIN) * (INC * PX));
                    const v4i B = Bq[I];
                    constexpr int lo = yy - 2 > 0 ? yy - 2 : 0, hi = yy < C3_WROWS - 1 ? yy : C3_WROWS - 1;  // output rows fed
#pragma unroll
                    for (int y = lo; y <= hi; ++y) acc31[y] = mfma16(f31[s][yy - y], B, acc31[y]);
                    if constexpr (s == 1 && yy >= 1 && yy <= C3_WROWS) {
                        acc32[0][yy - 1] = mfma16(f32[0], B, acc32[0][yy - 1]);
                        acc32[1][yy - 1] = mfma16(f32[1], B, acc32[1][yy - 1]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
                STAMP();
                const int jj = cb + c;
                const int gx = x0 - 1 + jj, gy0 = y0 - 1 + R0;
                unsigned char *dstbase = sC3 + (R0 * C3C + jj) * PX + 12 * g;
                auto quant_row = [&](int y, unsigned &d0, unsigned &d1, unsigned &d2) {
                    if constexpr (!EXACT) {
                        requant12_int(acc32[0][y], acc32[1][y], acc31[y], q32, q31, d1, d2, d0);
                    } else {
                        d0 = quant4<EXACT>(acc31[y][0], acc31[y][1], acc31[y][2], acc31[y][3], bf[0], bf[1], bf[2], bf[3], q31);
                        d1 = quant4<EXACT>(acc32[0][y][0], acc32[0][y][1], acc32[0][y][2], acc32[0][y][3], bf[4], bf[5], bf[6], bf[7],
                                           q32);
                        d2 = quant4<EXACT>(acc32[1][y][0], acc32[1][y][1], acc32[1][y][2], acc32[1][y][3], bf[8], bf[9], bf[10],
                                           bf[11], q32);
                    }
                };
                if (gy0 >= 0 && gy0 + C3_WROWS <= H && x0 - 1 + cb >= 0 && x0 - 1 + cb + 16 <= W) {
                    // the wave's region pixels are all inside the frame
#pragma unroll
                    for (int y = 0; y < C3_WROWS; ++y) {
                        unsigned d0, d1, d2;
                        quant_row(y, d0, d1, d2);
                        unsigned *dst = reinterpret_cast<unsigned *>(dstbase + y * (C3C * PX));
                        dst[0] = d0;
                        dst[1] = d1;
                        dst[2] = d2;
                    }
                } else {
#pragma unroll
                    for (int y = 0; y < C3_WROWS; ++y) {
                        unsigned d0, d1, d2;
                        quant_row(y, d0, d1, d2);
                        const int gy = gy0 + y;
                        if (!(gy >= 0 && gy < H && gx >= 0 && gx < W)) d0 = d1 = d2 = 0;  // zero padding of the C4 input
                        unsigned *dst = reinterpret_cast<unsigned *>(dstbase + y * (C3C * PX));
                        dst[0] = d0;
                        dst[1] = d1;
                        dst[2] = d2;
                    }
                }
            }
            STAMP();
            step_barrier_b();
        }
    }
}

hipError_t launch_fused_c3c4(const FusedArgs &a, bool exact, hipStream_t s)
{
    const int ntiles = ((a.W + kb::TW - 1) / kb::TW) * ((a.H + kb::TH - 1) / kb::TH) * a.nframes;
    if (ntiles < 1) return hipSuccess;
    const int ncu = a.ncu > 0 ? a.ncu : 256;
    dim3 grid(ntiles < ncu ? ntiles : ncu);
    if (exact) hipLaunchKernelGGL(k_c3c4<true>, grid, dim3(KB_THREADS), 0, s, a);
    else hipLaunchKernelGGL(k_c3c4<false>, grid, dim3(KB_THREADS), 0, s, a);
    return hipGetLastError();
}

}  // namespace qcnn
