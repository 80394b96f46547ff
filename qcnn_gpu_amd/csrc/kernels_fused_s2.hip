// Kernel B of the fused forward pass, streaming form with TWO column tiles per wave (the default; QCNN_B_WIDE=0
// selects the one-tile form of kernels_fused_s.hip, which this file follows step by step).
//
// A wave owns a strip of 30 output columns: two 16-pixel MFMA column tiles side by side (region pixels 0..31 = image
// columns x0-1 .. x0+30, conc1 columns x0-2 .. x0+31 = 34 pixels = 1632 contiguous bytes per row of the bordered
// image).  Against the one-tile form a strip reads 34 conc1 columns for 30 outputs instead of 18 for 14 (x 1.13
// instead of x 1.29), the twelve weight fragments serve both tiles, and the scalar bookkeeping, the waits and the
// row requests are paid once per 30 columns.  The price is registers (two sets of accumulators and fragments: two
// waves per SIMD instead of three) and LDS (a ring row is 1632 bytes: 20 KB per wave, so workgroups are single
// waves and eight fit a CU, with a Y ring of R instead of 2 R slots).
//
// What differs from the one-tile form:
//   * a conc1 row is fetched with two LDS-direct loads (pieces 0..63 and 64..101), the Y pixels of 30 outputs with
//     one; a row issues five vector memory instructions (two byte stores, two conc1 loads, one Y load).  The Y
//     pixels of output row y are requested R rows ahead (LDS has no room for more), so the hand-counted wait at the
//     top of a row is vmcnt(25) = 5 * 5: everything but the last five rows' worth has landed (10 in the last two
//     steps, whose rows issue two or three instructions);
//   * the C4 gathers of tile 0 reach into tile 1: out[p] = P0[p] + P1[p+1] + P2[p+2] needs P1[16], P2[16], P2[17] for
//     p = 14, 15.  The lanes of tile 0 that would hold P1[0], P2[0], P2[1] serve no output of this strip (they belong
//     to the column left of it), so they are overwritten with tile 1's values of the same lanes and the gather
//     addresses wrap modulo 16: one select, and tile 0 yields all 16 of its outputs; tile 1 yields 14.
//   * every phase of the software-pipelined row runs for tile 0, then for tile 1.
//
// Scheduling (measured with the diagnostic builds of tools/stamps_s2.py):
//   * a launch is sized to ONE round of waves where it can be (2048 units on 256 CUs), and the launch lasts as long
//     as its slowest wave.  The two waves of a SIMD are not equals: the older one wins every arbitration and used to
//     finish after 0.70 of the time the younger one needed, which then ran alone on a SIMD it cannot fill.  The
//     younger wave therefore runs every other row at priority 1: both finish together.
//   * the strips at the left and right frame edge used to run every row through the tested form and ended 20 % after
//     the others; they now run the steps in the middle of a band through a form that only masks the region columns
//     outside the frame.
//   * on the integer path the requantiser's two instructions per value sit one value behind every MFMA (rowi below),
//     which is what the matrix pipe lets the vector port issue in an MFMA's shadow.
#define STAMP_ROW (wv + 12)
#include <climits>

#include "fused_common.h"
#ifdef QCNN_STAMPS_NOPH  // diagnostic builds that time whole workgroups only
#undef STAMP
#define STAMP() do { } while (0)
#endif

namespace qcnn {

namespace ks2 {
constexpr int NT = 2;                         // column tiles per wave
constexpr int SW = 30;                        // output columns of a strip
constexpr int CW = SW + 4;                    // conc1 columns a strip reads (halo 2)
constexpr int R = 6;                          // conc1 rows per step of the unrolled row loop
constexpr int ROW_BYTES = CW * 48;            // one ring row: 1632 contiguous bytes of the HBM image
constexpr int ROW_PIECES = ROW_BYTES / 16;    // 102 16-byte pieces: one load of 64 lanes, one of 38
constexpr int GROUP_BYTES = R * ROW_BYTES;
constexpr int RING_BYTES = 2 * GROUP_BYTES + 64;  // two groups of R rows (+ slack: a fragment read runs 16 bytes past its pixel)
constexpr int YSLOT = SW * 4;                 // Y pixels of one output row of a strip, a dword each
constexpr int YRING_BYTES = R * YSLOT;        // Y ring: R slots (the Y pixels travel R rows ahead of their use, conc1 2 R rows)
constexpr int WG_LDS = RING_BYTES + YRING_BYTES;  // a workgroup is one wave
constexpr int WAVES_PER_CU = 8;               // two per SIMD: what the registers allow, and LDS just holds (8 x 20 368 bytes)
static_assert(ROW_PIECES > 64 && ROW_PIECES <= 128, "a row is fetched with two instructions");
static_assert(WAVES_PER_CU * WG_LDS <= 160 * 1024, "eight waves per CU");
}  // namespace ks2

namespace {
using namespace ks2;

// lane (of lane row g) that holds region pixel `pix` (0..15) of a tile: the inverse of kb_pixel_of_lane
__device__ __forceinline__ int lane_of_pixel2(int pix, int g)
{
    const int c = (pix & 1) == 0 ? 4 + (pix >> 1) : pix < 8 ? (pix >> 1) : 12 + ((pix - 9) >> 1);
    return 16 * g + c;
}
}  // namespace

template <bool EXACT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_c3c4s2(const FusedArgs a, int nstrips, int nseg, int seg_rows, int total_units, int second_from)
{
#ifdef QCNN_ABL_LONE  // diagnostic builds only: LDS for four workgroups per CU, i.e. a wave has its SIMD to itself
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * WG_LDS];
#else
    __shared__ __attribute__((aligned(16))) unsigned char smem[WG_LDS];
#endif
    const int lane = threadIdx.x & 63;
    // Workgroups (single waves) are dealt to the 8 XCDs round robin; XCD x takes the x-th eighth of the units, so that
    // strips that share conc1 columns run on one XCD at about the same time and meet in its L2.
    const int per_xcd = gridDim.x >> 3;
    int unit = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (unit >= total_units) return;
    // In a launch of one round the workgroups an XCD receives first land one on every SIMD, the later ones second:
    // those are the younger waves (second_from = SIMDs of an XCD; INT_MAX in a launch of several rounds).
    const bool young = (int)(blockIdx.x >> 3) >= second_from;
#ifdef QCNN_STAMPS  // diagnostic builds only: when and where this workgroup ran
    unsigned long long blk_t0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(blk_t0)::"memory");
#endif
    // (the divisions run on the vector ALU: readfirstlane brings the wave-uniform results back to scalar registers,
    // which the buffer descriptors and scalar offsets below need)
    const int strip = __builtin_amdgcn_readfirstlane(unit % nstrips);
    unit /= nstrips;
    const int seg = __builtin_amdgcn_readfirstlane(unit % nseg), f = __builtin_amdgcn_readfirstlane(unit / nseg);
    const int H = a.H, W = a.W;
    const int x0 = strip * SW, ya = seg * seg_rows;
    const int yb = ya + seg_rows < H ? ya + seg_rows : H;  // output rows ya .. yb-1
    if (ya >= yb) return;
    unsigned char *ring = smem;                 // conc1 rows: two groups of R
    unsigned char *yring = smem + RING_BYTES;   // Y rows: R x 30 dwords
    const int g = lane >> 4;
    const int pix = kb_pixel_of_lane(lane & 15);  // pixel of this lane inside its tile: image column x0 - 1 + 16 T + pix
    const int gc = kb_chunk_of_kgroup(g);

    // A fragments: resident for the whole strip, shared by both tiles
    v4i f31[3][3], f32[2], f4;
    {
        const v4i *w31 = reinterpret_cast<const v4i *>(a.wpack + PackedOffsets::C3W31) + lane;
        const v4i *w32 = reinterpret_cast<const v4i *>(a.wpack + PackedOffsets::C3W32) + lane;
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int r = 0; r < 3; ++r) f31[s][r] = w31[(s * 3 + r) * 64];
        f32[0] = w32[0];
        f32[1] = w32[64];
        f4 = reinterpret_cast<const v4i *>(a.wpack + PackedOffsets::C4W9)[lane];
    }
    const Requant q31 = a.rq[3], q32 = a.rq[4];
    float bf[12];
    v4i b0 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0}, b2 = {0, 0, 0, 0};  // integer path: bias starts the accumulator
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
        b0 = *reinterpret_cast<const v4i *>(a.bias_i + BIAS_OFF[3] + 4 * g);
        b1 = *reinterpret_cast<const v4i *>(a.bias_i + BIAS_OFF[4] + 4 * g);
        b2 = *reinterpret_cast<const v4i *>(a.bias_i + BIAS_OFF[4] + 16 + 4 * g);
    }
    const float b4 = a.bias_f[BIAS_OFF[5]];
    const int b4i = EXACT ? 0 : a.bias_i[BIAS_OFF[5]];
    const int mul = a.rq[5].mul, shift = a.rq[5].shift;

    // Staging as in the one-tile form: conc1 row m of this band is row ya + m of the bordered image, the strip's first
    // column x0 - 2 is column x0 of it; the row goes into slot m % 2R of the ring, requested while conc1 row m - 2R is
    // processed.  Every vector memory instruction of the main loop is inline assembly and the waits are counted by
    // hand: a row issues exactly five (store of tile 0, store of tile 1, conc1 pieces 0..63, conc1 pieces 64..101, Y),
    // completion is in issue order, and row m needs the conc1 loads row m - 11 issued and the Y load row m - R
    // issued, the last instruction of its row: behind it 5 * 5 = 25 instructions may still be in flight.  (A strip at the
    // right edge of the frame reads up to 29 pixels past the end of an image row, the last row of an image past the
    // image: the buffer descriptor ends with the frame's image and answers such pieces with zeros; they feed region
    // columns outside the frame, which are masked.)
    const int row_bytes = conc_wp(W) * 48;
    const __amdgpu_buffer_rsrc_t conc_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<int8_t *>(a.conc1 + QCNN_CONC_OFF(a, f)), 0, (int)a.conc_stride, 0x00020000);
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t *>(a.y + (size_t)f * a.frame_stride), 0, H * W, 0x00020000);
    const __amdgpu_buffer_rsrc_t rec_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(a.rec + (size_t)f * a.frame_stride, 0, H * W, 0x00020000);
    constexpr int OOB = 0x7f000000;
    const int lane16 = lane * 16, lane16b = lane * 16 + 1024;
    const int y_voff = (lane < SW && x0 + lane < W) ? lane : OOB;
    constexpr unsigned long long tail_lanes = (1ull << (ROW_PIECES - 64)) - 1, y_lanes = (1ull << SW) - 1;

    const unsigned ring_s = __builtin_amdgcn_readfirstlane((unsigned)(size_t)ring);  // LDS byte addresses
    const unsigned yring_s = __builtin_amdgcn_readfirstlane((unsigned)(size_t)yring);
    // requests conc1 row `crow` of the bordered image into row j of ring group `grp` (with_conc) and the Y pixels of
    // output row `yrow` into Y slot j
    auto request_impl = [&](const bool with_conc, int grp, int j, int crow, int yrow, bool clamp) __attribute__((always_inline)) {
        if (clamp) {  // rows past the border / outside the frame are never used: any row inside will do
            crow = crow <= H + 3 ? crow : H + 3;
            yrow = yrow < 0 ? 0 : yrow >= H ? H - 1 : yrow;
        }
#ifdef QCNN_ABL_L2LOADS  // diagnostic builds only: every load hits a few rows that stay in L2
        crow &= 7;
#endif
        const unsigned lds_c = ring_s + grp * GROUP_BYTES + j * ROW_BYTES, lds_y = yring_s + j * YSLOT;
        const int off_c = crow * row_bytes + x0 * 48, off_y = yrow * W + x0;
        unsigned long long saved;
        // (m0 is written up to three times and is listed as clobbered; between an s_mov to m0 and the LDS-direct load
        // that reads it sits one other instruction, the wait state this hazard needs.  The compiler calls m0 a reserved
        // register and warns about seeing it in a clobber list; the clobber still becomes an implicit definition.)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
        if (with_conc)
            asm volatile("s_mov_b64 %0, exec\n\t"
                         "s_mov_b32 m0, %1\n\t"
                         "s_mov_b64 exec, -1\n\t"
                         "buffer_load_dwordx4 %3, %5, %6 offen nt lds\n\t"
                         "s_mov_b32 m0, %2\n\t"
                         "s_mov_b64 exec, %11\n\t"
                         "buffer_load_dwordx4 %4, %5, %6 offen nt lds\n\t"
                         "s_mov_b32 m0, %7\n\t"
                         "s_mov_b64 exec, %12\n\t"
                         "buffer_load_ubyte %8, %9, %10 offen lds\n\t"
                         "s_mov_b64 exec, %0"
                         : "=&s"(saved)
                         : "s"(lds_c), "s"(lds_c + 1024), "v"(lane16), "v"(lane16b), "s"(conc_rsrc), "s"(off_c), "s"(lds_y),
                           "v"(y_voff), "s"(y_rsrc), "s"(off_y), "s"(tail_lanes), "s"(y_lanes)
                         : "memory", "m0");
        else
            asm volatile("s_mov_b64 %0, exec\n\t"
                         "s_mov_b32 m0, %1\n\t"
                         "s_mov_b64 exec, %5\n\t"
                         "buffer_load_ubyte %2, %3, %4 offen lds\n\t"
                         "s_mov_b64 exec, %0"
                         : "=&s"(saved)
                         : "s"(lds_y), "v"(y_voff), "s"(y_rsrc), "s"(off_y), "s"(y_lanes)
                         : "memory", "m0");
#pragma clang diagnostic pop
    };
    // the lane's read addresses: B fragments of a conc1 row (+ group * GROUP_BYTES + row * ROW_BYTES + (16 T + s) * 48),
    // the Y pixel of its output column (+ slot * YSLOT)
    const unsigned char *rd = ring + pix * 48 + gc * 16;
    const unsigned char *rdy[NT] = {yring + pix * 4, yring + (pix < SW - 16 ? 16 + pix : 0) * 4};

    // output: lane (pixel o, lane row 0) of tile T owns output column x0 + 16 T + o; tile 0 yields 16, tile 1 14
    int out_off[NT], colmask[NT];
#pragma unroll
    for (int T = 0; T < NT; ++T) {
        const bool outlane = g == 0 && 16 * T + pix < SW && x0 + 16 * T + pix < W;
        out_off[T] = outlane ? x0 + 16 * T + pix : OOB;
        // conc2 pixels outside the frame are the zero padding of C4: a region column outside the frame is masked once,
        // in the sum the gathers read
        const int cx = x0 - 1 + 16 * T + pix;
        colmask[T] = (cx >= 0 && cx < W) ? -1 : 0;
    }
    // gathers: pixel + 1 of lane row 1, pixel + 2 of lane row 2, modulo 16 (tile 0 finds tile 1's first pixels there)
    const int addr1 = 4 * lane_of_pixel2((pix + 1) & 15, 1);
    const int addr2 = 4 * lane_of_pixel2((pix + 2) & 15, 2);
    // lanes of tile 0 whose value serves no output of this strip: (pixel 0, row 1), (pixels 0 and 1, row 2)
    const bool borrow = (g == 1 && pix == 0) || (g == 2 && pix < 2);
    const bool edge_strip = x0 == 0 || x0 + SW >= W;  // uniform: some region column lies outside the frame
    const int round4 = (1 << (shift - 1)) + b4i * mul;
    int round4v;  // in a vector register: the multiply-add takes one scalar operand only
    asm volatile("v_mov_b32 %0, %1" : "=v"(round4v) : "s"(round4));

    const int T_steps = (yb - ya + 5 + R - 1) / R;
    // the ring's first 2R rows, all landed before the first row starts
#pragma unroll
    // (the Y ring has R slots: the second round of requests fetches the same Y rows again)
    for (int i = 0; i < 2 * R; ++i) request_impl(true, i / R, i % R, ya + i, ya - 5 + i % R, true);
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
    asm volatile("" ::: "memory");

    v4i acc[NT][3], a32[NT][2][2], D4p[NT];
    int Pa[NT], Pb[NT];
#pragma unroll
    for (int T = 0; T < NT; ++T) {
        acc[T][0] = acc[T][1] = acc[T][2] = v4i{0, 0, 0, 0};
        a32[T][0][0] = a32[T][0][1] = a32[T][1][0] = a32[T][1][1] = v4i{0, 0, 0, 0};
        D4p[T] = v4i{0, 0, 0, 0};
        Pa[T] = Pb[T] = 0;
    }
    v4i Bq[2][NT][3];  // B fragments of the current conc1 row (slot row & 1) and of the next one
#pragma unroll
    for (int T = 0; T < NT; ++T)
#pragma unroll
        for (int s = 0; s < 3; ++s) Bq[0][T][s] = *reinterpret_cast<const v4i *>(rd + (16 * T + s) * 48);

    int g3 = 0, g3n = 1;  // ring groups of step t: read (g3) and next (g3n); this step's requests refill g3
    STAMP_INIT();
    const int wv = 0;
    (void)wv;
    // One conc1 row for both tiles; see the one-tile form for the software pipeline.  MODE 0: a step whose six conc2
    // rows lie inside the frame and whose six output rows lie inside the band, in a strip with no column outside the
    // frame; MODE 2: such a step in a strip at the frame edge (region columns outside the frame masked); MODE 1:
    // everything tested.  This form serves the float path; the integer path runs rowi below.
    // (always inlined: called from several loops, and as a function of its own it would take its wave-uniform
    // operands in vector registers)
    auto prio = [&](auto jc) __attribute__((always_inline)) {
        if (young) {
            if constexpr (decltype(jc)::value & 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
    };
    auto row = [&](auto jc, auto modec, int t) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        constexpr int MODE = decltype(modec)::value;
        constexpr bool CHECK = MODE == 1, COLS = MODE != 0;
        const int m0 = R * t;
        prio(jc);
        const unsigned char *rdc = rd + g3 * GROUP_BYTES, *rdo = rd + g3n * GROUP_BYTES;
        static_assert(R % 2 == 0 && R % 3 == 0, "the fragment and accumulator slots repeat with the step");
        v4i(&B)[NT][3] = Bq[j & 1];
        v4i(&Bn)[NT][3] = Bq[(j + 1) & 1];
        if (CHECK && t >= T_steps - 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(25)" ::: "memory");
        int yv[NT];
#pragma unroll
        for (int T = 0; T < NT; ++T) yv[T] = *reinterpret_cast<const int *>(rdy[T] + j * YSLOT);
        // the fragments of the next row (the first row of the other group after the last row of this one)
#pragma unroll
        for (int T = 0; T < NT; ++T)
#pragma unroll
            for (int s = 0; s < 3; ++s)
                Bn[T][s] = *reinterpret_cast<const v4i *>((j + 1 < R ? rdc + (j + 1) * ROW_BYTES : rdo) + (16 * T + s) * 48);
        __builtin_amdgcn_sched_barrier(0);
        STAMP();
        constexpr int s0 = j % 3, s1 = (j + 2) % 3, s2 = (j + 1) % 3;
        const int y2 = ya - 3 + m0 + j;
#pragma unroll
        for (int T = 0; T < NT; ++T) {
            acc[T][s2] = mfma16(f31[0][2], B[T][0], acc[T][s2]);
            acc[T][s1] = mfma16(f31[0][1], B[T][0], acc[T][s1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        // C4 of conc2 row y2 - 1 (D4p): kernel row r of output row y2 - 1 - r + 1; output row y3 = y2 - 2 is complete
        // with it.  Kernel columns: out[o] = P_0[o] + P_1[o + 1] + P_2[o + 2] (lane rows 0, 1, 2).
        int P[NT], v1[NT], v2[NT];
#pragma unroll
        for (int T = 0; T < NT; ++T) {
            P[T] = Pa[T] + D4p[T][2];
            if constexpr (COLS) P[T] &= colmask[T];
            Pa[T] = Pb[T] + D4p[T][1];
            Pb[T] = D4p[T][0];
        }
        {
            const int M0 = borrow ? P[1] : P[0];  // tile 0's gathers wrap into tile 1's first pixels
            v1[0] = __builtin_amdgcn_ds_bpermute(addr1, M0);
            v2[0] = __builtin_amdgcn_ds_bpermute(addr2, M0);
            v1[1] = __builtin_amdgcn_ds_bpermute(addr1, P[1]);
            v2[1] = __builtin_amdgcn_ds_bpermute(addr2, P[1]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int T = 0; T < NT; ++T) {
            acc[T][s2] = mfma16(f31[1][2], B[T][1], acc[T][s2]);
            acc[T][s1] = mfma16(f31[1][1], B[T][1], acc[T][s1]);
            acc[T][s2] = mfma16(f31[2][2], B[T][2], acc[T][s2]);
            acc[T][s1] = mfma16(f31[2][1], B[T][2], acc[T][s1]);
            acc[T][s0] = mfma16(f31[0][0], B[T][0], b0);
            a32[T][j % 2][0] = mfma16(f32[0], B[T][1], b1);  // C3_2 of conc2 row y1 (its centre pixel): used one row later
        }
        __builtin_amdgcn_sched_barrier(0);
        v4i c31[NT], c320[NT], c321[NT];
        unsigned d0[NT], d1[NT], d2[NT];
#pragma unroll
        for (int T = 0; T < NT; ++T) c31[T] = acc[T][s2], c320[T] = a32[T][(j + 1) % 2][0], c321[T] = a32[T][(j + 1) % 2][1];
#define REQ_PIECE2(I, LAST)                                                                                              \
    do {                                                                                                                 \
        requant12_piece<I, false>(c320[0][I], c321[0][I], c31[0][I], q32, q31, d1[0], d2[0], d0[0]);                      \
        requant12_piece<I, LAST>(c320[1][I], c321[1][I], c31[1][I], q32, q31, d1[1], d2[1], d0[1]);                       \
    } while (0)
        if constexpr (!EXACT) REQ_PIECE2(0, false);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int T = 0; T < NT; ++T) {
            acc[T][s0] = mfma16(f31[1][0], B[T][1], acc[T][s0]);
            a32[T][j % 2][1] = mfma16(f32[1], B[T][1], b2);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!EXACT) REQ_PIECE2(1, false);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int T = 0; T < NT; ++T) acc[T][s0] = mfma16(f31[2][0], B[T][2], acc[T][s0]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!EXACT) REQ_PIECE2(2, false);
        __builtin_amdgcn_sched_barrier(0);
        STAMP();
        // ---- applyRes_y of output row y3 = y2 - 2 (stored only if it lies in the band), then the requests for the
        // slots this row has emptied
        const int y3 = y2 - 2;
#pragma unroll
        for (int T = 0; T < NT; ++T) {
            const int sum = P[T] + v1[T] + v2[T];  // C4 accumulator (its bias sits in round4)
            int out;
            if constexpr (EXACT) {
                const float u = (float)sum + b4;  // C4.u, inference/cnn.cu:148-155 (b4i is 0 on this path)
                int temp = (int)u;                // applyRes_GPU_y, inference/cnn.cu:515-520
                temp = (int)((unsigned)temp * (unsigned)mul + (1u << (shift - 1))) >> shift;
                const short rec = (short)(yv[T] + temp);
                out = rec > 255 ? 255 : rec < 0 ? 0 : rec;
            } else {
                const int temp = (__mul24(sum, mul) + round4v) >> shift;
                out = min(max(yv[T] + temp, 0), 255);
            }
            const int voff = (!CHECK || (y3 >= ya && y3 < yb)) ? out_off[T] : OOB;
            asm volatile("buffer_store_byte %0, %1, %2, %3 offen" ::"v"(out), "v"(voff), "s"(rec_rsrc), "s"(y3 * W) : "memory");
        }
        // (rows past the last one the band walks are not requested: when the wave ends and its LDS is handed to
        // the next workgroup, no load may still be on its way into it)
        // the last step but one still fetches the Y pixels of the last step
        if (!CHECK || t <= T_steps - 3) request_impl(true, g3, j, ya + m0 + j + 2 * R, y3 + R, CHECK);
        else if (t == T_steps - 2) request_impl(false, g3, j, 0, y3 + R, true);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!EXACT) {
            REQ_PIECE2(3, true);
        } else {
#pragma unroll
            for (int T = 0; T < NT; ++T) {
                d0[T] = quant4<EXACT>(c31[T][0], c31[T][1], c31[T][2], c31[T][3], bf[0], bf[1], bf[2], bf[3], q31);
                d1[T] = quant4<EXACT>(c320[T][0], c320[T][1], c320[T][2], c320[T][3], bf[4], bf[5], bf[6], bf[7], q32);
                d2[T] = quant4<EXACT>(c321[T][0], c321[T][1], c321[T][2], c321[T][3], bf[8], bf[9], bf[10], bf[11], q32);
            }
        }
#undef REQ_PIECE2
        __builtin_amdgcn_sched_barrier(0);
        // C4 of conc2 row y2: lane (pixel, s) register r = w4[r][s] . conc2[y2][pixel]; read in the next row.
        // A row outside the frame is the zero padding of C4.
#pragma unroll
        for (int T = 0; T < NT; ++T) {
            if (!CHECK || (unsigned)y2 < (unsigned)H) D4p[T] = mfma16(f4, v4i{(int)d0[T], (int)d1[T], (int)d2[T], 0}, v4i{0, 0, 0, 0});
            else D4p[T] = v4i{0, 0, 0, 0};
        }
        STAMP();
    };
    // The row of the integer path: the same work with the requantiser spread over the matrix instructions, one value
    // (two vector instructions) behind every MFMA, and applyRes_y one instruction per gap.  The C4 MFMAs of a conc2 row
    // move to the start of the next row (dq carries the packed row instead of D4p its products).
    unsigned dq[NT][3] = {{0, 0, 0}, {0, 0, 0}};
#if defined(QCNN_STAMPS) && !defined(QCNN_STAMPS_NOPH)  // diagnostic builds only: cycles per phase of a row, summed over the MODE 0 rows of one wave
#define QCNN_PH 1
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ph_prev = 0, ph_rows = 0;
#define PH(K)                                                                                     \
    do {                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        unsigned long long t_;                                                                    \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                \
        if (MODE == 0) ph[K] += t_ - ph_prev;                                                     \
        ph_prev = t_;                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                        \
    } while (0)
#else
#define PH(K) do { } while (0)
#endif
    auto rowi = [&](auto jc, auto modec, int t) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        constexpr int MODE = decltype(modec)::value;
        constexpr bool CHECK = MODE == 1, COLS = MODE != 0;
        const int m0 = R * t;
        const unsigned char *rdc = rd + g3 * GROUP_BYTES, *rdo = rd + g3n * GROUP_BYTES;
        v4i(&B)[NT][3] = Bq[j & 1];
        v4i(&Bn)[NT][3] = Bq[(j + 1) & 1];
        prio(jc);
        PH(7);
        if (CHECK && t >= T_steps - 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(25)" ::: "memory");
        PH(0);
        constexpr int s0 = j % 3, s1 = (j + 2) % 3, s2 = (j + 1) % 3;
        constexpr int cur = j % 2, prv = (j + 1) % 2;
        const int y2 = ya - 3 + m0 + j;
#define SB() __builtin_amdgcn_sched_barrier(0)
#ifdef QCNN_ABL_FINE_NORQ  // diagnostic builds only (garbage results)
#define RQ32(T, I, HALF) dq[T][1 + HALF] ^= a32[T][prv][HALF][I]
#define RQ31(T, I) dq[T][0] ^= acc[T][s2][I]
#else
#define RQ32(T, I, HALF) requant1<I>(a32[T][prv][HALF][I], q32.mulk, dq[T][1 + HALF])
#define RQ31(T, I) requant1<I>(acc[T][s2][I], q31.mulk, dq[T][0])
#endif
#ifdef QCNN_ABL_FINE_NOMFMA  // diagnostic builds only (garbage results)
#define mfma16 [](v4i A_, v4i B_, v4i C_) { return v4i{C_[0] ^ B_[0], C_[1] ^ A_[1], C_[2], C_[3]}; }
#endif
        // C4 of conc2 row y2 - 1, requantised by the previous row (a row outside the frame is C4's zero padding)
        v4i D4[NT];
#pragma unroll
        for (int T = 0; T < NT; ++T) {
            if (!CHECK || (unsigned)(y2 - 1) < (unsigned)H)
                D4[T] = mfma16(f4, v4i{(int)dq[T][0], (int)dq[T][1], (int)dq[T][2], 0}, v4i{0, 0, 0, 0});
            else D4[T] = v4i{0, 0, 0, 0};
        }
        int yv[NT];
#pragma unroll
        for (int T = 0; T < NT; ++T) yv[T] = *reinterpret_cast<const int *>(rdy[T] + j * YSLOT);
#ifdef QCNN_ABL_FINE_NOBREAD  // diagnostic builds only (garbage results): no fragment reads
#pragma unroll
        for (int T = 0; T < NT; ++T)
#pragma unroll
            for (int s = 0; s < 3; ++s) Bn[T][s] = B[T][s];
#else
#pragma unroll
        for (int T = 0; T < NT; ++T)
#pragma unroll
            for (int s = 0; s < 3; ++s)
                Bn[T][s] = *reinterpret_cast<const v4i *>((j + 1 < R ? rdc + (j + 1) * ROW_BYTES : rdo) + (16 * T + s) * 48);
#endif
        SB();
        // third kernel row of conc2 row y2 (complete with it)
        acc[0][s2] = mfma16(f31[0][2], B[0][0], acc[0][s2]); SB(); RQ32(0, 0, 0); SB();
        acc[1][s2] = mfma16(f31[0][2], B[1][0], acc[1][s2]); SB(); RQ32(1, 0, 0); SB();
        acc[0][s2] = mfma16(f31[1][2], B[0][1], acc[0][s2]); SB(); RQ32(0, 0, 1); SB();
        acc[1][s2] = mfma16(f31[1][2], B[1][1], acc[1][s2]); SB(); RQ32(1, 0, 1); SB();
        // C4 sums of output row y3 = y2 - 2: out[o] = P_0[o] + P_1[o + 1] + P_2[o + 2] (lane rows 0, 1, 2)
        int P[NT], v1[NT], v2[NT];
        acc[0][s2] = mfma16(f31[2][2], B[0][2], acc[0][s2]); SB();
#pragma unroll
        for (int T = 0; T < NT; ++T) {
            P[T] = Pa[T] + D4[T][2];
            if constexpr (COLS) P[T] &= colmask[T];
        }
        SB();
        acc[1][s2] = mfma16(f31[2][2], B[1][2], acc[1][s2]); SB();
        PH(1);
#pragma unroll
        for (int T = 0; T < NT; ++T) {
            Pa[T] = Pb[T] + D4[T][1];
            Pb[T] = D4[T][0];
        }
        {
            const int M0 = borrow ? P[1] : P[0];  // tile 0's gathers wrap into tile 1's first pixels
#ifdef QCNN_ABL_FINE_NOBPERM  // diagnostic builds only (garbage results): no gathers
            v1[0] = M0, v2[0] = M0 ^ addr1, v1[1] = P[1], v2[1] = P[1] ^ addr2;
#else
#ifdef QCNN_ABL_FINE_NOBPERM  // diagnostic builds only (garbage results): no gathers
            v1[0] = M0, v2[0] = M0 ^ addr1, v1[1] = P[1], v2[1] = P[1] ^ addr2;
#else
            v1[0] = __builtin_amdgcn_ds_bpermute(addr1, M0);
            v2[0] = __builtin_amdgcn_ds_bpermute(addr2, M0);
            v1[1] = __builtin_amdgcn_ds_bpermute(addr1, P[1]);
            v2[1] = __builtin_amdgcn_ds_bpermute(addr2, P[1]);
#endif
#endif
        }
        SB();
        const int y3 = y2 - 2;
        // second kernel row of conc2 row y2 + 1; from here on the C3_1 channels of row y2 are requantised too
        acc[0][s1] = mfma16(f31[0][1], B[0][0], acc[0][s1]); SB(); RQ32(0, 1, 0); SB();
        acc[1][s1] = mfma16(f31[0][1], B[1][0], acc[1][s1]); SB(); RQ32(1, 1, 0); SB();
        acc[0][s1] = mfma16(f31[1][1], B[0][1], acc[0][s1]); SB(); RQ32(0, 1, 1); RQ31(0, 0); SB();
        acc[1][s1] = mfma16(f31[1][1], B[1][1], acc[1][s1]); SB(); RQ32(1, 1, 1); RQ31(1, 0); SB();
        acc[0][s1] = mfma16(f31[2][1], B[0][2], acc[0][s1]); SB(); RQ32(0, 2, 0); RQ31(0, 1); SB();
        acc[1][s1] = mfma16(f31[2][1], B[1][2], acc[1][s1]); SB(); RQ32(1, 2, 0); RQ31(1, 1); SB();
        PH(2);
        // first kernel row of conc2 row y2 + 2 (the bias starts the accumulator); applyRes_y of output row y3, one
        // instruction per gap
        int sum[NT], temp[NT], out[NT];
        acc[0][s0] = mfma16(f31[0][0], B[0][0], b0); SB(); RQ32(0, 2, 1); RQ31(0, 2); SB();
        acc[1][s0] = mfma16(f31[0][0], B[1][0], b0); SB(); RQ32(1, 2, 1); RQ31(1, 2); SB();
        acc[0][s0] = mfma16(f31[1][0], B[0][1], acc[0][s0]); SB(); RQ32(0, 3, 0); sum[0] = P[0] + v1[0] + v2[0]; SB();
        acc[1][s0] = mfma16(f31[1][0], B[1][1], acc[1][s0]); SB(); RQ32(1, 3, 0); sum[1] = P[1] + v1[1] + v2[1]; SB();
        acc[0][s0] = mfma16(f31[2][0], B[0][2], acc[0][s0]); SB(); RQ32(0, 3, 1); temp[0] = __mul24(sum[0], mul) + round4v; SB();
        acc[1][s0] = mfma16(f31[2][0], B[1][2], acc[1][s0]); SB(); RQ32(1, 3, 1); temp[1] = __mul24(sum[1], mul) + round4v; SB();
        PH(3);
        // C3_2 of conc2 row y2 + 1 (its centre pixel): requantised by the next row
        a32[0][cur][0] = mfma16(f32[0], B[0][1], b1); SB(); RQ31(0, 3); temp[0] >>= shift; SB();
        a32[1][cur][0] = mfma16(f32[0], B[1][1], b1); SB(); RQ31(1, 3); temp[1] >>= shift; SB();
        a32[0][cur][1] = mfma16(f32[1], B[0][1], b2); SB(); temp[0] += yv[0]; temp[1] += yv[1]; SB();
        a32[1][cur][1] = mfma16(f32[1], B[1][1], b2); SB(); out[0] = min(max(temp[0], 0), 255); out[1] = min(max(temp[1], 0), 255); SB();
        PH(4);
        // the stores (only if the row lies in the band)
#pragma unroll
        for (int T = 0; T < NT; ++T) {
            const int voff = (!CHECK || (y3 >= ya && y3 < yb)) ? out_off[T] : OOB;
#ifndef QCNN_ABL_FINE_NOSTORE  // (diagnostic builds leave the stores out: garbage results)
            asm volatile("buffer_store_byte %0, %1, %2, %3 offen" ::"v"(out[T]), "v"(voff), "s"(rec_rsrc), "s"(y3 * W) : "memory");
#else
            asm volatile("" ::"v"(out[T]), "v"(voff));
#endif
        }
        PH(5);
        // then the requests for the slots this row has emptied (rows past the last one the band walks are not
        // requested: when the wave ends and its LDS is handed to the next workgroup, no load may still be on its way
        // into it; the last step but one still fetches the Y pixels of the last step)
#ifdef QCNN_ABL_FINE_NOLOAD  // diagnostic builds only (garbage results)
        if (!CHECK || t <= T_steps - 2) request_impl(false, g3, j, 0, y3 + R, true);
#else
        if (!CHECK || t <= T_steps - 3) request_impl(true, g3, j, ya + m0 + j + 2 * R, y3 + R, CHECK);
        else if (t == T_steps - 2) request_impl(false, g3, j, 0, y3 + R, true);
#endif
        SB();
        PH(6);
#ifdef QCNN_PH
        if (MODE == 0) ++ph_rows;
#endif
#undef SB
#undef RQ32
#undef RQ31
#ifdef QCNN_ABL_FINE_NOMFMA
#undef mfma16
#endif
    };
    // Steps [t_lo, t_hi) need no test of rows (see the one-tile form): conc2 rows inside the frame, output rows inside
    // the band, requested rows inside the frame / the bordered image.
    int t_lo, t_hi;
    {
        int last = (H + 3 - R - ya < yb + 5 - R - ya ? H + 3 - R - ya : yb + 5 - R - ya);
        last = last < H + 5 - 3 * R - ya ? last : H + 5 - 3 * R - ya;
        last = last < H + 4 - 3 * R - ya ? last : H + 4 - 3 * R - ya;
        t_lo = ya >= 3 ? 1 : (3 - ya + R - 1) / R > 1 ? (3 - ya + R - 1) / R : 1;
        t_hi = last >= 0 ? last / R + 1 : 0;
        if (t_hi > T_steps - 2) t_hi = T_steps - 2 > 0 ? T_steps - 2 : 0;  // the last two steps request nothing
        if (t_lo > t_hi) t_lo = t_hi;
    }
    // (four explicit loops: with the step wrapped in a lambda of its own the compiler spills registers)
#define QCNN_STEP(M)                                                            \
    do {                                                                        \
        static_for<R>([&](auto jc) {                                            \
            if constexpr (EXACT) row(jc, std::integral_constant<int, M>{}, t);  \
            else rowi(jc, std::integral_constant<int, M>{}, t);                 \
        });                                                                     \
        g3 ^= 1, g3n ^= 1;                                                      \
    } while (0)
    int t = 0;
    for (; t < t_lo; ++t) QCNN_STEP(1);
    if (edge_strip)
        for (; t < t_hi; ++t) QCNN_STEP(2);
    for (; t < t_hi; ++t) QCNN_STEP(0);
    for (; t < T_steps; ++t) QCNN_STEP(1);
#undef QCNN_STEP
    __builtin_amdgcn_s_setprio(0);
#ifdef QCNN_STAMPS
    if (a.dbg && blockIdx.x < 16384 && lane == 0) {
        unsigned long long blk_t1;
        unsigned hwid, xcc;
        asm volatile("s_memtime %0\n\ts_getreg_b32 %1, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %2, hwreg(HW_REG_XCC_ID)\n\ts_waitcnt lgkmcnt(0)"
                     : "=s"(blk_t1), "=s"(hwid), "=s"(xcc)::"memory");
        unsigned long long *o = a.dbg + 24 * 128 + 3 * (size_t)blockIdx.x;
        o[0] = blk_t0;
        o[1] = blk_t1;
        o[2] = ((unsigned long long)xcc << 32) | hwid;
    }
#endif
#ifdef QCNN_PH
    if constexpr (!EXACT)
        if (a.dbg && (int)blockIdx.x == a.dbg_block && lane == 0) {
            for (int k = 0; k < 8; ++k) a.dbg[20 * 128 + k] = ph[k];
            a.dbg[20 * 128 + 8] = ph_rows;
        }
#endif
}

hipError_t launch_fused_c3c4_stream2(const FusedArgs &a, bool exact, hipStream_t s)
{
    if (a.nframes < 1 || a.H < 1 || a.W < 1) return hipSuccess;
    const int nstrips = (a.W + ks2::SW - 1) / ks2::SW;
    const int ncu = a.ncu > 0 ? a.ncu : 256;
    const double slots = (double)ks2::WAVES_PER_CU * ncu;  // waves the device keeps resident
    // Bands: every band of rows costs 4 extra conc1 rows (and up to R - 1 wasted ones), every round of waves the
    // device cannot fill completely costs its idle share.  Pick the count with the best product of the two.
    int seg_rows = 0;
    if (const char *sr = getenv("QCNN_SEG_ROWS")) seg_rows = atoi(sr);
    if (seg_rows < 1) {
        double best = -1;
        const int maxseg = a.H / 24 > 1 ? a.H / 24 : 1;
        for (int n = 1; n <= maxseg && n <= 256; ++n) {
            const int rows = (a.H + n - 1) / n;
            const int nn = (a.H + rows - 1) / rows;
            const double rounds = (double)a.nframes * nstrips * nn / slots;
            const double full = rounds / std::ceil(rounds);
            const double eff = full * rows / (rows + 4.0 + (ks2::R - 1) / 2.0);
            if (eff > best + 1e-9) best = eff, seg_rows = rows;
        }
    }
    if (seg_rows > a.H) seg_rows = a.H;
    const int nseg = (a.H + seg_rows - 1) / seg_rows;
    const long long units = (long long)a.nframes * nstrips * nseg;
    if (units > (1ll << 30)) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((units + 7) / 8 * 8));  // a multiple of 8: see the unit map
    // one round of waves: the workgroups past the first ncu * 4 / 8 of an XCD are the second ones on their SIMDs
    const int second_from = (double)grid.x <= slots ? ncu * 4 / 8 : INT_MAX;
    if (exact) hipLaunchKernelGGL(k_c3c4s2<true>, grid, dim3(64), 0, s, a, nstrips, nseg, seg_rows, (int)units, second_from);
    else hipLaunchKernelGGL(k_c3c4s2<false>, grid, dim3(64), 0, s, a, nstrips, nseg, seg_rows, (int)units, second_from);
    return hipGetLastError();
}

}  // namespace qcnn
