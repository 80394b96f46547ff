// Kernel A of the fused forward pass: Y -> I1.ppro -> C1 -> requant -> (C2_1 || C2_2) -> requant ->
// concat, written to HBM as [H][W][48] int8 (see fused_common.h for the overall design).
//
// Persistent, wave-specialised: one 768-thread workgroup per CU walks a list of 20 x 32 output
// tiles.  Waves 0-3 are PRODUCERS (Y tile, im2col, C1 MFMAs, C1 requant epilogue: VALU-heavy),
// waves 4-11 are CONSUMERS (C2 MFMAs, C2 requant epilogue: MFMA-heavy).  Waves w, w+4 and w+8 share
// a SIMD, so each SIMD always has VALU-bound and MFMA-bound instruction streams to interleave: a
// wave's own vector and matrix instructions barely overlap, those of different waves do.  The C1 tile is double-buffered in LDS; in step n the producers fill buffer n&1
// with tile n while the consumers read buffer (n-1)&1; one workgroup barrier ends each step.
// A workgroup walks its tiles DOWN a 32-pixel column of the frame: the top 4 rows of the C1 region
// of a tile are the bottom 4 rows of the tile above it, so the producers copy them from the other
// LDS buffer and compute 20 new rows instead of 24 (a "warm" tile; the first tile of a workgroup
// and the top tile of a column are "cold" and compute all 24).
#include "fused_common.h"

namespace qcnn {

// Load plan of the C2 stage for a wave that owns ROWS output rows x 16 columns of the convolutions in CONV (bit 0:
// C2_2, bit 1: C2_1; both: 48 channels).  A step issues every MFMA that uses the B fragment of (input row yy, kernel
// column s) of the wave's IN = ROWS + 4 input rows (C2_1 alone needs kernel columns 1..3 and input rows 1..IN-2
// only).  The LDS reads (B fragments, 25 C2_2 and 18 C2_1 A fragments) are numbered in the order the steps first
// need them; need[I] is the number of reads that must have been issued before step I.  The kernel issues the
// reads of step I + C2_PREFETCH while it runs the MFMAs of step I.
template <int ROWS, int CONV = 3>
struct C2Plan {
    static constexpr int IN = ROWS + 4, MAXSTEPS = 5 * IN, ITEMS = MAXSTEPS + 25 + 18;  // (upper bounds)
    int kind[ITEMS];  // 0: B fragment (s, yy); 1: C2_2 A fragment (s, r); 2: C2_1 A fragment (s-1, r-1, t)
    int a0[ITEMS], a1[ITEMS], a2[ITEMS];
    int need[MAXSTEPS], step_s[MAXSTEPS], step_yy[MAXSTEPS];
    int iB[5][IN], i22[5][5], i21[3][3][2];
    int n, steps;
};
template <int ROWS, int CONV>
constexpr C2Plan<ROWS, CONV> make_c2_plan()
{
    C2Plan<ROWS, CONV> p{};
    int n = 0, steps = 0;
    for (int s = 0; s < 5; ++s) {
        if (!(CONV & 1) && (s < 1 || s > 3)) continue;
        for (int yy = 0; yy < ROWS + 4; ++yy) {
            if (!(CONV & 1) && (yy < 1 || yy > ROWS + 2)) continue;
            p.iB[s][yy] = n;
            p.kind[n] = 0, p.a0[n] = s, p.a1[n] = yy, ++n;
            if (yy < 5) {  // kernel row r = yy is first used here (output row 0)
                if (CONV & 1) {
                    p.i22[s][yy] = n;
                    p.kind[n] = 1, p.a0[n] = s, p.a1[n] = yy, ++n;
                }
                if ((CONV & 2) && s >= 1 && s <= 3 && yy >= 1 && yy <= 3)
                    for (int t = 0; t < 2; ++t) {
                        p.i21[s - 1][yy - 1][t] = n;
                        p.kind[n] = 2, p.a0[n] = s - 1, p.a1[n] = yy - 1, p.a2[n] = t, ++n;
                    }
            }
            p.step_s[steps] = s, p.step_yy[steps] = yy;
            p.need[steps++] = n;
        }
    }
    p.n = n, p.steps = steps;
    return p;
}
template <int ROWS, int CONV = 3>
struct C2PlanOf {
    static constexpr C2Plan<ROWS, CONV> P = make_c2_plan<ROWS, CONV>();
    static_assert(CONV != 3 || P.n == C2Plan<ROWS, CONV>::ITEMS, "B fragments + 25 + 18 A fragments");
};
#ifndef C2_PREFETCH_N
#define C2_PREFETCH_N 4
#endif
constexpr int C2_PREFETCH = C2_PREFETCH_N;  // steps of look-ahead of the LDS reads
#ifndef C2_PREFETCH_START_N
#define C2_PREFETCH_START_N 6
#endif
// The reads of the first C2_PREFETCH_START steps are issued before the stage's first MFMA (the caller
// puts other work between them and the MFMAs, so they have landed by then).
constexpr int C2_PREFETCH_START = C2_PREFETCH_START_N;
static_assert(C2_PREFETCH_START >= C2_PREFETCH, "the start covers at least the steady look-ahead");
template <int ROWS, int CONV = 3>
constexpr int c2_issued(int I)  // reads issued once step I has issued its own (I = -1: before step 0)
{
    constexpr int STEPS = C2PlanOf<ROWS, CONV>::P.steps;
    const int start = C2PlanOf<ROWS, CONV>::P.need[C2_PREFETCH_START - 1];
    if (I < 0) return start;
    const int n = C2PlanOf<ROWS, CONV>::P.need[I + C2_PREFETCH < STEPS ? I + C2_PREFETCH : STEPS - 1];
    return n > start ? n : start;
}

// Per-lane constants of the C2 stage: biases of the lane's channels (int32 folded into the
// accumulators on the integer path, float32 for the float pipeline) and the requantisers.
struct C2Consts {
    v4i b0, b1, b2;
    float bf[12];
    Requant q21, q22;
#ifdef QCNN_STAMPS
    unsigned long long *dbg;
    int dbg_block, row, *idx;
#endif
};
#ifdef QCNN_STAMPS
#define C2_STAMP() stamp(k.dbg, k.dbg_block, k.row, *k.idx)
#else
#define C2_STAMP() do { } while (0)
#endif

// Instantiations of kernel A: integer epilogues (proved at load), the literal float32 pipeline, integer epilogues
// with the C2 accumulators clamped first (fused_common.h, requant12_wide).
enum { A_INT = 0, A_FLOAT = 1, A_WIDE = 2 };

// Accumulators of the C2 stage of one wave: ROWS output rows x 16 columns x 48 channels.
template <int ROWS>
struct C2Acc {
    v4i a21[2][ROWS], a22[ROWS];
};

// Requant of the accumulators and the store into conc1 (frame `concf`): rows R0 .. R0+ROWS-1, columns
// cb .. cb+15 of the tile at (x0, y0).  Lane (pixel c, g) owns channels 4g..4g+3 of each 16-channel
// tile = 12 consecutive bytes at position 12g of the pixel in the internal channel order.
template <int MODE, int ROWS, int CONV = 3>
__device__ __forceinline__ void c2_store(const C2Acc<ROWS> &acc, int R0, int cb, int c, int g, const C2Consts &k, int8_t *concf,
                                         int x0, int y0, int H, int W)
{
#ifdef QCNN_ABL_L2STORES  // diagnostic builds only (garbage results): every workgroup keeps storing one tile, which stays in L2
    x0 = (blockIdx.x % 60) * ka::TW, y0 = (blockIdx.x / 60) * ka::TH;
#endif
#ifdef QCNN_ABL_LINEARSTORES  // diagnostic builds only (garbage results): lane l stores the 12 bytes at 12 l of the wave's 768
    {
        const int l = c + 16 * g;
        c = l >> 2, g = l & 3;
    }
#endif
    const int gx = x0 + cb + c, gy0 = y0 + R0;
    // the frame sits at rows / columns 2.. of the bordered conc1 image
    const size_t row_bytes = (size_t)conc_wp(W) * 48;  // row y sits y * row_bytes further
    int8_t *out = concf + (size_t)(gy0 + 2) * row_bytes + (size_t)(gx + 2) * 48 + 12 * g;
    constexpr bool EXACT = MODE == A_FLOAT;
    static_assert(CONV == 3 || MODE == A_INT, "single-convolution waves exist on the integer path only");
    auto quant_row = [&](int y, unsigned &d0, unsigned &d1, unsigned &d2) {
#ifdef QCNN_ABL_NOREQ2  // diagnostic builds only (garbage results): the consumers' requantisation switched off
        d0 = acc.a21[0][y][0] ^ acc.a21[0][y][1] ^ acc.a21[0][y][2] ^ acc.a21[0][y][3];
        d1 = acc.a21[1][y][0] ^ acc.a21[1][y][1] ^ acc.a21[1][y][2] ^ acc.a21[1][y][3];
        d2 = acc.a22[y][0] ^ acc.a22[y][1] ^ acc.a22[y][2] ^ acc.a22[y][3];
        return;
#endif
        if constexpr (MODE == A_INT && CONV == 1) {
            d2 = requant4_int(acc.a22[y][0], acc.a22[y][1], acc.a22[y][2], acc.a22[y][3], k.q22);
        } else if constexpr (MODE == A_INT && CONV == 2) {
            requant8_int(acc.a21[0][y], acc.a21[1][y], k.q21, d0, d1);
        } else if constexpr (MODE == A_INT) {
            requant12_int(acc.a21[0][y], acc.a21[1][y], acc.a22[y], k.q21, k.q22, d0, d1, d2);
        } else if constexpr (MODE == A_WIDE) {
            requant12_wide(acc.a21[0][y], acc.a21[1][y], acc.a22[y], k.q21, k.q22, k.q21.blu + k.q21.bias, k.q22.blu + k.q22.bias, d0,
                           d1, d2);
        } else {
            d0 = quant4<EXACT>(acc.a21[0][y][0], acc.a21[0][y][1], acc.a21[0][y][2], acc.a21[0][y][3], k.bf[0], k.bf[1], k.bf[2],
                               k.bf[3], k.q21);
            d1 = quant4<EXACT>(acc.a21[1][y][0], acc.a21[1][y][1], acc.a21[1][y][2], acc.a21[1][y][3], k.bf[4], k.bf[5], k.bf[6],
                               k.bf[7], k.q21);
            d2 = quant4<EXACT>(acc.a22[y][0], acc.a22[y][1], acc.a22[y][2], acc.a22[y][3], k.bf[8], k.bf[9], k.bf[10], k.bf[11],
                               k.q22);
        }
    };
#ifdef QCNN_ABL_STRIPMAJOR  // diagnostic builds only (garbage image): the stores of a strip-major conc1 image -- pixel x goes
    // to column x % 14 + 2 of strip x / 14 (864-byte rows, the rows of a strip contiguous), and the four columns two
    // neighbouring strips share are stored twice
    const int sm_strip = gx / 14, sm_col = gx - sm_strip * 14;
    int8_t *sm_out = concf + ((size_t)(sm_strip % 106) * (H + 4) + gy0 + 2) * 864 + (sm_col + 2) * 48 + 12 * g;
    const int sm_other = sm_col < 2 ? sm_strip - 1 : sm_col >= 12 ? sm_strip + 1 : -1;
    int8_t *sm_out2 = concf + ((size_t)((sm_other < 0 ? 0 : sm_other) % 106) * (H + 4) + gy0 + 2) * 864 +
                      (sm_col < 2 ? sm_col + 16 : sm_col - 12) * 48 + 12 * g;
#endif
    if (gy0 + ROWS <= H && x0 + cb + 16 <= W) {  // the wave's pixels are all inside the frame
#pragma unroll
        for (int y = 0; y < ROWS; ++y) {
            unsigned d0, d1, d2;
            quant_row(y, d0, d1, d2);
            unsigned *dst = reinterpret_cast<unsigned *>(out + y * row_bytes);
#ifdef QCNN_ABL_STRIPMAJOR
            dst = reinterpret_cast<unsigned *>(sm_out + y * 864);
            if (sm_other >= 0) {
                unsigned *dst2 = reinterpret_cast<unsigned *>(sm_out2 + y * 864);
                dst2[0] = d0, dst2[1] = d1, dst2[2] = d2;
            }
#endif
#ifdef QCNN_STAMP_ROWS  // diagnostic builds: a stamp behind the requantisation of every row (ahead of its store)
            C2_STAMP();
#endif
            if constexpr (CONV == 1) {
                dst[2] = d2;
                continue;
            } else if constexpr (CONV == 2) {
                dst[0] = d0;
                dst[1] = d1;
                continue;
            }
#ifdef QCNN_CONC_STORE_POLICY  // experiment builds: cache policy bits of the conc1 stores (1 nt, 2 sc1, 3 sc0 sc1, 4 sc0 sc1 nt)
            typedef unsigned v3u __attribute__((ext_vector_type(3)));
            const v3u d = {d0, d1, d2};
#if QCNN_CONC_STORE_POLICY == 1
            asm volatile("global_store_dwordx3 %0, %1, off nt" ::"v"(dst), "v"(d) : "memory");
#elif QCNN_CONC_STORE_POLICY == 2
            asm volatile("global_store_dwordx3 %0, %1, off sc1" ::"v"(dst), "v"(d) : "memory");
#elif QCNN_CONC_STORE_POLICY == 3
            asm volatile("global_store_dwordx3 %0, %1, off sc0 sc1" ::"v"(dst), "v"(d) : "memory");
#else
            asm volatile("global_store_dwordx3 %0, %1, off sc0 sc1 nt" ::"v"(dst), "v"(d) : "memory");
#endif
#else
            dst[0] = d0;
            dst[1] = d1;
            dst[2] = d2;
#endif
        }
    } else {
#pragma unroll
        for (int y = 0; y < ROWS; ++y) {
            unsigned d0, d1, d2;
            quant_row(y, d0, d1, d2);
            if (gy0 + y < H && gx < W) {
                unsigned *dst = reinterpret_cast<unsigned *>(out + y * row_bytes);
                if constexpr (CONV & 2) {
                    dst[0] = d0;
                    dst[1] = d1;
                }
                if constexpr (CONV & 1) dst[2] = d2;
            }
        }
    }
}

// C2_1 (3x3, 64 -> 32) and C2_2 (5x5, 64 -> 16) for ROWS output rows x 16 columns of the tile in `buf`
// (rows R0 .. R0+ROWS-1, columns cb .. cb+15) as 16x16x64 MFMAs into `acc`.  LDS row = output row + r,
// LDS column = output column + s (the 3x3 taps are r-1, s-1 of the 5x5 window).  The LDS reads of
// the first steps are issued, then `between()` runs (it may still read the previous contents of
// `acc`), then the accumulators are set to the biases and the MFMA steps run.
template <bool EXACT, int ROWS, int CONV = 3, typename F>
__device__ __forceinline__ void c2_mfma(C2Acc<ROWS> &acc, const unsigned char *buf, int R0, int cb, int c, int g, const v4i *w22,
                                        const v4i *w21, const C2Consts &k, F &&between)
{
    using namespace ka;
    using PO = C2PlanOf<ROWS, CONV>;  // PO::P is the plan
    constexpr int IN = C2Plan<ROWS, CONV>::IN, STEPS = PO::P.steps;
    const unsigned char *rowbase = buf + (R0 * C1C + cb + c) * 64;
    const unsigned char *colbase[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) colbase[s] = rowbase + s * 64 + swz(g, cb + c + s);
    v4i item[C2Plan<ROWS, CONV>::ITEMS];
    auto issue = [&](auto jc) {
        constexpr int J = decltype(jc)::value;
        if constexpr (PO::P.kind[J] == 0)
            item[J] = *reinterpret_cast<const v4i *>(colbase[PO::P.a0[J]] + PO::P.a1[J] * (C1C * 64));
        else if constexpr (PO::P.kind[J] == 1)
            item[J] = w22[(PO::P.a0[J] * 5 + PO::P.a1[J]) * 64];
        else
            item[J] = w21[((PO::P.a0[J] * 3 + PO::P.a1[J]) * 2 + PO::P.a2[J]) * 64];
    };
    static_range<0, c2_issued<ROWS, CONV>(-1)>(issue);
    __builtin_amdgcn_sched_barrier(0);
    between();
    __builtin_amdgcn_sched_barrier(0);
    C2_STAMP();
#pragma unroll
    for (int y = 0; y < ROWS; ++y) {
        if constexpr (CONV & 2) {
            acc.a21[0][y] = k.b0;
            acc.a21[1][y] = k.b1;
        }
        if constexpr (CONV & 1) acc.a22[y] = k.b2;
    }
    static_for<STEPS>([&](auto ic) {
        constexpr int I = decltype(ic)::value;
        constexpr int s = PO::P.step_s[I], yy = PO::P.step_yy[I];
        static_range<c2_issued<ROWS, CONV>(I - 1), c2_issued<ROWS, CONV>(I)>(issue);
        const v4i B = item[PO::P.iB[s][yy]];
        static_for<5>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            constexpr int y = yy - r;
            if constexpr (y >= 0 && y < ROWS) {
                if constexpr (CONV & 1) acc.a22[y] = mfma16(item[PO::P.i22[s][r]], B, acc.a22[y]);
                if constexpr ((CONV & 2) && s >= 1 && s <= 3 && r >= 1 && r <= 3) {
                    acc.a21[0][y] = mfma16(item[PO::P.i21[s - 1][r - 1][0]], B, acc.a21[0][y]);
                    acc.a21[1][y] = mfma16(item[PO::P.i21[s - 1][r - 1][1]], B, acc.a21[1][y]);
                }
            }
        });
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (I + 1 == STEPS || PO::P.step_s[I + 1 < STEPS ? I + 1 : I] != s) C2_STAMP();
    });
}

// Rows of a SIMD's 10 x 16 block (C2_BLOCK_ROWS x 16) that its consumers compute themselves; the PRODUCER wave on the same
// SIMD would compute the rest after its C1 work.  Measured: with 7 the step does not get shorter (the
// MFMA pipe and the vector issue port are shared by the two waves either way, and the producer's
// one-row share reuses no fragment), so the consumer keeps all 8.
constexpr int C2_BLOCK_ROWS = ka::TH / 2;  // rows of the 16-column block the consumers of one SIMD compute
#ifndef C2_CONS_ROWS_N
#define C2_CONS_ROWS_N C2_BLOCK_ROWS
#endif
constexpr int C2_CONS_ROWS = C2_CONS_ROWS_N;
// When the producer runs its share: 1 = at the start of the step, next to the consumer's requant + store phase;
// 0 = at the end of the step, after its C1 work.
#ifndef C2_SHARE_FIRST
#define C2_SHARE_FIRST 1
#endif
// Consumer waves per SIMD.  With two, consumer 0 takes the upper half of the SIMD's 10 x 16 block and stores a
// tile at the start of the next step (as the single consumer does), consumer 1 takes the lower half and stores
// at the end of its own step: one of them always has matrix instructions to issue while the other requantises.
#ifndef C2_CONSUMERS_N
#define C2_CONSUMERS_N 2
#endif
constexpr int C2_CONSUMERS = C2_CONSUMERS_N;
// Which of the two consumers of a SIMD keeps its accumulators over the step boundary (requantises and stores first,
// then runs its matrix instructions): 0 = the older wave (4..7), 1 = the younger one (8..11).
#ifndef C2_DEFER_YOUNGER
#define C2_DEFER_YOUNGER 1
#endif
// Experiment: wave priorities (s_setprio) of a consumer in its matrix phase and in its requant + store phase, per
// role (D = the deferring consumer, N = the other one); -1 = leave the priority alone.  Producers stay at 0.
#ifndef C2_PRIO_D_MFMA
#define C2_PRIO_D_MFMA -1
#endif
#ifndef C2_PRIO_D_VALU
#define C2_PRIO_D_VALU -1
#endif
#ifndef C2_PRIO_N_MFMA
#define C2_PRIO_N_MFMA -1
#endif
#ifndef C2_PRIO_N_VALU
#define C2_PRIO_N_VALU -1
#endif
#ifndef C2_PRIO_PRODUCER
#define C2_PRIO_PRODUCER -1
#endif
// Experiment: the two consumers of a SIMD split its block by convolution instead of by rows (1: the older wave
// takes C2_2, 2: C2_1); integer path only.
#ifndef C2_SPLIT_CONV
#define C2_SPLIT_CONV 0
#endif
template <int P>
__device__ __forceinline__ void set_prio()
{
    if constexpr (P >= 0) asm volatile("s_setprio %0" ::"n"(P));
}
static_assert(C2_CONSUMERS == 1 || (C2_CONSUMERS == 2 && C2_CONS_ROWS == C2_BLOCK_ROWS && C2_BLOCK_ROWS % 2 == 0),
              "two consumers split all rows of the block");
constexpr int A_WAVES = 4 + 4 * C2_CONSUMERS, A_THREADS = 64 * A_WAVES;

// End of a pipeline step: every LDS write of this wave has landed, then the workgroup barrier.
// (A raw s_barrier: global loads and stores in flight are deliberately not waited for.)
__device__ __forceinline__ void step_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// Barrier inside a step: the producers wait here until the consumers have stored their previous
// tile, so that the two vector-heavy phases of a SIMD's wave pair do not run side by side.
// (Not with two consumers per SIMD: the second one has matrix work from the start of the step.)
#ifndef C2_MID_BARRIER
#define C2_MID_BARRIER (C2_CONSUMERS_N == 1)
#endif
__device__ __forceinline__ void mid_barrier()
{
#if C2_MID_BARRIER
    asm volatile("s_barrier" ::: "memory");
#endif
}

// Experiment (-DQCNN_A_FLAGS=1; measured slower, see below): hand-offs without a workgroup barrier, through two
// counters in LDS that only grow.  produced = producer waves that have completed a tile (4 per tile), consumed =
// consumer waves that have read a tile's last fragment (4 per consumer of a SIMD and tile).  A wave that waits polls
// the word with a sleep between two reads; the bound on the number of polls only keeps a logic error from hanging
// the device (it is never reached: tens of milliseconds).  Results are bit-identical; sustained on 512 frames the
// kernel takes 70.6 us per frame against 68.2 with the barrier (2.11 M against 1.96 M cycles per 16 frames): the
// older consumer of a SIMD runs a tile ahead and then waits for the producers, which wait for the younger one it
// starves; with the younger one at priority 1 72.0, with neither consumer deferring its stores 69.4.
#ifndef QCNN_A_FLAGS
#define QCNN_A_FLAGS 0
#endif
#ifndef QCNN_A_FLAG_SLEEP
#define QCNN_A_FLAG_SLEEP 1
#endif
__device__ __forceinline__ void flag_wait(const unsigned *flag, unsigned target, bool &alive)
{
    const unsigned addr = (unsigned)(size_t)flag;
    if (!alive) return;  // (a wave that has given up once stops waiting: the launch ends, with garbage)
    for (int spin = 0; spin < (1 << 20); ++spin) {
        unsigned v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
        if ((int)(__builtin_amdgcn_readfirstlane(v) - target) >= 0) return;
        __builtin_amdgcn_s_sleep(QCNN_A_FLAG_SLEEP);
    }
    alive = false;
}
// (every LDS access of this wave has completed before the counter moves)
__device__ __forceinline__ void flag_signal(unsigned *flag, int lane)
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
}

// Runs of tiles of the workgroups of one launch: workgroup b walks tiles start[b] .. start[b + 1] - 1.
constexpr int A_MAX_GROUPS = 320;
struct TileRuns {
    int start[A_MAX_GROUPS + 1];
};

template <int PATH>
__global__ __launch_bounds__(A_THREADS) __attribute__((amdgpu_waves_per_eu(A_WAVES / 4, A_WAVES / 4)))
void k_c1c2(const FusedArgs a, const TileRuns runs)
{
    using namespace ka;
    constexpr bool EXACT = PATH == A_FLOAT;
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
    // [2][C1R][C1C][64] int8 C1 output tiles (chunk-swizzled), one private Y tile per producer,
    // then all 43 C2 A fragments (loaded once per workgroup, read by every consumer tile)
    unsigned char *sYall = smem + 2 * SC1_BYTES;
    unsigned char *sW = smem + 2 * SC1_BYTES + 4 * SYW_BYTES;
    unsigned *sProduced = reinterpret_cast<unsigned *>(smem + 2 * SC1_BYTES + 4 * SYW_BYTES + SW_BYTES), *sConsumed = sProduced + 1;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = a.H, W = a.W;
    const int tiles_x = (W + TW - 1) / TW;
    const int tpf = tiles_x * ((H + TH - 1) / TH);  // tiles per frame
    const int ntiles = tpf * a.nframes;
    // tiles in column-major order (frame, tile column, tile row); workgroup b takes the contiguous run
    // [start[b], start[b + 1]) of that order (the launcher cuts the runs to equal cost, not to equal length)
    const int tiles_y = (H + TH - 1) / TH;
    const int k0 = runs.start[blockIdx.x];
    const int N = runs.start[blockIdx.x + 1] - k0;  // tiles of this workgroup
    TilePos pos;  // the tile the role below handles next
    pos.f = k0 / tpf;
    pos.tx = (k0 - pos.f * tpf) / tiles_y;
    pos.ty = k0 - pos.f * tpf - pos.tx * tiles_y;
    pos.valid = true;
    auto advance = [&](TilePos &t) {
        if (++t.ty == tiles_y) {
            t.ty = 0;
            if (++t.tx == tiles_x) {
                t.tx = 0;
                ++t.f;
            }
        }
    };
    STAMP_INIT();
#ifdef QCNN_STAMPS  // diagnostic builds only: when this workgroup started
    unsigned long long blk_t0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(blk_t0)::"memory");
#endif

    // C2 weights -> LDS, once.  They are first read after the barrier that ends step 0.
    {
        const v4i *src = reinterpret_cast<const v4i *>(a.wpack + PackedOffsets::C2W22);
        v4i *dst = reinterpret_cast<v4i *>(sW);
        for (int i = tid; i < SW_BYTES / 16; i += A_THREADS) dst[i] = src[i];
    }
#if QCNN_A_FLAGS
    // the only workgroup barrier of the kernel: the weights and the zeroed counters are in LDS
    if (tid < 2) sProduced[tid] = 0;
    step_barrier();
    bool alive = true;
#endif

    // C2 constants of this lane, used by the consumer and by the producer's share of the C2 rows
    const int c = lane & 15, g = lane >> 4;
    const int blk = wv & 3;  // the C2_BLOCK_ROWS x 16 output block of this SIMD's consumers and producer
    const int cb = (blk & 1) * 16, R0 = (blk >> 1) * C2_BLOCK_ROWS;
    const v4i *w22 = reinterpret_cast<const v4i *>(sW) + lane;
    const v4i *w21 = reinterpret_cast<const v4i *>(sW + (PackedOffsets::C2W21 - PackedOffsets::C2W22)) + lane;
    C2Consts kc;
#ifdef QCNN_STAMPS
    kc.dbg = a.dbg, kc.dbg_block = a.dbg_block, kc.row = wv, kc.idx = &stamp_idx;
#endif
    kc.q21 = a.rq[1], kc.q22 = a.rq[2];
    kc.b0 = kc.b1 = kc.b2 = v4i{0, 0, 0, 0};
    if constexpr (!EXACT) {
        kc.b0 = *reinterpret_cast<const v4i *>(a.bias_i + BIAS_OFF[1] + 4 * g);
        kc.b1 = *reinterpret_cast<const v4i *>(a.bias_i + BIAS_OFF[1] + 16 + 4 * g);
        kc.b2 = *reinterpret_cast<const v4i *>(a.bias_i + BIAS_OFF[2] + 4 * g);
#pragma unroll
        for (int i = 0; i < 12; ++i) kc.bf[i] = 0.f;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            kc.bf[i] = a.bias_f[BIAS_OFF[1] + 4 * g + i];
            kc.bf[4 + i] = a.bias_f[BIAS_OFF[1] + 16 + 4 * g + i];
            kc.bf[8 + i] = a.bias_f[BIAS_OFF[2] + 4 * g + i];
        }
    }

    if (wv < 4) {
        // =============================== producer ===============================
        // Producer p owns C1 region rows 5p .. 5p+4 (all C1C columns): 180 pixels = 6 MFMA pixel
        // tiles of 32.  It needs Y tile rows 5p .. 5p+8, which it loads itself into a private LDS
        // image, so producers never wait for each other.
        //
        // The private image holds FOUR copies of the Y tile, copy c shifted left by c bytes, so the
        // 8 bytes Y[row][j .. j+7] of any column j are two aligned dwords of copy j&3: the im2col
        // B fragment of a pixel is read straight from LDS.  C1 is then one 32x32x32 MFMA per
        // 32-channel tile: k-group h holds kernel rows 3h .. 3h+2, five bytes of each packed into 15
        // K positions with four byte permutes; the 16th position and kernel rows >= 5 have zero
        // weights, so whatever those bytes hold is multiplied by zero.
        const int p = wv;
        set_prio<C2_PRIO_PRODUCER>();
        unsigned char *sYw = sYall + p * SYW_BYTES;  // [4][YWR][YWS] int8 (x - 128), 0 outside the frame
        const v4i *wp = reinterpret_cast<const v4i *>(a.wpack + PackedOffsets::C1W) + lane;
        const v4i A0 = wp[0], A1 = wp[64];  // one fragment per 32-channel tile
        const int h = lane >> 5;
        const Requant q = a.rq[0];
        v16i bias0 = {}, bias1 = {};
        float bf0[16], bf1[16];
        if constexpr (!EXACT) {
            bias0 = *reinterpret_cast<const v16i *>(a.bias_i + 16 * h);
            bias1 = *reinterpret_cast<const v16i *>(a.bias_i + 32 + 16 * h);
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                bf0[i] = a.bias_f[16 * h + i];
                bf1[i] = a.bias_f[32 + 16 * h + i];
            }
        }
        // per-lane constants of pixel group k, for a cold tile (mode 0: region rows 5p .. 5p+4, 180 pixels, 6
        // groups) and for a warm tile (mode 1: rows 4+4p .. 7+4p, 144 pixels, 5 groups): LDS read address
        // (k-group h = kernel rows 3h .. 3h+2), LDS write address, region coordinates.  Only the addresses of the
        // warm tiles are kept in registers; cold tiles (the first of a workgroup and of a tile column) and the
        // coordinates (frame-edge tiles only) are worked out where they are needed.
        constexpr int NKC = (PROD_ROWS * C1C + 31) / 32, NKW = (WARM_ROWS * C1C + 31) / 32;
        auto group_consts = [&](int mode, int k, int &rdk, int &wrk, int &ijk, bool keep = false) {
            const int rows = mode ? WARM_ROWS : PROD_ROWS, row0 = mode ? 4 + WARM_ROWS * p : PROD_ROWS * p;
            const int npix = rows * C1C;
            int l = lane;
            // (opaque to the compiler, or it would hoist the results out of the tile loop into registers again)
            if (!keep) asm volatile("" : "+v"(l));
            const int qi = k * 32 + (l & 31);
            const int qc = qi < npix ? qi : npix - 1;
            const int il = qc / C1C, j = qc - il * C1C;
            const int cc = j & 3;
            rdk = cc * YCOPY + (il + 3 * h) * YWS + (j - cc) + 4;
            const int i = row0 + il;
            wrk = (i * C1C + j) * 64 + swz(h, j);
            ijk = i | (j << 8);
        };
        int rdw[NKW], wrw[NKW];
#pragma unroll
        for (int k = 0; k < NKW; ++k) {
            int unused;
            group_consts(1, k, rdw[k], wrw[k], unused, true);
        }
        const bool last_valid[2] = {(NKC - 1) * 32 + (lane & 31) < PROD_ROWS * C1C, (NKW - 1) * 32 + (lane & 31) < WARM_ROWS * C1C};
        constexpr int YLOADS = (YROWS * YWC + 63) / 64;  // bytes of the private Y tile per lane
        // byte `it` of this lane (frame-edge tiles only): row and column relative to the tile's first halo pixel, LDS
        // offset in copy 0 (copy c sits c * (YCOPY - 1) further; -1: no byte)
        auto ybyte = [&](int it, int &row, int &col) {
            int l = lane;
            asm volatile("" : "+v"(l));  // (as above)
            const int idx = it * 64 + l;
            row = idx / YWC, col = idx - row * YWC;
            return idx < YROWS * YWC ? row * YWS + col + 4 : -1;
        };
        // The same tile in one load, for tiles whose halo (and 3 bytes beyond it) lies inside the frame:
        // lane (row, j), j = 0..4, loads the 12 bytes 8j .. 8j+11 of tile row `row` and derives from them
        // dwords 2j and 2j+1 of all four copies (dword i of copy c holds columns 4i+c .. 4i+c+3 = a byte
        // alignment of two neighbouring dwords), so the wave issues one global load and eight
        // conflict-free ds_write_b32 instead of a load and four byte stores per 64 bytes (columns >= YWC
        // only meet zero weights).
        static_assert(YWC == 40 && YCOPY % 4 == 0, "five 8-byte pieces per tile row");
        const int yrow4 = lane / 5, yj4 = lane - yrow4 * 5;
        const bool ylane4 = lane < 5 * YROWS;
        static_assert(5 * YROWS <= 64, "one lane per 8-byte piece of the private Y tile");
        const int goff4 = ylane4 ? yrow4 * W + 8 * yj4 : 0;
        const int lwr4 = yrow4 * YWS + 4 + 8 * yj4;  // copy c sits c * YCOPY further
        unsigned char ypre[YLOADS];
        unsigned ypre4[3];
        bool pre_dwords = false;  // which of the two staging sets the last fetch_y filled
        auto fetch_y = [&](const TilePos &tp, bool warm) {
            // first halo pixel of this producer's rows
            const int xa = tp.tx * TW - 4, ya = tp.ty * TH - 4 + (warm ? 4 + WARM_ROWS * p : PROD_ROWS * p);
            const uint8_t *yf = a.y + (size_t)tp.f * a.frame_stride;
            pre_dwords = xa >= 0 && xa + YWC + 4 <= W && ya >= 0 && ya + YROWS <= H;
            if (pre_dwords) {  // inside the frame: raw bytes, turned into x - 128 when they are written to LDS
                const uint8_t *org = yf + (size_t)ya * W + xa;
                if (ylane4) __builtin_memcpy(ypre4, org + goff4, 12);
            } else {  // frame edge: clamped addresses; I1.ppro of the zero padding is (x - 128) == 0
#pragma unroll
                for (int it = 0; it < YLOADS; ++it) {
                    int yr, yc;
                    ybyte(it, yr, yc);
                    const int gy = ya + yr, gx = xa + yc;
                    const int cy = gy < 0 ? 0 : gy >= H ? H - 1 : gy, cx = gx < 0 ? 0 : gx >= W ? W - 1 : gx;
                    const unsigned char v = yf[(size_t)cy * W + cx];
                    ypre[it] = (cy == gy && cx == gx) ? (v ^ 0x80) : 0;
                }
            }
        };
        TilePos nxt = pos, prev = pos;  // tiles n + 1 and n - 1 while tile n is computed
        if (N > 0) fetch_y(pos, false);
        advance(nxt);
        // this wave's share of the C2 stage of tile n-1: the last rows of its consumer's block
        auto c2_share = [&](int n) {
            if constexpr (C2_CONS_ROWS < C2_BLOCK_ROWS) {
                if (n >= 1) {
                    C2Acc<C2_BLOCK_ROWS - C2_CONS_ROWS> acc;
                    c2_mfma<EXACT, C2_BLOCK_ROWS - C2_CONS_ROWS>(acc, smem + ((n - 1) & 1) * SC1_BYTES, R0 + C2_CONS_ROWS, cb, c, g, w22, w21, kc,
                                                     [] {});
                    c2_store<PATH, C2_BLOCK_ROWS - C2_CONS_ROWS>(acc, R0 + C2_CONS_ROWS, cb, c, g, kc,
                                                      a.conc1 + QCNN_CONC_OFF(a, prev.f), prev.tx * TW, prev.ty * TH, H, W);
                }
            }
        };
        for (int n = 0; n <= N; ++n) {
            STAMP();
#if C2_SHARE_FIRST
            // first, while the consumer of this SIMD requantises and stores (vector ALU only, matrix pipe idle)
            c2_share(n);
#endif
            mid_barrier();
            if (n < N) {
                const int x0 = pos.tx * TW, y0 = pos.ty * TH;
                const bool warm = n > 0 && pos.ty > 0;  // the tile above was this workgroup's previous tile
                unsigned char *buf = smem + (n & 1) * SC1_BYTES;
#if QCNN_A_FLAGS
                static_assert(C2_CONS_ROWS == C2_BLOCK_ROWS, "the producers' share of the C2 rows is not handed off by counters");
                // the tile above is complete (its last rows are copied), and every consumer has read the tile that
                // sat in this buffer
                if (warm) flag_wait(sProduced, 4u * n, alive);
                if (n >= 2) flag_wait(sConsumed, 4u * C2_CONSUMERS * (n - 1), alive);
#endif
                if (pre_dwords) {
                    if (ylane4) {
                        const unsigned d0 = ypre4[0] ^ 0x80808080u, d1 = ypre4[1] ^ 0x80808080u, d2 = ypre4[2] ^ 0x80808080u;
#pragma unroll
                        for (int cc = 0; cc < 4; ++cc) {
                            unsigned *dst = reinterpret_cast<unsigned *>(sYw + cc * YCOPY + lwr4);
                            dst[0] = cc ? __builtin_amdgcn_alignbyte(d1, d0, cc) : d0;
                            dst[1] = cc ? __builtin_amdgcn_alignbyte(d2, d1, cc) : d1;
                        }
                    }
                } else {
#pragma unroll
                    for (int it = 0; it < YLOADS; ++it) {
                        int yr, yc;
                        const int ywr = ybyte(it, yr, yc);
                        if (ywr >= 0) {
#pragma unroll
                            for (int cc = 0; cc < 4; ++cc) sYw[ywr + cc * (YCOPY - 1)] = ypre[it];
                        }
                    }
                }
                if (n + 1 < N) fetch_y(nxt, nxt.ty > 0);  // next tile's bytes travel while this one is computed
                // frame-edge tiles zero the C1 outputs that lie outside the frame (zero padding of C2)
                const bool edge = y0 < 2 || y0 - 2 + C1R > H || x0 < 2 || x0 - 2 + C1C > W;
                auto produce = [&](auto modec) {
                    constexpr int MODE = decltype(modec)::value;
                    constexpr int NK = MODE ? NKW : NKC;
                    // C1 MFMAs of pixel group k: the B fragments come straight from the shifted copies
                    auto c1_mfma = [&](int k, v16i &d0, v16i &d1) {
                        // rows 3h, 3h+1, 3h+2 of the window, 8 bytes each; the first 5 of each row are
                        // packed into 15 consecutive K positions (the 16th has zero weight)
                        int rdk, wrk, ijk;
                        if constexpr (MODE) rdk = rdw[k];
                        else group_consts(0, k, rdk, wrk, ijk);
                        const unsigned *src = reinterpret_cast<const unsigned *>(sYw + rdk);
                        const unsigned r0l = src[0], r0h = src[1];
                        const unsigned r1l = src[YWS / 4], r1h = src[YWS / 4 + 1];
                        const unsigned r2l = src[YWS / 2], r2h = src[YWS / 2 + 1];
                        v4i B;
                        B.x = (int)r0l;
                        B.y = (int)__builtin_amdgcn_perm(r1l, r0h, 0x06050400u);  // r0[4] r1[0] r1[1] r1[2]
                        B.z = (int)__builtin_amdgcn_perm(r2l, __builtin_amdgcn_alignbyte(r1h, r1l, 3), 0x05040100u);  // r1[3] r1[4] r2[0] r2[1]
                        B.w = (int)__builtin_amdgcn_alignbyte(r2h, r2l, 2);        // r2[2] r2[3] r2[4] (zero weight)
                        d0 = mfma32(A0, B, bias0);
                        d1 = mfma32(A1, B, bias1);
                    };
                    // requant + store of pixel group k
                    auto c1_store = [&](int k, const v16i &d0, const v16i &d1) {
                        v4i o0, o1;
#ifdef QCNN_ABL_NOC1REQ  // diagnostic builds only (garbage results): the producers' requantisation switched off
                        if constexpr (true) {
                            o0 = v4i{d0[0] ^ d0[4] ^ d0[8] ^ d0[12], d0[1] ^ d0[5] ^ d0[9] ^ d0[13], d0[2] ^ d0[6] ^ d0[10] ^ d0[14], d0[3] ^ d0[7] ^ d0[11] ^ d0[15]};
                            o1 = v4i{d1[0] ^ d1[4] ^ d1[8] ^ d1[12], d1[1] ^ d1[5] ^ d1[9] ^ d1[13], d1[2] ^ d1[6] ^ d1[10] ^ d1[14], d1[3] ^ d1[7] ^ d1[11] ^ d1[15]};
                        } else
#endif
                        if constexpr (!EXACT) {
                            o0 = requant16_int(d0, q);
                            o1 = requant16_int(d1, q);
                        } else {
                            o0.x = quant4<true>(d0[0], d0[1], d0[2], d0[3], bf0[0], bf0[1], bf0[2], bf0[3], q);
                            o0.y = quant4<true>(d0[4], d0[5], d0[6], d0[7], bf0[4], bf0[5], bf0[6], bf0[7], q);
                            o0.z = quant4<true>(d0[8], d0[9], d0[10], d0[11], bf0[8], bf0[9], bf0[10], bf0[11], q);
                            o0.w = quant4<true>(d0[12], d0[13], d0[14], d0[15], bf0[12], bf0[13], bf0[14], bf0[15], q);
                            o1.x = quant4<true>(d1[0], d1[1], d1[2], d1[3], bf1[0], bf1[1], bf1[2], bf1[3], q);
                            o1.y = quant4<true>(d1[4], d1[5], d1[6], d1[7], bf1[4], bf1[5], bf1[6], bf1[7], q);
                            o1.z = quant4<true>(d1[8], d1[9], d1[10], d1[11], bf1[8], bf1[9], bf1[10], bf1[11], q);
                            o1.w = quant4<true>(d1[12], d1[13], d1[14], d1[15], bf1[12], bf1[13], bf1[14], bf1[15], q);
                        }
                        int rdk, wrk, ijk;
                        if constexpr (MODE) wrk = wrw[k];
                        else group_consts(0, k, rdk, wrk, ijk);
                        if (edge) {
                            group_consts(MODE, k, rdk, wrk, ijk);
                            const int gy = y0 - 2 + (ijk & 0xff), gx = x0 - 2 + (ijk >> 8);
                            if (!(gy >= 0 && gy < H && gx >= 0 && gx < W)) {
                                o0 = v4i{0, 0, 0, 0};
                                o1 = v4i{0, 0, 0, 0};
                            }
                        }
#ifdef QCNN_ABL_NOC1WRITE  // diagnostic builds only (garbage results): the C1 results are computed and not written to LDS
                        asm volatile("" ::"v"(o0), "v"(o1));
                        if (wrk < 0)
#else
                        if (k + 1 < NK || last_valid[MODE])
#endif
                        {
                            *reinterpret_cast<v4i *>(buf + wrk) = o0;         // channels 16h .. 16h+15
                            *reinterpret_cast<v4i *>(buf + (wrk ^ 32)) = o1;  // channels 32+16h .. 32+16h+15
                        }
                    };
                    // software pipeline: the MFMAs of group k+1 are in flight while group k is requantised
                    v16i dA0, dA1, dB0, dB1;
                    c1_mfma(0, dA0, dA1);
                    static_for<NK>([&](auto kc) {
                        constexpr int k = decltype(kc)::value;
                        if constexpr (k + 1 < NK) {
                            if constexpr (k & 1) c1_mfma(k + 1, dA0, dA1);
                            else c1_mfma(k + 1, dB0, dB1);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (k & 1) c1_store(k, dB0, dB1);
                        else c1_store(k, dA0, dA1);
                    });
                };
                if (warm) {
                    // region rows 0..3 = rows TH..TH+3 of the tile above, which sits in the other buffer: a
                    // straight copy of 4 x 36 x 64 bytes (the chunk rotation depends on the column only)
                    const v4i *src = reinterpret_cast<const v4i *>(smem + ((n - 1) & 1) * SC1_BYTES) + TH * C1C * 4;
                    v4i *dst = reinterpret_cast<v4i *>(buf);
                    constexpr int PIECES = 4 * C1C * 4;
#pragma unroll
                    for (int i0 = 0; i0 < PIECES; i0 += 256) {
                        const int i = i0 + p * 64 + lane;
                        if (i < PIECES) dst[i] = src[i];
                    }
                }
#ifndef QCNN_ABL_NOPRODUCE  // diagnostic builds only: the consumers alone (on whatever LDS holds)
                if (warm) produce(std::integral_constant<int, 1>{});
                else produce(std::integral_constant<int, 0>{});
#endif
            }
            STAMP();
#if !C2_SHARE_FIRST
            c2_share(n);
#endif
#if QCNN_A_FLAGS
            if (n < N) flag_signal(sProduced, lane);
#endif
            if (n < N) {
                prev = pos;
                pos = nxt;
                advance(nxt);
            }
            STAMP();
#if !QCNN_A_FLAGS
            step_barrier();
#endif
        }
    } else {
        // =============================== consumer ===============================
        // rows R0c .. R0c + C2_ROWS - 1 of the SIMD's block of tile n-1.
        // A deferring consumer, in step n, issues the first LDS reads of tile n-1, requantises and stores tile
        // n-2 (whose accumulators it still holds) while they travel, then runs the MFMAs of tile n-1.  The second
        // consumer of a SIMD (and the float pipeline, which needs its registers for the requant) stores a tile
        // right after its MFMAs.
        // (convc / rowsc: the convolutions and the number of rows of the block this wave computes)
        auto consumer = [&](auto deferc, int half, auto convc, auto rowsc) {
            constexpr bool DEFER = decltype(deferc)::value;
            constexpr int CONV = decltype(convc)::value, C2_ROWS = decltype(rowsc)::value;
            const int R0c = R0 + half * C2_ROWS;
            C2Acc<C2_ROWS> acc;
            TilePos held = pos;  // the tile whose results `acc` holds
            bool have = false;
            auto store_held = [&] {
                if (have)
                    c2_store<PATH, C2_ROWS, CONV>(acc, R0c, cb, c, g, kc, a.conc1 + QCNN_CONC_OFF(a, held.f), held.tx * TW,
                                                   held.ty * TH, H, W);
            };
            auto store_then_release = [&] {
                store_held();
                mid_barrier();
                set_prio<C2_PRIO_D_MFMA>();
            };
            for (int n = 0; n <= N; ++n) {
                STAMP();
                if (n >= 1) {
#ifdef QCNN_ABL_NOCONSUME  // diagnostic builds only: the producers alone
                    mid_barrier();
#else
                    const unsigned char *buf = smem + ((n - 1) & 1) * SC1_BYTES;
#if QCNN_A_FLAGS
                    flag_wait(sProduced, 4u * n, alive);  // tile n-1 is complete
#endif
                    if constexpr (DEFER) {
                        c2_mfma<EXACT, C2_ROWS, CONV>(acc, buf, R0c, cb, c, g, w22, w21, kc, store_then_release);
#if QCNN_A_FLAGS
                        flag_signal(sConsumed, lane);  // (the last fragment of the tile was read before its last MFMA)
#endif
                        set_prio<C2_PRIO_D_VALU>();
                        held = pos;
                        have = true;
                    } else {
                        set_prio<C2_PRIO_N_MFMA>();
                        c2_mfma<EXACT, C2_ROWS, CONV>(acc, buf, R0c, cb, c, g, w22, w21, kc, [] { mid_barrier(); });
#if QCNN_A_FLAGS
                        flag_signal(sConsumed, lane);
#endif
                        set_prio<C2_PRIO_N_VALU>();
                        c2_store<PATH, C2_ROWS, CONV>(acc, R0c, cb, c, g, kc, a.conc1 + QCNN_CONC_OFF(a, pos.f), pos.tx * TW,
                                                       pos.ty * TH, H, W);
                    }
#endif
                    advance(pos);
                } else {
                    mid_barrier();
                }
                STAMP();
#if !QCNN_A_FLAGS
                step_barrier();
#endif
            }
            if constexpr (DEFER) store_held();
        };
        constexpr std::integral_constant<int, 3> both{};
        constexpr std::integral_constant<int, C2_CONS_ROWS / C2_CONSUMERS> part{};  // rows of one consumer wave
        if constexpr (EXACT) {
            consumer(std::false_type{}, (wv - 4) >> 2, both, part);
        } else if constexpr (C2_CONSUMERS == 1) {
            consumer(std::true_type{}, 0, both, part);
        } else if constexpr (C2_SPLIT_CONV != 0 && PATH == A_INT) {
            // experiment: the two consumers of a SIMD split the block by convolution, all its rows each
            constexpr std::integral_constant<int, C2_BLOCK_ROWS> all{};
            constexpr std::integral_constant<int, C2_SPLIT_CONV == 1 ? 1 : 2> conv_old{};  // 1: the older wave takes C2_2
            constexpr std::integral_constant<int, C2_SPLIT_CONV == 1 ? 2 : 1> conv_young{};
            if (wv < 8) consumer(std::false_type{}, 0, conv_old, all);
            else consumer(std::true_type{}, 0, conv_young, all);
        } else {
#if C2_DEFER_YOUNGER == 2  // experiment: neither consumer defers (both matrix phases, then both requant phases, side by side)
            consumer(std::false_type{}, wv < 8 ? 0 : 1, both, part);
#elif C2_DEFER_YOUNGER
            if (wv < 8) consumer(std::false_type{}, 0, both, part);
            else consumer(std::true_type{}, 1, both, part);
#else
            if (wv < 8) consumer(std::true_type{}, 0, both, part);
            else consumer(std::false_type{}, 1, both, part);
#endif
        }
    }
#ifdef QCNN_STAMPS  // diagnostic builds only: start, end and tile count per wave 0 / 4 / 8 of every workgroup
    if (a.dbg && lane == 0 && (wv == 0 || wv == 4 || wv == 8) && blockIdx.x < 2048) {
        unsigned long long blk_t1;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(blk_t1)::"memory");
        unsigned long long *o = a.dbg + 24 * 128 + 3 * (size_t)(8192 + (wv >> 2) * 2048 + blockIdx.x);
        o[0] = blk_t0;
        o[1] = blk_t1;
        o[2] = (unsigned long long)N;
    }
#endif
}

// mode: 0 integer epilogues, 1 float pipeline, 2 integer epilogues with clamped C2 accumulators
hipError_t launch_fused_c1c2(const FusedArgs &a, int mode, hipStream_t s)
{
    const int ntiles = ((a.W + ka::TW - 1) / ka::TW) * ((a.H + ka::TH - 1) / ka::TH) * a.nframes;
    if (ntiles < 1) return hipSuccess;
    int ncu = a.ncu > 0 ? a.ncu : 256;
    if (ncu > A_MAX_GROUPS) ncu = A_MAX_GROUPS;
    const int groups = ntiles < ncu ? ntiles : ncu;
    // One persistent workgroup per CU; the launch lasts as long as the workgroup with the most work.  A tile in the
    // leftmost or rightmost tile column of a frame costs more than the others (all four producers stage its Y pixels
    // byte by byte with clamped addresses): 1.042 times as much, from a least-squares fit of the workgroups' durations
    // on their tile counts by kind (tools/stamps_a3.py).  The runs are cut at equal cost: with runs of equal length
    // the workgroups that hold two such columns ended 2.4 % after the median one.
    const int tiles_x = (a.W + ka::TW - 1) / ka::TW, tiles_y = (a.H + ka::TH - 1) / ka::TH;
    const double edge_cost = getenv("QCNN_A_EQUAL_RUNS") ? 1.0 : 1.042;
    TileRuns runs;
    {
        const long long columns = (long long)tiles_x * a.nframes;
        auto column_cost = [&](long long col) {
            const int tx = (int)(col % tiles_x);
            return (tx == 0 || tx == tiles_x - 1) ? edge_cost : 1.0;
        };
        double total = 0;
        for (long long c = 0; c < columns; ++c) total += column_cost(c) * tiles_y;
        long long col = 0;
        double before = 0;  // cost of the columns in front of `col`
        runs.start[0] = 0;
        for (int b = 1; b < groups; ++b) {
            const double target = total * b / groups;
            while (col < columns && before + column_cost(col) * tiles_y <= target) before += column_cost(col++) * tiles_y;
            long long k = col * tiles_y + (col < columns ? (long long)((target - before) / column_cost(col) + 0.5) : 0);
            if (k < runs.start[b - 1] + 1) k = runs.start[b - 1] + 1;  // (every workgroup gets a tile: groups <= ntiles)
            if (k > ntiles - (groups - b)) k = ntiles - (groups - b);
            runs.start[b] = (int)k;
        }
        runs.start[groups] = ntiles;
    }
    dim3 grid(groups);
    if (mode == A_FLOAT) hipLaunchKernelGGL(k_c1c2<A_FLOAT>, grid, dim3(A_THREADS), 0, s, a, runs);
    else if (mode == A_WIDE) hipLaunchKernelGGL(k_c1c2<A_WIDE>, grid, dim3(A_THREADS), 0, s, a, runs);
    else hipLaunchKernelGGL(k_c1c2<A_INT>, grid, dim3(A_THREADS), 0, s, a, runs);
    return hipGetLastError();
}

}  // namespace qcnn
