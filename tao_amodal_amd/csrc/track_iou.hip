// 3D IoU of track pairs (reference tao_amodal/eval.py:15-117, 306-335), gfx950.
//
//   track_iou_task_kernel  the planned path: one wavefront per task (<= 32
//                          tracks, <= 64 track pairs), frames staged in LDS
//                          chunk by chunk of the timeline from the padded frame
//                          table, lane = pair adds the per-frame terms in order
//   track_pad_*            CSR frame lists -> padded frame table
//   track_iou_kernel       plan-less fallback: lane per pair, two-pointer merge
//
// fp64 elementwise / compare work, bound by instruction issue and HBM; nothing
// is shaped into a GEMM.
#include <algorithm>
#include <vector>

#include "common.hpp"
#include "pyset.hpp"

using namespace taoamd;

// ------------------------------------------------------------------ 3D IoU
// upper_bound(off, n+1 entries, p) - 1: the cell whose pair range holds p
__device__ __forceinline__ int64_t find_cell(const int64_t *__restrict__ off,
                                             int64_t n_cells, int64_t p)
{
    int64_t lo = 0, hi = n_cells;  // invariant: off[lo] <= p < off[hi]
    while (hi - lo > 1) {
        int64_t mid = (lo + hi) >> 1;
        if (off[mid] <= p) lo = mid; else hi = mid;
    }
    return lo;
}

// A track's frame list read through a 3-deep register queue: the loads of
// frame p+2 are issued when frame p becomes current, so the two-pointer merge
// below never waits on a load it has just issued (the merge is a chain of
// data-dependent steps; without the queue every step pays two serialized
// memory round trips).
struct FrameQueue {
    const int32_t *__restrict__ pos;
    const double4 *__restrict__ box;
    int32_t p, e;
    int32_t f0, f1, f2;
    double4 b0, b1, b2;

    __device__ __forceinline__ void load(int32_t q, int32_t &f, double4 &b) const
    {
        if (q < e) { f = pos[q]; b = box[q]; } else { f = INT32_MAX; }
    }
    __device__ __forceinline__ void init(const int32_t *pos_, const double *box_,
                                         int32_t start, int32_t end)
    {
        pos = pos_; box = reinterpret_cast<const double4 *>(box_);
        p = start; e = end;
        b0 = b1 = b2 = make_double4(0, 0, 0, 0);
        load(p, f0, b0); load(p + 1, f1, b1); load(p + 2, f2, b2);
    }
    __device__ __forceinline__ void advance()
    {
        p++;
        f0 = f1; b0 = b1;
        f1 = f2; b1 = b2;
        load(p + 2, f2, b2);
    }
};

__global__ __launch_bounds__(256) void track_iou_kernel(
    int64_t n_cells, const int32_t *__restrict__ cell_dt_off,
    const int32_t *__restrict__ cell_gt_off,
    const int64_t *__restrict__ cell_iou_off, int64_t n_pairs,
    const int32_t *__restrict__ dfoff, const int32_t *__restrict__ dfpos,
    const double *__restrict__ dfbox, const int32_t *__restrict__ gfoff,
    const int32_t *__restrict__ gfpos, const double *__restrict__ gfbox,
    double *__restrict__ iou, unsigned long long *__restrict__ pair_frames,
    int mode)
{
    int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    unsigned long long common = 0;
    bool mine = p < n_pairs;
    int64_t c = 0;
    if (mine) {
        c = find_cell(cell_iou_off, n_cells, p);
    }
    if (mine) {
        const int32_t G = cell_gt_off[c + 1] - cell_gt_off[c];
        const int64_t local = p - cell_iou_off[c];
        const int32_t d = (int32_t)(local / G), g = (int32_t)(local - (int64_t)d * G);
        const int32_t td = cell_dt_off[c] + d, tg = cell_gt_off[c] + g;
        FrameQueue qd, qg;
        qd.init(dfpos, dfbox, dfoff[td], dfoff[td + 1]);
        qg.init(gfpos, gfbox, gfoff[tg], gfoff[tg + 1]);
        double i = 0.0, u = 0.0, acc = 0.0, cnt = 0.0;
        // ascending timeline order; per frame exactly the arithmetic of
        // reference tao_amodal/eval.py:32-48 and :87-94
        while (qd.f0 != INT32_MAX || qg.f0 != INT32_MAX) {
            const double4 B = qd.b0, A = qg.b0;
            cnt += 1.0;
            if (qd.f0 == qg.f0) {
                double w = fmin(B.x + B.z, A.x + A.z) - fmax(B.x, A.x);
                double h = fmin(B.y + B.w, A.y + A.w) - fmax(B.y, A.y);
                w = w > 0 ? w : 0.0;
                h = h > 0 ? h : 0.0;
                const double i_ = w * h;
                const double u_ = B.z * B.w + A.z * A.w - i_;
                i += i_;
                u += u_;
                if (mode == 1) acc += u_ > 0 ? i_ / u_ : 0.0;
                if (mode == 2 && i_ > 0.5 * u_) acc += 1.0;
                common++;
                qd.advance();
                qg.advance();
            } else if (qg.f0 < qd.f0) {
                u += A.z * A.w;
                qg.advance();
            } else {
                u += B.z * B.w;
                qd.advance();
            }
        }
        iou[p] = mode == 0 ? (u > 0 ? i / u : 0.0) : acc / cnt;
    }
    if (pair_frames != nullptr) {
        for (int s = WAVE / 2; s > 0; s >>= 1)
            common += __shfl_down(common, s, WAVE);
        if (lane_id() == 0 && common) atomicAdd(pair_frames, common);
    }
}

// Every track ONE frame (the stress shape: thousands of one-frame videos): a
// pair's 3D IoU is one box IoU when the two frames are the same image, 0
// otherwise -- one term, so there is no order of summation either.  A thread
// per detection track walks the ground-truth tracks of its cell (dt_group:
// {first GT, GT count, place in the cell, cell}); consecutive threads read
// consecutive frames and write consecutive rows of the IoU matrices.  The task
// kernel's LDS pipeline is all prologue here: 1.05 ms for 2.9 M tracks.
__global__ __launch_bounds__(256) void track_iou_single_kernel(
    int64_t n_dt, const int4 *__restrict__ dt_group,
    const int64_t *__restrict__ cell_iou_off, const int32_t *__restrict__ dpos,
    const double4 *__restrict__ dbox, const int32_t *__restrict__ gpos,
    const double4 *__restrict__ gbox, double *__restrict__ iou,
    unsigned long long *__restrict__ pair_frames, int mode)
{
    // a fixed grid walks the tracks (consecutive threads, consecutive tracks):
    // the common-frame counter is then a few hundred atomic adds on one word
    // instead of one per wavefront (45 k of them at 2.9 M tracks: 0.5 ms of
    // serialised read-modify-writes, most of the kernel's first version)
    unsigned long long common = 0;
    for (int64_t d = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; d < n_dt;
         d += gridDim.x * (int64_t)blockDim.x) {
        const int4 grp = dt_group[d];
        const int32_t g0 = grp.x, G = grp.y;
        double *__restrict__ out = iou + cell_iou_off[grp.w] + (int64_t)grp.z * G;
        const int32_t pd = dpos[d];
        const double4 B = dbox[d];
        for (int32_t g = 0; g < G; g++) {
            double v = 0.0;
            if (gpos[g0 + g] == pd) {
                // reference tao_amodal/eval.py:32-48 on the one common frame
                const double4 A = gbox[g0 + g];
                double w = fmin(B.x + B.z, A.x + A.z) - fmax(B.x, A.x);
                double h = fmin(B.y + B.w, A.y + A.w) - fmax(B.y, A.y);
                w = w > 0 ? w : 0.0;
                h = h > 0 ? h : 0.0;
                const double i_ = w * h;
                const double u_ = B.z * B.w + A.z * A.w - i_;
                if (mode == 2) v = i_ > 0.5 * u_ ? 1.0 : 0.0;
                else v = u_ > 0 ? i_ / u_ : 0.0;
                common++;
            }
            out[g] = v;
        }
    }
    if (pair_frames != nullptr) {
        __shared__ unsigned long long s_common[4];
        for (int s = WAVE / 2; s > 0; s >>= 1)
            common += __shfl_down(common, s, WAVE);
        if (lane_id() == 0) s_common[threadIdx.x >> 6] = common;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long t = s_common[0] + s_common[1] + s_common[2] + s_common[3];
            if (t) atomicAdd(pair_frames, t);
        }
    }
}

// Task variant (the path every planned call takes).  A task = up to TT_ROWS
// tracks (of one or several cells) and up to 64 (detection track, GT track)
// pairs among them, run by a workgroup of THREE wavefronts with separate jobs:
//
//   stagers (waves 0, 1) fetch the tracks' frames and park them in LDS, each
//           the rows of every other round (tt_stager<0 / 1>),
//   adder   (wave 2) lane = track pair, adds the per-frame terms in order.
//
// They meet at one barrier per chunk of TT_P timeline positions; the frames
// live in two LDS buffers, so the stager fills chunk k + 1 while the adder
// walks chunk k, and the loads of chunks k + 2 and k + 3 are in flight
// meanwhile (two register sets).  The adder's instruction stream -- the one that bounds the
// kernel: 13 fp64 operations per pair and position, a chain of dependent adds
// -- therefore never waits for global memory or for the staging arithmetic.
// 24 KB of LDS per task: 6 tasks = 18 wavefronts per CU.
//
// Tracks are read from the PADDED frame table (taoamd_track_pad): the frames
// of a track occupy consecutive slots first .. last of the timeline, a
// position the track skips holds the "far box" (x = y = 1e300, w = h = 0), and
// so does slot 0 of the table.  The frame of track t at position p is
// padded[basem_t + p] -- no cursor, no search, no dependence between chunks.
//
//   stage   TT_P consecutive lanes = TT_P consecutive positions of one track
//           (contiguous bytes; positions outside first .. last read slot 0);
//           the boxes are parked as (x1, y1, x2 = x + w, y2 = y + h, area =
//           w * h) at [track][position - p0].  Rows that do not reach into
//           the chunk are not touched (wiped once if they held frames
//           before); the plan lists a task's tracks by first position, so
//           whole rounds of rows drop out.
//   add     i_ = max(min(x2) - max(x1), 0) * max(.., 0), u_ = (da + ga) - i_,
//           u += u_, i += i_ in ascending position -- the reference's
//           per-frame arithmetic (tao_amodal/eval.py:32-48, 87-94) with no
//           case split: against the far box the intersection is exactly 0 and
//           the union term is exactly the other box's area (x + 0 = x), and
//           two far boxes give (0, 0), whose addition is exact (u, i >= +0).
//           So the sequence of roundings equals the reference's walk over the
//           union of the two tracks' frames in timeline order.
// Chunks in which no track of the task has a frame are skipped by the adder;
// chunks that only hold GT frames add the areas alone.
#define TT_P 8                   // timeline positions per chunk
#define TT_ROWS 32               // tracks of a task
#define TT_RS (5 * TT_P + 2)     // doubles per row: rows 16-byte aligned, 20 banks apart
#define TT_GROUPS (64 / TT_P)    // rows served by one round of the stager
#define TT_ROUNDS ((TT_ROWS + TT_GROUPS - 1) / TT_GROUPS)
#define TT_SLOTS (TT_ROUNDS * TT_GROUPS)
#define TT_FAR 1e300
#define TT_SETS 2                // chunks between a load and its use (even)

// v_max_f64 / v_min_f64 as such: fmax() / fmin() make the compiler canonicalise
// every operand it cannot prove free of signalling NaNs (one extra v_max_f64
// per value read from LDS: 8 of 21 fp64 instructions per position)
__device__ __forceinline__ double tt_max(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double tt_min(double a, double b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double tt_pos(double a)       // max(a, 0)
{
    double r;
    asm("v_max_f64 %0, %1, 0" : "=v"(r) : "v"(a));
    return r;
}

// The chunk barrier of the two waves.  __syncthreads() would also drain the
// stager's global loads (its fence covers global memory: s_waitcnt vmcnt(0)),
// i.e. wait at every chunk for the prefetch that was just issued; the waves
// only exchange LDS data, so LDS traffic is all the barrier has to order.
__device__ __forceinline__ void tt_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// The stager's work of one chunk -- own rounds of rows only -- for stager S of
// two: rounds S, S + 2, ... (a phase trace of the kernel with ONE stager: a
// chunk took 2250 cycles for ~200 stager and ~160 adder instructions, two
// wavefronts per SIMD, i.e. one dependent instruction every ~10 cycles: the
// stager was the longer of the two chains).
template <int S, bool MASKS>
__device__ __forceinline__ void tt_stager(
    const double4 *__restrict__ frames, int64_t sbase, double (*rows)[TT_ROWS * TT_RS],
    const int4 *meta, int32_t p_first, int n_chunks, int lane)
{
    constexpr int NR = (TT_ROUNDS - S + 1) / 2;      // my rounds
    const int grp = lane / TT_P, j = lane % TT_P;
    // lane (grp, j) serves position j of the rows grp, TT_GROUPS + grp, ...
    uint32_t bit[NR];
#pragma unroll
    for (int i = 0; i < NR; i++) bit[i] = 1u << (TT_GROUPS * (S + 2 * i) + grp);
    // lanes 0 .. TT_SLOTS - 1: first / last of row = lane, for the chunk's
    // mask of present rows (one ballot per chunk, the same in both stagers)
    static_assert(TT_SLOTS <= 32, "one 32-bit mask of a task's rows");
    const int4 mrow = meta[lane & (TT_SLOTS - 1)];
    // (rows with frames, as a mask: a track without frames has first = INT32_MAX, last = -1)
    const uint32_t has_rows = (uint32_t)__builtin_amdgcn_ballot_w64(lane < TT_SLOTS && mrow.y >= mrow.x);
    // TT_SETS register sets: the loads of chunk k + TT_SETS are issued when
    // chunk k has been staged.  Every round loads (lanes out of range read
    // slot 0), so the number of loads in flight is known at compile time
    // and a chunk waits for ITS loads only (s_waitcnt vmcnt(n > 0)).
    double4 B[TT_SETS][NR];
    // Round 6 (the stagers' instruction chain is the kernel's critical one):
    // the mask of the rows present in a chunk is worked out ONCE, when the
    // chunk's loads are issued, and kept with its register set -- staging tests
    // "any row of this round" on the scalar unit and a lane's own bit with one
    // AND; the loads take a 32-bit byte offset from the task's own stretch of
    // the stream (a lane without a piece reads the stretch's first bytes: what
    // it loads is never parked), no 64-bit address arithmetic per lane.
    uint32_t pmv[TT_SETS];
    const char *const tbase = reinterpret_cast<const char *>(frames + sbase);
    uint32_t cur = 0;                 // bytes of the task's stretch requested so far (uniform)
    const uint32_t joff = (uint32_t)j * 32u;

    auto issue = [&](double4 *Bx, uint32_t &pm_set, int32_t pc) {
        // rows whose span reaches into the chunk own a piece of TT_P slots in
        // the chunk's stretch of the stream, in row order
        const uint32_t pm = (uint32_t)(__builtin_amdgcn_ballot_w64(mrow.x < pc + TT_P) &
                                       __builtin_amdgcn_ballot_w64(mrow.y >= pc)) & has_rows;
        pm_set = pm;
#pragma unroll
        for (int i = 0; i < NR; i++) {
            // (a lane whose row has no piece in the chunk computes the place of
            // the next present row's piece, or of the bytes behind the chunk's
            // stretch -- the stream ends in a margin: what it loads is never
            // parked, and no select is spent on it)
            const uint32_t off = cur + ((uint32_t)__popc(pm & (bit[i] - 1u)) << 8) + joff;
#ifdef TT_ABLATE_LOADS     // (timing experiment: every load hits the stretch's first bytes)
            Bx[i] = *reinterpret_cast<const double4 *>(tbase + (off & 0u));
#elif defined(TT_ABLATE_CACHED)   // (timing experiment: the same requests inside 1 MB)
            Bx[i] = *reinterpret_cast<const double4 *>(
                reinterpret_cast<const char *>(frames) + (((uint32_t)(sbase * 32) + off) & 0xfffe0u));
#else
            Bx[i] = *reinterpret_cast<const double4 *>(tbase + off);
#endif
        }
        cur += (uint32_t)__popc(pm) << 8;
    };
    auto stage = [&](const double4 *Bx, uint32_t pm, int b) {
        // A row is parked only where its span reaches into the chunk -- the one
        // case in which the adder reads it (it branches on the same test).
#pragma unroll
        for (int i = 0; i < NR; i++) {
            constexpr uint32_t ALL = (1u << TT_GROUPS) - 1u;
            if ((pm & (ALL << (TT_GROUPS * (S + 2 * i)))) == 0) continue;     // (scalar)
            const int r = TT_GROUPS * (S + 2 * i) + grp;
            const bool ov = (pm & bit[i]) != 0;               // same for the row's lanes
            const double4 bx = Bx[i];
            uint64_t ball = 0;
            if (MASKS) ball = __builtin_amdgcn_ballot_w64(bx.x != TT_FAR);
            if (ov) {
                double *rb = rows[b] + r * TT_RS + j;
                rb[0] = bx.x;
                rb[TT_P] = bx.y;
                rb[2 * TT_P] = bx.x + bx.z;
                rb[3 * TT_P] = bx.y + bx.w;
                rb[4 * TT_P] = bx.z * bx.w;
                if (MASKS && j == 0)     // positions of the chunk that hold a frame: the row's padding
                    *reinterpret_cast<uint32_t *>(rb + 5 * TT_P) =
                        (uint32_t)(ball >> (TT_P * grp)) & ((1u << TT_P) - 1);
            }
        }
    };
    // chunk c lives in register set c % TT_SETS and in LDS buffer c & 1
#pragma unroll
    for (int c = 0; c < TT_SETS; c++) issue(B[c], pmv[c], p_first + c * TT_P);
    if (n_chunks > 0) stage(B[0], pmv[0], 0);
    issue(B[0], pmv[0], p_first + TT_SETS * TT_P);
    tt_barrier();
    for (int k = 0; k < n_chunks; k += TT_SETS) {
#pragma unroll
        for (int a = 1; a <= TT_SETS; a++) {
            // the adder is on chunk k + a - 1: fill the other buffer with
            // chunk k + a
            const int c = k + a;
            if (c - 1 >= n_chunks) break;
            if (c < n_chunks) stage(B[a % TT_SETS], pmv[a % TT_SETS], a & 1);
            issue(B[a % TT_SETS], pmv[a % TT_SETS], p_first + (c + TT_SETS) * TT_P);
            tt_barrier();
        }
    }
}

// (six wavefronts per SIMD: 78 instead of 84 VGPRs, no spill, and seven tasks
// of 21.5 KB LDS per CU instead of six: 0.329 -> 0.314 ms at 2000 videos)
// COUNT: the pairs' common frames are counted (pair_frames).  The 3D IoU's
// arithmetic needs no frame masks -- far boxes make absent frames exact -- so a
// pass that does not count (every pass after a problem's first: the count is a
// constant of the problem) keeps them out of the stagers and the adder: 5 % of
// the kernel.
template <int MODE, bool COUNT = true>
__global__ __launch_bounds__(192) __attribute__((amdgpu_waves_per_eu(MODE == 2 ? 4 : 6, MODE == 2 ? 4 : 6))) void track_iou_task_kernel(
    const int4 *__restrict__ tasks, const int32_t *__restrict__ task_rows,
    const int32_t *__restrict__ task_pairs, const int64_t *__restrict__ task_out,
    const double4 *__restrict__ frames, const int32_t *__restrict__ task_base,
    const int4 *__restrict__ trk_meta,
    double *__restrict__ iou, unsigned long long *__restrict__ pair_frames)
{
    __shared__ __align__(16) double rows[2][TT_ROWS * TT_RS];
    __shared__ int4 meta[TT_SLOTS];
    __shared__ int32_t span[2];               // first chunk start, last position

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;        // 0, 1: stagers; 2: adder
    const bool stager = wave < 2;             // wave-uniform
    const int4 tk = tasks[blockIdx.x];        // {first row, rows, first pair, pairs}
    const int n_rows = tk.y, n_pairs = tk.w;

    if (wave == 0) {
        // ---- {first, last, base - first, is detection} of the task's tracks
        int32_t p_lo = INT32_MAX, p_hi = -1;
        if (lane < TT_SLOTS) {
            int4 m = make_int4(INT32_MAX, -1, 0, 0);       // no track: never in range
            if (lane < n_rows) m = trk_meta[task_rows[tk.x + lane]];
            if (m.y >= m.x) {
                p_lo = m.x;
                p_hi = m.y;
            } else {
                m.x = INT32_MAX;       // a track without frames: in no chunk
                m.y = -1;
            }
            meta[lane] = m;
        }
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) {
            p_lo = min(p_lo, __shfl_xor(p_lo, s));
            p_hi = max(p_hi, __shfl_xor(p_hi, s));
        }
        if (lane == 0) {
            span[0] = p_lo & ~(TT_P - 1);
            span[1] = p_hi;
        }
    }
    __syncthreads();
    const int32_t p_first = span[0], p_hi = span[1];
    const int n_chunks = p_hi < 0 ? 0 : (p_hi - p_first) / TT_P + 1;

    if (stager) {
        const int64_t sbase = (int64_t)task_base[blockIdx.x] * TT_P;
        constexpr bool MASKS = MODE != 0 || COUNT;
        if (wave == 0)
            tt_stager<0, MASKS>(frames, sbase, rows, meta, p_first, n_chunks, lane);
        else
            tt_stager<1, MASKS>(frames, sbase, rows, meta, p_first, n_chunks, lane);
        return;
    }

    // ---- adder
    int32_t pr = 0;
    if (lane < n_pairs) pr = task_pairs[tk.z + lane];
    const int rowd = pr & 0xFF, rowg = (pr >> 8) & 0xFF;
    // first / last position of my pair's two tracks: which of them reaches into
    // a chunk follows from registers (a lane without a pair: neither, ever)
    int32_t Fd = INT32_MAX, Ld = -1, Fg = INT32_MAX, Lg = -1;
    if (lane < n_pairs) {
        const int4 md = meta[rowd], mg = meta[rowg];
        Fd = md.x;
        Ld = md.y;
        Fg = mg.x;
        Lg = mg.y;
    }
    const int offd = rowd * (TT_RS * 8), offg = rowg * (TT_RS * 8);     // bytes
    double u = 0.0, i = 0.0;
    unsigned long long common = 0;
    auto add = [&](int b, int32_t pc) {
#ifdef TT_ABLATE_ADDER     // (timing experiment: the adder only keeps the barriers)
        return;
#endif
        // Round 6: the stagers park a row exactly where its span first .. last
        // reaches into the chunk, and the adder branches on the same test --
        // from four registers, where it used to wait for two LDS round trips
        // (the chunk's flags, then its rows' frame masks) before it could ask
        // for the first box: the adder's chain of dependent waits, not its
        // arithmetic, was a third of the kernel (a build without the general
        // step: -8 %; without the adder's skeleton: -40 %).  A chunk that lies
        // in a hole of a track finds far boxes there, whose terms are exact
        // zeros / the other box's area: the same bits as not reading it.
        const bool dp = Fd < pc + TT_P && Ld >= pc;
        const bool gp = Fg < pc + TT_P && Lg >= pc;
        if (__ballot(dp || gp) == 0) return;
#ifdef TT_ABLATE_ADD_BODY  // (timing experiment: 1 = no area-only step, 2 = no general step, 3 = neither)
        if (TT_ABLATE_ADD_BODY == 3) return;
#endif
        // (row offsets in two registers; the buffer's base folds into the LDS
        // instructions' immediate offsets)
        const char *rbase = reinterpret_cast<const char *>(rows[b]);
        const double *__restrict__ dr = reinterpret_cast<const double *>(rbase + offd);
        const double *__restrict__ gr = reinterpret_cast<const double *>(rbase + offg);
        if (MODE == 0) {
            // At most ONE of the pair's tracks in the chunk (two thirds of a
            // task's (pair, chunk) steps): that track's areas alone -- against
            // far boxes (x = 1e300, area 0) the general formula gives i_ = 0
            // and u_ = (a + 0) - 0 = a, the same bits.  Lanes branch: fewer
            // lanes, fewer LDS reads (round 4).
#ifdef TT_ABLATE_ADD_BODY
            if (dp != gp && !(TT_ABLATE_ADD_BODY & 1)) {
#else
            if (dp != gp) {
#endif
                const double2 *__restrict__ a2 =
                    reinterpret_cast<const double2 *>(rbase + (dp ? offd : offg) + 4 * TT_P * 8);
#pragma unroll
                for (int pp = 0; pp < TT_P / 2; pp++) {
                    const double2 ar = a2[pp];
                    u += ar.x;
                    u += ar.y;
                }
#ifdef TT_ABLATE_ADD_BODY
            } else if (dp && gp && !(TT_ABLATE_ADD_BODY & 2)) {
#else
            } else if (dp) {
#endif
                // two positions per 16-byte LDS read of every field
                const double2 *__restrict__ d2 = reinterpret_cast<const double2 *>(dr);
                const double2 *__restrict__ g2 = reinterpret_cast<const double2 *>(gr);
                uint32_t dm = 0, gm = 0;
                if (COUNT) {
                    dm = *reinterpret_cast<const uint32_t *>(dr + 5 * TT_P);
                    gm = *reinterpret_cast<const uint32_t *>(gr + 5 * TT_P);
                }
#pragma unroll
                for (int pp = 0; pp < TT_P / 2; pp++) {
                    const double2 dx1 = d2[pp], gx1 = g2[pp];
                    const double2 dy1 = d2[TT_P / 2 + pp], gy1 = g2[TT_P / 2 + pp];
                    const double2 dx2 = d2[TT_P + pp], gx2 = g2[TT_P + pp];
                    const double2 dy2 = d2[3 * TT_P / 2 + pp], gy2 = g2[3 * TT_P / 2 + pp];
                    const double2 da = d2[2 * TT_P + pp], ga = g2[2 * TT_P + pp];
                    {
                        const double w = tt_pos(tt_min(dx2.x, gx2.x) - tt_max(dx1.x, gx1.x));
                        const double h = tt_pos(tt_min(dy2.x, gy2.x) - tt_max(dy1.x, gy1.x));
                        const double i_ = w * h;
                        const double u_ = da.x + ga.x - i_;
                        u += u_;
                        i += i_;
                    }
                    {
                        const double w = tt_pos(tt_min(dx2.y, gx2.y) - tt_max(dx1.y, gx1.y));
                        const double h = tt_pos(tt_min(dy2.y, gy2.y) - tt_max(dy1.y, gy1.y));
                        const double i_ = w * h;
                        const double u_ = da.y + ga.y - i_;
                        u += u_;
                        i += i_;
                    }
                }
                if (COUNT) common += __popc(dm & gm);
            }
        } else {
            // avg_iou / imagenetvid: u = sum of per-frame scores, i = frames
            // (a row that does not reach into the chunk was not parked: its
            // frame mask counts as empty, what is read of it is not used)
            const uint32_t dm = dp ? *reinterpret_cast<const uint32_t *>(dr + 5 * TT_P) : 0u;
            const uint32_t gm = gp ? *reinterpret_cast<const uint32_t *>(gr + 5 * TT_P) : 0u;
#pragma unroll
            for (int pp = 0; pp < TT_P; pp++) {
                const bool both = ((dm & gm) >> pp) & 1u, either = ((dm | gm) >> pp) & 1u;
                const double x1 = tt_max(dr[pp], gr[pp]);
                const double y1 = tt_max(dr[TT_P + pp], gr[TT_P + pp]);
                const double x2 = tt_min(dr[2 * TT_P + pp], gr[2 * TT_P + pp]);
                const double y2 = tt_min(dr[3 * TT_P + pp], gr[3 * TT_P + pp]);
                const double w = tt_pos(x2 - x1), h = tt_pos(y2 - y1);
                const double i_ = w * h;
                const double u_ = dr[4 * TT_P + pp] + gr[4 * TT_P + pp] - i_;
                double tx;
                if (MODE == 1) tx = both ? (u_ > 0 ? i_ / u_ : 0.0) : 0.0;
                else tx = (both && i_ > 0.5 * u_) ? 1.0 : 0.0;
                u += tx;
                i += either ? 1.0 : 0.0;
            }
            common += __popc(dm & gm);
        }
    };
    tt_barrier();
    for (int k = 0; k < n_chunks; k += 2) {
        add(0, p_first + k * TT_P);
        tt_barrier();
        if (k + 1 >= n_chunks) break;
        add(1, p_first + (k + 1) * TT_P);
        tt_barrier();
    }
    if (lane < n_pairs)       // (the place is fetched here: two registers less across the loop)
        iou[task_out[tk.z + lane]] = MODE == 0 ? (u > 0 ? i / u : 0.0) : u / i;
    if (COUNT && pair_frames != nullptr) {
        for (int s_ = WAVE / 2; s_ > 0; s_ >>= 1)
            common += __shfl_down(common, s_, WAVE);
        if (lane == 0 && common) atomicAdd(pair_frames, common);
    }
}

// Frames of CSR tracks -> slots of the padded table (pre-filled with far boxes)
__global__ void track_pad_kernel(int64_t n_frames, int64_t n_trk,
                                 const int32_t *__restrict__ foff,
                                 const int32_t *__restrict__ fpos,
                                 const double4 *__restrict__ fbox,
                                 const int4 *__restrict__ meta,
                                 double4 *__restrict__ padded,
                                 int32_t *__restrict__ inexact)
{
    const int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (k >= n_frames) return;
    int64_t lo = 0, hi = n_trk;       // track of frame k: foff[lo] <= k < foff[hi]
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (foff[mid] <= k) lo = mid; else hi = mid;
    }
    const double4 b = fbox[k];
    padded[(int64_t)meta[lo].z + fpos[k]] = b;
    // integer coordinates below 2^20: every per-frame product (< 2^40) and
    // every sum of up to 2^13 of them is exact in fp64, so the order frames are
    // added in cannot matter (taoamd_track_iou_near has nothing to guard)
    if (inexact) {
        const double lim = 1048576.0;
        const bool exact = b.x == rint(b.x) && b.y == rint(b.y) && b.z == rint(b.z) &&
                           b.w == rint(b.w) && fabs(b.x) < lim && fabs(b.y) < lim &&
                           fabs(b.z) < lim && fabs(b.w) < lim;
        if (!exact && *inexact == 0) atomicOr(inexact, 1);
    }
}

// Guard of the one documented deviation: frames are added in timeline order,
// the reference adds them in CPython set order (tao_amodal/eval.py:83-94).
// With inexact per-frame terms the two sums may differ in the last bits, which
// only matters where a comparison can flip: an IoU within max_ulp of one of the
// ten thresholds, or of another GT's IoU in the same row (the greedy takes the
// best).  Those pairs are listed; the host recomputes them in set order.
__global__ void track_iou_near_kernel(int64_t n_cells,
                                      const int32_t *__restrict__ cell_gt_off,
                                      const int64_t *__restrict__ cell_iou_off,
                                      int64_t n_pairs, const double *__restrict__ iou,
                                      IouThr thr, int32_t max_ulp, int32_t cap,
                                      int32_t *__restrict__ count,
                                      int64_t *__restrict__ list)
{
    const int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    const double v = iou[p];
    if (!(v > 0)) return;          // no intersection at all: 0 in any order
    auto close = [max_ulp](double a, double b) {
        const int64_t d = __double_as_longlong(a) - __double_as_longlong(b);
        return (d < 0 ? -d : d) <= max_ulp;
    };
    bool near = false;
#pragma unroll
    for (int t = 0; t < N_THR; t++) near |= close(v, thr.v[t]);
    if (!near) {
        const int64_t c = find_cell(cell_iou_off, n_cells, p);
        const int32_t G = cell_gt_off[c + 1] - cell_gt_off[c];
        const int64_t local = p - cell_iou_off[c];
        const int64_t row = cell_iou_off[c] + (local / G) * G;
        for (int32_t g = 0; g < G && !near; g++) {
            const double o = iou[row + g];
            near = row + g != p && o > 0 && close(v, o);
        }
    }
    if (near) {
        const int32_t k = atomicAdd(count, 1);
        if (k < cap) list[k] = p;
    }
}

// The listed pairs recomputed ON THE DEVICE in the reference's order: a thread
// per pair restates CPython's ``set(gt.keys()) | set(dt.keys())`` (pyset.hpp)
// in its own scratch tables, walks the union's slots and adds the per-frame
// terms as the reference does -- no host round trip between the 3D IoU and the
// match.  Listed pairs are rare (an IoU within the reordering bound of a
// threshold or of a rival), so this is a handful of serial threads; a grid-
// stride loop over the list whose length is read from device memory.
__global__ __launch_bounds__(64) void track_iou_setorder_kernel(
    int64_t n_cells, const int32_t *__restrict__ cell_dt_off,
    const int32_t *__restrict__ cell_gt_off, const int64_t *__restrict__ cell_iou_off,
    const int32_t *__restrict__ cell_unit, const int64_t *__restrict__ tl_vid_start,
    const int64_t *__restrict__ tl_image_id, const int32_t *__restrict__ dfoff,
    const int32_t *__restrict__ dfpos, const double *__restrict__ dfbox,
    const int32_t *__restrict__ gfoff, const int32_t *__restrict__ gfpos,
    const double *__restrict__ gfbox, int mode, const int32_t *__restrict__ count,
    int32_t list_cap, const int64_t *__restrict__ list, int32_t *__restrict__ scratch,
    uint32_t table_cap, double *__restrict__ iou, int32_t *__restrict__ status)
{
    const int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t W = gridDim.x * (int64_t)blockDim.x;
    int64_t n = *count;
    if (n > list_cap) n = list_cap;
    int32_t *buf = scratch + 3 * w * (int64_t)table_cap;
    for (int64_t k = w; k < n; k += W) {
        const int64_t p = list[k];
        const int64_t c = find_cell(cell_iou_off, n_cells, p);
        const int32_t G = cell_gt_off[c + 1] - cell_gt_off[c];
        const int64_t local = p - cell_iou_off[c];
        const int32_t td = cell_dt_off[c] + (int32_t)(local / G);
        const int32_t tg = cell_gt_off[c] + (int32_t)(local % G);
        const pyset::Frames D{dfpos + dfoff[td], dfbox + 4 * (int64_t)dfoff[td],
                              dfoff[td + 1] - dfoff[td]};
        const pyset::Frames Gf{gfpos + gfoff[tg], gfbox + 4 * (int64_t)gfoff[tg],
                               gfoff[tg + 1] - gfoff[tg]};
        if (pyset::table_size(4ull * ((uint64_t)D.n + Gf.n)) > table_cap) {
            atomicOr(status, 1);            // scratch too small for this pair
            continue;
        }
        iou[p] = pyset::set_order_iou(tl_image_id + tl_vid_start[cell_unit[c]], D, Gf,
                                      mode, buf, table_cap);
    }
}

// Padded table -> the tasks' frame stream (taoamd_track_stream).  A workgroup
// per task walks the task's chunks exactly as tt_stager::issue does: thread
// (row, position) of a row whose span reaches into the chunk writes its slot
// of the row's piece -- the frame's box, or the far box outside first .. last
// (holes inside the span hold far boxes in the padded table already).
__global__ __launch_bounds__(TT_SLOTS * TT_P) void track_stream_kernel(
    const int4 *__restrict__ tasks, const int32_t *__restrict__ task_rows,
    const int4 *__restrict__ trk_meta, const int32_t *__restrict__ task_base,
    const double4 *__restrict__ padded, double4 *__restrict__ out)
{
    __shared__ int4 meta[TT_SLOTS];
    const int4 tk = tasks[blockIdx.x];
    const int tid = threadIdx.x;
    if (tid < TT_SLOTS)
        meta[tid] = tid < tk.y ? trk_meta[task_rows[tk.x + tid]] : make_int4(INT32_MAX, -1, 0, 0);
    if (blockIdx.x == 0 && tid < TT_P) out[tid] = make_double4(TT_FAR, TT_FAR, 0.0, 0.0);
    __syncthreads();
    int32_t p_lo = INT32_MAX, p_hi = -1;
    for (int r = 0; r < TT_SLOTS; r++) {
        const int4 m = meta[r];
        if (m.y >= m.x) {
            p_lo = min(p_lo, m.x);
            p_hi = max(p_hi, m.y);
        }
    }
    if (p_hi < 0) return;
    const int r = tid / TT_P, j = tid % TT_P;
    const int4 mine = meta[r];
    const bool has = mine.y >= mine.x;
    int64_t cursor = (int64_t)task_base[blockIdx.x] * TT_P;
    for (int32_t pc = p_lo & ~(TT_P - 1); pc <= p_hi; pc += TT_P) {
        uint32_t pm = 0;
        for (int q = 0; q < TT_SLOTS; q++) {
            const int4 m = meta[q];
            if (m.y >= m.x && m.x < pc + TT_P && m.y >= pc) pm |= 1u << q;
        }
        if (has && ((pm >> r) & 1u)) {
            const int32_t p = pc + j;
            const bool in = p >= mine.x && p <= mine.y;
            out[cursor + (__popc(pm & ((1u << r) - 1u)) * TT_P + j)] =
                in ? padded[(int64_t)mine.z + p] : make_double4(TT_FAR, TT_FAR, 0.0, 0.0);
        }
        cursor += __popc(pm) * TT_P;
    }
}

__global__ void track_pad_fill_kernel(int64_t n, double4 *__restrict__ padded)
{
    const int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (k < n) padded[k] = make_double4(TT_FAR, TT_FAR, 0.0, 0.0);
}

extern "C" int taoamd_track_iou(int64_t n_cells, const int32_t *cell_dt_off,
                                const int32_t *cell_gt_off,
                                const int64_t *cell_iou_off, int64_t n_pairs,
                                const int32_t *dt_frame_off,
                                const int32_t *dt_frame_pos,
                                const double *dt_frame_box,
                                const int32_t *gt_frame_off,
                                const int32_t *gt_frame_pos,
                                const double *gt_frame_box, int32_t mode,
                                double *iou, int64_t *pair_frames, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (mode < 0 || mode > 2) return TAOAMD_ERR_ARG;
    if (pair_frames) TAO_HIP(hipMemsetAsync(pair_frames, 0, 8, s));
    if (n_pairs == 0) return TAOAMD_OK;
    TAO_TIMED("track_iou_kernel", s, track_iou_kernel<<<(unsigned)((n_pairs + 255) / 256), 256, 0, s>>>(
        n_cells, cell_dt_off, cell_gt_off, cell_iou_off, n_pairs, dt_frame_off,
        dt_frame_pos, dt_frame_box, gt_frame_off, gt_frame_pos, gt_frame_box,
        iou, (unsigned long long *)pair_frames, mode));
    TAO_LAUNCH_CHECK();
    return TAOAMD_OK;
}

extern "C" int taoamd_track_iou_single(int64_t n_dt, const int32_t *dt_group,
                                       const int64_t *cell_iou_off,
                                       const int32_t *dt_frame_pos,
                                       const double *dt_frame_box,
                                       const int32_t *gt_frame_pos,
                                       const double *gt_frame_box, int32_t mode,
                                       double *iou, int64_t *pair_frames, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (mode < 0 || mode > 2 || n_dt < 0) return TAOAMD_ERR_ARG;
    if (pair_frames) TAO_HIP(hipMemsetAsync(pair_frames, 0, 8, s));
    if (n_dt == 0) return TAOAMD_OK;
    if (!dt_group || !cell_iou_off || !dt_frame_pos || !dt_frame_box || !gt_frame_pos ||
        !gt_frame_box || !iou)
        return TAOAMD_ERR_ARG;
    TAO_TIMED("track_iou_single_kernel", s,
              track_iou_single_kernel<<<(unsigned)std::min<int64_t>((n_dt + 255) / 256, 2048), 256, 0, s>>>(
                  n_dt, (const int4 *)dt_group, cell_iou_off, dt_frame_pos,
                  (const double4 *)dt_frame_box, gt_frame_pos, (const double4 *)gt_frame_box,
                  iou, (unsigned long long *)pair_frames, mode));
    TAO_LAUNCH_CHECK();
    return TAOAMD_OK;
}

extern "C" int taoamd_track_iou_planned(int64_t n_tasks, const int32_t *tasks,
                                        const int32_t *task_rows,
                                        const int32_t *task_pairs,
                                        const int64_t *task_out,
                                        const double *frames,
                                        const int32_t *task_base,
                                        const int32_t *trk_meta, int32_t mode,
                                        double *iou, int64_t *pair_frames,
                                        void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (mode < 0 || mode > 2 || n_tasks < 0) return TAOAMD_ERR_ARG;
    if (pair_frames) TAO_HIP(hipMemsetAsync(pair_frames, 0, 8, s));
    if (n_tasks == 0) return TAOAMD_OK;
    if (!tasks || !task_rows || !task_pairs || !task_out || !frames || !task_base ||
        !trk_meta || !iou)
        return TAOAMD_ERR_ARG;
    unsigned long long *pf = (unsigned long long *)pair_frames;
    // (TAOAMD_TT_LDS_PAD: unused dynamic LDS per task, i.e. fewer tasks per CU --
    // a schedule experiment: room for the other level's workgroups beside this kernel)
    static const unsigned lds_pad = getenv("TAOAMD_TT_LDS_PAD") ? (unsigned)atoi(getenv("TAOAMD_TT_LDS_PAD")) : 0u;
#define TT_LAUNCH(M)                                                           \
    TAO_TIMED("track_iou_task_kernel", s,                                      \
              track_iou_task_kernel<M><<<(unsigned)n_tasks, 192, lds_pad, s>>>( \
                  (const int4 *)tasks, task_rows, task_pairs, task_out,        \
                  (const double4 *)frames, task_base, (const int4 *)trk_meta, iou, pf))
    if (mode == 0 && pf == nullptr)
        TAO_TIMED("track_iou_task_kernel", s,
                  (track_iou_task_kernel<0, false><<<(unsigned)n_tasks, 192, lds_pad, s>>>(
                      (const int4 *)tasks, task_rows, task_pairs, task_out,
                      (const double4 *)frames, task_base, (const int4 *)trk_meta, iou, pf)));
    else if (mode == 0) TT_LAUNCH(0);
    else if (mode == 1) TT_LAUNCH(1);
    else TT_LAUNCH(2);
#undef TT_LAUNCH
    TAO_LAUNCH_CHECK();
    return TAOAMD_OK;
}

extern "C" int taoamd_track_pad(int64_t n_trk, int64_t n_frames,
                                const int32_t *frame_off,
                                const int32_t *frame_pos,
                                const double *frame_box, const int32_t *meta,
                                int64_t slot_first, int64_t n_slots,
                                double *padded, int32_t *inexact, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (n_trk < 0 || n_frames < 0 || n_slots < 0 || slot_first < 0)
        return TAOAMD_ERR_ARG;
    if (n_slots > 0) {
        if (!padded) return TAOAMD_ERR_ARG;
        TAO_TIMED("track_pad_fill_kernel", s, track_pad_fill_kernel<<<(unsigned)((n_slots + 255) / 256), 256, 0, s>>>(
            n_slots, (double4 *)padded + slot_first));
    }
    if (n_frames > 0) {
        if (!frame_off || !frame_pos || !frame_box || !meta || !padded)
            return TAOAMD_ERR_ARG;
        TAO_TIMED("track_pad_kernel", s, track_pad_kernel<<<(unsigned)((n_frames + 255) / 256), 256, 0, s>>>(
            n_frames, n_trk, frame_off, frame_pos, (const double4 *)frame_box,
            (const int4 *)meta, (double4 *)padded, inexact));
    }
    TAO_LAUNCH_CHECK();
    return TAOAMD_OK;
}

extern "C" int taoamd_track_stream(int64_t n_tasks, const int32_t *tasks,
                                   const int32_t *task_rows, const int32_t *trk_meta,
                                   const int32_t *task_base, const double *padded,
                                   double *frames, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (n_tasks < 0) return TAOAMD_ERR_ARG;
    if (n_tasks == 0) return TAOAMD_OK;
    if (!tasks || !task_rows || !trk_meta || !task_base || !padded || !frames)
        return TAOAMD_ERR_ARG;
    TAO_TIMED("track_stream_kernel", s,
              track_stream_kernel<<<(unsigned)n_tasks, TT_SLOTS * TT_P, 0, s>>>(
                  (const int4 *)tasks, task_rows, (const int4 *)trk_meta, task_base,
                  (const double4 *)padded, (double4 *)frames));
    TAO_LAUNCH_CHECK();
    return TAOAMD_OK;
}

extern "C" int taoamd_track_iou_near(int64_t n_cells, const int32_t *cell_gt_off,
                                     const int64_t *cell_iou_off, int64_t n_pairs,
                                     const double *iou, int32_t max_ulp,
                                     int32_t capacity, int32_t *count,
                                     int64_t *list, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (n_cells < 0 || n_pairs < 0 || max_ulp < 0 || capacity < 0 || !count)
        return TAOAMD_ERR_ARG;
    TAO_HIP(hipMemsetAsync(count, 0, sizeof(int32_t), s));
    if (n_pairs == 0) return TAOAMD_OK;
    if (!cell_gt_off || !cell_iou_off || !iou || (capacity > 0 && !list))
        return TAOAMD_ERR_ARG;
    TAO_TIMED("track_iou_near_kernel", s, track_iou_near_kernel<<<(unsigned)((n_pairs + 255) / 256), 256, 0, s>>>(
        n_cells, cell_gt_off, cell_iou_off, n_pairs, iou, iou_thr(), max_ulp, capacity,
        count, list));
    TAO_LAUNCH_CHECK();
    return TAOAMD_OK;
}

extern "C" int64_t taoamd_track_iou_setorder_table(int64_t max_dt_frames,
                                                  int64_t max_gt_frames)
{
    if (max_dt_frames < 0 || max_gt_frames < 0) return -1;
    return pyset::table_size(4ull * (uint64_t)(max_dt_frames + max_gt_frames));
}

extern "C" int taoamd_track_iou_setorder(
    int64_t n_cells, const int32_t *cell_dt_off, const int32_t *cell_gt_off,
    const int64_t *cell_iou_off, const int32_t *cell_unit, const int64_t *tl_vid_start,
    const int64_t *tl_image_id, const int32_t *dt_frame_off, const int32_t *dt_frame_pos,
    const double *dt_frame_box, const int32_t *gt_frame_off, const int32_t *gt_frame_pos,
    const double *gt_frame_box, int32_t mode, const int32_t *count, int32_t capacity,
    const int64_t *list, double *iou, int32_t *scratch, int64_t scratch_slots,
    int64_t table_cap, int32_t *status, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (mode < 0 || mode > 1 || n_cells < 0 || capacity < 0 || table_cap < 8 ||
        table_cap > (1ll << 30) || !count || !status)
        return TAOAMD_ERR_ARG;
    if (n_cells == 0 || capacity == 0) return TAOAMD_OK;
    if (!cell_dt_off || !cell_gt_off || !cell_iou_off || !cell_unit || !tl_vid_start ||
        !tl_image_id || !dt_frame_off || !dt_frame_pos || !dt_frame_box || !gt_frame_off ||
        !gt_frame_pos || !gt_frame_box || !list || !iou || !scratch)
        return TAOAMD_ERR_ARG;
    const int64_t workers = scratch_slots / (3 * table_cap);
    if (workers < 1) return TAOAMD_ERR_ARG;
    int64_t blocks = workers / 64;
    unsigned threads = 64;
    if (blocks == 0) { blocks = 1; threads = (unsigned)workers; }
    if (blocks > 4096) blocks = 4096;
    TAO_TIMED("track_iou_setorder_kernel", s,
              track_iou_setorder_kernel<<<(unsigned)blocks, threads, 0, s>>>(
                  n_cells, cell_dt_off, cell_gt_off, cell_iou_off, cell_unit, tl_vid_start,
                  tl_image_id, dt_frame_off, dt_frame_pos, dt_frame_box, gt_frame_off,
                  gt_frame_pos, gt_frame_box, mode, count, capacity, list, scratch,
                  (uint32_t)table_cap, iou, status));
    TAO_LAUNCH_CHECK();
    return TAOAMD_OK;
}

// ---- host statements of the same text, for the CPU tests (tests/test_pyset.py)
extern "C" int taoamd_set_order_iou_host(const int64_t *tl_image_id, int32_t n_dt,
                                         const int32_t *dt_pos, const double *dt_box,
                                         int32_t n_gt, const int32_t *gt_pos,
                                         const double *gt_box, int32_t mode, double *out)
{
    if (!tl_image_id || n_dt < 0 || n_gt < 0 || (n_dt && (!dt_pos || !dt_box)) ||
        (n_gt && (!gt_pos || !gt_box)) || mode < 0 || mode > 1 || !out || n_dt + n_gt == 0)
        return TAOAMD_ERR_ARG;
    const uint32_t cap = pyset::table_size(4ull * ((uint64_t)n_dt + n_gt));
    std::vector<int32_t> buf(3 * (size_t)cap);
    const pyset::Frames D{dt_pos, dt_box, n_dt}, G{gt_pos, gt_box, n_gt};
    *out = pyset::set_order_iou(tl_image_id, D, G, mode, buf.data(), cap);
    return TAOAMD_OK;
}

extern "C" int taoamd_pyset_union_order_host(int64_t n_a, const int64_t *a, int64_t n_b,
                                             const int64_t *b, int64_t *out, int64_t *n_out)
{
    if (n_a < 0 || n_b < 0 || (n_a && !a) || (n_b && !b) || !out || !n_out ||
        n_a + n_b > (1 << 27))
        return TAOAMD_ERR_ARG;
    // codes: 1 + index into the list of keys (a then b); equal keys share the
    // code of their first occurrence
    std::vector<int64_t> keys(a, a + n_a);
    keys.insert(keys.end(), b, b + n_b);
    std::vector<int32_t> code(keys.size());
    {
        std::vector<int64_t> idx(keys.size());
        for (size_t i = 0; i < idx.size(); i++) idx[i] = (int64_t)i;
        std::stable_sort(idx.begin(), idx.end(),
                         [&](int64_t x, int64_t y) { return keys[x] < keys[y]; });
        for (size_t i = 0; i < idx.size(); i++) {
            if (keys[idx[i]] < 0 || keys[idx[i]] >= (1ll << 61) - 1) return TAOAMD_ERR_ARG;
            code[idx[i]] = (i && keys[idx[i - 1]] == keys[idx[i]]) ? code[idx[i - 1]]
                                                                  : (int32_t)idx[i] + 1;
        }
    }
    const uint32_t cap = pyset::table_size(4ull * (uint64_t)(n_a + n_b));
    std::vector<int32_t> buf(3 * (size_t)cap);
    auto hash = [&](int32_t c) { return (uint64_t)keys[c - 1]; };
    auto ca = [&](uint32_t k) { return code[k]; };
    auto cb = [&](uint32_t k) { return code[n_a + k]; };
    const pyset::Set R = pyset::set_union((uint32_t)n_a, ca, (uint32_t)n_b, cb, hash,
                                          buf.data(), cap);
    int64_t n = 0;
    for (uint32_t i = 0; i <= R.mask; i++)
        if (R.t[i]) out[n++] = keys[R.t[i] - 1];
    *n_out = n;
    return TAOAMD_OK;
}

// ---------------------------------------------------------------- the plan
// Cells are visited by descending end of their timeline (tasks that run
// longest start first, and the cells that share a task have similar spans);
// the detection tracks of a cell by first position.  A task is filled track
// by track: a detection track brings one pair per GT track of its cell's GT
// block (<= 32 tracks), the block's rows are added when the task does not
// hold them yet; a task is closed when the next track would exceed 64 pairs or
// TT_ROWS rows -- so a big cell spills into the next task and the tail of one
// cell shares a wavefront with the head of the next.
extern "C" int taoamd_track_iou_plan_host(
    int64_t n_cells, const int32_t *cell_dt_off, const int32_t *cell_gt_off,
    const int64_t *cell_iou_off, const int32_t *trk_meta, int64_t *sizes,
    int32_t *tasks, int32_t *task_rows, int32_t *task_pairs, int64_t *task_out)
{
    if (n_cells < 0 || !cell_dt_off || !cell_gt_off || !cell_iou_off ||
        !trk_meta || !sizes)
        return TAOAMD_ERR_ARG;
    const bool fill = tasks != nullptr;
    if (fill && (!task_rows || !task_pairs || !task_out)) return TAOAMD_ERR_ARG;
    const int64_t n_dt = cell_dt_off[n_cells];
    auto first_of = [&](int64_t t) { return trk_meta[4 * t]; };
    auto last_of = [&](int64_t t) { return trk_meta[4 * t + 1]; };
    std::vector<int64_t> order;
    std::vector<int32_t> cell_hi(n_cells, -1);
    for (int64_t c = 0; c < n_cells; c++) {
        if (cell_dt_off[c + 1] <= cell_dt_off[c] || cell_gt_off[c + 1] <= cell_gt_off[c])
            continue;
        int32_t hi = -1;
        for (int64_t t = cell_dt_off[c]; t < cell_dt_off[c + 1]; t++)
            hi = std::max(hi, last_of(t));
        for (int64_t t = cell_gt_off[c]; t < cell_gt_off[c + 1]; t++)
            hi = std::max(hi, last_of(n_dt + t));
        cell_hi[c] = hi;
        order.push_back(c);
    }
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) {
        return cell_hi[a] > cell_hi[b];
    });
    int64_t nt = 0, nr = 0, np = 0;
    // the open task
    int32_t rows[TT_ROWS], n_rows = 0, n_pairs = 0;
    int32_t pair_d[64], pair_g[64];
    int64_t pair_out[64];
    auto close = [&]() {
        if (n_pairs == 0) { n_rows = 0; return; }
        if (fill) {
            // rows by first position (stable), pairs renumbered
            int32_t perm[TT_ROWS], inv[TT_ROWS];
            for (int r = 0; r < n_rows; r++) perm[r] = r;
            std::stable_sort(perm, perm + n_rows, [&](int32_t a, int32_t b) {
                return first_of(rows[a]) < first_of(rows[b]);
            });
            for (int r = 0; r < n_rows; r++) inv[perm[r]] = r;
            tasks[4 * nt] = (int32_t)nr;
            tasks[4 * nt + 1] = n_rows;
            tasks[4 * nt + 2] = (int32_t)np;
            tasks[4 * nt + 3] = n_pairs;
            for (int r = 0; r < n_rows; r++) task_rows[nr + r] = rows[perm[r]];
            for (int k = 0; k < n_pairs; k++) {
                task_pairs[np + k] = inv[pair_d[k]] | (inv[pair_g[k]] << 8);
                task_out[np + k] = pair_out[k];
            }
        }
        nt++;
        nr += n_rows;
        np += n_pairs;
        n_rows = n_pairs = 0;
    };
    std::vector<int32_t> dts;
    for (int64_t c : order) {
        const int32_t d0 = cell_dt_off[c], D = cell_dt_off[c + 1] - d0;
        const int32_t g0 = cell_gt_off[c], G = cell_gt_off[c + 1] - g0;
        dts.resize(D);
        for (int32_t d = 0; d < D; d++) dts[d] = d;
        std::stable_sort(dts.begin(), dts.end(), [&](int32_t a, int32_t b) {
            return first_of(d0 + a) < first_of(d0 + b);
        });
        const int32_t gblocks = (G + 31) / 32;
        for (int32_t gb = 0; gb < gblocks; gb++) {
            const int32_t ga = (int32_t)((int64_t)gb * G / gblocks);
            const int32_t ng = (int32_t)((int64_t)(gb + 1) * G / gblocks) - ga;
            int32_t grow0 = -1;          // the block's first row in the open task
            for (int32_t k = 0; k < D; k++) {
                const int32_t d = dts[k];
                if (n_pairs + ng > 64 || n_rows + (grow0 < 0 ? ng : 0) + 1 > TT_ROWS) {
                    close();
                    grow0 = -1;
                }
                if (grow0 < 0) {
                    grow0 = n_rows;
                    for (int32_t g = 0; g < ng; g++)
                        rows[n_rows++] = (int32_t)(n_dt + g0 + ga + g);
                }
                rows[n_rows] = d0 + d;
                for (int32_t g = 0; g < ng; g++) {
                    pair_d[n_pairs] = n_rows;
                    pair_g[n_pairs] = grow0 + g;
                    pair_out[n_pairs] = cell_iou_off[c] + (int64_t)d * G + ga + g;
                    n_pairs++;
                }
                n_rows++;
            }
        }
    }
    close();
    if (nr >= INT32_MAX || np >= INT32_MAX) return TAOAMD_ERR_ARG;
    sizes[0] = nt;
    sizes[1] = nr;
    sizes[2] = np;
    return TAOAMD_OK;
}
