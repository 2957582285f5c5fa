// IoU and greedy-assignment kernels (gfx950).
//
//   bb_iou_kernel       one thread per (d, g) entry of a single bbIou matrix
//   *_ranges_kernel     per-GT / per-detection range masks + num_gt histogram
//   (the 3D track IoU lives in track_iou.hip)
//   match_kernel        one wavefront per (cell, 64-combo word): lane = one
//                       (range, IoU threshold) combo running the sequential
//                       greedy of the reference; the IoU tile of the cell
//                       lives in one VGPR pair spread across the wave and is
//                       broadcast with v_readlane (no LDS, no HBM round trip
//                       for the LVIS matrix); the per-detection result of the
//                       64 combos is packed with one ballot
//   match_big_kernel    cells with more than 64 ground truths: IoU row and
//                       per-lane "taken" bitsets staged in LDS
//
// HBM-bound integer/compare work: nothing here is shaped into a GEMM.
#include <cstdlib>

#include "common.hpp"

using namespace taoamd;

// The IoU thresholds as the match kernels compare with them: min(thr, 1 - 1e-10)
// (reference lvis_amodal/eval.py:250 / tao_amodal/eval.py:401), clamped once on
// the host instead of by every wavefront.
static IouThr match_thr()
{
    IouThr t = iou_thr();
    for (int q = 0; q < N_THR; q++) t.v[q] = t.v[q] < 1 - 1e-10 ? t.v[q] : 1 - 1e-10;
    return t;
}


// --------------------------------------------------------------------- bbIou
__global__ void bb_iou_kernel(const double *__restrict__ dt,
                              const double *__restrict__ gt, size_t m,
                              size_t n, const unsigned char *__restrict__ crowd,
                              double *__restrict__ o)
{
    size_t total = m * n;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        size_t g = i / m, d = i - g * m;
        const double4 D = reinterpret_cast<const double4 *>(dt)[d];
        const double4 G = reinterpret_cast<const double4 *>(gt)[g];
        double v;
        if (crowd != nullptr && crowd[g]) {
            double da = D.z * D.w;
            double w = raw_fmin(D.z + D.x, G.z + G.x) - raw_fmax(D.x, G.x);
            double h = raw_fmin(D.w + D.y, G.w + G.y) - raw_fmax(D.y, G.y);
            v = (w <= 0 || h <= 0) ? 0.0 : (w * h) / da;
        } else {
            v = box_iou(D.x, D.y, D.z, D.w, G.x, G.y, G.z, G.w);
        }
        o[i] = v;
    }
}

// -------------------------------------------------------------------- ranges
__global__ void lvis_ranges_kernel(int64_t n_gt, const double *__restrict__ vis,
                                   const uint8_t *__restrict__ gflags,
                                   const int32_t *__restrict__ gcat,
                                   int64_t n_dt,
                                   const uint8_t *__restrict__ dflags,
                                   uint32_t *__restrict__ gt_rng,
                                   uint32_t *__restrict__ dt_rng,
                                   int32_t *__restrict__ num_gt, RangeTab tab)
{
    // visibility ranges of reference lvis_amodal/eval.py:567-574 (params.
    // visibility_rng, read at run time: eval.py:205)
    const double *lo = tab.vis_lo, *hi = tab.vis_hi;
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n_gt) {
        uint32_t m = 0;
        bool ign = gflags[i] & TAOAMD_GT_IGNORE;
        double v = vis[i];
#pragma unroll
        for (int r = 0; r < 5; r++)
            if (ign || v < lo[r] || v > hi[r]) m |= 1u << r;
        if (ign || !(gflags[i] & TAOAMD_GT_OOF)) m |= 1u << 5;
        gt_rng[i] = m;
        if (num_gt != nullptr) {
            int32_t *row = num_gt + (int64_t)gcat[i] * TAOAMD_LVIS_RNG;
#pragma unroll
            for (int r = 0; r < TAOAMD_LVIS_RNG; r++)
                if (!((m >> r) & 1u)) atomicAdd(row + r, 1);
        }
    }
    if (i < n_dt)
        dt_rng[i] = (dflags[i] & TAOAMD_DT_IGNORE_UNMATCHED) ? 0x3fu : 0u;
}

__global__ void tao_ranges_kernel(
    int64_t n_gt, const double *__restrict__ garea,
    const int32_t *__restrict__ glen, const int32_t *__restrict__ gnhp,
    const uint8_t *__restrict__ gflags, const int32_t *__restrict__ gcat,
    int64_t n_dt, const double *__restrict__ darea,
    const int32_t *__restrict__ dlen, const uint8_t *__restrict__ dflags,
    uint32_t *__restrict__ gt_rng, uint32_t *__restrict__ dt_rng,
    int32_t *__restrict__ num_gt, RangeTab tab)
{
    // area / duration ranges of reference tao_amodal/eval.py:735-744 (params.
    // area_rng / time_rng, read at run time: eval.py:272-275)
    const double *alo = tab.area_lo, *ahi = tab.area_hi;
    const double *tlo = tab.time_lo, *thi = tab.time_hi;
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n_gt) {
        uint32_t m = 0;
        bool ign = gflags[i] & TAOAMD_GT_IGNORE;
        double a_ = garea[i], len = (double)glen[i];
        bool few_hp = gnhp[i] <= 5;
#pragma unroll
        for (int a = 0; a < 5; a++)
#pragma unroll
            for (int t = 0; t < 4; t++) {
                bool bad = ign || a_ < alo[a] || a_ > ahi[a] || len < tlo[t] ||
                           len > thi[t] || (a == 4 && few_hp);
                if (bad) m |= 1u << (a * 4 + t);
            }
        gt_rng[i] = m;
        if (num_gt != nullptr) {
            int32_t *row = num_gt + (int64_t)gcat[i] * TAOAMD_TAO_RNG;
            for (int r = 0; r < TAOAMD_TAO_RNG; r++)
                if (!((m >> r) & 1u)) atomicAdd(row + r, 1);
        }
    }
    if (i < n_dt) {
        uint32_t m = 0;
        bool nel = dflags[i] & TAOAMD_DT_IGNORE_UNMATCHED;
        double a_ = darea[i], len = (double)dlen[i];
#pragma unroll
        for (int a = 0; a < 5; a++)
#pragma unroll
            for (int t = 0; t < 4; t++)
                if (nel || a_ < alo[a] || a_ > ahi[a] || len < tlo[t] ||
                    len > thi[t])
                    m |= 1u << (a * 4 + t);
        dt_rng[i] = m;
    }
}

// num_gt without global atomics: GTs grouped by category, one workgroup per
// category; the four wavefronts stride over its GTs 64 at a time (four loads
// in flight each), lane r < n_rng ends up owning the count of range r
__global__ __launch_bounds__(256) void count_gt_kernel(
    int32_t n_cat, int32_t n_rng, const int32_t *__restrict__ gt_cat_off,
    const uint32_t *__restrict__ gt_rng, int32_t *__restrict__ num_gt)
{
    __shared__ int32_t part[4][32];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int32_t k = blockIdx.x;
    const int lane = lane_id();
    const int32_t b = gt_cat_off[k], e = gt_cat_off[k + 1];
    int32_t mine = 0;
    for (int32_t base = b + wave * WAVE; base < e; base += 4 * 4 * WAVE) {
        uint32_t m[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int32_t i = base + u * 4 * WAVE + lane;
            m[u] = i < e ? gt_rng[i] : 0xffffffffu;
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            for (int r = 0; r < n_rng; r++) {
                const int c = __popcll(__ballot(!((m[u] >> r) & 1u)));
                if (lane == r) mine += c;
            }
    }
    if (lane < 32) part[wave][lane] = mine;
    __syncthreads();
    if (threadIdx.x < (unsigned)n_rng)
        num_gt[(int64_t)k * n_rng + threadIdx.x] =
            part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] +
            part[3][threadIdx.x];
}

// ------------------------------------------------------------------- greedy
struct MatchArgs {
    int64_t n_cells;
    const int32_t *cell_dt_off;
    const int32_t *cell_gt_off;
    const int64_t *cell_iou_off;
    const double *dt_box;   // fused LVIS IoU when non-null
    const double *gt_box;
    const double *iou;      // precomputed (TAO) otherwise
    const uint32_t *gt_rng;
    const uint32_t *dt_rng;
    const uint8_t *gt_flags;
    const uint8_t *dt_flags;
    const int32_t *dst;
    uint64_t *matched;
    uint64_t *ignored;
    int32_t *match_gt;
    double *ious_out;
    int32_t n_rng;
    int32_t n_words;
    int64_t out_stride;  // words between consecutive output rows
    int32_t paired;      // ignored == matched + 1: word w of a row is the pair at 2 * w
    int32_t wide;        // ... and every pair is 16-byte aligned: one store
    // optional launch plan (built by the host from the cell table)
    const int32_t *dt_group;  // [n_dt][4] first GT / GT count of the cell, position in it, cell
    const uint32_t *dt_meta;  // optional [n_dt]: flags | first GT - run's << 8 | GTs << 14 | position << 18
    const int32_t *groups;    // [n_groups][4] first detection, count, first GT, count of a run
    const int32_t *singles;   // [n_singles] cells handled one per wavefront
    int32_t n_groups, n_singles;
    int32_t xcd;              // blocks renumbered per XCD (xcd_block)
};

// range mask of detection d: the caller's table, or -- image level, where the
// mask is "every range iff the detection is ignored when unmatched"
// (lvis_ranges_kernel) -- straight from its flags: 120 MB less written and
// 120 MB less read per pass at 30 M detections
// one row's word: two 8-byte stores into the two tables, or -- when the tables
// are the two halves of (matched, ignored) pairs -- one 16-byte store (the rows
// are scattered to their sorted places: every store is its own cache line)
__device__ __forceinline__ void store_row(const MatchArgs &a, int64_t row, int word,
                                          uint64_t m, uint64_t i)
{
    if (a.wide) {
        ulonglong2 v;
        v.x = m;
        v.y = i;
        *reinterpret_cast<ulonglong2 *>(a.matched + row * a.out_stride + 2 * word) = v;
    } else if (a.paired) {
        // (a 24-byte exchange record: pairs at 8 mod 16)
        a.matched[row * a.out_stride + 2 * word] = m;
        a.matched[row * a.out_stride + 2 * word + 1] = i;
    } else {
        a.matched[row * a.out_stride + word] = m;
        a.ignored[row * a.out_stride + word] = i;
    }
}

__device__ __forceinline__ uint32_t dt_rng_of(const MatchArgs &a, int64_t d, uint32_t flags)
{
    if (a.dt_rng != nullptr) return a.dt_rng[d];
    return (flags & TAOAMD_DT_IGNORE_UNMATCHED) ? 0xffffffffu >> (32 - a.n_rng) : 0u;
}

#define GRP_GCAP 8   // most GTs of one cell inside a multi-cell group

// Fast path: G <= 64.  One wavefront per (cell, word).
template <bool FUSED>
__global__ __launch_bounds__(256) void match_kernel(MatchArgs a, IouThr thr)
{
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t item = (int64_t)blockIdx.x * 4 + wave;
    const int64_t n_items = a.singles != nullptr ? a.n_singles : a.n_cells;
    if (item >= n_items * a.n_words) return;
    const int64_t ci = item / a.n_words;
    const int word = (int)(item - ci * a.n_words);
    const int64_t cell = a.singles != nullptr ? a.singles[ci] : ci;
    const int32_t d0 = __builtin_amdgcn_readfirstlane(a.cell_dt_off[cell]);
    const int32_t D = __builtin_amdgcn_readfirstlane(a.cell_dt_off[cell + 1]) - d0;
    const int32_t g0 = __builtin_amdgcn_readfirstlane(a.cell_gt_off[cell]);
    const int32_t G = __builtin_amdgcn_readfirstlane(a.cell_gt_off[cell + 1]) - g0;
    if (D == 0 || G > WAVE) return;
    const int n_combo = a.n_rng * N_THR;
    const int combo = word * WAVE + lane;
    const bool active = combo < n_combo;
    const int r = active ? combo / N_THR : 0;
    const int t = active ? combo - r * N_THR : 0;
    const double thr0 = thr.v[t];
    const int64_t ioff = (FUSED && a.ious_out == nullptr && a.iou == nullptr)
                             ? 0 : a.cell_iou_off[cell];

    // per-lane set of GTs ignored in this lane's range; wave-uniform set of
    // GTs whose id equals the "unmatched" sentinel
    uint64_t IG = 0, HID = 0;
    for (int g = 0; g < G; g++) {
        uint32_t m = a.gt_rng[g0 + g];
        IG |= (uint64_t)((m >> r) & 1u) << g;
        HID |= (uint64_t)((a.gt_flags[g0 + g] & TAOAMD_GT_ID_HIDDEN) ? 1 : 0) << g;
    }
    // this lane's GT box when it holds column (lane % G) of the IoU tile
    const int TD = G > 0 ? WAVE / G : WAVE;  // detections per register tile
    double gx = 0, gy = 0, gw = 0, gh = 0;
    const int my_dd = G > 0 ? lane / G : 0;
    const int my_g = G > 0 ? lane - my_dd * G : 0;
    if (FUSED && G > 0 && my_dd < TD) {
        const double4 B = reinterpret_cast<const double4 *>(a.gt_box)[g0 + my_g];
        gx = B.x; gy = B.y; gw = B.z; gh = B.w;
    }

    uint64_t taken = 0;
    for (int32_t base = 0; base < D; base += TD) {
        const int nd = min(TD, D - base);
        // ---- IoU tile: lane l holds entry (dd = l / G, g = l % G)
        double v_tile = 0.0;
        if (G > 0 && my_dd < nd) {
            if (FUSED) {
                const double4 B =
                    reinterpret_cast<const double4 *>(a.dt_box)[d0 + base + my_dd];
                v_tile = box_iou(B.x, B.y, B.z, B.w, gx, gy, gw, gh);
                if (a.ious_out != nullptr && word == 0)
                    a.ious_out[ioff + (int64_t)(base + my_dd) * G + my_g] = v_tile;
            } else {
                v_tile = a.iou[ioff + (int64_t)(base + my_dd) * G + my_g];
            }
        }
        // per-detection inputs of the tile: one coalesced load, then
        // v_readlane inside the sequential loop (no memory in the loop)
        int32_t t_flags = 0, t_rng = 0;
        int64_t t_row = 0;
        if (lane < nd) {
            const int32_t d = d0 + base + lane;
            t_flags = a.dt_flags[d];
            t_rng = (int32_t)dt_rng_of(a, d, (uint32_t)t_flags);
            t_row = a.dst != nullptr ? a.dst[d] : d;
        }
        uint64_t my_m = 0, my_i = 0;
        for (int dd = 0; dd < nd; dd++) {
            const int32_t d = d0 + base + dd;
            double best1 = thr0, best2 = thr0;
            int m1 = -1, m2 = -1;
            const uint64_t free1 = ~taken & ~IG, free2 = ~taken & IG;
            for (int g = 0; g < G; g++) {
                const double v = readlane_f64(v_tile, dd * G + g);
                const bool ok1 = ((free1 >> g) & 1) && !(v < best1);
                const bool ok2 = ((free2 >> g) & 1) && !(v < best2);
                best1 = ok1 ? v : best1;  m1 = ok1 ? g : m1;
                best2 = ok2 ? v : best2;  m2 = ok2 ? g : m2;
            }
            const int m = m1 >= 0 ? m1 : m2;
            const uint32_t df = (uint32_t)__builtin_amdgcn_readlane(t_flags, dd);
            const uint32_t drng = (uint32_t)__builtin_amdgcn_readlane(t_rng, dd);
            if (m >= 0 && !(df & TAOAMD_DT_NO_CONSUME)) taken |= 1ull << m;
            const bool vis = m >= 0 && !((HID >> m) & 1);
            bool ig = m >= 0 && ((IG >> m) & 1);
            if (!vis && ((drng >> r) & 1u)) ig = true;
            const uint64_t mw = __ballot(active && vis);
            const uint64_t iw = __ballot(active && ig);
            if (lane == dd) { my_m = mw; my_i = iw; }
            if (a.match_gt != nullptr && active)
                a.match_gt[(int64_t)d * n_combo + combo] = m;
        }
        if (lane < nd) {
            store_row(a, t_row, word, my_m, my_i);
        }
    }
}

// Multi-cell groups.  Most cells are tiny (a handful of detections, one or two
// GTs): one wavefront per cell spends its life waiting on three dependent,
// nearly empty memory round trips.  Here a wavefront takes a run of
// consecutive cells with at most 64 detections and 64 GTs in total (each cell
// at most GRP_GCAP GTs): detections and GTs of the run are contiguous, so
// lane = detection / lane = GT loads are coalesced and issued once per run.
//   * lane = detection: IoU against the (<= GRP_GCAP) GTs of its own cell,
//     GT boxes broadcast from LDS -> IoU values to LDS
//   * lane = combo: the sequential greedy over the run's detections; the
//     "ignored" / "taken" sets are 64-bit masks over the run's GTs, so moving
//     from one cell to the next needs no reset at all.
// FAST (round 4): the image level's production layout -- boxes fused, dt_meta,
// dst, range masks from the flags, no IoU output -- known at compile time, so
// that every load that depends on the run descriptor alone is issued in ONE
// batch: with the layout decided by run-time (uniform) branches the compiler
// put an s_waitcnt vmcnt(0) behind each of them, four HBM round trips in a
// row where two are needed (descriptor, then everything).
template <bool FUSED, bool FAST = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void match_group_kernel(MatchArgs a, IouThr thr)
{
    __shared__ double4 s_gt[4][WAVE];
    // IoU tile of the run: only where the IoUs come from memory.  The fused
    // kernel keeps a detection's candidate in registers and, in the rare cell
    // that needs the sequential loop, computes the IoU again from the boxes --
    // without the 4 KB tile per wavefront it is the registers (7 wavefronts per
    // SIMD) and no longer the LDS (6) that bound its occupancy.
    __shared__ double s_iou[4][FUSED ? 1 : WAVE * GRP_GCAP];
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // runs are category-major and a run's rows are scattered into its
    // category's segment: consecutive runs behind one L2 (common.hpp)
    const uint32_t blk = a.xcd ? xcd_block(blockIdx.x, gridDim.x) : blockIdx.x;
    const int64_t item = (int64_t)blk * 4 + wave;
    if (item >= (int64_t)a.n_groups * a.n_words) return;
    // (n_words is 1 or 4: no 64-bit division in the prologue)
    const int64_t grp = a.n_words == 1 ? item : a.n_words == 4 ? item >> 2 : item / a.n_words;
    const int word = (int)(item - grp * a.n_words);
    // Everything a wavefront needs is at most two dependent loads away: the
    // run descriptor, then the per-detection / per-GT rows (the host resolved
    // cell -> GT range per detection, so no chain through the cell table).
    const int4 run = reinterpret_cast<const int4 *>(a.groups)[grp];
    const int32_t d0 = __builtin_amdgcn_readfirstlane(run.x);
    const int32_t nD = __builtin_amdgcn_readfirstlane(run.y);
    const int32_t g0 = __builtin_amdgcn_readfirstlane(run.z);
    const int32_t nG = __builtin_amdgcn_readfirstlane(run.w);
    if (nD == 0) return;
    const int n_combo = a.n_rng * N_THR;
    // (lane = combo -- its range, its threshold, the ignore mask of its range's
    // GTs -- is the view of the SEQUENTIAL loop at the end, which few runs
    // reach: worked out there, round 6; 90 of the kernel's 427 VALU
    // instructions per wavefront were spent on it in every run)

    // ---- lane = detection of the run
    int32_t t_flags = 0, t_rng = 0, gb = 0, dloc = 0, Gc = 0;
    int64_t t_row = 0, t_ioff = 0;
    double4 B = make_double4(0, 0, 0, 0);
    uint32_t grng = 0xffffffffu;
    bool ghid = false;
    if (FAST) {
        // lanes past the run's detections / GTs re-read its last one
        const int32_t d = d0 + min(lane, nD - 1);
        const int32_t g = g0 + min(lane, max(nG, 1) - 1);
        const uint32_t mt = a.dt_meta[d];
#ifdef TAOAMD_ABLATE_SCATTER     // (timing experiment: rows stored in detection order)
        const int32_t row = d + (a.dst[d] & 0);
#else
        const int32_t row = a.dst[d];
#endif
        const double4 box = reinterpret_cast<const double4 *>(a.dt_box)[d];
        uint32_t gr = 0xffffffffu;
        uint8_t gf = 0;
        double4 gbx = make_double4(0, 0, 0, 0);
        if (nG > 0) {
            gr = a.gt_rng[g];
            gf = a.gt_flags[g];
            gbx = reinterpret_cast<const double4 *>(a.gt_box)[g];
        }
        if (lane < nD) {
            t_flags = (int32_t)(mt & 0xffu);
            gb = (int32_t)((mt >> 8) & 63u);
            Gc = (int32_t)((mt >> 14) & 15u);
            dloc = (int32_t)(mt >> 18);
            t_rng = (int32_t)((t_flags & TAOAMD_DT_IGNORE_UNMATCHED)
                                  ? 0xffffffffu >> (32 - a.n_rng) : 0u);
            t_row = row;
            B = box;
        }
        if (lane < nG) {
            grng = gr;
            ghid = gf & TAOAMD_GT_ID_HIDDEN;
            s_gt[wave][lane] = gbx;
        }
    } else if (lane < nD) {
        const int32_t d = d0 + lane;
        if (FUSED && a.dt_meta != nullptr) {
            // one word instead of the 16-byte row and the flag byte (the cell
            // index is only needed to find IoUs in memory)
            static_assert(GRP_GCAP <= 15 && WAVE <= 64, "fields of dt_meta");
            const uint32_t mt = a.dt_meta[d];
            t_flags = (int32_t)(mt & 0xffu);
            gb = (int32_t)((mt >> 8) & 63u);
            Gc = (int32_t)((mt >> 14) & 15u);
            dloc = (int32_t)(mt >> 18);
        } else {
            const int4 dg = reinterpret_cast<const int4 *>(a.dt_group)[d];
            t_flags = a.dt_flags[d];
            Gc = dg.y;
            gb = dg.x - g0;
            dloc = dg.z;
            if (a.cell_iou_off != nullptr) t_ioff = a.cell_iou_off[dg.w];
        }
        t_rng = (int32_t)dt_rng_of(a, d, (uint32_t)t_flags);
        t_row = a.dst != nullptr ? a.dst[d] : d;
        if (FUSED) B = reinterpret_cast<const double4 *>(a.dt_box)[d];
    }
    // ---- lane = GT of the run
    if (!FAST && lane < nG) {
        grng = a.gt_rng[g0 + lane];
        ghid = a.gt_flags[g0 + lane] & TAOAMD_GT_ID_HIDDEN;
        if (FUSED) s_gt[wave][lane] = reinterpret_cast<const double4 *>(a.gt_box)[g0 + lane];
    }
    const uint64_t HID = __ballot(ghid);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // ---- IoU of every detection against the GTs of its own cell, and its
    // CANDIDATES on the way (see below)
    const double tmin = thr.v[0];
    int cand = -1, ncand = 0;
    double vc = 0.0;
    // (IoUs from memory: a detection's row of up to GRP_GCAP values in one batch
    // of loads -- inside the loop below every load was waited for on its own)
    double vmem[FUSED ? 1 : GRP_GCAP];
    if (!FUSED) {
        const int64_t base = t_ioff + (int64_t)dloc * Gc;
#pragma unroll
        for (int k = 0; k < GRP_GCAP; k++)
            vmem[k] = a.iou[lane < nD && k < Gc ? base + k : 0];
    }
#pragma unroll
    for (int k = 0; k < GRP_GCAP; k++) {
        const bool has = lane < nD && k < Gc;
        if (__ballot(has) == 0) break;
        if (has) {
            double v;
            if (FUSED) {
                const double4 A = s_gt[wave][gb + k];
                v = box_iou(B.x, B.y, B.z, B.w, A.x, A.y, A.z, A.w);
                if (!FAST && a.ious_out != nullptr && word == 0)
                    a.ious_out[t_ioff + (int64_t)dloc * Gc + k] = v;
            } else {
                v = vmem[k];
                s_iou[wave][lane * GRP_GCAP + k] = v;
            }
            if (!(v < tmin)) { ncand++; cand = k; vc = v; }
        }
    }
    if (!FUSED) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    // ---- closed form, lane = detection, for every cell in which no detection
    // has more than one CANDIDATE -- a GT whose IoU reaches the lowest
    // threshold (a GT below it can never be chosen).  With at most one
    // candidate per detection the greedy has no choice to make: a detection
    // takes its candidate iff the IoU reaches the threshold and no earlier
    // *consuming* detection of the cell with the same candidate did (whether
    // the GT is ignored only changes the `ignored` bit, never the
    // assignment); without a candidate nothing matches.  Cells with zero or
    // one GT are the special case "candidate = GT 0".  One ballot per
    // (threshold, candidate index) answers that for all detections of the
    // run at once; only cells where some detection overlaps two GTs by the
    // lowest threshold or more go through the sequential loop below.
    uint64_t my_m = 0, my_i = 0;
    // lanes [ca, ce) hold the detections of my cell
    const uint64_t starts = __ballot(lane < nD && dloc == 0);
    const int ca = lane - dloc;
    const uint64_t above = lane >= 63 ? 0ull : starts & ~((2ull << lane) - 1);
    const int ce = above ? __builtin_ctzll(above) : nD;
    const uint64_t below_ca = (1ull << ca) - 1;
    const uint64_t cell_all = (ce >= 64 ? ~0ull : ((1ull << ce) - 1)) & ~below_ca;
    const uint64_t multi = __ballot(lane < nD && ncand >= 2);
    const bool simple = lane < nD && (multi & cell_all) == 0;
    {
        uint32_t mygrng = 0;
        bool myghid = false;
        // the candidate's range mask / hidden flag sit in the GT lane gb + cand
        const uint32_t grng_c = (uint32_t)__shfl((int)grng, (gb + max(cand, 0)) & 63);
        if (simple && cand >= 0) {
            mygrng = grng_c;
            myghid = (HID >> (gb + cand)) & 1ull;
        }
        const bool consumes = !(t_flags & TAOAMD_DT_NO_CONSUME);
        // The thresholds ascend, so a detection passes the first `qpass` of
        // them, and its candidate is taken at threshold q by an EARLIER
        // consuming detection of the cell with the same candidate iff one of
        // those passes q, i.e. iff q < M, the largest qpass among them: the
        // detection matches at the thresholds M <= q < qpass.  M is a
        // segmented exclusive prefix maximum over the cell's lanes, one scan
        // per candidate index in use (before: two ballots per threshold and
        // candidate index -- 45 % of the kernel's time in a wave-lifetime trace).
        int qpass = 0;
        if (simple && cand >= 0) {
#pragma unroll
            for (int q = 0; q < N_THR; q++) qpass += !(vc < thr.v[q]) ? 1 : 0;
        }
        const int val = consumes ? qpass : 0;
        int M = 0;
        // The scan in DPP moves (round 4): a cell's number in the bits above the
        // value makes the segmented maximum an ordinary one -- a later cell's
        // entries outrank every earlier cell's, so the running maximum at a lane
        // is that of its own cell's lanes up to it -- and an ordinary inclusive
        // maximum over the wavefront is four row shifts and two row broadcasts
        // on the VALU (v_max_u32 with a DPP operand) instead of six ds_bpermute
        // round trips through the LDS crossbar with a compare and a select each.
        const uint32_t cell_no = (uint32_t)__popcll(starts & (lane >= 63 ? ~0ull : ((2ull << lane) - 1)));
        for (int c = 0; c < GRP_GCAP; c++) {
            if (__ballot(simple && cand >= c) == 0) break;          // wave-uniform
            static_assert(N_THR < 16, "value bits of the keyed scan");
            uint32_t x = (lane < nD ? cell_no << 4 : 0u) | (uint32_t)(cand == c ? val : 0);
#define TAOAMD_SCAN_STEP(CTRL, ROWS)                                                         \
            x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROWS, 0xF, false))
            TAOAMD_SCAN_STEP(0x111, 0xF);        // row_shr:1 (a lane without a source keeps 0)
            TAOAMD_SCAN_STEP(0x112, 0xF);        // row_shr:2
            TAOAMD_SCAN_STEP(0x114, 0xF);        // row_shr:4
            TAOAMD_SCAN_STEP(0x118, 0xF);        // row_shr:8
            TAOAMD_SCAN_STEP(0x142, 0xA);        // row_bcast:15 into rows 1 and 3
            TAOAMD_SCAN_STEP(0x143, 0xC);        // row_bcast:31 into rows 2 and 3
#undef TAOAMD_SCAN_STEP
            // the lane before me (wave_shr:1; lane 0 has none): in my cell iff it
            // carries my cell's number
            const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x138, 0xF, 0xF, true);
            if (cand == c) M = (prev >> 4) == cell_no ? (int)(prev & 15u) : 0;
        }
        const uint32_t m10 = qpass > M ? ((1u << qpass) - 1) & ~((1u << M) - 1) : 0u;
        if (FAST) {
            // One combo word, at most six ranges (the image level): range q's ten
            // bits are (igr_q ? m10 : 0) | (dig_q ? X : 0) at bit 10 q, i.e.
            //     ignored = m10 * S(mygrng) | X * S(t_rng),  matched = m10 * S(all)
            // with S(g) = sum of 2^(10 q) over the set bits q of g -- products of
            // a ten-bit value with bits ten apart: no two partial products
            // overlap, nothing carries.  S(g) itself is (g * C) & S(all) for
            // C = sum of 2^(9 i): bit j of g lands at 9 i + j, which is a
            // multiple of ten only for i = j.  A dozen VALU instructions where
            // the loop over the ranges below, with its run-time 64-bit shifts,
            // took 144 (round 6).
            static_assert(N_THR == 10, "bit layout of the fast word");
            constexpr uint64_t S_ALL = 1ull | 1ull << 10 | 1ull << 20 | 1ull << 30 |
                                       1ull << 40 | 1ull << 50;
            constexpr uint64_t C9 = 1ull | 1ull << 9 | 1ull << 18 | 1ull << 27 |
                                    1ull << 36 | 1ull << 45;
            const uint32_t all10 = (1u << N_THR) - 1;
            const uint32_t rmask = 0xffffffffu >> (32 - a.n_rng);       // (uniform)
            const uint64_t sall = ((uint64_t)rmask * C9) & S_ALL;
            const uint64_t sg = ((uint64_t)(mygrng & rmask) * C9) & S_ALL;
            const uint64_t sd = ((uint64_t)((uint32_t)t_rng & rmask) * C9) & S_ALL;
            const uint32_t X = myghid ? all10 : (~m10 & all10);
            if (simple) {
                my_m = myghid ? 0ull : (uint64_t)m10 * sall;
                my_i = (uint64_t)m10 * sg | (uint64_t)X * sd;
            }
        } else if (simple) {
            const uint32_t all10 = (1u << N_THR) - 1;
            const int r_lo = (word * WAVE) / N_THR;
            const int r_hi = min(a.n_rng - 1, (word * WAVE + WAVE - 1) / N_THR);
            for (int q = r_lo; q <= r_hi; q++) {
                const bool igr = (mygrng >> q) & 1u, dig = ((uint32_t)t_rng >> q) & 1u;
                const uint64_t mb = myghid ? 0u : m10;
                const uint64_t ib = (igr ? m10 : 0u) |
                                    (dig ? (myghid ? all10 : (~m10 & all10)) : 0u);
                const int o = q * N_THR - word * WAVE;
                my_m |= o >= 0 ? mb << o : mb >> (-o);
                my_i |= o >= 0 ? ib << o : ib >> (-o);
            }
            if (a.match_gt != nullptr)
                for (int cc = 0; cc < WAVE && word * WAVE + cc < n_combo; cc++)
                    a.match_gt[(int64_t)(d0 + lane) * n_combo + word * WAVE + cc] =
                        ((m10 >> ((word * WAVE + cc) % N_THR)) & 1u) ? cand : -1;
        }
    }
    // ---- lane = combo: sequential greedy over the detections of the remaining
    // cells.  Per detection one readlane brings a packed word
    // (first GT of the cell | GT count | flags) to the scalar unit; the masks
    // are narrowed to the cell's <= GRP_GCAP GTs so the inner loop is 32-bit.
    const uint64_t todo = __ballot(lane < nD && !simple);
    if (todo == 0) {                  // (wave-uniform: the common case)
        if (lane < nD) store_row(a, t_row, word, my_m, my_i);
        return;
    }
    const int combo = word * WAVE + lane;
    const bool active = combo < n_combo;
    const int r = active ? combo / N_THR : 0;
    const int t = active ? combo - r * N_THR : 0;
    const double thr0 = thr.v[t];
    uint64_t IG = 0;
    for (int q = 0; q < a.n_rng; q++) {
        const uint64_t m = __ballot(lane < nG && ((grng >> q) & 1u));
        IG = (q == r) ? m : IG;
    }
    const int32_t meta = (gb & 0xff) | ((Gc & 0xff) << 8) | ((t_flags & 0xff) << 16);
    uint64_t taken = 0;
    for (uint64_t rest = todo; rest != 0; rest &= rest - 1) {
        const int i = __builtin_ctzll(rest);
        const uint32_t mt = (uint32_t)__builtin_amdgcn_readlane(meta, i);
        const int gbi = mt & 0xff, gci = (mt >> 8) & 0xff;
        const uint32_t df = (mt >> 16) & 0xff;
        const uint32_t drng = (uint32_t)__builtin_amdgcn_readlane(t_rng, i);
        const uint32_t cellbits = (1u << gci) - 1;
        const uint32_t igl = (uint32_t)(IG >> gbi) & cellbits;
        const uint32_t freel = ~(uint32_t)(taken >> gbi) & cellbits;
        const uint32_t free1 = freel & ~igl, free2 = freel & igl;
        double best1 = thr0, best2 = thr0;
        int m1 = -1, m2 = -1;
        const double *__restrict__ row = &s_iou[wave][FUSED ? 0 : i * GRP_GCAP];
        double bx = 0, by = 0, bw = 0, bh = 0;
        if (FUSED) {
            bx = readlane_f64(B.x, i); by = readlane_f64(B.y, i);
            bw = readlane_f64(B.z, i); bh = readlane_f64(B.w, i);
        }
        for (int g = 0; g < gci; g++) {
            double v;
            if (FUSED) {
                // (the same operands through the same function as above: the same bits)
                const double4 A = s_gt[wave][gbi + g];
                v = box_iou(bx, by, bw, bh, A.x, A.y, A.z, A.w);
            } else {
                v = row[g];
            }
            const bool ok1 = ((free1 >> g) & 1u) && !(v < best1);
            const bool ok2 = ((free2 >> g) & 1u) && !(v < best2);
            best1 = ok1 ? v : best1;  m1 = ok1 ? g : m1;
            best2 = ok2 ? v : best2;  m2 = ok2 ? g : m2;
        }
        const int m = m1 >= 0 ? m1 : m2;            // cell-local index
        if (m >= 0 && !(df & TAOAMD_DT_NO_CONSUME)) taken |= 1ull << (gbi + m);
        const bool vis = m >= 0 && !((HID >> (gbi + (m >= 0 ? m : 0))) & 1);
        bool ig = m >= 0 && ((igl >> (m >= 0 ? m : 0)) & 1u);
        if (!vis && ((drng >> r) & 1u)) ig = true;
        const uint64_t mw = __ballot(active && vis);
        const uint64_t iw = __ballot(active && ig);
        if (lane == i) { my_m = mw; my_i = iw; }
        if (a.match_gt != nullptr && active)
            a.match_gt[(int64_t)(d0 + i) * n_combo + combo] = m;
    }
    if (lane < nD) {
        store_row(a, t_row, word, my_m, my_i);
    }
}

// Slow path: cells with more than 64 ground truths.  One wavefront (one
// 64-thread block) per (cell, word); dynamic LDS = IoU row [Gmax] doubles +
// taken bitsets [ceil(Gmax/32)][64] words (lane-minor: conflict free).
template <bool FUSED>
__global__ __launch_bounds__(64) void match_big_kernel(MatchArgs a, IouThr thr,
                                                       int32_t g_cap)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *row = reinterpret_cast<double *>(smem);
    uint32_t *takenw = reinterpret_cast<uint32_t *>(smem + (size_t)g_cap * 8);
    const int lane = lane_id();
    const int64_t item = blockIdx.x;
    const int64_t ci = item / a.n_words;
    const int word = (int)(item - ci * a.n_words);
    const int64_t cell = a.singles != nullptr ? a.singles[ci] : ci;
    const int32_t d0 = a.cell_dt_off[cell], D = a.cell_dt_off[cell + 1] - d0;
    const int32_t g0 = a.cell_gt_off[cell], G = a.cell_gt_off[cell + 1] - g0;
    if (D == 0 || G <= WAVE) return;
    const int n_combo = a.n_rng * N_THR;
    const int combo = word * WAVE + lane;
    const bool active = combo < n_combo;
    const int r = active ? combo / N_THR : 0;
    const int t = active ? combo - r * N_THR : 0;
    const double thr0 = thr.v[t];
    const int64_t ioff = a.cell_iou_off != nullptr ? a.cell_iou_off[cell] : 0;
    const int n_tw = (G + 31) / 32;
    for (int w = 0; w < n_tw; w++) takenw[w * WAVE + lane] = 0;
    for (int32_t dd = 0; dd < D; dd++) {
        const int32_t d = d0 + dd;
        __syncthreads();  // single-wave block: orders the LDS row reuse
        if (FUSED) {
            const double4 B = reinterpret_cast<const double4 *>(a.dt_box)[d];
            for (int g = lane; g < G; g += WAVE) {
                const double4 A = reinterpret_cast<const double4 *>(a.gt_box)[g0 + g];
                const double v = box_iou(B.x, B.y, B.z, B.w, A.x, A.y, A.z, A.w);
                row[g] = v;
                if (a.ious_out != nullptr && word == 0)
                    a.ious_out[ioff + (int64_t)dd * G + g] = v;
            }
        } else {
            for (int g = lane; g < G; g += WAVE)
                row[g] = a.iou[ioff + (int64_t)dd * G + g];
        }
        __syncthreads();
        double best1 = thr0, best2 = thr0;
        int m1 = -1, m2 = -1;
        uint32_t tw = 0;
        for (int g = 0; g < G; g++) {
            if ((g & 31) == 0) tw = takenw[(g >> 5) * WAVE + lane];
            const double v = row[g];
            const bool ign = (a.gt_rng[g0 + g] >> r) & 1u;
            const bool fr = !((tw >> (g & 31)) & 1u);
            const bool ok1 = fr && !ign && !(v < best1);
            const bool ok2 = fr && ign && !(v < best2);
            best1 = ok1 ? v : best1;  m1 = ok1 ? g : m1;
            best2 = ok2 ? v : best2;  m2 = ok2 ? g : m2;
        }
        const int m = m1 >= 0 ? m1 : m2;
        const uint8_t df = a.dt_flags[d];
        if (m >= 0 && !(df & TAOAMD_DT_NO_CONSUME))
            takenw[(m >> 5) * WAVE + lane] |= 1u << (m & 31);
        const bool vis = m >= 0 && !(a.gt_flags[g0 + max(m, 0)] & TAOAMD_GT_ID_HIDDEN);
        bool ig = m >= 0 && ((a.gt_rng[g0 + max(m, 0)] >> r) & 1u);
        if (!vis && ((dt_rng_of(a, d, df) >> r) & 1u)) ig = true;
        const uint64_t mw = __ballot(active && vis);
        const uint64_t iw = __ballot(active && ig);
        if (lane == 0) {
            const int64_t rowi = a.dst != nullptr ? a.dst[d] : d;
            store_row(a, rowi, word, mw, iw);
        }
        if (a.match_gt != nullptr && active)
            a.match_gt[(int64_t)d * n_combo + combo] = m;
    }
}

// ---------------------------------------------------------------- host side
extern "C" int taoamd_bb_iou(const double *dt, const double *gt, size_t m,
                             size_t n, const unsigned char *iscrowd, double *o,
                             void *stream)
{
    if (m == 0 || n == 0) return TAOAMD_OK;
    if (!dt || !gt || !o) return TAOAMD_ERR_ARG;
    size_t total = m * n;
    unsigned blocks = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    TAO_TIMED("bb_iou_kernel", (hipStream_t)stream, bb_iou_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(dt, gt, m, n, iscrowd, o));
    TAO_LAUNCH_CHECK();
    return TAOAMD_OK;
}

extern "C" int taoamd_bb_iou_host(const double *dt, const double *gt, size_t m,
                                  size_t n, const unsigned char *iscrowd,
                                  double *o)
{
    if (m == 0 || n == 0) return TAOAMD_OK;
    if (!dt || !gt || !o) return TAOAMD_ERR_ARG;
    double *d_dt = nullptr, *d_gt = nullptr, *d_o = nullptr;
    unsigned char *d_c = nullptr;
    TAO_HIP(hipMalloc(&d_dt, m * 32));
    TAO_HIP(hipMalloc(&d_gt, n * 32));
    TAO_HIP(hipMalloc(&d_o, m * n * 8));
    TAO_HIP(hipMemcpy(d_dt, dt, m * 32, hipMemcpyHostToDevice));
    TAO_HIP(hipMemcpy(d_gt, gt, n * 32, hipMemcpyHostToDevice));
    if (iscrowd) {
        TAO_HIP(hipMalloc(&d_c, n));
        TAO_HIP(hipMemcpy(d_c, iscrowd, n, hipMemcpyHostToDevice));
    }
    int st = taoamd_bb_iou(d_dt, d_gt, m, n, d_c, d_o, nullptr);
    if (st == TAOAMD_OK) {
        TAO_HIP(hipMemcpy(o, d_o, m * n * 8, hipMemcpyDeviceToHost));
    }
    (void)hipFree(d_dt); (void)hipFree(d_gt); (void)hipFree(d_o);
    if (d_c) (void)hipFree(d_c);
    return st;
}

extern "C" int taoamd_lvis_ranges(int64_t n_gt, const double *gt_vis,
                                  const uint8_t *gt_flags,
                                  const int32_t *gt_cat,
                                  const int32_t *gt_cat_off, int64_t n_dt,
                                  const uint8_t *dt_flags, int32_t n_cat,
                                  uint32_t *gt_rng, uint32_t *dt_rng,
                                  int32_t *num_gt, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    const bool grouped = gt_cat_off != nullptr;
    if (!grouped)
        TAO_HIP(hipMemsetAsync(num_gt, 0, sizeof(int32_t) * (size_t)n_cat * TAOAMD_LVIS_RNG, s));
    // dt_rng == NULL: the caller lets taoamd_match derive it from dt_flags
    if (dt_rng == nullptr) n_dt = 0;
    int64_t n = n_gt > n_dt ? n_gt : n_dt;
    if (n > 0) {
        TAO_TIMED("lvis_ranges_kernel", s, lvis_ranges_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(
            n_gt, gt_vis, gt_flags, gt_cat, n_dt, dt_flags, gt_rng, dt_rng,
            grouped ? nullptr : num_gt, range_tab()));
    }
    if (grouped)
        TAO_TIMED("count_gt_kernel", s, count_gt_kernel<<<(unsigned)n_cat, 256, 0, s>>>(
            n_cat, TAOAMD_LVIS_RNG, gt_cat_off, gt_rng, num_gt));
    TAO_LAUNCH_CHECK();
    return TAOAMD_OK;
}

extern "C" int taoamd_tao_ranges(int64_t n_gt, const double *gt_area,
                                 const int32_t *gt_len, const int32_t *gt_nhp,
                                 const uint8_t *gt_flags, const int32_t *gt_cat,
                                 const int32_t *gt_cat_off, int64_t n_dt,
                                 const double *dt_area, const int32_t *dt_len,
                                 const uint8_t *dt_flags, int32_t n_cat,
                                 uint32_t *gt_rng, uint32_t *dt_rng,
                                 int32_t *num_gt, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    const bool grouped = gt_cat_off != nullptr;
    if (!grouped)
        TAO_HIP(hipMemsetAsync(num_gt, 0, sizeof(int32_t) * (size_t)n_cat * TAOAMD_TAO_RNG, s));
    int64_t n = n_gt > n_dt ? n_gt : n_dt;
    if (n > 0) {
        TAO_TIMED("tao_ranges_kernel", s, tao_ranges_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(
            n_gt, gt_area, gt_len, gt_nhp, gt_flags, gt_cat, n_dt, dt_area,
            dt_len, dt_flags, gt_rng, dt_rng, grouped ? nullptr : num_gt, range_tab()));
    }
    if (grouped)
        TAO_TIMED("count_gt_kernel", s, count_gt_kernel<<<(unsigned)n_cat, 256, 0, s>>>(
            n_cat, TAOAMD_TAO_RNG, gt_cat_off, gt_rng, num_gt));
    TAO_LAUNCH_CHECK();
    return TAOAMD_OK;
}

// Launch plan of taoamd_match from HOST copies of the cell offsets: runs of
// consecutive small cells (each <= cap_d detections and <= cap_cell_g ground
// truths, the run <= cap_d detections and <= cap_g ground truths in total)
// become groups {first cell, end cell} for match_group_kernel, every other
// cell that holds detections a single.  Call with groups == NULL for the
// sizes (sizes[0] = groups, sizes[1] = singles).
extern "C" int taoamd_match_plan_host(int64_t n_cells, const int32_t *cell_dt_off,
                                      const int32_t *cell_gt_off, int32_t cap_d,
                                      int32_t cap_g, int32_t cap_cell_g,
                                      int64_t *sizes, int32_t *groups,
                                      int32_t *singles)
{
    if (n_cells < 0 || !cell_dt_off || !cell_gt_off || !sizes) return TAOAMD_ERR_ARG;
    if ((groups == nullptr) != (singles == nullptr)) return TAOAMD_ERR_ARG;
    int64_t ng = 0, ns = 0, c = 0;
    auto small = [&](int64_t k) {
        return cell_dt_off[k + 1] - cell_dt_off[k] <= cap_d &&
               cell_gt_off[k + 1] - cell_gt_off[k] <= cap_cell_g;
    };
    while (c < n_cells) {
        if (!small(c)) {
            if (cell_dt_off[c + 1] > cell_dt_off[c]) {
                if (singles) singles[ns] = (int32_t)c;
                ns++;
            }
            c++;
            continue;
        }
        int64_t e = c + 1;
        while (e < n_cells && small(e) &&
               cell_dt_off[e + 1] - cell_dt_off[c] <= cap_d &&
               cell_gt_off[e + 1] - cell_gt_off[c] <= cap_g)
            e++;
        if (cell_dt_off[e] > cell_dt_off[c]) {
            if (groups) {
                groups[2 * ng] = (int32_t)c;
                groups[2 * ng + 1] = (int32_t)e;
            }
            ng++;
        }
        c = e;
    }
    sizes[0] = ng;
    sizes[1] = ns;
    return TAOAMD_OK;
}

extern "C" int taoamd_match(int64_t n_cells, const int32_t *cell_dt_off,
                            const int32_t *cell_gt_off,
                            const int64_t *cell_iou_off,
                            int32_t max_gt_per_cell, const double *dt_box,
                            const double *gt_box, const double *iou,
                            int32_t n_rng, const uint32_t *gt_rng,
                            const uint32_t *dt_rng, const uint8_t *gt_flags,
                            const uint8_t *dt_flags, const int32_t *dst,
                            int64_t out_stride, uint64_t *matched,
                            uint64_t *ignored, int32_t *match_gt,
                            double *ious_out, const int32_t *dt_group,
                            const uint32_t *dt_meta,
                            const int32_t *groups, int32_t n_groups,
                            const int32_t *singles, int32_t n_singles,
                            void *stream)
{
    if (n_cells == 0) return TAOAMD_OK;
    if (n_rng < 1 || n_rng > 32) return TAOAMD_ERR_ARG;
    const bool fused = dt_box != nullptr;
    if (fused ? (gt_box == nullptr) : (iou == nullptr)) return TAOAMD_ERR_ARG;
    if (max_gt_per_cell > TAOAMD_MAX_GT_PER_CELL) return TAOAMD_ERR_TOO_LARGE;
    if ((!fused || ious_out) && cell_iou_off == nullptr) return TAOAMD_ERR_ARG;
    const bool planned = groups != nullptr;
    if (planned && (dt_group == nullptr || (n_singles > 0 && singles == nullptr)))
        return TAOAMD_ERR_ARG;
    MatchArgs a;
    a.n_cells = n_cells; a.cell_dt_off = cell_dt_off; a.cell_gt_off = cell_gt_off;
    a.cell_iou_off = cell_iou_off; a.dt_box = dt_box; a.gt_box = gt_box;
    a.iou = iou; a.gt_rng = gt_rng; a.dt_rng = dt_rng; a.gt_flags = gt_flags;
    a.dt_flags = dt_flags; a.dst = dst; a.matched = matched; a.ignored = ignored;
    a.match_gt = match_gt; a.ious_out = ious_out; a.n_rng = n_rng;
    a.n_words = (n_rng * N_THR + 63) / 64;
    a.paired = matched != nullptr && ignored == matched + 1;
    a.out_stride = out_stride > 0 ? out_stride : (a.paired ? 2 : 1) * a.n_words;
    if (a.out_stride < (a.paired ? 2 : 1) * a.n_words) return TAOAMD_ERR_ARG;
    a.wide = a.paired && (((uintptr_t)matched) & 15) == 0 && (a.out_stride & 1) == 0;
    // (the packed word carries no cell index: not where IoUs are read or written)
    a.dt_meta = fused && ious_out == nullptr ? dt_meta : nullptr;
    a.dt_group = dt_group; a.groups = groups; a.n_groups = planned ? n_groups : 0;
    a.singles = planned ? singles : nullptr; a.n_singles = planned ? n_singles : 0;
    static const int xcd_env = getenv("TAOAMD_XCD") ? atoi(getenv("TAOAMD_XCD")) : 1;
    a.xcd = xcd_env;
    hipStream_t s = (hipStream_t)stream;
    if (planned && n_groups > 0) {
        const unsigned gb = (unsigned)(((int64_t)n_groups * a.n_words + 3) / 4);
        // (one combo word of at most six ranges: the fast kernel's word assembly)
        const bool fast = fused && a.dt_meta && a.dst && !a.dt_rng && !a.ious_out && !a.match_gt &&
                          a.n_words == 1 && n_rng <= 6;
        if (fast) TAO_TIMED("match_group_kernel", s, (match_group_kernel<true, true><<<gb, 256, 0, s>>>(a, match_thr())));
        else if (fused) TAO_TIMED("match_group_kernel", s, match_group_kernel<true><<<gb, 256, 0, s>>>(a, match_thr()));
        else TAO_TIMED("match_group_kernel", s, match_group_kernel<false><<<gb, 256, 0, s>>>(a, match_thr()));
    }
    const int64_t cells = planned ? n_singles : n_cells;
    const int64_t items = cells * a.n_words;
    if (items > 0) {
        const unsigned blocks = (unsigned)((items + 3) / 4);
        if (fused) TAO_TIMED("match_kernel", s, match_kernel<true><<<blocks, 256, 0, s>>>(a, match_thr()));
        else TAO_TIMED("match_kernel", s, match_kernel<false><<<blocks, 256, 0, s>>>(a, match_thr()));
        if (max_gt_per_cell > WAVE) {
            const int32_t cap = (max_gt_per_cell + 31) / 32 * 32;
            const size_t lds = (size_t)cap * 8 + (size_t)(cap / 32) * WAVE * 4;
            if (fused)
                TAO_TIMED("match_big_kernel", s, match_big_kernel<true><<<(unsigned)items, 64, lds, s>>>(a, match_thr(), cap));
            else
                TAO_TIMED("match_big_kernel", s, match_big_kernel<false><<<(unsigned)items, 64, lds, s>>>(a, match_thr(), cap));
        }
    }
    TAO_LAUNCH_CHECK();
    return TAOAMD_OK;
}

// ------------------------------------------------------------- row gather
__global__ void gather_rows_kernel(int64_t n, int32_t n_words,
                                   const uint64_t *__restrict__ sm,
                                   const uint64_t *__restrict__ si,
                                   int64_t stride,
                                   const int32_t *__restrict__ order,
                                   uint64_t *__restrict__ dm,
                                   uint64_t *__restrict__ di)
{
    const int64_t total = n * n_words;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i / n_words;
        const int w = (int)(i - p * n_words);
        const int64_t src = (int64_t)order[p] * stride + w;
        dm[i] = sm[src];
        di[i] = si[src];
    }
}

extern "C" int taoamd_gather_rows(int64_t n, int32_t n_words,
                                  const uint64_t *src_matched,
                                  const uint64_t *src_ignored,
                                  int64_t src_stride, const int32_t *order,
                                  uint64_t *dst_matched, uint64_t *dst_ignored,
                                  void *stream)
{
    if (n == 0) return TAOAMD_OK;
    if (n_words < 1 || src_stride < n_words) return TAOAMD_ERR_ARG;
    // the destination tables are DENSE [n][n_words]; the interleaved pair
    // layout of taoamd_match / taoamd_accumulate (ignored == matched + 1) is
    // not written here: refused rather than silently scrambled
    if (dst_ignored == dst_matched + 1 && n * n_words > 1) return TAOAMD_ERR_ARG;
    const int64_t total = n * n_words;
    unsigned blocks = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    TAO_TIMED("gather_rows_kernel", (hipStream_t)stream, gather_rows_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(
        n, n_words, src_matched, src_ignored, src_stride, order, dst_matched,
        dst_ignored));
    TAO_LAUNCH_CHECK();
    return TAOAMD_OK;
}
