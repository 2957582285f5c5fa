// accumulate(): TP/FP sweep over the score-sorted detections of every
// category, precision envelope and 101-point recall sampling (gfx950).
// Reference: lvis_amodal/eval.py:339-417 == tao_amodal/eval.py:496-573.
//
// Input rows are already in (category, descending score, concatenation) order
// (sort.hip + the scatter of match_kernel), so a category is a contiguous row
// range [cat_off[k], cat_off[k+1]).  Lane = one (range, IoU threshold) combo,
// exactly as in match_kernel, so a detection's two 64-bit words are consumed
// with wave-uniform (scalar) loads and one bit test per lane.
//
// A category is cut into chunks of ACC_CH rows so that long categories
// (e.g. "person") spread over the whole chip:
//
//   acc_chunks     chunk table from cat_off (block scan)
//   acc_count      per chunk: #TP, #FP per combo
//   acc_prefix     per category: exclusive prefix of the chunk counts; recall
//   acc_chunkmax   per chunk: max precision at a TP row (needs the prefix)
//   acc_sufmax     per category: reverse exclusive max over chunks
//   acc_emit       per chunk, walking backwards: running max = precision
//                  envelope; each recall threshold is written by the chunk
//                  holding the TP that first reaches it  -> val[k][r][t][j]
//   acc_transpose  val -> precision[T][R][K][A] (reference layout), -1 fill,
//                  LDS-tiled so that both sides are coalesced
//
// Precision at a TP row is tp / (fp + tp + eps) in fp64, the very expression
// of the reference; the envelope is a max of those values, so the order in
// which chunks are combined cannot change a bit of the result.
//
// Rows are consumed 64 at a time through a 64 x 64 bit transpose across the
// wavefront (six butterfly stages of __shfl_xor): lane = row holds the row's
// TP word (bit = combo); after the transpose lane = combo holds a word whose
// bit q says whether row q is a TP for that combo.  Counting is then a
// popcount, and the envelope / emission sweeps visit only the TP rows of each
// combo (ctz / clz), taking the TP and FP counts at that row from popcounts of
// the masks below it.  acc_count writes the transposed words once; the two
// later sweeps read them back (lane-contiguous).
//
// Two exact shortcuts keep fp64 divisions off the per-row paths:
//  * the envelope is tracked as the integer pair (tp, n = tp + fp) and
//    compared by cross-multiplication.  fl(tp / (n + eps)) is monotone in the
//    rational tp/n, and n + eps == n for n >= 2, so the pair with the largest
//    rational (ties: larger n, which ranks (1,1) -> 1/(1+eps) below (k,k) -> 1)
//    yields the largest fp64 value; the division happens once, when a recall
//    threshold is emitted;
//  * recall thresholds are crossed at integer TP counts: cj[j] = the smallest
//    c with fl(c / num_gt) >= rec_thrs[j] is tabulated per (category, range)
//    (in acc_prefix_kernel / acc_fused_kernel) with the reference's fp64 comparison, and the sweep
//    compares integers.
#include <atomic>
#include <cstdlib>
#include <cstring>

#include "common.hpp"

using namespace taoamd;

#ifndef ACC_CH
#define ACC_CH 256
#endif
#ifndef ACC_INLINE_CHUNKS
#define ACC_INLINE_CHUNKS 64    // categories up to 16384 rows scan their chunks inline
#endif

struct AccArgs {
    int64_t n_dt;
    int32_t n_cat, n_rng, n_words, n_chunks_max;
    int32_t paired;          // rows are (matched, ignored) pairs: ignored == matched + 1
    int32_t wide;            // ... 16-byte aligned: one load per pair
    const int32_t *cat_off;
    const uint64_t *matched;
    const uint64_t *ignored;
    const int32_t *order;    // optional: sorted position -> row of matched / ignored
    const int32_t *num_gt;
    int32_t *cat_chunk_off;  // [n_cat + 1]
    const int32_t *chunk_tab;  // [chunk] category of the chunk; null: searched per wavefront
    uint32_t *cnt_tp, *cnt_fp;   // [chunk][word][64] counts inside the chunk
    uint32_t *pre_tp, *pre_fp;   // exclusive prefix inside the category
    uint64_t *t_tp, *t_fp;       // [chunk][word][4 blocks][64] transposed TP / FP words
    uint64_t *cmax;              // chunk max as (tp << 32 | n), then reverse-exclusive max
    int32_t *cj;                 // [n_cat][n_rng][N_REC] TP count crossing each recall thr
    uint64_t *val;               // [n_cat][n_rng][N_THR][N_REC] (tp << 32 | n) records
    double *rec;                 // [n_cat][n_rng][N_THR]
    int32_t k_begin, k_end;      // categories swept by this call
    int32_t fused_rows;          // categories up to this many rows take acc_fused_kernel
    int32_t fused_lo;            // ... and more than this many (size class of the launch)
    int32_t inline_scans;        // short categories: no acc_prefix / acc_sufmax launches
    // one-pass sweep (acc_sweep_kernel): the chunk table then is a table of
    // super-chunks of sc_rows rows (cat_chunk_off / chunk_tab / cnt_* per SC)
    int32_t sc_rows;             // 0: the chunked kernels
    uint32_t sc_gen;             // generation of this call's status words
    uint64_t *sc_stat;           // [SC][word][64][2] look-back status {tp, fp}
    uint64_t *sc_max;            // [SC][word][64] largest precision record of the SC
    uint8_t *sc_jhi;             // [SC][word][64] recall thresholds reached up to its end
    uint32_t *sc_error;          // set when a look-back gave up (never observed)
    uint32_t *sc_giveups;        // caller's counter of such waits, never cleared here (may be null)
    int32_t *xcd_start;          // [9] first SC of each XCD's run (null: SC = workgroup number)
    int32_t sc_spin;             // polls a look-back waits for one predecessor; < 0: it gives
                                 // up at once (fault injection, taoamd_accumulate_spin_limit)
};

__global__ __launch_bounds__(256) void acc_chunks_kernel(AccArgs a)
{
    __shared__ int32_t part[256];
    const int per = (a.n_cat + 255) / 256;
    const int lo = threadIdx.x * per, hi = min(lo + per, a.n_cat);
    int32_t s = 0;
    // (categories short enough for acc_fused_kernel get no chunks here)
    auto chunks_of = [&](int k) {
        const int32_t rows = a.cat_off[k + 1] - a.cat_off[k];
        if (a.fused_rows > 0 && rows <= a.fused_rows) return 0;
        // without acc_prefix_kernel nobody visits a category that has no
        // chunk: one of no rows carries its recall 0 / precision 0
        if ((a.inline_scans || a.sc_rows) && rows == 0) return 1;
        const int32_t per = a.sc_rows ? a.sc_rows : ACC_CH;
        return (rows + per - 1) / per;
    };
    for (int k = lo; k < hi; k++) s += chunks_of(k);
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        int32_t v = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int32_t run = part[threadIdx.x] - s;
    for (int k = lo; k < hi; k++) {
        a.cat_chunk_off[k] = run;
        run += chunks_of(k);
    }
    if (threadIdx.x == 255) a.cat_chunk_off[a.n_cat] = part[255];
    if (a.xcd_start == nullptr) return;
    // eight runs of SCs for the eight XCDs, cut at category boundaries: run x
    // starts at the first category that begins at or behind x eighths
    __syncthreads();
    if (threadIdx.x <= N_XCD) {
        const int32_t total = part[255];
        const int32_t want = (int32_t)((int64_t)total * threadIdx.x / N_XCD);
        int32_t lo = 0, hi = a.n_cat;              // first k with cat_chunk_off[k] >= want
        while (lo < hi) {
            const int32_t mid = (lo + hi) >> 1;
            if (a.cat_chunk_off[mid] < want) lo = mid + 1; else hi = mid;
        }
        a.xcd_start[threadIdx.x] = threadIdx.x == N_XCD ? total : a.cat_chunk_off[lo];
    }
}

// category owning chunk c: last k with cat_chunk_off[k] <= c (uniform)
__device__ __forceinline__ int32_t chunk_cat(const int32_t *__restrict__ off,
                                             int32_t n_cat, int32_t c)
{
    int32_t lo = 0, hi = n_cat;
    while (hi - lo > 1) {
        int32_t mid = (lo + hi) >> 1;
        if (off[mid] <= c) lo = mid; else hi = mid;
    }
    return lo;
}

// chunk -> category table of a prepared plan (taoamd_accumulate_prepare)
__global__ void acc_chunktab_kernel(AccArgs a, int32_t *__restrict__ tab)
{
    const int32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < a.cat_chunk_off[a.n_cat]) tab[c] = chunk_cat(a.cat_chunk_off, a.n_cat, c);
}

__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int lane)
{
    uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
    return ((uint64_t)hi << 32) | lo;
}

// Rows [base, base+n) of a chunk: lane q loads row q (coalesced), returning
// the TP and FP words of that row; the sequential loops then broadcast row q
// with v_readlane, so there is no memory access inside them.
// `first` = sorted position of the block's first row.  With a.order the rows
// still lie where the match kernel produced them (cell order: its stores are
// full wavefront runs and it does not wait for the sort) and are gathered here,
// the one place that reads them.
__device__ __forceinline__ void load_rows(const AccArgs &a, int64_t first, int word,
                                          int n, int lane, uint64_t &tpw,
                                          uint64_t &fpw)
{
    uint64_t m = 0, i = ~0ull;
    if (lane < n) {
        int64_t r = first + lane;
        if (a.order) r = a.order[r];
        if (a.wide) {
            // one 16-byte load of the pair the match stored with one store
            const ulonglong2 v =
                *reinterpret_cast<const ulonglong2 *>(a.matched + 2 * (r * a.n_words + word));
            m = v.x;
            i = v.y;
        } else if (a.paired) {
            m = a.matched[2 * (r * a.n_words + word)];
            i = a.matched[2 * (r * a.n_words + word) + 1];
        } else {
            m = a.matched[r * a.n_words + word];
            i = a.ignored[r * a.n_words + word];
        }
    }
    tpw = m & ~i;
    fpw = ~m & ~i;
}

// The ACC_BLK blocks of a chunk's rows, all loads issued before the first is
// waited for.  Round 4: load_rows() block by block compiled to a chain -- every
// block's loads sit behind the run-time layout branches (order / wide / paired)
// and an `if (lane < n)`, and the compiler put an s_waitcnt vmcnt(0) between
// them: eight dependent round trips to HBM per wavefront in the counting,
// fused and one-pass kernels.  The common layout (pairs, 16-byte aligned, no
// gather) gets ACC_BLK unconditional 16-byte loads -- a lane past the chunk's
// rows re-reads the chunk's first row -- and the others keep load_rows().
template <int ACC_BLK_N>
__device__ __forceinline__ void load_chunk(const AccArgs &a, int64_t start, int word, int len,
                                           int lane, uint64_t (&tpw)[ACC_BLK_N],
                                           uint64_t (&fpw)[ACC_BLK_N])
{
    if (a.wide && a.order == nullptr && len > 0) {
        ulonglong2 v[ACC_BLK_N];
#pragma unroll
        for (int blk = 0; blk < ACC_BLK_N; blk++) {
            const int i = blk * WAVE + lane;
            const int64_t r = start + (i < len ? i : 0);
            v[blk] = *reinterpret_cast<const ulonglong2 *>(a.matched + 2 * (r * a.n_words + word));
        }
#pragma unroll
        for (int blk = 0; blk < ACC_BLK_N; blk++) {
            const bool in = blk * WAVE + lane < len;
            const uint64_t m = in ? v[blk].x : 0, i_ = in ? v[blk].y : ~0ull;
            tpw[blk] = m & ~i_;
            fpw[blk] = ~m & ~i_;
        }
        return;
    }
    if (a.wide && a.order != nullptr && len > 0) {
        // rows in cell order, gathered through the sort's order[]: the places
        // first (coalesced), then all the 16-byte gathers in flight together
        int32_t at[ACC_BLK_N];
#pragma unroll
        for (int blk = 0; blk < ACC_BLK_N; blk++) {
            const int i = blk * WAVE + lane;
            at[blk] = a.order[start + (i < len ? i : 0)];
        }
        ulonglong2 v[ACC_BLK_N];
#pragma unroll
        for (int blk = 0; blk < ACC_BLK_N; blk++)
            v[blk] = *reinterpret_cast<const ulonglong2 *>(
                a.matched + 2 * ((int64_t)at[blk] * a.n_words + word));
#pragma unroll
        for (int blk = 0; blk < ACC_BLK_N; blk++) {
            const bool in = blk * WAVE + lane < len;
            const uint64_t m = in ? v[blk].x : 0, i_ = in ? v[blk].y : ~0ull;
            tpw[blk] = m & ~i_;
            fpw[blk] = ~m & ~i_;
        }
        return;
    }
#pragma unroll
    for (int blk = 0; blk < ACC_BLK_N; blk++)
        load_rows(a, start + blk * WAVE, word, max(0, min(WAVE, len - blk * WAVE)), lane,
                  tpw[blk], fpw[blk]);
}

// value of lane (i ^ S): CDNA4 lane permutes that need neither an address VGPR
// nor a round trip through the LDS crossbar queue (what __shfl_xor compiles to,
// ds_bpermute_b32): v_permlane32_swap / v_permlane16_swap for the two widest
// strides, DPP row_ror:8 and quad_perm for 8 / 2 / 1, ds_swizzle for 4
template <int S>
__device__ __forceinline__ uint32_t xor_lane(uint32_t x, int lane)
{
    if constexpr (S == 32) {
        auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
        return (lane & 32) ? r[0] : r[1];
    } else if constexpr (S == 16) {
        auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
        return (lane & 16) ? r[0] : r[1];
    } else if constexpr (S == 8) {
        return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x128, 0xf, 0xf, false);   // row_ror:8
    } else if constexpr (S == 4) {
        return (uint32_t)__builtin_amdgcn_ds_swizzle((int)x, 0x101f);                // xor 4
    } else if constexpr (S == 2) {
        return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x4e, 0xf, 0xf, false);    // [2,3,0,1]
    } else {
        return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xb1, 0xf, 0xf, false);    // [1,0,3,2]
    }
}

// 64 x 64 bit-matrix transpose across the wavefront: in: lane i holds row i
// (bit j = element (i, j)); out: lane j holds column j (bit i = element (i, j)).
// Stage S swaps the off-diagonal S x S blocks between lanes i and i ^ S.
//
// Round 4: ~30 VALU instructions instead of ~100 (the ternaries of the first
// version compiled to exec-masked branches, both sides executed; the sweep is
// bound by VALU issue -- a wave64 instruction occupies its SIMD for four
// cycles -- and the transposes were 0.12 ms of chip time at 21 M rows):
//   S = 32  the two words change places between the wave's halves: ONE
//           v_permlane32_swap (lower lanes' hi <-> upper lanes' lo);
//   S = 16  half words move: see transpose64 (one v_permlane16_swap for both
//           words, four v_perm_b32 with constant selectors);
//   S = 8   whole bytes move: the partner's word (DPP row_ror:8) and ONE
//           v_perm_b32 with a per-lane selector;
//   S = 4, 2, 1  the partner's word rotated so that the bits it hands over
//           sit where they go (v_alignbit, per-lane amount) and ONE v_bfi.
struct TransposeConsts {
    uint32_t sel8;                 // v_perm_b32 selector
    uint32_t keep4, keep2, keep1;  // bits of my own word that stay
    uint32_t rot4, rot2, rot1;     // right-rotation of the partner's word
};
__device__ __forceinline__ TransposeConsts transpose_consts(int lane)
{
    TransposeConsts c;
    // lower lane of a pair keeps the low part and takes the partner's low part
    // into its high part; the upper lane the other way round
    c.sel8 = (lane & 8) ? 0x03070105u : 0x06020400u;
    c.keep4 = (lane & 4) ? 0xf0f0f0f0u : 0x0f0f0f0fu;
    c.keep2 = (lane & 2) ? 0xccccccccu : 0x33333333u;
    c.keep1 = (lane & 1) ? 0xaaaaaaaau : 0x55555555u;
    c.rot4 = (lane & 4) ? 4 : 28;      // upper: partner >> S; lower: partner << S
    c.rot2 = (lane & 2) ? 2 : 30;
    c.rot1 = (lane & 1) ? 1 : 31;
    return c;
}
template <int S>
__device__ __forceinline__ uint32_t transpose_bits(uint32_t x, int lane, uint32_t keep,
                                                   uint32_t rot)
{
    const uint32_t y = xor_lane<S>(x, lane);
    const uint32_t r = __builtin_amdgcn_alignbit(y, y, rot);
    return (x & keep) | (r & ~keep);                       // v_bfi_b32
}
__device__ __forceinline__ uint64_t transpose64(uint64_t x, int lane, const TransposeConsts &c)
{
    uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    {
        // vdst lanes 32..63 <-> src lanes 0..31
        auto r = __builtin_amdgcn_permlane32_swap(lo, hi, false, false);
        lo = r[0];
        hi = r[1];
    }
    {
        // S = 16 (round 6): the halves that change lanes gathered into ONE
        // register first -- P = the low halves of (lo, hi), Q = the high ones;
        // the lower lane of a pair keeps P and needs its partner's P, the upper
        // one keeps Q and needs its partner's Q -- so that a single
        // v_permlane16_swap (odd rows of P <-> even rows of Q) serves both
        // words and both directions, and the two words are put together again
        // with the same selectors in every lane: 5 instructions for the stage
        // instead of 8 (a copy, a swap, a per-lane select and a v_perm a word).
        const uint32_t p = __builtin_amdgcn_perm(hi, lo, 0x05040100u);
        const uint32_t q = __builtin_amdgcn_perm(hi, lo, 0x07060302u);
        auto r = __builtin_amdgcn_permlane16_swap(p, q, false, false);
        lo = __builtin_amdgcn_perm(r[1], r[0], 0x05040100u);
        hi = __builtin_amdgcn_perm(r[1], r[0], 0x07060302u);
    }
    lo = __builtin_amdgcn_perm(xor_lane<8>(lo, lane), lo, c.sel8);
    hi = __builtin_amdgcn_perm(xor_lane<8>(hi, lane), hi, c.sel8);
    lo = transpose_bits<4>(lo, lane, c.keep4, c.rot4);
    hi = transpose_bits<4>(hi, lane, c.keep4, c.rot4);
    lo = transpose_bits<2>(lo, lane, c.keep2, c.rot2);
    hi = transpose_bits<2>(hi, lane, c.keep2, c.rot2);
    lo = transpose_bits<1>(lo, lane, c.keep1, c.rot1);
    hi = transpose_bits<1>(hi, lane, c.keep1, c.rot1);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t transpose64(uint64_t x, int lane)
{
    return transpose64(x, lane, transpose_consts(lane));
}

// The one-pass sweep's form of load_chunk: the TP word and the word of the rows
// that COUNT (not ignored: TP | FP) of every row -- the sweep needs the
// transposes of exactly these two (T and T | F; the FP count of a lane is the
// difference of their popcounts).  A FULL chunk of the common layout (all but a
// category's last) is fetched from one uniform base address with one 32-bit
// offset per lane (global_load_dwordx4 v, v_off, s[base]) and no in-range
// selects: ~190 VALU instructions of address arithmetic, selects and and-nots
// a chunk become ~50.
template <int ACC_BLK_N>
__device__ __forceinline__ void load_chunk_tv(const AccArgs &a, int64_t start, int word, int len,
                                              int lane, uint64_t (&tpw)[ACC_BLK_N],
                                              uint64_t (&vw)[ACC_BLK_N])
{
    if (a.wide && a.order == nullptr && len == ACC_BLK_N * WAVE) {
        const char *base = reinterpret_cast<const char *>(a.matched + 2 * (start * a.n_words + word));
        const uint32_t lane_off = (uint32_t)lane * (uint32_t)a.n_words * 16u;
        const uint32_t blk_off = (uint32_t)WAVE * (uint32_t)a.n_words * 16u;
        ulonglong2 v[ACC_BLK_N];
#pragma unroll
        for (int blk = 0; blk < ACC_BLK_N; blk++)
            v[blk] = *reinterpret_cast<const ulonglong2 *>(base + ((uint32_t)blk * blk_off + lane_off));
#pragma unroll
        for (int blk = 0; blk < ACC_BLK_N; blk++) {
            tpw[blk] = v[blk].x & ~v[blk].y;
            vw[blk] = ~v[blk].y;
        }
        return;
    }
    uint64_t fpw[ACC_BLK_N];
    load_chunk(a, start, word, len, lane, tpw, fpw);
#pragma unroll
    for (int blk = 0; blk < ACC_BLK_N; blk++) vw[blk] = tpw[blk] | fpw[blk];
}

#define ACC_BLK (ACC_CH / WAVE)   // 64-row blocks per chunk

struct ChunkInfo {
    int32_t k, first, last;  // category, is first / last chunk of it
    int64_t start;
    int32_t len, c, word;
    bool valid;
};

__device__ __forceinline__ ChunkInfo chunk_info(const AccArgs &a)
{
    ChunkInfo ci;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t item = (int64_t)blockIdx.x * 4 + wave;
    ci.c = (int32_t)(item / a.n_words);
    ci.word = (int32_t)(item - (int64_t)ci.c * a.n_words);
    const int32_t total = a.cat_chunk_off[a.n_cat];
    ci.valid = ci.c < total;
    if (!ci.valid) return ci;
    // (a wave-lifetime trace of the emission kernel: the eleven dependent
    // scalar loads of the search were 3.8 us of a 23 us wavefront)
    ci.k = a.chunk_tab ? a.chunk_tab[ci.c] : chunk_cat(a.cat_chunk_off, a.n_cat, ci.c);
    const int32_t j = ci.c - a.cat_chunk_off[ci.k];
    ci.start = (int64_t)a.cat_off[ci.k] + (int64_t)j * ACC_CH;
    const int64_t end = a.cat_off[ci.k + 1];
    ci.len = (int32_t)min((int64_t)ACC_CH, end - ci.start);
    ci.first = j == 0;
    ci.last = ci.c + 1 == a.cat_chunk_off[ci.k + 1];
    return ci;
}

__global__ __launch_bounds__(256) void acc_count_kernel(AccArgs a)
{
    const ChunkInfo ci = chunk_info(a);
    if (!ci.valid) return;
    const int lane = lane_id();
    uint32_t tp = 0, fp = 0;
    const int64_t tb = (((int64_t)ci.c * a.n_words + ci.word) * ACC_BLK) * WAVE + lane;
    // every load of the chunk ahead of its first store: gfx9 counts loads and
    // stores in one in-order counter (vmcnt), so a load issued after a store
    // waits for that store's acknowledgement as well
    uint64_t tpw[ACC_BLK], fpw[ACC_BLK];
    load_chunk(a, ci.start, ci.word, ci.len, lane, tpw, fpw);
#pragma unroll
    for (int blk = 0; blk < ACC_BLK; blk++) {
        uint64_t T = 0, F = 0;
        if (blk * WAVE < ci.len) {
            T = transpose64(tpw[blk], lane);
            F = transpose64(fpw[blk], lane);
        }
        a.t_tp[tb + (int64_t)blk * WAVE] = T;
        a.t_fp[tb + (int64_t)blk * WAVE] = F;
        tp += (uint32_t)__popcll(T);
        fp += (uint32_t)__popcll(F);
    }
    const int64_t o = ((int64_t)ci.c * a.n_words + ci.word) * WAVE + lane;
    a.cnt_tp[o] = tp;
    a.cnt_fp[o] = fp;
}

// one workgroup per (category, word): the four wavefronts scan a quarter of
// the category's chunks each, the quarters are stitched through LDS (long
// categories -- a rank's share of a multi-GPU job -- would otherwise be one
// serial chain of dependent loads)
__global__ __launch_bounds__(256) void acc_prefix_kernel(AccArgs a, RecThr rec)
{
    __shared__ uint32_t s_tp[4][WAVE], s_fp[4][WAVE];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t item = blockIdx.x;
    const int32_t k = a.k_begin + (int32_t)(item / a.n_words);
    const int word = (int)(item % a.n_words);
    const int lane = lane_id();
    if (a.fused_rows > 0 && a.cat_off[k + 1] - a.cat_off[k] <= a.fused_rows)
        return;                                    // acc_fused_kernel's
    // cj[k][r][j]: smallest TP count c with fl(c / num_gt) >= rec_thrs[j]
    // (np.searchsorted(rc, rec_thrs, side="left") on rc = tp / num_gt,
    // reference lvis_amodal/eval.py:386,406-408) -- tabulated here, by the
    // word-0 workgroup of the category, for the emission sweep
    if (word == 0)
        for (int i = threadIdx.x; i < a.n_rng * N_REC; i += 256) {
            const int64_t kr = (int64_t)k * a.n_rng + i / N_REC;
            const int32_t ng = a.num_gt[kr];
            if (ng > 0)
                a.cj[kr * N_REC + i % N_REC] = recall_crossing(rec.v[i % N_REC], ng);
        }
    const int32_t c0 = a.cat_chunk_off[k], c1 = a.cat_chunk_off[k + 1];
    const int32_t q = (c1 - c0 + 3) / 4;
    const int32_t lo = min(c1, c0 + wave * q), hi = min(c1, lo + q);
    uint32_t tp = 0, fp = 0;
    // eight chunks per step: their sixteen loads are in flight together, then
    // the sixteen stores (one by one every load waited for the stores before
    // it: loads and stores share vmcnt)
    for (int32_t c = lo; c < hi; c += 8) {
        uint32_t t_[8], f_[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int64_t o = ((int64_t)min(c + u, hi - 1) * a.n_words + word) * WAVE + lane;
            t_[u] = a.cnt_tp[o];
            f_[u] = a.cnt_fp[o];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (c + u < hi) {
                const int64_t o = ((int64_t)(c + u) * a.n_words + word) * WAVE + lane;
                a.pre_tp[o] = tp;
                a.pre_fp[o] = fp;
                tp += t_[u];
                fp += f_[u];
            }
        }
    }
    s_tp[wave][lane] = tp;
    s_fp[wave][lane] = fp;
    __syncthreads();
    uint32_t btp = 0, bfp = 0;
    for (int w = 0; w < wave; w++) { btp += s_tp[w][lane]; bfp += s_fp[w][lane]; }
    if (wave > 0 && (__ballot(btp | bfp) != 0))
        for (int32_t c = lo; c < hi; c++) {
            const int64_t o = ((int64_t)c * a.n_words + word) * WAVE + lane;
            a.pre_tp[o] += btp;
            a.pre_fp[o] += bfp;
        }
    if (wave != 3) return;
    tp += btp;                                     // whole category
    const int combo = word * WAVE + lane;
    if (combo < a.n_rng * N_THR) {
        const int r = combo / N_THR, t = combo - r * N_THR;
        const int32_t ng = a.num_gt[(int64_t)k * a.n_rng + r];
        if (ng > 0) {
            const int64_t kr = (int64_t)k * a.n_rng + r;
            a.rec[kr * N_THR + t] = (double)tp / (double)ng;
            // a category without detections still has precision 0 / recall 0
            // where it has evaluated GT (reference lvis_amodal/eval.py:412-417)
            if (c0 == c1) {
                uint64_t *out = a.val + (kr * N_THR + t) * N_REC;
                for (int j = 0; j < N_REC; j++) out[j] = PR_ZERO;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// The two walks over the TP rows of a 64-row block.  A TP row directly followed
// by a TP row cannot hold the maximum precision ((tp+1)/(n+1) >= tp/n, and on
// a tie the larger n wins, pr_better), so only the LAST row of each run of
// consecutive TP rows is visited: the bits of m = T & ~(T >> 1).  The block's
// 64-bit words are walked as two 32-bit halves: q, the "rows above / below q"
// mask and the two popcounts are then one instruction each instead of a
// 64-bit pair (32 -> 18 VALU instructions per visited row).
// ---------------------------------------------------------------------------

// forward: best (tp, n) at a TP row of the block; tp0 / n0 = counts before it
__device__ __forceinline__ void best_of_block(uint64_t T, uint64_t TF, uint32_t &tp0,
                                              uint32_t &n0, uint64_t &best)
{
    const uint32_t Th[2] = {(uint32_t)T, (uint32_t)(T >> 32)};
    const uint32_t TFh[2] = {(uint32_t)TF, (uint32_t)(TF >> 32)};
    // run ends; row 31's successor is row 32 (bit 0 of the high half)
    const uint32_t mh[2] = {Th[0] & ~((Th[0] >> 1) | (Th[1] << 31)), Th[1] & ~(Th[1] >> 1)};
#pragma unroll
    for (int h = 0; h < 2; h++) {
        for (uint32_t m = mh[h]; m != 0; m &= m - 1) {
            const int q = __builtin_ctz(m);
            const uint32_t le = (2u << q) - 1;                 // rows <= q (q = 31: all)
            const uint32_t tp = tp0 + (uint32_t)__popc(Th[h] & le);
            const uint32_t n = n0 + (uint32_t)__popc(TFh[h] & le);
            if (pr_better(tp, n, best)) best = pr_pack(tp, n);
        }
        tp0 += (uint32_t)__popc(Th[h]);
        n0 += (uint32_t)__popc(TFh[h]);
    }
}

// INLINE: the category is short (the host says so): the chunk adds up the
// counts of the chunks before it by itself -- at most ACC_INLINE_CHUNKS - 1
// loads per lane, which is cheaper than a launch between two dependent
// kernels (acc_prefix_kernel) on the critical chain of the step.
template <bool INLINE>
__global__ __launch_bounds__(256) void acc_chunkmax_kernel(AccArgs a, RecThr rec)
{
    const ChunkInfo ci = chunk_info(a);
    if (!ci.valid) return;
    const int lane = lane_id();
    const int64_t o = ((int64_t)ci.c * a.n_words + ci.word) * WAVE + lane;
    uint32_t tp0, n0, pre_t = 0, pre_f = 0;
    if (INLINE) {
        uint32_t fp0 = 0;
        tp0 = 0;
        // (one chunk per step: most chunks have a handful of predecessors, and a
        // step of eight with clamped loads was 2.4 times slower at Config 2)
        for (int32_t c = a.cat_chunk_off[ci.k]; c < ci.c; c++) {
            const int64_t oc = ((int64_t)c * a.n_words + ci.word) * WAVE + lane;
            tp0 += a.cnt_tp[oc];
            fp0 += a.cnt_fp[oc];
        }
        pre_t = tp0;
        pre_f = fp0;
        n0 = tp0 + fp0;
    } else {
        tp0 = a.pre_tp[o];
        n0 = tp0 + a.pre_fp[o];
    }
    uint64_t best = PR_ZERO;
    const int64_t tb = (((int64_t)ci.c * a.n_words + ci.word) * ACC_BLK) * WAVE + lane;
    // (blocks past the chunk's rows hold zero words: acc_count_kernel wrote them)
    uint64_t Tb[ACC_BLK], TFb[ACC_BLK];
#pragma unroll
    for (int blk = 0; blk < ACC_BLK; blk++) {
        Tb[blk] = a.t_tp[tb + (int64_t)blk * WAVE];
        TFb[blk] = Tb[blk] | a.t_fp[tb + (int64_t)blk * WAVE];
    }
#pragma unroll
    for (int blk = 0; blk < ACC_BLK; blk++) {
        best_of_block(Tb[blk], TFb[blk], tp0, n0, best);
    }
    a.cmax[o] = best;
    if (INLINE) {
        a.pre_tp[o] = pre_t;
        a.pre_fp[o] = pre_f;
    }
    if (INLINE && ci.last) {                        // recall of the whole category
        const int combo = ci.word * WAVE + lane;
        if (combo < a.n_rng * N_THR) {
            const int r = combo / N_THR, t = combo - r * N_THR;
            const int64_t kr = (int64_t)ci.k * a.n_rng + r;
            const int32_t ng = a.num_gt[kr];
            if (ng > 0) a.rec[kr * N_THR + t] = (double)tp0 / (double)ng;
        }
    }
    if (INLINE) {
        // (last: stores, and nothing here is read back in this kernel)
        // the recall crossings (acc_prefix_kernel's table) depend on num_gt
        // only: every wavefront of the launch tabulates an equal slice of the
        // whole table, whatever category its chunk belongs to (tabulating a
        // category's crossings in each of its emission wavefronts cost 33 us,
        // in the wavefront of its first chunk 15 us: two fp64 divisions each)
        const int64_t total = (int64_t)a.n_cat * a.n_rng * N_REC;
        const int64_t items = (int64_t)a.cat_chunk_off[a.n_cat] * a.n_words;
        const int64_t per = (total + items - 1) / items;
        const int64_t lo = ((int64_t)ci.c * a.n_words + ci.word) * per;
        const int64_t hi = min(total, lo + per);
        for (int64_t i = lo + lane; i < hi; i += WAVE) {
            const int64_t kr = i / N_REC;
            const int32_t ng = a.num_gt[kr];
            if (ng > 0) a.cj[i] = recall_crossing(rec.v[i - kr * N_REC], ng);
        }
    }
}

// reverse exclusive maximum over the chunks of a category; one workgroup per
// (category, word), quarters stitched like acc_prefix_kernel
__global__ __launch_bounds__(256) void acc_sufmax_kernel(AccArgs a)
{
    __shared__ uint64_t s_max[4][WAVE];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t item = blockIdx.x;
    const int32_t k = a.k_begin + (int32_t)(item / a.n_words);
    const int word = (int)(item % a.n_words);
    const int lane = lane_id();
    const int32_t c0 = a.cat_chunk_off[k], c1 = a.cat_chunk_off[k + 1];
    const int32_t q = (c1 - c0 + 3) / 4;
    const int32_t lo = min(c1, c0 + wave * q), hi = min(c1, lo + q);
    uint64_t run = PR_ZERO;
    for (int32_t c = hi - 1; c >= lo; c -= 8) {           // (eight at a time, as above)
        uint64_t v[8];
#pragma unroll
        for (int u = 0; u < 8; u++)
            v[u] = a.cmax[((int64_t)max(c - u, lo) * a.n_words + word) * WAVE + lane];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (c - u >= lo) {
                a.cmax[((int64_t)(c - u) * a.n_words + word) * WAVE + lane] = run;
                if (pr_better((uint32_t)(v[u] >> 32), (uint32_t)v[u], run)) run = v[u];
            }
        }
    }
    s_max[wave][lane] = run;
    __syncthreads();
    uint64_t later = PR_ZERO;                      // maximum of the later quarters
    for (int w = 3; w > wave; w--) {
        const uint64_t v = s_max[w][lane];
        if (pr_better((uint32_t)(v >> 32), (uint32_t)v, later)) later = v;
    }
    if (wave < 3 && __ballot(later != PR_ZERO) != 0)
        for (int32_t c = lo; c < hi; c++) {
            const int64_t o = ((int64_t)c * a.n_words + word) * WAVE + lane;
            const uint64_t v = a.cmax[o];
            if (pr_better((uint32_t)(later >> 32), (uint32_t)later, v)) a.cmax[o] = later;
        }
}

// Emission burst: thresholds jcur-1, jcur-2, ... whose crossing count cj[.]
// satisfies the condition (GT: cj > x, otherwise cj == x) take the value v.
// cj is non-decreasing, so these are consecutive; the caller has checked the
// first (cnext = cj[jcur-1]).  W = 4 (the fused sweep of short categories):
// four candidates are fetched per step -- a track level category crosses ten
// thresholds at one TP row, and a step per threshold is a chain of dependent
// LDS reads (acc_fused_kernel at Config 2's track level: 100 -> 52 us with the
// smaller workgroups).  The chunked sweep of long categories (bursts of one or
// two) is faster with the plain loop, W = 1 (85 vs 99 us).
template <bool GT, int W>
__device__ __forceinline__ void emit_burst(uint64_t *__restrict__ out,
                                           const int32_t *__restrict__ cj, int &jcur,
                                           int32_t &cnext, int32_t x, uint64_t v)
{
    auto ok = [&](int32_t y) { return GT ? y > x : y == x; };
    if (W == 1) {                     // long categories: a threshold or two per TP row
        do {
            out[--jcur] = v;
            cnext = jcur > 0 ? cj[jcur - 1] : -1;
        } while (ok(cnext));
        return;
    }
    do {
        int32_t c[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int idx = jcur - 2 - k;
            c[k] = idx >= 0 ? cj[idx] : -1;
        }
        out[jcur - 1] = v;
        int n = 1;
        if (ok(c[0])) {
            out[jcur - 2] = v;
            n = 2;
            if (ok(c[1])) {
                out[jcur - 3] = v;
                n = 3;
                if (ok(c[2])) {
                    out[jcur - 4] = v;
                    n = 4;
                }
            }
        }
        cnext = c[n - 1];
        jcur -= n;
    } while (ok(cnext));
}

// backward: the emission sweep of one block.  tp / n = counts at the END of
// the block on entry, at its start on return; run = envelope of everything
// behind.  Thresholds reached at a skipped row (inside a run) take the
// envelope of the rows above it, i.e. the running value BEFORE the next
// visited row; thresholds reached below the lowest visited row of the block
// see the envelope after it.
template <int W>
__device__ __forceinline__ void emit_block(uint64_t T, uint64_t TF, uint32_t &tp,
                                           uint32_t &n, uint64_t &run, int &jcur,
                                           int32_t &cnext, uint64_t *__restrict__ out,
                                           const int32_t *__restrict__ cj)
{
    const uint32_t Th[2] = {(uint32_t)T, (uint32_t)(T >> 32)};
    const uint32_t TFh[2] = {(uint32_t)TF, (uint32_t)(TF >> 32)};
    const uint32_t mh[2] = {Th[0] & ~((Th[0] >> 1) | (Th[1] << 31)), Th[1] & ~(Th[1] >> 1)};
    if (W == 1) {
        // The walk of the long categories' sweeps, bound by VALU issue (round 5:
        // 24 -> 17 VALU instructions per visited row).  The envelope lives in
        // two registers; a row's counts come from the counts BEFORE the half
        // plus a popcount (v_bcnt with its addend); `run` only ever holds a
        // LATER row, whose n is larger, so on equal rationals it stays
        // (pr_better's tie rule can never favour the earlier row) and the
        // comparison is one 64-bit compare of the cross products.
        uint32_t rt = (uint32_t)(run >> 32), rn = (uint32_t)run;
#pragma unroll
        for (int h = 1; h >= 0; h--) {
            const uint32_t tps = tp - (uint32_t)__popc(Th[h]);     // counts before the half
            const uint32_t ns = n - (uint32_t)__popc(TFh[h]);
            for (uint32_t m = mh[h]; m != 0;) {
                const uint32_t le = 0xffffffffu >> __builtin_clz(m);       // rows <= q
                const uint32_t tpq = tps + (uint32_t)__popc(Th[h] & le);   // incl. row q
                const uint32_t nq = ns + (uint32_t)__popc(TFh[h] & le);
                const bool better = (uint64_t)tpq * rn > (uint64_t)rt * nq;
                if (cnext >= (int32_t)tpq) {
                    // thresholds reached above row q (crossing count > tpq) take
                    // the envelope before this row, those reached exactly at it
                    // (== tpq) the envelope including it
                    const uint64_t above = pr_pack(rt, rn);
                    const uint64_t at = better ? pr_pack(tpq, nq) : above;
                    do {
                        out[--jcur] = cnext > (int32_t)tpq ? above : at;
                        cnext = jcur > 0 ? cj[jcur - 1] : -1;
                    } while (cnext >= (int32_t)tpq);
                }
                rt = better ? tpq : rt;
                rn = better ? nq : rn;
                m &= le >> 1;
            }
            tp = tps;
            n = ns;
        }
        run = pr_pack(rt, rn);
        if (cnext > (int32_t)tp) emit_burst<true, W>(out, cj, jcur, cnext, (int32_t)tp, run);
        return;
    }
#pragma unroll
    for (int h = 1; h >= 0; h--) {
        for (uint32_t m = mh[h]; m != 0;) {
            const int q = 31 - __builtin_clz(m);
            const uint32_t gt = 0xfffffffeu << q;              // rows > q (q = 31: none)
            const uint32_t tpq = tp - (uint32_t)__popc(Th[h] & gt);    // incl. row q
            const uint32_t nq = n - (uint32_t)__popc(TFh[h] & gt);
            if (cnext > (int32_t)tpq)          // reached above row q
                emit_burst<true, W>(out, cj, jcur, cnext, (int32_t)tpq, run);
            if (pr_better(tpq, nq, run)) run = pr_pack(tpq, nq);
            if (cnext == (int32_t)tpq)         // reached exactly at row q
                emit_burst<false, W>(out, cj, jcur, cnext, (int32_t)tpq, run);
            m &= ~(1u << q);
        }
        tp -= (uint32_t)__popc(Th[h]);
        n -= (uint32_t)__popc(TFh[h]);
    }
    if (cnext > (int32_t)tp) emit_burst<true, W>(out, cj, jcur, cnext, (int32_t)tp, run);
}

#define EMIT_RMAX 8   // ranges that can overlap one 64-combo word

// INLINE (short categories): the envelope of the later chunks is gathered
// here instead of by acc_sufmax_kernel.
template <bool INLINE>
__global__ __launch_bounds__(256) void acc_emit_kernel(AccArgs a)
{
    __shared__ int32_t s_cj[4][EMIT_RMAX][N_REC];
    const ChunkInfo ci = chunk_info(a);
    if (!ci.valid) return;
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int combo = ci.word * WAVE + lane;
    const bool active = combo < a.n_rng * N_THR;
    const int r = active ? combo / N_THR : 0;
    const int t = active ? combo - r * N_THR : 0;
    const int r_lo = (ci.word * WAVE) / N_THR;
    const int r_hi = min(a.n_rng - 1, (ci.word * WAVE + WAVE - 1) / N_THR);
    // Every global load of the set-up in one batch, ahead of the first wait
    // (the trace of this kernel showed the set-up as five memory round trips in
    // a row -- crossings to LDS, counts, transposed words, the later chunks'
    // maxima -- 11 us of a 23 us wavefront; the crossings of a row without
    // evaluated ground truth were never written: loaded anyway, not used).
    int32_t cjv[EMIT_RMAX][2];
    int32_t ngq[EMIT_RMAX];
#pragma unroll
    for (int q = 0; q < EMIT_RMAX; q++) {
        const int rq = min(r_lo + q, r_hi);
        const int64_t base = ((int64_t)ci.k * a.n_rng + rq) * N_REC;
        ngq[q] = a.num_gt[(int64_t)ci.k * a.n_rng + rq];
        cjv[q][0] = a.cj[base + lane];
        cjv[q][1] = a.cj[base + min(lane + WAVE, N_REC - 1)];
    }
    const int32_t ng = active ? a.num_gt[(int64_t)ci.k * a.n_rng + r] : 0;
    const bool live = active && ng > 0;
    const int64_t o = ((int64_t)ci.c * a.n_words + ci.word) * WAVE + lane;
    uint32_t tp = a.pre_tp[o] + a.cnt_tp[o];
    uint32_t n = tp + a.pre_fp[o] + a.cnt_fp[o];
    // (blocks past the chunk's rows hold zero words)
    const int64_t tb = (((int64_t)ci.c * a.n_words + ci.word) * ACC_BLK) * WAVE + lane;
    uint64_t Tb[ACC_BLK], TFb[ACC_BLK];
#pragma unroll
    for (int blk = 0; blk < ACC_BLK; blk++) {
        const uint64_t t_ = a.t_tp[tb + (int64_t)blk * WAVE];
        const uint64_t f_ = a.t_fp[tb + (int64_t)blk * WAVE];
        Tb[blk] = live ? t_ : 0;
        TFb[blk] = live ? (t_ | f_) : 0;
    }
    uint64_t run;
    if (INLINE) {
        // four later chunks per step: their loads are in flight together
        run = PR_ZERO;
        const int32_t c_end = a.cat_chunk_off[ci.k + 1];
        for (int32_t c = ci.c + 1; c < c_end; c += 4) {
            uint64_t v[4];
#pragma unroll
            for (int u = 0; u < 4; u++)
                v[u] = a.cmax[((int64_t)min(c + u, c_end - 1) * a.n_words + ci.word) * WAVE + lane];
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (pr_better((uint32_t)(v[u] >> 32), (uint32_t)v[u], run)) run = v[u];
        }
    } else {
        run = a.cmax[o];
    }
#pragma unroll
    for (int q = 0; q < EMIT_RMAX; q++) {
        if (r_lo + q <= r_hi) {
            const bool has = ngq[q] > 0;
            s_cj[wave][q][lane] = has ? cjv[q][0] : 0;
            if (lane + WAVE < N_REC) s_cj[wave][q][lane + WAVE] = has ? cjv[q][1] : 0;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    const int32_t *__restrict__ cj = s_cj[wave][active ? r - r_lo : 0];
    uint64_t *__restrict__ out =
        a.val + (((int64_t)ci.k * a.n_rng + r) * N_THR + t) * N_REC;
    // thresholds already reached by the TP count at the end of this chunk:
    // jcur = #{j : cj[j] <= tp} (cj is non-decreasing in j)
    int jcur = 0;
    if (live) {
        int lo = 0, hi = N_REC;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cj[mid] <= (int32_t)tp) lo = mid + 1; else hi = mid;
        }
        jcur = lo;
    }
    if (ci.last) {
        // thresholds the category never reaches get precision 0.  The tail of
        // a lane's row is contiguous: the wavefront writes the tails row by
        // row with coalesced stores instead of every lane walking its own row
        // (64 scattered 8-byte stores per step).
        // (a category without detections: precision 0 everywhere it has
        // evaluated GT, reference lvis_amodal/eval.py:412-417)
        const int jz = live ? (ci.len == 0 ? 0 : jcur) : N_REC;
        for (uint64_t need = __ballot(jz < N_REC); need != 0; need &= need - 1) {
            const int l = __builtin_ctzll(need);
            const int jl = __builtin_amdgcn_readlane(jz, l);
            uint64_t *__restrict__ row = (uint64_t *)readlane_u64((uint64_t)out, l);
            for (int j = jl + lane; j < N_REC; j += WAVE) row[j] = PR_ZERO;
        }
    }
    // next threshold to write, cached in a register: cj[jcur - 1], or -1
    int32_t cnext = jcur > 0 ? cj[jcur - 1] : -1;
#pragma unroll
    for (int blk = ACC_BLK - 1; blk >= 0; blk--) {
        if (blk * WAVE >= ci.len) continue;
        emit_block<1>(Tb[blk], TFb[blk], tp, n, run, jcur, cnext, out, cj);
    }
    if (live && ci.first && jcur > 0 && ci.len > 0) {
        const uint64_t v = run;
        while (jcur > 0) {
            out[jcur - 1] = v;
            jcur--;
        }
    }
}

// ---------------------------------------------------------------------------
// Fused sweep for problems whose categories all fit one workgroup
// (chunks x words <= ACC_FUSED_WAVES wavefronts, i.e. <= 4096 rows at one
// combo word, <= 1024 rows at four): ONE kernel, one workgroup per category,
// wavefront = (chunk, word).  Counts, prefixes, chunk maxima and the recall
// crossings live in LDS, the transposed TP / FP words of a wavefront's 256
// rows stay in its registers from the counting pass to the emission pass --
// none of the intermediates of the general path (cnt, pre, cmax, t_tp, t_fp,
// cj: ~100 MB of traffic and six more launches at Config 2) exists.
// The arithmetic is the general path's, statement by statement.
#define ACC_FUSED_WAVES 16       // the largest workgroup; 4 and 8 serve shorter categories

template <int FW>   // wavefronts of the workgroup
__global__ __launch_bounds__(FW * WAVE) void acc_fused_kernel(AccArgs a, RecThr rec)
{
    __shared__ uint32_t s_tp[FW][WAVE], s_fp[FW][WAVE];
    __shared__ uint64_t s_max[FW][WAVE];
    __shared__ int32_t s_cj[32 * N_REC];            // [n_rng][N_REC]
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int32_t k = a.k_begin + (int32_t)blockIdx.x;
    const int32_t sb = a.cat_off[k], se = a.cat_off[k + 1];
    // another size class, or a long category (the chunked kernels' job)
    if (se - sb > a.fused_rows || se - sb <= a.fused_lo) return;
    const int nw = a.n_words;
    const int nch = (se - sb + ACC_CH - 1) / ACC_CH;
    const int j = wave / nw, word = wave - j * nw;   // my chunk, my combo word
    const bool mine = j < nch;                        // wave-uniform
    const int32_t start = sb + j * ACC_CH;
    const int len = mine ? min(ACC_CH, se - start) : 0;
    // ---- recall crossings of the category's ranges
    for (int i = threadIdx.x; i < a.n_rng * N_REC; i += FW * WAVE) {
        const int q = i / N_REC;
        const int32_t ngq = a.num_gt[(int64_t)k * a.n_rng + q];
        s_cj[i] = ngq > 0 ? recall_crossing(rec.v[i - q * N_REC], ngq) : 0;
    }
    // ---- counting pass: transposed words of my rows -> registers
    uint64_t T[ACC_BLK], TF[ACC_BLK];
    uint32_t tp_own = 0, fp_own = 0;
    {
        uint64_t tpw[ACC_BLK], fpw[ACC_BLK];
        load_chunk(a, (int64_t)start, word, len, lane, tpw, fpw);
#pragma unroll
        for (int blk = 0; blk < ACC_BLK; blk++) {
            const int base = blk * WAVE;
            uint64_t t_ = 0, f_ = 0;
            if (base < len) {
                t_ = transpose64(tpw[blk], lane);
                f_ = transpose64(fpw[blk], lane);
            }
            T[blk] = t_;
            TF[blk] = t_ | f_;
            tp_own += (uint32_t)__popcll(t_);
            fp_own += (uint32_t)__popcll(f_);
        }
    }
    s_tp[wave][lane] = tp_own;
    s_fp[wave][lane] = fp_own;
    __syncthreads();
    // ---- prefix over the earlier chunks of my word; recall; empty category
    uint32_t tp0 = 0, fp0 = 0;
    for (int jj = 0; jj < j && jj < nch; jj++) {
        tp0 += s_tp[jj * nw + word][lane];
        fp0 += s_fp[jj * nw + word][lane];
    }
    const int combo = word * WAVE + lane;
    const bool active = combo < a.n_rng * N_THR;
    const int r = active ? combo / N_THR : 0;
    const int t = active ? combo - r * N_THR : 0;
    const int32_t ng = active ? a.num_gt[(int64_t)k * a.n_rng + r] : 0;
    const bool live = active && ng > 0;
    const int64_t kr = (int64_t)k * a.n_rng + r;
    uint64_t *__restrict__ out = a.val + (kr * N_THR + t) * N_REC;
    const bool last = mine && j == nch - 1, first = mine && j == 0;
    if (live && (last || (nch == 0 && j == 0 && wave < nw))) {
        a.rec[kr * N_THR + t] = (double)(tp0 + tp_own) / (double)ng;
        // a category without detections still has precision 0 / recall 0
        // where it has evaluated GT (reference lvis_amodal/eval.py:412-417)
        if (nch == 0)
            for (int jj = 0; jj < N_REC; jj++) out[jj] = PR_ZERO;
    }
    // ---- largest precision at a TP row of my chunk (see acc_chunkmax_kernel)
    uint64_t best = PR_ZERO;
    {
        uint32_t tpb = tp0, nb = tp0 + fp0;
#pragma unroll
        for (int blk = 0; blk < ACC_BLK; blk++) {
            best_of_block(T[blk], TF[blk], tpb, nb, best);
        }
    }
    s_max[wave][lane] = best;
    __syncthreads();
    if (!mine) return;
    // ---- envelope of the later chunks, then the emission sweep of mine
    // (acc_emit_kernel)
    uint64_t run = PR_ZERO;
    for (int jj = nch - 1; jj > j; jj--) {
        const uint64_t v = s_max[jj * nw + word][lane];
        if (pr_better((uint32_t)(v >> 32), (uint32_t)v, run)) run = v;
    }
    const int32_t *__restrict__ cj = s_cj + r * N_REC;
    uint32_t tp = tp0 + tp_own;
    uint32_t n = tp + fp0 + fp_own;
    int jcur = 0;
    if (live) {
        int lo = 0, hi = N_REC;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cj[mid] <= (int32_t)tp) lo = mid + 1; else hi = mid;
        }
        jcur = lo;
    }
    if (last) {
        const int jz = live ? jcur : N_REC;
        for (int l = 0; l < WAVE; l++) {
            const int jl = __builtin_amdgcn_readlane(jz, l);
            if (jl >= N_REC) continue;
            const int cl = word * WAVE + l;
            const int rl = cl / N_THR, tl = cl - rl * N_THR;
            uint64_t *__restrict__ row =
                a.val + (((int64_t)k * a.n_rng + rl) * N_THR + tl) * N_REC;
            for (int jj = jl + lane; jj < N_REC; jj += WAVE) row[jj] = PR_ZERO;
        }
    }
    int32_t cnext = jcur > 0 ? cj[jcur - 1] : -1;
#pragma unroll
    for (int blk = ACC_BLK - 1; blk >= 0; blk--) {
        if (blk * WAVE >= len) continue;
        emit_block<4>(live ? T[blk] : 0, live ? TF[blk] : 0, tp, n, run, jcur, cnext, out, cj);
    }
    if (live && first && jcur > 0) {
        const uint64_t v = run;
        while (jcur > 0) {
            out[jcur - 1] = v;
            jcur--;
        }
    }
}

// ===========================================================================
// One-pass sweep of long categories (round 4).
//
// The chunked kernels above read the rows three times (count, chunk maxima,
// emission: the transposed copy written once and read twice) in six dependent
// launches.  Here ONE kernel reads every row once:
//
//   super-chunk (SC) = SW consecutive chunks of a category = one workgroup of
//   SW wavefronts, wavefront = chunk; the transposed words of a chunk never
//   leave the wavefront's registers.
//
//   forward dependency (TP / FP counts of everything before the chunk): inside
//   the workgroup through LDS, between the SCs of a category by a decoupled
//   look-back over 8-byte status words {generation | flag | count}: an SC
//   publishes its own totals (flag AGG) as soon as its rows are counted, then
//   the prefix that includes it (flag PRE); a later SC adds up the words it
//   finds until it meets a PRE.  An SC's number is its workgroup number (a
//   ticket atomic ahead of the first load was measured slower): an SC waits
//   for lower-numbered workgroups, which the dispatcher starts first on an idle
//   GPU.  That is a matter of speed, not of correctness: a wait that exceeds
//   the poll limit flags the pass (taoamd_accumulate_error) and the caller
//   sweeps it again with the chunked kernels (taoamd_accumulate_chunked).
//
//   backward dependency (the precision envelope of everything AFTER a row)
//   cannot be waited for -- the later SCs may not have started.  It is not
//   needed while walking: the envelope is a maximum, so a chunk emits every
//   threshold it reaches with the envelope of ITS OWN later rows, and the
//   maxima of what lies behind are folded in afterwards, which cannot change
//   a bit of the result (max of the same set of (tp, n) pairs under the total
//   order of pr_better):
//     * later chunks of the same SC: after a barrier, from LDS, by the
//       wavefront that wrote the records (it re-reads its own stores);
//     * later SCs: acc_raise_kernel, a wavefront per (category, range,
//       threshold) row, from two small per-SC tables: sc_max (largest
//       precision record of the SC) and sc_jhi (recall thresholds reached up
//       to its end).
//
// The walk itself (emit_block) is the chunked path's, statement by statement.
// ===========================================================================
// -DSW_ABLATE=<bits> (timing experiments, results wrong): 1 no walk, 2 no
// transposes, 4 no look-back, 8 leave behind the rows' counts, 16 leave behind
// the counts of the earlier chunks.  Round 6, 21.4 M rows, one box: whole kernel
// 0.260 ms; rows fetched and counted without transposes 0.069 (5.0 TB/s), with
// them 0.103; up to the earlier chunks' counts 0.113; everything but the walk
// 0.160 (0.159 without the transposes too: behind the first barrier they are
// hidden); without the look-back 0.245.  So the walk is 0.10 ms of the kernel
// and the bookkeeping between the counts and the walk's end 0.05.
#define SC_SPIN_LIMIT (1 << 18)     // a fraction of a second of polling
#define SWEEP_NB 8                   // blocks of 64 rows per wavefront of the one-pass sweep
#ifndef SWEEP_SW
#define SWEEP_SW 4                   // wavefronts (chunks) per super-chunk: 16 and 8 were
                                     // measured slower (big workgroups hold their CU's wave
                                     // slots until the slowest wavefront is done); round 5,
                                     // -DSWEEP_SW=2 / 3 / 6 at 21 M rows: 0.287 / 0.270 /
                                     // 0.297 against 0.264 ms
#endif
#define SC_FLAG_AGG 1ull
#define SC_FLAG_PRE 2ull
__device__ __forceinline__ uint64_t sc_status(uint32_t gen, uint64_t flag, uint32_t v)
{
    return ((uint64_t)gen << 34) | (flag << 32) | v;
}
__device__ __forceinline__ void sc_publish(uint64_t *p, uint64_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t sc_peek(const uint64_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// records [jlo, jhi) of `out` raised to `later` where it is the better pair
__device__ __forceinline__ void sc_raise(uint64_t *__restrict__ out, int jlo, int jhi,
                                         uint64_t later)
{
    const uint32_t lt = (uint32_t)(later >> 32), ln = (uint32_t)later;
    for (int j = jlo; j < jhi; j++) {
        const uint64_t v = out[j];
        if (pr_better(lt, ln, v)) out[j] = later;
    }
}

// cj[k][r][j]: smallest TP count c with fl(c / num_gt) >= rec_thrs[j]
// (np.searchsorted(rc, rec_thrs, side="left") on rc = tp / num_gt, reference
// lvis_amodal/eval.py:386,406-408); 0 where the range has no ground truth
__global__ void acc_cj_kernel(AccArgs a, RecThr rec)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)a.n_cat * a.n_rng * N_REC) return;
    const int64_t kr = i / N_REC;
    const int32_t ng = a.num_gt[kr];
    a.cj[i] = ng > 0 ? recall_crossing(rec.v[i - kr * N_REC], ng) : 0;
}

// MODE 0: look-back (one pass); MODE 1: the SC totals come from
// acc_sccount_kernel (two passes over the rows, no flags); MODE 2: that
// counting pass itself (rows -> SC totals, nothing else)
// NB = 64-row blocks of a wavefront's chunk (8: 512 rows; the chunked kernels'
// (NB * WAVE) = 256 rows would be 4 -- 512 is 0.281 against 0.297 ms at 21 M rows:
// half the wavefronts, half the per-wavefront set-up, still 64 VGPRs)
template <int SW, int MODE, int NB = SWEEP_NB>
__global__ __launch_bounds__(SW * WAVE) __attribute__((amdgpu_waves_per_eu(8, 8)))
void acc_sweep_kernel(AccArgs a, RecThr rec)
{
    __shared__ uint32_t s_tp[SW][WAVE], s_fp[SW][WAVE];
    __shared__ uint64_t s_max[SW][WAVE];
    __shared__ uint32_t s_pre[2][WAVE];
    __shared__ int32_t s_cj[EMIT_RMAX][N_REC];
    __shared__ uint16_t s_jr[SW][WAVE];          // thresholds a chunk emitted: jlo | jhi << 8
    __shared__ int32_t s_done;
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x == 0) s_done = 0;
    // SC = workgroup number.  A workgroup only ever waits for LOWER-numbered
    // ones, which the dispatcher has started before it (workgroups leave the
    // queue of their XCD in order); should that ever not hold the wait below
    // gives up after SC_SPIN_LIMIT polls and flags the pass instead of hanging.
    // XCD-aware order (round 5): workgroup b runs on XCD b % 8; XCD x takes the
    // x-th of eight runs of consecutive SCs that START AT CATEGORY BOUNDARIES
    // (xcd_start, acc_chunks_kernel), in order.  A category's SCs then sit
    // behind ONE L2: the 8-byte records its chunks store into the category's
    // rows of `val` meet there and leave for HBM as whole lines (WRITE_SIZE
    // 162 -> 120 MB, the kernel's traffic 675 -> 523 MB at 21 M rows), and an
    // SC still only ever waits for workgroups its own XCD started before it.
    int32_t item = (int32_t)blockIdx.x;
    if (a.xcd_start != nullptr) {
        const int32_t x = item % (int32_t)N_XCD, i = item / (int32_t)N_XCD;
        item = a.xcd_start[x] * a.n_words + i;
        if (item >= a.xcd_start[x + 1] * a.n_words) return;
    }
    const int32_t sc = item / a.n_words;
    const int word = item - sc * a.n_words;
    if (sc >= a.cat_chunk_off[a.n_cat]) return;
    const int32_t k = a.chunk_tab ? a.chunk_tab[sc] : chunk_cat(a.cat_chunk_off, a.n_cat, sc);
    const int32_t jsc = sc - a.cat_chunk_off[k];
    const int64_t cat_begin = a.cat_off[k], cat_end = a.cat_off[k + 1];
    const int64_t start = cat_begin + ((int64_t)jsc * SW + wave) * (NB * WAVE);
    const int len = (int)max((int64_t)0, min((int64_t)(NB * WAVE), cat_end - start));
    const bool swept = k >= a.k_begin && k < a.k_end;       // (uniform)
    // ---- rows of my chunk: every load ahead of anything else
    uint64_t tpw[NB], vw[NB];
    load_chunk_tv(a, start, word, len, lane, tpw, vw);
    const int r_lo = (word * WAVE) / N_THR;
    const int r_hi = min(a.n_rng - 1, (word * WAVE + WAVE - 1) / N_THR);
    if (MODE != 2) {
        // recall crossings of the ranges this word overlaps: tabulated once per
        // pass by acc_cj_kernel (they depend on num_gt alone; every workgroup
        // working them out for itself was a tenth of this kernel's time)
        const int32_t *__restrict__ src = a.cj + ((int64_t)k * a.n_rng + r_lo) * N_REC;
        for (int i = threadIdx.x; i < (r_hi - r_lo + 1) * N_REC; i += SW * WAVE)
            (&s_cj[0][0])[i] = src[i];
    }
    uint64_t T[NB], TF[NB];
    uint32_t tp_own = 0, n_own = 0;
#pragma unroll
    for (int blk = 0; blk < NB; blk++) {
        uint64_t t_ = 0, v_ = 0;
        if (blk * WAVE < len) {
#if defined(SW_ABLATE) && (SW_ABLATE & 2)     // (timing experiment: no transposes)
            t_ = tpw[blk];
            v_ = vw[blk];
#else
            t_ = transpose64(tpw[blk], lane);
            v_ = transpose64(vw[blk], lane);
#endif
        }
        T[blk] = t_;
        TF[blk] = v_;
        tp_own += (uint32_t)__popcll(t_);
        n_own += (uint32_t)__popcll(v_);
    }
    const uint32_t fp_own = n_own - tp_own;       // (TP and FP rows are disjoint)
#if defined(SW_ABLATE) && (SW_ABLATE & 8)     // (timing experiment: rows fetched and counted, nothing else)
    if (tp_own + n_own == (uint32_t)a.sc_spin) a.sc_max[0] = n_own;
    return;
#endif
    s_tp[wave][lane] = tp_own;
    s_fp[wave][lane] = fp_own;
    __syncthreads();
    uint32_t tp0 = 0, fp0 = 0;
    for (int w = 0; w < wave; w++) {
        tp0 += s_tp[w][lane];
        fp0 += s_fp[w][lane];
    }
    const int64_t so = ((int64_t)sc * a.n_words + word) * WAVE + lane;
    if (MODE == 2) {
        if (wave == SW - 1) {
            a.cnt_tp[so] = tp0 + tp_own;
            a.cnt_fp[so] = fp0 + fp_own;
        }
        return;
    }
    // ---- counts of the category's earlier SCs
    if (wave == SW - 1) {
        const uint32_t tot_t = tp0 + tp_own, tot_f = fp0 + fp_own;
        uint32_t acc_t = 0, acc_f = 0;
        if (MODE == 0) {
            uint64_t *st = a.sc_stat + 2 * so;              // {tp word, fp word} per lane
            const uint64_t mine = jsc == 0 ? SC_FLAG_PRE : SC_FLAG_AGG;
            sc_publish(st, sc_status(a.sc_gen, mine, tot_t));
            sc_publish(st + 1, sc_status(a.sc_gen, mine, tot_f));
            // look back over the category's earlier SCs until one carries its
            // prefix.  A pair is usable when both words are of this call and of
            // one kind (the publisher replaces AGG by PRE word by word)
            bool open = jsc > 0;
#if defined(SW_ABLATE) && (SW_ABLATE & 4)     // (timing experiment: no look-back)
            open = false;
#endif
            const int64_t stride = 2 * (int64_t)a.n_words * WAVE;
            const uint64_t *p = st - stride;
            for (int32_t jj = jsc - 1; jj >= 0 && __ballot(open) != 0; jj--, p -= stride) {
                uint64_t vt, vf;
                for (int spin = 0;; spin++) {
                    vt = sc_peek(p);
                    vf = sc_peek(p + 1);
                    const uint32_t ft = (uint32_t)(vt >> 32) & 3, ff = (uint32_t)(vf >> 32) & 3;
                    const bool ready = (uint32_t)(vt >> 34) == a.sc_gen &&
                                       (uint32_t)(vf >> 34) == a.sc_gen && ft != 0 && ft == ff;
                    if (a.sc_spin >= 0 && __ballot(open && !ready) == 0) break;
                    if (spin >= a.sc_spin) {
                        if (lane == 0) {
                            atomicOr(a.sc_error, 1u);
                            if (a.sc_giveups) atomicAdd(a.sc_giveups, 1u);
                        }
                        vt = vf = 0;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
                if (open) {
                    acc_t += (uint32_t)vt;
                    acc_f += (uint32_t)vf;
                    if (((vt >> 32) & 3) == SC_FLAG_PRE) open = false;
                }
            }
            if (jsc > 0) {
                sc_publish(st, sc_status(a.sc_gen, SC_FLAG_PRE, acc_t + tot_t));
                sc_publish(st + 1, sc_status(a.sc_gen, SC_FLAG_PRE, acc_f + tot_f));
            }
        } else {
            for (int32_t jj = 0; jj < jsc; jj++) {
                const int64_t o = ((int64_t)(sc - jsc + jj) * a.n_words + word) * WAVE + lane;
                acc_t += a.cnt_tp[o];
                acc_f += a.cnt_fp[o];
            }
        }
        s_pre[0][lane] = acc_t;
        s_pre[1][lane] = acc_f;
    }
    __syncthreads();
    tp0 += s_pre[0][lane];
    fp0 += s_pre[1][lane];
#if defined(SW_ABLATE) && (SW_ABLATE & 16)    // (timing experiment: ... and the counts before the chunk)
    if (tp0 + fp0 == (uint32_t)a.sc_spin) a.sc_max[0] = T[0] ^ TF[NB - 1];
    return;
#endif
    // ---- my lane's combo
    const int combo = word * WAVE + lane;
    const bool active = combo < a.n_rng * N_THR;
    const int r = active ? combo / N_THR : 0;
    const int t = active ? combo - r * N_THR : 0;
    const int64_t kr = (int64_t)k * a.n_rng + r;
    const int32_t ng = active ? a.num_gt[kr] : 0;
    const bool live = active && ng > 0 && swept;
    const int32_t *__restrict__ cj = s_cj[active ? r - r_lo : 0];
    uint64_t *__restrict__ out = a.val + (kr * N_THR + t) * N_REC;
    uint32_t tp = tp0 + tp_own;
    uint32_t n = tp + fp0 + fp_own;
    // the chunk that holds the category's last row (a category without rows:
    // the first wavefront of its one SC)
    const bool is_last = len > 0 ? start + len == cat_end : (cat_end == cat_begin && wave == 0);
    const bool is_first = jsc == 0 && wave == 0;
    int jcur = 0;
    if (live) {
        int lo = 0, hi = N_REC;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cj[mid] <= (int32_t)tp) lo = mid + 1; else hi = mid;
        }
        jcur = lo;
    }
    const int jhi = jcur;                      // thresholds reached up to my chunk's end
    if (is_last) {
        if (live) a.rec[kr * N_THR + t] = (double)tp / (double)ng;
        // thresholds the category never reaches: precision 0 (acc_emit_kernel)
        const int jz = live ? (len == 0 ? 0 : jcur) : N_REC;
        for (uint64_t need = __ballot(jz < N_REC); need != 0; need &= need - 1) {
            const int l = __builtin_ctzll(need);
            const int jl = __builtin_amdgcn_readlane(jz, l);
            uint64_t *__restrict__ row = (uint64_t *)readlane_u64((uint64_t)out, l);
            for (int j = jl + lane; j < N_REC; j += WAVE) row[j] = PR_ZERO;
        }
    }
    uint64_t run = PR_ZERO;
    int32_t cnext = jcur > 0 ? cj[jcur - 1] : -1;
#pragma unroll
    for (int blk = NB - 1; blk >= 0; blk--) {
        if (blk * WAVE >= len) continue;
#if defined(SW_ABLATE) && (SW_ABLATE & 1)     // (timing experiment: no walk)
        run ^= T[blk] + TF[blk];
#else
        emit_block<1>(live ? T[blk] : 0, live ? TF[blk] : 0, tp, n, run, jcur, cnext, out, cj);
#endif
    }
    if (live && is_first && jcur > 0 && len > 0) {
        const uint64_t v = run;
        while (jcur > 0) {
            out[jcur - 1] = v;
            jcur--;
        }
    }
    // ---- envelope of the SC's later chunks into the records of the earlier
    // ones.  No barrier: the walks of an SC's chunks differ by a factor of
    // five in length and a wavefront that is done leaves; the LAST one to
    // finish folds the chunk maxima into the records of all (a few per chunk).
    s_max[wave][lane] = run;
    s_jr[wave][lane] = (uint16_t)((len > 0 ? jcur : jhi) | (jhi << 8));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    int done = 0;
    if (lane == 0) done = atomicAdd(&s_done, 1);
    done = __builtin_amdgcn_readfirstlane(done);
    if (done != SW - 1) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    uint64_t later = PR_ZERO;
    for (int w = SW - 1; w >= 0; w--) {
        const int jr = s_jr[w][lane];
        if (live && later != PR_ZERO) sc_raise(out, jr & 0xff, jr >> 8, later);
        const uint64_t v = s_max[w][lane];
        if (pr_better((uint32_t)(v >> 32), (uint32_t)v, later)) later = v;
    }
    a.sc_max[so] = live ? later : PR_ZERO;
    // thresholds reached up to the end of this SC (its last wavefront's)
    a.sc_jhi[so] = (uint8_t)(live ? s_jr[SW - 1][lane] >> 8 : 0);
}

// The envelope of the later SCs folded into val: one wavefront per (category,
// range, threshold) row.  Pass 1, lane = SC (64 at a time, from the category's
// last SC backwards): the SCs' maxima become suffix maxima by a wave scan, and
// SC s, which emitted the thresholds [jhi(s - 1), jhi(s)), notes for each of
// them the maximum of the SCs behind it in LDS.  Pass 2, lane = threshold: the
// row's records are raised with coalesced loads and stores.
__device__ __forceinline__ uint64_t pr_max(uint64_t a, uint64_t b)
{
    return pr_better((uint32_t)(a >> 32), (uint32_t)a, b) ? a : b;
}
__global__ __launch_bounds__(256) void acc_raise_kernel(AccArgs a)
{
    __shared__ uint64_t s_env[4][N_REC + 3];
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    const int64_t rows = (int64_t)(a.k_end - a.k_begin) * a.n_rng * N_THR;
    if (row >= rows) return;
    const int32_t k = a.k_begin + (int32_t)(row / (a.n_rng * N_THR));
    const int combo = (int)(row % (a.n_rng * N_THR));
    const int32_t s0 = a.cat_chunk_off[k], s1 = a.cat_chunk_off[k + 1];
    if (s1 - s0 < 2) return;
    if (a.num_gt[(int64_t)k * a.n_rng + combo / N_THR] <= 0) return;
    uint64_t *env = s_env[wave];
    for (int j = lane; j < N_REC; j += WAVE) env[j] = PR_ZERO;
    const int64_t step = (int64_t)a.n_words * WAVE;
    const int64_t o0 = ((int64_t)combo / WAVE) * WAVE + combo % WAVE;
    uint64_t behind = PR_ZERO;            // maximum of the SCs of the later passes
    for (int32_t top = s1; top > s0; top -= WAVE) {
        // lane l holds SC top - 1 - l: the scan runs towards higher lanes
        const int32_t sc = top - 1 - lane;
        const bool have = sc >= s0;
        const int64_t o = (int64_t)(have ? sc : s0) * step + o0;
        uint64_t v = have ? a.sc_max[o] : PR_ZERO;
        const int jh = have ? (int)a.sc_jhi[o] : 0;
        const int jl = (have && sc > s0) ? (int)a.sc_jhi[o - step] : 0;
        // inclusive prefix maximum over the lanes below (= the later SCs)
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) {
            const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)v, d, WAVE);
            const uint32_t hi = (uint32_t)__shfl_up((int)(uint32_t)(v >> 32), d, WAVE);
            const uint64_t u = ((uint64_t)hi << 32) | lo;
            if (lane >= d) v = pr_max(u, v);
        }
        // exclusive: what lies behind SC `sc`
        const uint32_t elo = (uint32_t)__shfl_up((int)(uint32_t)v, 1, WAVE);
        const uint32_t ehi = (uint32_t)__shfl_up((int)(uint32_t)(v >> 32), 1, WAVE);
        uint64_t after = lane > 0 ? (((uint64_t)ehi << 32) | elo) : PR_ZERO;
        after = pr_max(after, behind);
        if (have && after != PR_ZERO)
            for (int j = jl; j < jh; j++) env[j] = after;
        behind = pr_max(behind, readlane_u64(v, WAVE - 1));
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    uint64_t *__restrict__ out = a.val + ((int64_t)k * a.n_rng * N_THR + combo) * N_REC;
    for (int j = lane; j < N_REC; j += WAVE) {
        const uint64_t e = env[j];
        if (e != PR_ZERO) {
            const uint64_t v = out[j];
            if (pr_better((uint32_t)(e >> 32), (uint32_t)e, v)) out[j] = e;
        }
    }
}

// val[KR][T*R] -> precision[T*R][KR], rec[KR][T] -> recall[T][KR], -1 fill
struct FinArgs {
    int32_t n_cat, n_rng;
    const int32_t *num_gt;
    const uint64_t *val;
    const double *rec;
    double *precision, *recall;
};

// One workgroup: 64 (k, r) rows x 64 (t, j) columns.  Rows without evaluated
// ground truth (most of them: a category rarely has tracks in every area x
// duration range) are never read.  The output, 8 bytes x 1010 x K x A, is the
// traffic that cannot be avoided; a wavefront stores 64 consecutive rows of
// one column (512 B) at a time.  Measured alone at Config 2's track-level
// shape (1203 x 21 rows, one in six live): 53 us, a plain fill of the same
// buffer 34 us, the 32 x 32 tiling this replaces 83 us.  Tried and slower:
// row windows shifted to 128-byte lines (64 us), dead rows stored ahead of the
// barrier (72 us, partial lines), lanes gathering their own row without LDS
// (56 us sparse, 122 us dense), 128/256-row tiles.
#define FIN_RT 64
#define FIN_CT 64

__global__ __launch_bounds__(256) void acc_finalize_kernel(FinArgs a)
{
    __shared__ double tile[FIN_RT][FIN_CT + 1];
    const int64_t KR = (int64_t)a.n_cat * a.n_rng;
    const int64_t COLS = (int64_t)N_THR * N_REC;
    const int64_t row0 = (int64_t)blockIdx.x * FIN_RT;   // (k, r) rows of val
    const int64_t col0 = (int64_t)blockIdx.y * FIN_CT;   // (t, j) columns of val
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int ncol = (int)min((int64_t)FIN_CT, COLS - col0);
    const int64_t orow = row0 + lane;                    // lane = row of the tile
    const bool olive = orow < KR && a.num_gt[orow] > 0;
    const uint64_t live_rows = __ballot(olive);          // same in every wavefront
    // rows i = wave (mod 4) are mine to fetch; only the live ones are read,
    // four loads in flight per step (one row in six is live at Config 2: a
    // wavefront has two or three to fetch, and waiting for them one by one is
    // what the kernel's time was)
    static_assert(FIN_RT == 64 && FIN_CT == 64, "one mask word, one lane per column");
    if (live_rows != 0) {
        uint64_t m = live_rows & (0x1111111111111111ull << wave);
        const int64_t col = col0 + lane;
        const uint64_t *__restrict__ src =
            a.val + row0 * COLS + (lane < ncol ? col : col0);
        while (m != 0) {
            int idx[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                idx[q] = m != 0 ? __builtin_ctzll(m) : -1;
                if (m != 0) m &= m - 1;
            }
            uint64_t v[4];
#pragma unroll
            for (int q = 0; q < 4; q++)
                v[q] = idx[q] >= 0 ? src[(int64_t)idx[q] * COLS] : PR_ZERO;
            // the sweeps leave (tp, n) records: the one fp64 division of a
            // precision value, tp / (n + eps), happens here, on dense lanes,
            // instead of inside their divergent emission branches
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (idx[q] >= 0) tile[idx[q]][lane] = pr_value(v[q]);
        }
        __syncthreads();
    }
    if (orow < KR)
        for (int c = wave; c < ncol; c += 4)
            a.precision[(col0 + c) * KR + orow] = olive ? tile[lane][c] : -1.0;
    // recall: the first column-block also transposes rec[KR][T]
    if (blockIdx.y == 0) {
        for (int i = threadIdx.x; i < FIN_RT * N_THR; i += 256) {
            const int64_t row = row0 + i / N_THR;
            const int t = i % N_THR;
            if (row < KR)
                a.recall[(int64_t)t * KR + row] =
                    a.num_gt[row] > 0 ? a.rec[row * N_THR + t] : -1.0;
        }
    }
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static int32_t max_chunks(int64_t n_dt, int32_t n_cat)
{
    return (int32_t)((n_dt + ACC_CH - 1) / ACC_CH + n_cat);
}

// ---- which sweep long categories take:
//   "chunked"  the six chunked kernels;  "lookback"  acc_sweep_kernel in one
//   pass;  "twopass"  acc_sweep_kernel behind a counting pass.
// taoamd_accumulate_sweep_mode() sets it for the process (the tests run every
// parity case under each); unset, the environment's TAOAMD_SWEEP (read once);
// without either: the one-pass sweep from SWEEP_ONEPASS_MIN rows up.  At
// 21 M rows (2000 videos) it takes 0.39 ms against 0.52 for the chunked kernels
// (rows read once instead of three times plus a transposed copy); at 2 M rows
// (Config 2) everything is latency and the chunked kernels' three short
// launches win, 0.12 against 0.16 ms.
enum { SWEEP_AUTO = -1, SWEEP_CHUNKED = 0, SWEEP_LOOKBACK = 1, SWEEP_TWOPASS = 2 };
#define SWEEP_ONEPASS_MIN 6000000
static std::atomic<int> g_sweep_set{SWEEP_AUTO};
static std::atomic<int> g_sweep_spin{0};       // 0: SC_SPIN_LIMIT
static std::atomic<uint32_t *> g_sweep_giveups{nullptr};   // taoamd_accumulate_giveup_counter
static int sweep_mode(int64_t n_dt)
{
    static const int env = [] {
        const char *e = getenv("TAOAMD_SWEEP");
        if (e && !strcmp(e, "lookback")) return (int)SWEEP_LOOKBACK;
        if (e && !strcmp(e, "twopass")) return (int)SWEEP_TWOPASS;
        if (e && !strcmp(e, "chunked")) return (int)SWEEP_CHUNKED;
        return (int)SWEEP_AUTO;
    }();
    const int set = g_sweep_set.load(std::memory_order_relaxed);
    const int m = set != SWEEP_AUTO ? set : env;
    if (m != SWEEP_AUTO) return m;
    return n_dt >= SWEEP_ONEPASS_MIN ? (int)SWEEP_LOOKBACK : (int)SWEEP_CHUNKED;
}

extern "C" int taoamd_accumulate_sweep_mode(int32_t mode)
{
    if (mode < SWEEP_AUTO || mode > SWEEP_TWOPASS) return TAOAMD_ERR_ARG;
    g_sweep_set.store(mode, std::memory_order_relaxed);
    return TAOAMD_OK;
}

// an explicit one-pass mode (setter or environment) also takes the categories
// the fused single-workgroup sweep would have taken: the parity tests drive the
// small reference fixtures through the look-back that way
static bool sweep_mode_explicit_onepass()
{
    const int set = g_sweep_set.load(std::memory_order_relaxed);
    const char *e = set == SWEEP_AUTO ? getenv("TAOAMD_SWEEP") : nullptr;
    return set == SWEEP_LOOKBACK || set == SWEEP_TWOPASS ||
           (e && (!strcmp(e, "lookback") || !strcmp(e, "twopass")));
}

// Which kind of plan taoamd_accumulate_prepare builds for these sizes under
// the current mode: 0 none (fused sweep), 1 chunk table, 2 / 3 super-chunk
// table (look-back / two passes).  A prepared workspace serves
// taoamd_accumulate_prepared while this value is the one it was built under.
extern "C" int taoamd_accumulate_plan_kind(int64_t n_dt, int32_t n_rng, int32_t max_segment)
{
    if (n_rng < 1 || n_rng > 32) return -1;
    const int32_t n_words = (n_rng * N_THR + 63) / 64;
    const int32_t fused_cap = ACC_FUSED_WAVES / n_words * ACC_CH;
    if (max_segment > 0 && max_segment <= fused_cap && !sweep_mode_explicit_onepass()) return 0;
    return 1 + sweep_mode(n_dt);
}

extern "C" int taoamd_accumulate_spin_limit(int32_t polls)
{
    g_sweep_spin.store(polls, std::memory_order_relaxed);
    return TAOAMD_OK;
}

// A caller-owned device word that counts the look-backs that gave up, over
// every pass launched while it is registered (nullptr: none).  The workspace's
// own flag is per pass -- an unprepared pass clears it when it starts --, so a
// caller that runs many passes and synchronises once (bench.py's timed region
// through the multi-GPU plans) reads this instead.
extern "C" int taoamd_accumulate_giveup_counter(uint32_t *device_word)
{
    g_sweep_giveups.store(device_word, std::memory_order_relaxed);
    return TAOAMD_OK;
}
// (rows, whatever their width: at 2.9 M track-level rows of four words -- the
// stress shape -- the chunked kernels are as fast, 0.275 against 0.32 ms)
#define SC_TICKETS 64           // header words of the SC tables (word 0: the error flag)
static std::atomic<uint32_t> g_sc_gen{0};

static size_t max_scs(int64_t n_dt, int32_t n_cat)
{
    return (size_t)(n_dt / (SWEEP_SW * SWEEP_NB * WAVE) + n_cat + 1);
}

static size_t sc_workspace(int64_t n_dt, int32_t n_cat, size_t nw)
{
    const size_t ns = max_scs(n_dt, n_cat);
    return align256(SC_TICKETS * 4) + align256(ns * nw * WAVE * 16) +
           align256(ns * nw * WAVE * 8) + align256(ns * nw * WAVE);
}

static size_t base_workspace(int64_t n_dt, int32_t n_cat, int32_t n_rng)
{
    const size_t nw = (size_t)(n_rng * N_THR + 63) / 64;
    const size_t nc = (size_t)max_chunks(n_dt, n_cat);
    return sc_workspace(n_dt, n_cat, nw) +
           align256(((size_t)n_cat + 1) * 4) + align256(64) + align256(nc * 4) +
           4 * align256(nc * nw * WAVE * 4) +
           align256(nc * nw * WAVE * 8) +
           2 * align256(nc * nw * ACC_BLK * WAVE * 8) +
           align256((size_t)n_cat * n_rng * N_REC * 4) + 4096;
}

extern "C" size_t taoamd_compact_elems(int32_t n_cat, int32_t n_rng)
{
    return (size_t)n_cat * n_rng * N_THR * N_REC;
}

extern "C" size_t taoamd_accumulate_workspace(int64_t n_dt, int32_t n_cat,
                                              int32_t n_rng)
{
    return base_workspace(n_dt, n_cat, n_rng) +
           align256(taoamd_compact_elems(n_cat, n_rng) * 8) +
           align256((size_t)n_cat * n_rng * N_THR * 8);
}

// phases of accumulate_compact: the chunk table depends on cat_off alone, so a
// caller that sweeps the same categories again and again builds it once
// (taoamd_accumulate_prepare) and keeps one launch off its critical chain
enum { ACC_ALL = 0, ACC_PLAN = 1, ACC_SWEEP = 2 };

static int accumulate_compact(int64_t n_dt, int32_t n_cat, int32_t n_rng,
                              const int32_t *cat_off, const int32_t *order,
                              const uint64_t *matched, const uint64_t *ignored,
                              const int32_t *num_gt, int32_t k_begin, int32_t k_end,
                              int32_t max_segment, double *val, double *rec,
                              void *workspace, size_t workspace_bytes, void *stream,
                              int phase = ACC_ALL, int force_mode = SWEEP_AUTO)
{
    if (n_cat <= 0 || n_rng < 1 || n_rng > 32) return TAOAMD_ERR_ARG;
    if (k_begin < 0 || k_end > n_cat || k_begin > k_end) return TAOAMD_ERR_ARG;
    if (!cat_off || !workspace) return TAOAMD_ERR_ARG;
    if (phase != ACC_PLAN && (!num_gt || !val || !rec)) return TAOAMD_ERR_ARG;
    if (workspace_bytes < base_workspace(n_dt, n_cat, n_rng))
        return TAOAMD_ERR_WORKSPACE;
    if (k_begin == k_end) return TAOAMD_OK;
    hipStream_t s = (hipStream_t)stream;
    AccArgs a;
    a.cat_chunk_off = nullptr; a.chunk_tab = nullptr; a.cnt_tp = a.cnt_fp = a.pre_tp = a.pre_fp = nullptr;
    a.cmax = a.t_tp = a.t_fp = nullptr; a.cj = nullptr;
    a.n_dt = n_dt; a.n_cat = n_cat; a.n_rng = n_rng;
    a.n_words = (n_rng * N_THR + 63) / 64;
    a.n_chunks_max = max_chunks(n_dt, n_cat);
    a.cat_off = cat_off; a.matched = matched; a.ignored = ignored; a.order = order;
    a.paired = matched != nullptr && ignored == matched + 1;
    a.wide = a.paired && ((uintptr_t)matched & 15) == 0;
    a.num_gt = num_gt; a.val = (uint64_t *)val; a.rec = rec;
    a.k_begin = k_begin; a.k_end = k_end;
    a.inline_scans = 0;
    a.sc_rows = 0; a.sc_gen = 0; a.sc_stat = a.sc_max = nullptr; a.sc_jhi = nullptr;
    a.sc_error = nullptr;
    a.sc_giveups = g_sweep_giveups.load(std::memory_order_relaxed);
    a.xcd_start = nullptr;
    {
        const int sp = g_sweep_spin.load(std::memory_order_relaxed);
        a.sc_spin = sp == 0 ? SC_SPIN_LIMIT : sp;
    }
    // every category fits one workgroup (the host says so): the fused sweep.
    // Mixing the two paths per category was measured slower than the chunked
    // path alone when long categories exist (image level, Config 2: 162 vs
    // 126 us of kernel time), so it is all or nothing.
    // One launch per size class: categories of up to 4, 8 and 16 wavefronts'
    // worth of (chunk, word) pairs get a workgroup of that size -- a category
    // of 30 tracks at four combo words keeps 4 wavefronts busy, and in a
    // 16-wave workgroup the other 12 only hold the CU's wave slots (2
    // workgroups per CU).  Track level, 2000 videos: 305 -> 104 us.
    // Workgroups of the other classes leave at once.
    // (header of the workspace: word 0 = the one-pass sweep's error flag,
    // taoamd_accumulate_error -- zero whenever a plan or an unprepared pass
    // starts, whatever path it takes)
    if (phase != ACC_SWEEP)
        TAO_HIP(hipMemsetAsync((void *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255), 0,
                               align256(SC_TICKETS * 4), s));
    const int32_t fused_cap = ACC_FUSED_WAVES / a.n_words * ACC_CH;
    const bool all_fused = max_segment > 0 && max_segment <= fused_cap &&
                           (force_mode == SWEEP_CHUNKED || !sweep_mode_explicit_onepass());
    a.fused_rows = 0;
    a.fused_lo = -1;
    if (all_fused && phase == ACC_PLAN) return TAOAMD_OK;     // no chunk table
    if (all_fused) {
        const unsigned grid = (unsigned)(k_end - k_begin);
        int32_t lo = -1;
        for (int fw = 4; fw <= ACC_FUSED_WAVES; fw *= 2) {
            const int32_t hi = fw / a.n_words * ACC_CH;
            if (hi <= lo || hi == 0) continue;
            a.fused_lo = lo;
            a.fused_rows = hi;
            if (fw == 4) {
                TAO_TIMED("acc_fused_kernel", s, acc_fused_kernel<4><<<grid, 4 * WAVE, 0, s>>>(a, rec_thr()));
            } else if (fw == 8) {
                TAO_TIMED("acc_fused_kernel", s, acc_fused_kernel<8><<<grid, 8 * WAVE, 0, s>>>(a, rec_thr()));
            } else {
                TAO_TIMED("acc_fused_kernel", s, acc_fused_kernel<16><<<grid, 16 * WAVE, 0, s>>>(a, rec_thr()));
            }
            lo = hi;
            if (max_segment > 0 && max_segment <= hi) break;
        }
        TAO_LAUNCH_CHECK();
        return TAOAMD_OK;
    }
    // categories of at most ACC_INLINE_CHUNKS chunks: the two per-category
    // scan kernels are folded into their consumers (four launches on the
    // chain instead of six; all or nothing, like the fused sweep)
    a.inline_scans = max_segment > 0 && max_segment <= ACC_INLINE_CHUNKS * ACC_CH;
    unsigned char *w = (unsigned char *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    const size_t nc = (size_t)a.n_chunks_max, nw = (size_t)a.n_words;
    const int mode = force_mode != SWEEP_AUTO ? force_mode : sweep_mode(n_dt);
    const size_t ns = max_scs(n_dt, n_cat);
    uint32_t *tickets = (uint32_t *)w;   w += align256(SC_TICKETS * 4);
    a.sc_stat = (uint64_t *)w;           w += align256(ns * nw * WAVE * 16);
    a.sc_max = (uint64_t *)w;            w += align256(ns * nw * WAVE * 8);
    a.sc_jhi = (uint8_t *)w;             w += align256(ns * nw * WAVE);
    if (mode != SWEEP_CHUNKED) {
        a.sc_rows = SWEEP_SW * SWEEP_NB * WAVE;
        a.inline_scans = 0;
    }
    a.cat_chunk_off = (int32_t *)w; w += align256(((size_t)n_cat + 1) * 4);
    // (the runs' lengths are bounded only when the longest category is known)
    if (mode != SWEEP_CHUNKED && max_segment > 0) a.xcd_start = (int32_t *)w;
    w += align256(64);
    int32_t *chunk_tab = (int32_t *)w; w += align256(nc * 4);
    a.chunk_tab = phase == ACC_SWEEP ? chunk_tab : nullptr;
    a.cnt_tp = (uint32_t *)w; w += align256(nc * nw * WAVE * 4);
    a.cnt_fp = (uint32_t *)w; w += align256(nc * nw * WAVE * 4);
    a.pre_tp = (uint32_t *)w; w += align256(nc * nw * WAVE * 4);
    a.pre_fp = (uint32_t *)w; w += align256(nc * nw * WAVE * 4);
    a.cmax = (uint64_t *)w; w += align256(nc * nw * WAVE * 8);
    a.t_tp = (uint64_t *)w; w += align256(nc * nw * ACC_BLK * WAVE * 8);
    a.t_fp = (uint64_t *)w; w += align256(nc * nw * ACC_BLK * WAVE * 8);
    a.cj = (int32_t *)w;
    const unsigned chunk_blocks = (unsigned)((nc * nw + 3) / 4);
    const unsigned cat_blocks = (unsigned)((size_t)(k_end - k_begin) * nw);
    if (phase != ACC_SWEEP)
        TAO_TIMED("acc_chunks_kernel", s, acc_chunks_kernel<<<1, 256, 0, s>>>(a));
    if (phase == ACC_PLAN) {
        TAO_TIMED("acc_chunktab_kernel", s, acc_chunktab_kernel<<<(unsigned)((nc + 255) / 256), 256, 0, s>>>(a, chunk_tab));
        // (the look-back's ticket slots and status words start from zero)
        if (mode != SWEEP_CHUNKED)
            TAO_HIP(hipMemsetAsync(tickets, 0, align256(SC_TICKETS * 4) + align256(ns * nw * WAVE * 16), s));
        TAO_LAUNCH_CHECK();
        return TAOAMD_OK;
    }
    if (mode != SWEEP_CHUNKED) {
        {
            const int64_t n_cj = (int64_t)n_cat * n_rng * N_REC;
            TAO_TIMED("acc_cj_kernel", s, acc_cj_kernel<<<(unsigned)((n_cj + 255) / 256), 256, 0, s>>>(a, rec_thr()));
        }
        // (every category's last SC may be a partial one; an SC past the table leaves at once)
        // SC = workgroup number: every SC of the table.  XCD-aware: eight runs
        // cut at category boundaries, each at most an eighth of the table and
        // one category longer
        const size_t n_sc = (size_t)(n_dt / a.sc_rows) + (size_t)n_cat + 1;
        const unsigned grid = a.xcd_start == nullptr
            ? (unsigned)(n_sc * nw)
            : (unsigned)(N_XCD * (n_sc / N_XCD + (size_t)max_segment / a.sc_rows + 3) * nw);
        if (mode == SWEEP_LOOKBACK) {
            // an unprepared workspace may hold anything
            if (phase == ACC_ALL)
                TAO_HIP(hipMemsetAsync(tickets, 0, align256(SC_TICKETS * 4) + align256(ns * nw * WAVE * 16), s));
            uint32_t g = ++g_sc_gen;
            a.sc_gen = (g & 0x3fffffffu) ? (g & 0x3fffffffu) : (++g_sc_gen & 0x3fffffffu);
            a.sc_error = tickets;
            TAO_TIMED("acc_sweep_kernel", s, (acc_sweep_kernel<SWEEP_SW, 0><<<grid, SWEEP_SW * WAVE, 0, s>>>(a, rec_thr())));
        } else {
            TAO_TIMED("acc_sccount_kernel", s, (acc_sweep_kernel<SWEEP_SW, 2><<<grid, SWEEP_SW * WAVE, 0, s>>>(a, rec_thr())));
            TAO_TIMED("acc_sweep_kernel", s, (acc_sweep_kernel<SWEEP_SW, 1><<<grid, SWEEP_SW * WAVE, 0, s>>>(a, rec_thr())));
        }
        // the later SCs' envelope into the records (val final after this)
        const int64_t rows = (int64_t)(k_end - k_begin) * n_rng * N_THR;
        TAO_TIMED("acc_raise_kernel", s, acc_raise_kernel<<<(unsigned)((rows + 3) / 4), 256, 0, s>>>(a));
        TAO_LAUNCH_CHECK();
        return TAOAMD_OK;
    }
    TAO_TIMED("acc_count_kernel", s, acc_count_kernel<<<chunk_blocks, 256, 0, s>>>(a));
    if (a.inline_scans) {
        TAO_TIMED("acc_chunkmax_kernel", s, acc_chunkmax_kernel<true><<<chunk_blocks, 256, 0, s>>>(a, rec_thr()));
        TAO_TIMED("acc_emit_kernel", s, acc_emit_kernel<true><<<chunk_blocks, 256, 0, s>>>(a));
    } else {
        TAO_TIMED("acc_prefix_kernel", s, acc_prefix_kernel<<<cat_blocks, 256, 0, s>>>(a, rec_thr()));
        TAO_TIMED("acc_chunkmax_kernel", s, acc_chunkmax_kernel<false><<<chunk_blocks, 256, 0, s>>>(a, rec_thr()));
        TAO_TIMED("acc_sufmax_kernel", s, acc_sufmax_kernel<<<cat_blocks, 256, 0, s>>>(a));
        TAO_TIMED("acc_emit_kernel", s, acc_emit_kernel<false><<<chunk_blocks, 256, 0, s>>>(a));
    }
    TAO_LAUNCH_CHECK();
    return TAOAMD_OK;
}

// 0 / 1: a look-back of the one-pass sweep gave up waiting (SC_SPIN_LIMIT) in
// a pass on this workspace since it was prepared -- its tables are not to be
// trusted.  Synchronises with `stream`.
extern "C" int taoamd_accumulate_error(const void *workspace, void *stream, int32_t *flag_host)
{
    if (!workspace || !flag_host) return TAOAMD_ERR_ARG;
    const unsigned char *w = (const unsigned char *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    uint32_t f = 0;
    TAO_HIP(hipMemcpyAsync(&f, w, sizeof f, hipMemcpyDeviceToHost, (hipStream_t)stream));
    TAO_HIP(hipStreamSynchronize((hipStream_t)stream));
    *flag_host = (int32_t)f;
    return TAOAMD_OK;
}

extern "C" int taoamd_accumulate_compact(int64_t n_dt, int32_t n_cat,
                                         int32_t n_rng, const int32_t *cat_off,
                                         const uint64_t *matched,
                                         const uint64_t *ignored,
                                         const int32_t *num_gt, int32_t k_begin,
                                         int32_t k_end, int32_t max_segment,
                                         double *val, double *rec,
                                         void *workspace, size_t workspace_bytes,
                                         void *stream)
{
    return accumulate_compact(n_dt, n_cat, n_rng, cat_off, nullptr, matched, ignored,
                              num_gt, k_begin, k_end, max_segment, val, rec,
                              workspace, workspace_bytes, stream);
}

extern "C" int taoamd_finalize(int32_t n_cat, int32_t n_rng,
                               const int32_t *num_gt, const double *val,
                               const double *rec, double *precision,
                               double *recall, void *stream)
{
    if (n_cat <= 0 || n_rng < 1 || n_rng > 32) return TAOAMD_ERR_ARG;
    if (!num_gt || !val || !rec || !precision || !recall) return TAOAMD_ERR_ARG;
    FinArgs f;
    f.n_cat = n_cat; f.n_rng = n_rng; f.num_gt = num_gt; f.val = (const uint64_t *)val; f.rec = rec;
    f.precision = precision; f.recall = recall;
    const int64_t KR = (int64_t)n_cat * n_rng;
    dim3 grid((unsigned)((KR + FIN_RT - 1) / FIN_RT),
              (unsigned)((N_THR * N_REC + FIN_CT - 1) / FIN_CT));
    TAO_TIMED("acc_finalize_kernel", (hipStream_t)stream, acc_finalize_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(f));
    TAO_LAUNCH_CHECK();
    return TAOAMD_OK;
}

static int accumulate_all(int64_t n_dt, int32_t n_cat, int32_t n_rng,
                          const int32_t *cat_off, const int32_t *order,
                          const uint64_t *matched, const uint64_t *ignored,
                          const int32_t *num_gt, int32_t max_segment,
                          double *precision, double *recall, void *workspace,
                          size_t workspace_bytes, void *stream, int phase = ACC_ALL,
                          int force_mode = SWEEP_AUTO)
{
    if (n_cat <= 0 || n_rng < 1 || n_rng > 32) return TAOAMD_ERR_ARG;
    if (!workspace) return TAOAMD_ERR_ARG;
    if (workspace_bytes < taoamd_accumulate_workspace(n_dt, n_cat, n_rng))
        return TAOAMD_ERR_WORKSPACE;
    unsigned char *w = (unsigned char *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    const size_t base = base_workspace(n_dt, n_cat, n_rng);
    double *val = (double *)(w + base);
    double *rec = val + (align256(taoamd_compact_elems(n_cat, n_rng) * 8) / 8);
    int st = accumulate_compact(n_dt, n_cat, n_rng, cat_off, order, matched, ignored,
                                num_gt, 0, n_cat, max_segment, val, rec, w, base,
                                stream, phase, force_mode);
    if (st != TAOAMD_OK || phase == ACC_PLAN) return st;
    return taoamd_finalize(n_cat, n_rng, num_gt, val, rec, precision, recall, stream);
}

extern "C" int taoamd_accumulate(int64_t n_dt, int32_t n_cat, int32_t n_rng,
                                 const int32_t *cat_off,
                                 const uint64_t *matched,
                                 const uint64_t *ignored, const int32_t *num_gt,
                                 int32_t max_segment, double *precision,
                                 double *recall, void *workspace,
                                 size_t workspace_bytes, void *stream)
{
    return accumulate_all(n_dt, n_cat, n_rng, cat_off, nullptr, matched, ignored,
                          num_gt, max_segment, precision, recall, workspace,
                          workspace_bytes, stream);
}

// The same tables from the chunked kernels whatever the sweep mode: what a
// caller runs again when taoamd_accumulate_error reports that a look-back of
// the one-pass sweep gave up (the rows of the pass are still in place).  The
// workspace's prepared plan does not survive it (taoamd_accumulate_prepare
// again before the next taoamd_accumulate_prepared).
extern "C" int taoamd_accumulate_chunked(int64_t n_dt, int32_t n_cat, int32_t n_rng,
                                         const int32_t *cat_off,
                                         const uint64_t *matched,
                                         const uint64_t *ignored, const int32_t *num_gt,
                                         int32_t max_segment, double *precision,
                                         double *recall, void *workspace,
                                         size_t workspace_bytes, void *stream)
{
    return accumulate_all(n_dt, n_cat, n_rng, cat_off, nullptr, matched, ignored,
                          num_gt, max_segment, precision, recall, workspace,
                          workspace_bytes, stream, ACC_ALL, SWEEP_CHUNKED);
}

extern "C" int taoamd_accumulate_compact_chunked(int64_t n_dt, int32_t n_cat,
                                                 int32_t n_rng, const int32_t *cat_off,
                                                 const uint64_t *matched,
                                                 const uint64_t *ignored,
                                                 const int32_t *num_gt, int32_t k_begin,
                                                 int32_t k_end, int32_t max_segment,
                                                 double *val, double *rec,
                                                 void *workspace, size_t workspace_bytes,
                                                 void *stream)
{
    return accumulate_compact(n_dt, n_cat, n_rng, cat_off, nullptr, matched, ignored,
                              num_gt, k_begin, k_end, max_segment, val, rec,
                              workspace, workspace_bytes, stream, ACC_ALL, SWEEP_CHUNKED);
}

extern "C" int taoamd_accumulate_by_order(int64_t n_dt, int32_t n_cat, int32_t n_rng,
                                          const int32_t *cat_off,
                                          const int32_t *order,
                                          const uint64_t *matched,
                                          const uint64_t *ignored,
                                          const int32_t *num_gt, int32_t max_segment,
                                          double *precision, double *recall,
                                          void *workspace, size_t workspace_bytes,
                                          void *stream)
{
    if (!order) return TAOAMD_ERR_ARG;
    return accumulate_all(n_dt, n_cat, n_rng, cat_off, order, matched, ignored,
                          num_gt, max_segment, precision, recall, workspace,
                          workspace_bytes, stream);
}

extern "C" int taoamd_accumulate_prepare(int64_t n_dt, int32_t n_cat, int32_t n_rng,
                                         const int32_t *cat_off, int32_t max_segment,
                                         void *workspace, size_t workspace_bytes,
                                         void *stream)
{
    return accumulate_all(n_dt, n_cat, n_rng, cat_off, nullptr, nullptr, nullptr,
                          nullptr, max_segment, nullptr, nullptr, workspace,
                          workspace_bytes, stream, ACC_PLAN);
}

extern "C" int taoamd_accumulate_prepared(int64_t n_dt, int32_t n_cat, int32_t n_rng,
                                          const int32_t *cat_off,
                                          const uint64_t *matched,
                                          const uint64_t *ignored,
                                          const int32_t *num_gt, int32_t max_segment,
                                          double *precision, double *recall,
                                          void *workspace, size_t workspace_bytes,
                                          void *stream)
{
    return accumulate_all(n_dt, n_cat, n_rng, cat_off, nullptr, matched, ignored,
                          num_gt, max_segment, precision, recall, workspace,
                          workspace_bytes, stream, ACC_SWEEP);
}
