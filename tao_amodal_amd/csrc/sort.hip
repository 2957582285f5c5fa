// Stable LSD radix sort of detections by (category asc, score desc) (gfx950).
//
// The reference orders the detections of a category with
// np.argsort(-dt_scores, kind="mergesort") over the concatenation of the
// category's cells (lvis_amodal/eval.py:353-361, tao_amodal/eval.py:508-518):
// ties keep concatenation order.  The input of this sort IS in concatenation
// order, so any stable sort on the key (category, -score) reproduces it.
//
// Key digits, least significant first: 8 x 8 bits of the order-preserving
// transform of -score (descending), then 4 x 8 bits of the category index
// (any non-negative int32: the flatten stage sorts by cell keys with it; a
// NULL score sorts by the integer key alone).
// A pass whose digit is the same for every element (e.g. the sign/exponent
// byte of scores in (0,1)) is detected from the global digit histograms and
// skipped on the device without host involvement.
//
// Per pass: (1) per-block digit histogram, (2) one block per digit scans its
// row of block counts, (3) scatter with wave-level match-any ranking
// (8 ballots per 64 elements) so that equal digits keep their order.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "common.hpp"

using namespace taoamd;

#define RS_THREADS 256
#define RS_WAVES (RS_THREADS / WAVE)
#define RS_ITEMS 8                           // rounds of 64 per wave
#define RS_TILE (RS_THREADS * RS_ITEMS)      // 2048 elements per block
#define RS_BINS 256
#define RS_PASSES 12

struct SortBufs {
    uint64_t *key[2];
    int32_t *idx[2];
    const int32_t *cat;
    uint32_t *block_hist;   // [RS_BINS][n_blocks]
    uint32_t *digit_total;  // [RS_PASSES][RS_BINS] global totals
    int32_t *skip;          // [RS_PASSES]
    int32_t *sel;           // [RS_PASSES + 1] which buffer holds the data
    int64_t n;
    int32_t n_blocks;
};

__device__ __forceinline__ uint64_t desc_key(double s)
{
    s = s + 0.0;  // -0.0 -> +0.0: argsort(-score) sees them as equal
    uint64_t u = (uint64_t)__double_as_longlong(s);
    uint64_t asc = (u >> 63) ? ~u : (u | 0x8000000000000000ull);
    return ~asc;
}

__device__ __forceinline__ uint32_t digit_of(int pass, uint64_t key, int32_t idx,
                                             const int32_t *__restrict__ cat)
{
    if (pass < 8) return (uint32_t)(key >> (8 * pass)) & 255u;
    return ((uint32_t)cat[idx] >> (8 * (pass - 8))) & 255u;
}

// keys, identity payload and the global histograms of all 10 digits
__global__ __launch_bounds__(RS_THREADS) void rs_init_kernel(
    SortBufs b, const double *__restrict__ score)
{
    __shared__ uint32_t h[RS_PASSES][RS_BINS];
    for (int i = threadIdx.x; i < RS_PASSES * RS_BINS; i += RS_THREADS)
        (&h[0][0])[i] = 0;
    __syncthreads();
    for (int64_t i = blockIdx.x * (int64_t)RS_THREADS + threadIdx.x; i < b.n;
         i += (int64_t)gridDim.x * RS_THREADS) {
        uint64_t k = score ? desc_key(score[i]) : 0;
        b.key[0][i] = k;
        b.idx[0][i] = (int32_t)i;
#pragma unroll
        for (int p = 0; p < RS_PASSES; p++)
            atomicAdd(&h[p][digit_of(p, k, (int32_t)i, b.cat)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < RS_PASSES * RS_BINS; i += RS_THREADS) {
        uint32_t v = (&h[0][0])[i];
        if (v) atomicAdd(&b.digit_total[i], v);
    }
}

__global__ __launch_bounds__(RS_BINS) void rs_plan_kernel(SortBufs b)
{
    // a pass is skippable iff a single bin holds everything: thread d tests
    // bin d of every pass
    __shared__ int32_t single[RS_PASSES];
    if (threadIdx.x < RS_PASSES) single[threadIdx.x] = 0;
    __syncthreads();
    for (int p = 0; p < RS_PASSES; p++)
        if (b.digit_total[p * RS_BINS + threadIdx.x] == (uint32_t)b.n)
            single[p] = 1;
    __syncthreads();
    if (threadIdx.x == 0) {
        int cur = 0;
        b.sel[0] = 0;
        for (int p = 0; p < RS_PASSES; p++) {
            b.skip[p] = single[p];
            if (!single[p]) cur ^= 1;
            b.sel[p + 1] = cur;
        }
    }
}

__global__ __launch_bounds__(RS_THREADS) void rs_hist_kernel(SortBufs b, int pass)
{
    if (b.skip[pass]) return;
    __shared__ uint32_t h[RS_BINS];
    const int s = b.sel[pass];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RS_TILE;
    for (int k = 0; k < RS_ITEMS; k++) {
        int64_t i = base + k * RS_THREADS + threadIdx.x;
        if (i < b.n)
            atomicAdd(&h[digit_of(pass, b.key[s][i], b.idx[s][i], b.cat)], 1u);
    }
    __syncthreads();
    b.block_hist[(int64_t)threadIdx.x * b.n_blocks + blockIdx.x] = h[threadIdx.x];
}

// one block per digit: exclusive scan of that digit's per-block counts
__global__ __launch_bounds__(RS_THREADS) void rs_scan_kernel(SortBufs b, int pass)
{
    if (b.skip[pass]) return;
    __shared__ uint32_t part[RS_THREADS];
    uint32_t *row = b.block_hist + (int64_t)blockIdx.x * b.n_blocks;
    const int per = (b.n_blocks + RS_THREADS - 1) / RS_THREADS;
    const int lo = threadIdx.x * per, hi = min(lo + per, b.n_blocks);
    uint32_t s = 0;
    for (int i = lo; i < hi; i++) s += row[i];
    part[threadIdx.x] = s;
    __syncthreads();
    // Hillis-Steele inclusive scan of the 256 partial sums
    for (int off = 1; off < RS_THREADS; off <<= 1) {
        uint32_t v = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - s;
    for (int i = lo; i < hi; i++) {
        uint32_t c = row[i];
        row[i] = run;
        run += c;
    }
}

__global__ __launch_bounds__(RS_THREADS) void rs_scatter_kernel(SortBufs b, int pass)
{
    if (b.skip[pass]) return;
    __shared__ uint32_t wave_cnt[RS_WAVES][RS_BINS];
    __shared__ uint32_t digit_base[RS_BINS];
    const int s = b.sel[pass];
    const uint64_t *__restrict__ kin = b.key[s];
    const int32_t *__restrict__ iin = b.idx[s];
    uint64_t *__restrict__ kout = b.key[s ^ 1];
    int32_t *__restrict__ iout = b.idx[s ^ 1];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < RS_WAVES * RS_BINS; i += RS_THREADS)
        (&wave_cnt[0][0])[i] = 0;
    // exclusive prefix of the global digit totals: where each digit starts
    {
        uint32_t v = b.digit_total[pass * RS_BINS + threadIdx.x];
        digit_base[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < RS_BINS; off <<= 1) {
            uint32_t w = threadIdx.x >= off ? digit_base[threadIdx.x - off] : 0;
            __syncthreads();
            digit_base[threadIdx.x] += w;
            __syncthreads();
        }
        uint32_t incl = digit_base[threadIdx.x];
        __syncthreads();
        digit_base[threadIdx.x] = incl - v +
            b.block_hist[(int64_t)threadIdx.x * b.n_blocks + blockIdx.x];
    }
    __syncthreads();
    // wave w owns elements [base + w*512, base + (w+1)*512): 8 rounds of 64
    const int64_t base = (int64_t)blockIdx.x * RS_TILE + (int64_t)wave * (WAVE * RS_ITEMS);
    uint64_t key[RS_ITEMS];
    int32_t idx[RS_ITEMS];
    uint32_t dig[RS_ITEMS], rank[RS_ITEMS];
#pragma unroll
    for (int k = 0; k < RS_ITEMS; k++) {
        const int64_t i = base + k * WAVE + lane;
        const bool ok = i < b.n;
        key[k] = ok ? kin[i] : 0;
        idx[k] = ok ? iin[i] : 0;
        dig[k] = ok ? digit_of(pass, key[k], idx[k], b.cat) : 0xffffffffu;
    }
#pragma unroll
    for (int k = 0; k < RS_ITEMS; k++) {
        const bool ok = dig[k] != 0xffffffffu;
        // match-any: lanes holding the same digit
        uint64_t peers = __ballot(ok);
#pragma unroll
        for (int bit = 0; bit < 8; bit++) {
            const bool one = (dig[k] >> bit) & 1u;
            const uint64_t m = __ballot(one);
            peers &= one ? m : ~m;
        }
        const uint32_t below = (uint32_t)__popcll(peers & ((1ull << lane) - 1));
        uint32_t old = 0;
        if (ok) old = wave_cnt[wave][dig[k]];
        rank[k] = old + below;
        // the highest peer lane publishes the new count (LDS ops of one
        // wave execute in program order)
        if (ok && (peers >> lane) == 1ull)
            wave_cnt[wave][dig[k]] = old + below + 1;
    }
    __syncthreads();
    // offsets of this wave inside the block, per digit
#pragma unroll
    for (int k = 0; k < RS_ITEMS; k++) {
        if (dig[k] == 0xffffffffu) continue;
        uint32_t off = digit_base[dig[k]] + rank[k];
        for (int w = 0; w < wave; w++) off += wave_cnt[w][dig[k]];
        kout[off] = key[k];
        iout[off] = idx[k];
    }
}

// order[p] = idx[p]; dst[idx[p]] = p
__global__ void rs_finish_kernel(SortBufs b, int32_t *__restrict__ order,
                                 int32_t *__restrict__ dst)
{
    const int s = b.sel[RS_PASSES];
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < b.n;
         i += (int64_t)gridDim.x * blockDim.x) {
        int32_t d = b.idx[s][i];
        if (order) order[i] = d;
        if (dst) dst[d] = (int32_t)i;
    }
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" size_t taoamd_sort_workspace(int64_t n)
{
    if (n < 1) n = 1;
    size_t nb = (size_t)((n + RS_TILE - 1) / RS_TILE);
    return 2 * align256((size_t)n * 8) + 2 * align256((size_t)n * 4) +
           align256(nb * RS_BINS * 4) + align256(RS_PASSES * RS_BINS * 4) +
           align256(256) + 4096;
}

extern "C" int taoamd_sort_by_cat_score(int64_t n, const int32_t *dt_cat,
                                        const double *dt_score, int32_t *order,
                                        int32_t *dst, void *workspace,
                                        size_t workspace_bytes, void *stream)
{
    if (n == 0) return TAOAMD_OK;
    if (n > 0x7fffffff) return TAOAMD_ERR_TOO_LARGE;
    if (!dt_cat || !workspace) return TAOAMD_ERR_ARG;
    if (workspace_bytes < taoamd_sort_workspace(n)) return TAOAMD_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    unsigned char *w = (unsigned char *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    SortBufs b;
    b.n = n;
    b.n_blocks = (int32_t)((n + RS_TILE - 1) / RS_TILE);
    b.cat = dt_cat;
    b.key[0] = (uint64_t *)w; w += align256((size_t)n * 8);
    b.key[1] = (uint64_t *)w; w += align256((size_t)n * 8);
    b.idx[0] = (int32_t *)w;  w += align256((size_t)n * 4);
    b.idx[1] = (int32_t *)w;  w += align256((size_t)n * 4);
    b.block_hist = (uint32_t *)w; w += align256((size_t)b.n_blocks * RS_BINS * 4);
    b.digit_total = (uint32_t *)w; w += align256(RS_PASSES * RS_BINS * 4);
    b.skip = (int32_t *)w;
    b.sel = b.skip + RS_PASSES;
    TAO_HIP(hipMemsetAsync(b.digit_total, 0, RS_PASSES * RS_BINS * 4, s));
    unsigned init_blocks = (unsigned)(b.n_blocks < 2048 ? b.n_blocks : 2048);
    TAO_TIMED("rs_init_kernel", s, rs_init_kernel<<<init_blocks, RS_THREADS, 0, s>>>(b, dt_score));
    TAO_TIMED("rs_plan_kernel", s, rs_plan_kernel<<<1, RS_BINS, 0, s>>>(b));
    for (int p = 0; p < RS_PASSES; p++) {
        TAO_TIMED("rs_hist_kernel", s, rs_hist_kernel<<<b.n_blocks, RS_THREADS, 0, s>>>(b, p));
        TAO_TIMED("rs_scan_kernel", s, rs_scan_kernel<<<RS_BINS, RS_THREADS, 0, s>>>(b, p));
        TAO_TIMED("rs_scatter_kernel", s, rs_scatter_kernel<<<b.n_blocks, RS_THREADS, 0, s>>>(b, p));
    }
    TAO_TIMED("rs_finish_kernel", s, rs_finish_kernel<<<init_blocks, 256, 0, s>>>(b, order, dst));
    TAO_LAUNCH_CHECK();
    return TAOAMD_OK;
}


// ---------------------------------------------------------------------------
// Segment-local sort: when the detections are laid out category-major (the
// cell tables of flatten.py are), a category is a contiguous run and only the
// score order inside it is missing.
//   seg_tile_kernel   one workgroup sorts one tile (<= SEG_TILE elements of one
//                     category) with a stable LSD radix sort in LDS
//   seg_kmerge_kernel categories of 2..SEG_KMERGE_TILES tiles: one pass, every
//                     element sums its ranks in the other tiles
//   seg_mpass_kernel  longer categories: log2(#tiles) pairwise merge-path
//                     passes, the last one writes order / dst
// ---------------------------------------------------------------------------
#define SEG_TILE 2816
#define SEG_THREADS 256

struct SegArgs {
    const int32_t *cat_off;    // [n_cat + 1] element offsets
    const int32_t *tile_off;   // [n_cat + 1] tile offsets
    const int32_t *cat;        // [n] category of every element
    const double *score;
    uint64_t *key[2];
    int32_t *idx[2];
    int32_t *order, *dst;
    int32_t *bnd;              // splitter ranks per (bucket, tile) of the long categories
    int64_t n;
    int32_t n_cat, n_tiles;
};

// LDS radix sort of one tile, LSD, one byte of the descending-score key per
// pass.  Every pass keeps the elements in registers (<= 16 per lane), ranks
// equal digits with the wave-level match-any of rs_scatter_kernel, combines
// the four wavefronts' digit counts with one 256-wide scan and scatters back
// into LDS.  LSD passes are stable and the tile is loaded in input order, so
// the result is the stable order without carrying the position in the key.  A
// pass whose digit is the same for all elements (sign / exponent bytes of
// scores in (0,1)) moves nothing and is skipped.
//
// The tile is sorted by the four bytes of the HIGH key word only; the low
// word matters just inside runs of equal high words that hold an inversion:
//   * such a run of <= SEG_RUN_MAX members is repaired when the tile is
//     written out: every member goes to its rank in the run;
//   * a longer one (e.g. many scores within 1e-6 of each other) gets the
//     four low-byte passes, on its own index range only;
//   * more than SEG_LONG_MAX long runs: the tile starts over with 8 passes.
#define SEG_RUN_MAX 64
#define SEG_LONG_MAX 8
#define SEG_ROUNDS (SEG_TILE / SEG_THREADS)   // rounds of 64 per wavefront

struct SegLds {
    uint64_t key[SEG_TILE];
    uint16_t pos[SEG_TILE];     // bits 0-11 input position, 14/15 run marks
    uint16_t wcnt[4][RS_BINS];  // (a tile holds < 65536 elements)
    uint32_t dbase[RS_BINS];
    uint32_t wave_tot[4];
    int32_t flag;
    int32_t n_long;
    int32_t long_s[SEG_LONG_MAX], long_e[SEG_LONG_MAX];
};

// passes [p0, p1) over the elements [base, base + n) of the tile, in place
// in LDS (all threads of the workgroup; ends with a barrier)
__device__ __forceinline__ void seg_passes(SegLds &L, int base, int n, int p0, int p1)
{
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    // wavefront w owns the contiguous slice [w*per, (w+1)*per) of the range
    const int per = ((n + 4 * WAVE - 1) / (4 * WAVE)) * WAVE;
    const int rounds = per / WAVE;
    const int w0 = wave * per;
    uint64_t kr[SEG_ROUNDS];
    uint16_t pr[SEG_ROUNDS];
#pragma unroll
    for (int r = 0; r < SEG_ROUNDS; r++) {
        const int i = w0 + r * WAVE + lane;
        const bool ok = r < rounds && i < n;
        kr[r] = ok ? L.key[base + i] : 0;
        pr[r] = ok ? L.pos[base + i] : (uint16_t)0;
    }
    __syncthreads();
#pragma nounroll
    for (int pass = p0; pass < p1; pass++) {
        for (int i = threadIdx.x; i < 4 * RS_BINS; i += SEG_THREADS)
            (&L.wcnt[0][0])[i] = 0;
        if (threadIdx.x == 0) L.flag = 0;
        __syncthreads();
        // a byte that is the same for every element (the sign / exponent byte
        // of scores in (0,1)) moves nothing: found out with one vote, before
        // any ranking work
        {
            const uint32_t ref = (uint32_t)(L.key[base] >> (8 * pass)) & 255u;
            bool differs = false;
#pragma unroll
            for (int r = 0; r < SEG_ROUNDS; r++)
                if (r < rounds && w0 + r * WAVE + lane < n)
                    differs |= ((uint32_t)(kr[r] >> (8 * pass)) & 255u) != ref;
            if (__ballot(differs) != 0 && lane == 0) L.flag = 2;
        }
        __syncthreads();
        if (L.flag != 2) { __syncthreads(); continue; }
        uint32_t rank[SEG_ROUNDS];
#pragma unroll
        for (int r = 0; r < SEG_ROUNDS; r++) {
            if (r < rounds) {                       // block-uniform
                const int i = w0 + r * WAVE + lane;
                const bool ok = i < n;
                const uint32_t dig = (uint32_t)(kr[r] >> (8 * pass)) & 255u;
                uint64_t peers = __ballot(ok);
#pragma unroll
                for (int bit = 0; bit < 8; bit++) {
                    const bool one = (dig >> bit) & 1u;
                    const uint64_t m = __ballot(one);
                    peers &= one ? m : ~m;
                }
                const uint32_t below = (uint32_t)__popcll(peers & ((1ull << lane) - 1));
                uint32_t old = 0;
                if (ok) old = L.wcnt[wave][dig];
                rank[r] = old + below;
                if (ok && (peers >> lane) == 1ull)
                    L.wcnt[wave][dig] = (uint16_t)(old + below + 1);
            }
        }
        __syncthreads();
        // thread d: digit d.  totals over the four wavefronts, exclusive scan
        {
            const int d = threadIdx.x;
            const uint32_t c0 = L.wcnt[0][d], c1 = L.wcnt[1][d], c2 = L.wcnt[2][d],
                           c3 = L.wcnt[3][d];
            const uint32_t tot = c0 + c1 + c2 + c3;
            if (d == 0) L.flag = 0;
            uint32_t inc = tot;                    // inclusive scan in the wave
#pragma unroll
            for (int off = 1; off < WAVE; off <<= 1) {
                const uint32_t v = __shfl_up(inc, off, WAVE);
                if (lane >= off) inc += v;
            }
            if (lane == WAVE - 1) L.wave_tot[wave] = inc;
            __syncthreads();
            uint32_t before = 0;
            for (int w = 0; w < wave; w++) before += L.wave_tot[w];
            const uint32_t excl = before + inc - tot;
            L.dbase[d] = excl;
            // exclusive over wavefronts, in place
            L.wcnt[0][d] = 0; L.wcnt[1][d] = (uint16_t)c0;
            L.wcnt[2][d] = (uint16_t)(c0 + c1);
            L.wcnt[3][d] = (uint16_t)(c0 + c1 + c2);
            if (tot == (uint32_t)n) L.flag = 1;
        }
        __syncthreads();
        if (L.flag) { __syncthreads(); continue; }
#pragma unroll
        for (int r = 0; r < SEG_ROUNDS; r++) {
            if (r < rounds) {
                const int i = w0 + r * WAVE + lane;
                if (i < n) {
                    const uint32_t dig = (uint32_t)(kr[r] >> (8 * pass)) & 255u;
                    const uint32_t dst = L.dbase[dig] + L.wcnt[wave][dig] + rank[r];
                    L.key[base + dst] = kr[r];
                    L.pos[base + dst] = pr[r];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < SEG_ROUNDS; r++) {
            if (r < rounds) {
                const int i = w0 + r * WAVE + lane;
                if (i < n) { kr[r] = L.key[base + i]; pr[r] = L.pos[base + i]; }
            }
        }
        __syncthreads();
    }
}

// Stable sort of L.key[0 .. n) (with L.pos) in LDS: high-word passes, then the
// repair of runs of equal high words (see above).  `load` fills L.key[i] and
// L.pos[i] = i for all i (it is called again if the tile starts over with all
// eight passes).  Returns whether elements carry run marks (seg_final_slot).
template <class Load>
__device__ __forceinline__ bool seg_sort_lds(SegLds &L, int n, Load load)
{
    bool repaired = false;
#pragma nounroll
    for (int attempt = 0; attempt < 2; attempt++) {
        load();
        if (threadIdx.x == 0) L.n_long = 0;
        __syncthreads();
        seg_passes(L, 0, n, attempt == 0 ? 4 : 0, 8);
        if (attempt == 1) break;            // all 8 bytes done: sorted
        // ---- runs of equal high words holding an inversion of low words
        // (the tile is stably sorted by the high word: a run is contiguous
        // and in input order).  The element that sees an inversion marks its
        // run: bit 15 on every member of a short run, bit 14 on the first
        // member of a long one.
        if (threadIdx.x == 0) L.flag = 0;
        __syncthreads();
        {
            int state = 0;
#pragma nounroll
            for (int i = threadIdx.x + 1; i < n; i += SEG_THREADS) {
                const uint64_t prev = L.key[i - 1], cur = L.key[i];
                if ((prev >> 32) == (cur >> 32) && prev > cur) {
                    const uint32_t h = (uint32_t)(cur >> 32);
                    int s0 = i - 1, e0 = i + 1;
                    while (s0 > 0 && e0 - s0 <= SEG_RUN_MAX &&
                           (uint32_t)(L.key[s0 - 1] >> 32) == h) s0--;
                    while (e0 < n && e0 - s0 <= SEG_RUN_MAX &&
                           (uint32_t)(L.key[e0] >> 32) == h) e0++;
                    if (e0 - s0 > SEG_RUN_MAX) {
                        while (s0 > 0 && (uint32_t)(L.key[s0 - 1] >> 32) == h) s0--;
                        L.pos[s0] |= 0x4000u;
                        state = 2;
                    } else {
                        for (int j = s0; j < e0; j++) L.pos[j] |= 0x8000u;
                        state = max(state, 1);
                    }
                }
            }
            if (state) atomicMax(&L.flag, state);
        }
        __syncthreads();
        repaired = L.flag != 0;
        if (L.flag != 2) break;
        // ---- long runs: list them, then four low-byte passes on each
#pragma nounroll
        for (int i = threadIdx.x; i < n; i += SEG_THREADS) {
            if (L.pos[i] & 0x4000u) {
                L.pos[i] &= 0xbfffu;
                const int slot = atomicAdd(&L.n_long, 1);
                if (slot < SEG_LONG_MAX) { L.long_s[slot] = i; L.long_e[slot] = n; }
            }
        }
        __syncthreads();
        const int n_long = L.n_long;
        if (n_long > SEG_LONG_MAX) { repaired = false; continue; }    // start over
        {   // end of every long run: first index past it with another high word
            for (int q = 0; q < n_long; q++) {
                const int s0 = L.long_s[q];
                const uint32_t h = (uint32_t)(L.key[s0] >> 32);
#pragma nounroll
                for (int i = s0 + 1 + threadIdx.x; i < n; i += SEG_THREADS)
                    if ((uint32_t)(L.key[i] >> 32) != h) { atomicMin(&L.long_e[q], i); break; }
            }
        }
        __syncthreads();
        for (int q = 0; q < n_long; q++)
            seg_passes(L, L.long_s[q], L.long_e[q] - L.long_s[q], 0, 4);
        break;
    }
    return repaired;
}

// final place of the element at LDS position i: itself, or -- a marked member
// of a short run of equal high words -- its rank in the run by (key, position)
__device__ __forceinline__ int seg_final_slot(const SegLds &L, int i, int n, bool repaired)
{
    if (!(repaired && (L.pos[i] & 0x8000u))) return i;
    const uint64_t mine = L.key[i];
    const uint32_t h = (uint32_t)(mine >> 32);
    int s0 = i, e0 = i + 1;
    while (s0 > 0 && (uint32_t)(L.key[s0 - 1] >> 32) == h) s0--;
    while (e0 < n && (uint32_t)(L.key[e0] >> 32) == h) e0++;
    int rk = 0;
    for (int j = s0; j < e0; j++) {
        const uint64_t o = L.key[j];
        rk += (o < mine || (o == mine && j < i)) ? 1 : 0;
    }
    return s0 + rk;
}

__global__ __launch_bounds__(SEG_THREADS, 5) void seg_tile_kernel(SegArgs a)
{
    __shared__ SegLds L;
    // category owning this tile: last k with tile_off[k] <= blockIdx.x
    int32_t lo = 0, hi = a.n_cat;
    while (hi - lo > 1) {
        const int32_t mid = (lo + hi) >> 1;
        if (a.tile_off[mid] <= (int32_t)blockIdx.x) lo = mid; else hi = mid;
    }
    const int32_t k = lo, t = blockIdx.x - a.tile_off[k];
    const int32_t sb = a.cat_off[k], se = a.cat_off[k + 1];
    const int32_t b = sb + t * SEG_TILE;
    const int32_t n = min(SEG_TILE, se - b);
    if (n <= 0) return;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    if (n <= WAVE) {
        // tiny category: one wavefront, rank by counting -- element i goes to
        // the number of elements that precede it in (key, position) order
        if (wave != 0) return;
        const uint64_t mine = lane < n ? desc_key(a.score[b + lane]) : ~0ull;
        int rank = 0;
        for (int j = 0; j < n; j++) {
            const uint32_t lo_ = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mine, j);
            const uint32_t hi_ = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mine >> 32), j);
            const uint64_t other = ((uint64_t)hi_ << 32) | lo_;
            rank += (other < mine || (other == mine && j < lane)) ? 1 : 0;
        }
        if (lane < n) {
            if (se - sb <= SEG_TILE) {
                if (a.order) a.order[b + rank] = b + lane;
                if (a.dst) a.dst[b + lane] = b + rank;
            } else {    // short last tile of a long category: goes on to merge
                a.key[0][b + rank] = mine;
                a.idx[0][b + rank] = b + lane;
            }
        }
        return;
    }
    const bool repaired = seg_sort_lds(L, n, [&]() {
#pragma nounroll
        for (int i = threadIdx.x; i < n; i += SEG_THREADS) {
            L.key[i] = desc_key(a.score[b + i]);
            L.pos[i] = (uint16_t)i;
        }
    });
    // ---- output from LDS; a marked element goes to its rank in its run by
    // (key, position in the run)
    const bool single = se - sb <= SEG_TILE;
#pragma nounroll
    for (int i = threadIdx.x; i < n; i += SEG_THREADS) {
        const uint64_t mine = L.key[i];
        const int to = seg_final_slot(L, i, n, repaired);
        const int32_t d = b + (int32_t)(L.pos[i] & 0x0fffu);
        if (single) {
            if (a.order) a.order[b + to] = d;
            if (a.dst) a.dst[d] = b + to;
        } else {
            a.key[0][b + to] = mine;
            a.idx[0][b + to] = d;
        }
    }
}

// number of elements of run [b, e) that precede (k, i) in the unique order
__device__ __forceinline__ int32_t rank_in(const uint64_t *__restrict__ key,
                                           const int32_t *__restrict__ idx,
                                           int32_t b, int32_t e, uint64_t k,
                                           int32_t i)
{
    int32_t lo = b, hi = e;
    while (lo < hi) {
        const int32_t mid = (lo + hi) >> 1;
        const uint64_t km = key[mid];
        const bool less = km < k || (km == k && idx[mid] < i);
        if (less) lo = mid + 1; else hi = mid;
    }
    return lo - b;
}

// Categories of up to SEG_KMERGE_TILES tiles are finished in ONE pass: every element
// adds up its rank in each of the other sorted tiles of its category (one
// binary search per tile, unique (key, input position) order) -- that sum is
// its final place.  Longer categories take log2(tiles) pairwise merge-path
// passes (seg_mpass_kernel; measured: equal at 5 tiles, 20 % faster at 6).
#ifndef SEG_KMERGE_TILES
#define SEG_KMERGE_TILES 5      // longest category (in tiles) finished by seg_kmerge_kernel
#endif

__global__ __launch_bounds__(256) void seg_kmerge_kernel(SegArgs a)
{
    const int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (p >= a.n) return;
    const int32_t k = a.cat[p];
    const int32_t sb = a.cat_off[k], se = a.cat_off[k + 1];
    if (se - sb <= SEG_TILE) return;
    const uint64_t *__restrict__ kin = a.key[0];
    const int32_t *__restrict__ iin = a.idx[0];
    const uint64_t kx = kin[p];
    const int32_t ix = iin[p];
    const int32_t mine = (int32_t)((p - sb) / SEG_TILE);
    int64_t pos = sb;
    for (int32_t tb = sb, t = 0; tb < se; tb += SEG_TILE, t++) {
        const int32_t te = min(tb + SEG_TILE, se);
        pos += t == mine ? p - tb : rank_in(kin, iin, tb, te, kx, ix);
    }
    if (a.order) a.order[pos] = ix;
    if (a.dst) a.dst[ix] = (int32_t)pos;
}

// ---------------------------------------------------------------------------
// Long categories by SPLITTERS: a sample sort over the already sorted tiles,
// ONE more pass over the data whatever the number of tiles (the pairwise
// merge-path passes need log2(tiles) of them: 4 x 0.23 ms of a 2.9 ms step at
// 2000 videos).
//   seg_split_kernel   one workgroup per category of 2..SEG_SPLIT_TILES tiles:
//                      every s-th element of every sorted tile is a sample; the
//                      samples are sorted in LDS and every q-th of them is a
//                      splitter; the rank of every splitter in every tile (one
//                      binary search, unique (key, input position) order) cuts
//                      the tiles into B buckets;
//   seg_bucket_kernel  one workgroup per bucket: its pieces of the tiles are
//                      gathered in tile order (= input order among equal keys),
//                      sorted in LDS like a tile and written to their final
//                      places (the elements before the bucket are the sum of
//                      the ranks of its lower splitter).
// A bucket cannot overflow the LDS tile: between two consecutive splitters lie
// q samples, and a tile holds at most s elements between two of ITS samples, so
// the bucket has at most s * (q + tiles) elements; q = SEG_TILE / s - tiles
// makes that SEG_TILE.  s = 16 up to 16 tiles (buckets ~91-98 % full), 32 up to
// 32 tiles (64-80 %); longer categories take the merge-path passes.
#define SEG_SPLIT_TILES 32
#define SEG_BND_PER_SLOT 256    // ints of the rank table per SEG_TILE elements of input

struct SegLdsX {                // seg_split_kernel
    SegLds L;
    int32_t idx[SEG_TILE];
    uint64_t spl_key[2 * SEG_SPLIT_TILES];
    int32_t spl_idx[2 * SEG_SPLIT_TILES];
};
struct SegLdsB {                // seg_bucket_kernel: under 32 KB, five workgroups per CU
    SegLds L;
    int32_t piece_lo[SEG_SPLIT_TILES], piece_at[SEG_SPLIT_TILES + 1];
    int32_t out_base;
};

struct SplitGeom {
    int32_t sb, n, m, s, S, q, ns, B;
    bool on;
};

__device__ __forceinline__ SplitGeom split_geom(const SegArgs &a, int32_t k)
{
    SplitGeom g;
    g.sb = a.cat_off[k];
    g.n = a.cat_off[k + 1] - g.sb;
    g.m = (g.n + SEG_TILE - 1) / SEG_TILE;
    g.on = g.m >= 2 && g.m <= SEG_SPLIT_TILES;
    g.s = g.m <= 16 ? 16 : 32;
    g.S = SEG_TILE / g.s;
    g.q = g.S - g.m;
    const int32_t last_len = g.n - (g.m - 1) * SEG_TILE;
    g.ns = (g.m - 1) * g.S + last_len / g.s;
    g.B = (g.ns + g.q - 1) / g.q;
    return g;
}

__global__ __launch_bounds__(SEG_THREADS, 3) void seg_split_kernel(SegArgs a)
{
    __shared__ SegLdsX X;
    SegLds &L = X.L;
    const SplitGeom g = split_geom(a, (int32_t)blockIdx.x);
    if (!g.on) return;
    const uint64_t *__restrict__ kin = a.key[0];
    const int32_t *__restrict__ iin = a.idx[0];
    // sample i, tile-major: the last element of a group of s elements
    auto src_of = [&](int i) {
        const int t = i / g.S, j = i - t * g.S;
        return g.sb + t * SEG_TILE + (j + 1) * g.s - 1;
    };
    for (int i = threadIdx.x; i < g.ns; i += SEG_THREADS) X.idx[i] = iin[src_of(i)];
    const bool repaired = seg_sort_lds(L, g.ns, [&]() {
#pragma nounroll
        for (int i = threadIdx.x; i < g.ns; i += SEG_THREADS) {
            L.key[i] = kin[src_of(i)];
            L.pos[i] = (uint16_t)i;
        }
    });
    // splitter b = the sample at sorted place b * q - 1, b = 1 .. B - 1
#pragma nounroll
    for (int i = threadIdx.x; i < g.ns; i += SEG_THREADS) {
        const int to = seg_final_slot(L, i, g.ns, repaired) + 1;
        if (to % g.q == 0 && to / g.q < g.B) {
            X.spl_key[to / g.q] = L.key[i];
            X.spl_idx[to / g.q] = X.idx[L.pos[i] & 0x0fffu];
        }
    }
    __syncthreads();
    int32_t *__restrict__ bnd = a.bnd + (int64_t)(g.sb / SEG_TILE) * SEG_BND_PER_SLOT;
    for (int i = threadIdx.x; i < (g.B - 1) * g.m; i += SEG_THREADS) {
        const int b = 1 + i / g.m, t = i - (b - 1) * g.m;
        const int32_t tb = g.sb + t * SEG_TILE, te = min(tb + SEG_TILE, g.sb + g.n);
        // elements of tile t at or before the splitter in (key, position) order
        bnd[b * SEG_SPLIT_TILES + t] = rank_in(kin, iin, tb, te, X.spl_key[b], X.spl_idx[b] + 1);
    }
}

__global__ __launch_bounds__(SEG_THREADS, 5) void seg_bucket_kernel(SegArgs a)
{
    __shared__ SegLdsB X;
    SegLds &L = X.L;
    int32_t lo = 0, hi = a.n_cat;
    const int32_t tile = (int32_t)(blockIdx.x >> 1);
    while (hi - lo > 1) {
        const int32_t mid = (lo + hi) >> 1;
        if (a.tile_off[mid] <= tile) lo = mid; else hi = mid;
    }
    const int32_t k = lo;
    const SplitGeom g = split_geom(a, k);
    const int32_t b = (int32_t)blockIdx.x - 2 * a.tile_off[k];
    if (!g.on || b >= g.B) return;
    const uint64_t *__restrict__ kin = a.key[0];
    const int32_t *__restrict__ iin = a.idx[0];
    const int32_t *__restrict__ bnd = a.bnd + (int64_t)(g.sb / SEG_TILE) * SEG_BND_PER_SLOT;
    // ---- my piece of every tile (first wavefront: tiles <= 32)
    if (threadIdx.x < WAVE) {
        const int t = threadIdx.x;
        int32_t plo = 0, cnt = 0;
        if (t < g.m) {
            const int32_t len_t = min(SEG_TILE, g.n - t * SEG_TILE);
            plo = b > 0 ? bnd[b * SEG_SPLIT_TILES + t] : 0;
            const int32_t phi = b + 1 < g.B ? bnd[(b + 1) * SEG_SPLIT_TILES + t] : len_t;
            cnt = phi - plo;
        }
        int32_t inc = cnt, before = plo;
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            const int32_t v = __shfl_up(inc, off, WAVE), w = __shfl_xor(before, off, WAVE);
            if ((int)threadIdx.x >= off) inc += v;
            before += w;
        }
        if (t < g.m) {
            X.piece_lo[t] = plo;
            X.piece_at[t] = inc - cnt;
        }
        if (t == g.m - 1) X.piece_at[g.m] = inc;
        if (t == 0) X.out_base = g.sb + before;     // everything before the bucket
    }
    __syncthreads();
    const int32_t total = min(X.piece_at[g.m], (int32_t)SEG_TILE);   // (<= SEG_TILE by construction)
    if (total <= 0) return;
    auto src_of = [&](int i) {
        int t0 = 0, t1 = g.m;                  // piece holding gathered element i
        while (t1 - t0 > 1) {
            const int mid = (t0 + t1) >> 1;
            if (X.piece_at[mid] <= i) t0 = mid; else t1 = mid;
        }
        return g.sb + t0 * SEG_TILE + X.piece_lo[t0] + (i - X.piece_at[t0]);
    };
    const bool repaired = seg_sort_lds(L, total, [&]() {
#pragma nounroll
        for (int i = threadIdx.x; i < total; i += SEG_THREADS) {
            L.key[i] = kin[src_of(i)];
            L.pos[i] = (uint16_t)i;
        }
    });
    const int32_t out = X.out_base;
#pragma nounroll
    for (int i = threadIdx.x; i < total; i += SEG_THREADS) {
        const int32_t p = out + seg_final_slot(L, i, total, repaired);
        // (the input position is fetched now rather than carried through LDS)
        const int32_t d = iin[src_of((int)(L.pos[i] & 0x0fffu))];
        if (a.order) a.order[p] = d;
        if (a.dst) a.dst[d] = p;
    }
}

// ---------------------------------------------------------------------------
// Pairwise merge passes by MERGE PATH for categories of many tiles: a workgroup
// produces one tile-sized slice of a merged pair of runs.  Two searches along
// the slice's diagonals tell which pieces of the two runs feed it; the pieces
// are staged in LDS with coalesced loads, every thread then finds its own
// diagonal in LDS and merges 12 elements sequentially.  Ties take the element
// of the left run (it holds the earlier input positions), so the pass is
// stable with key comparisons only.  Per pass every element is read once and
// written once, instead of one 12-step search per element and other tile.
#define MP_PER 11       // outputs per thread: 256 * 11 = SEG_TILE

// number of elements of A among the first o outputs of merge(A, B)
template <class KA, class KB>
__device__ __forceinline__ int32_t merge_split(KA A, int32_t na, KB B, int32_t nb, int32_t o)
{
    int32_t lo = max(0, o - nb), hi = min(o, na);
    while (lo < hi) {
        const int32_t mid = (lo + hi) >> 1;
        if (A[mid] <= B[o - mid - 1]) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void seg_mpass_kernel(SegArgs a, int pass, int last,
                                                        int32_t longer_than)
{
    __shared__ uint64_t m_key[SEG_TILE];
    __shared__ int32_t m_idx[SEG_TILE];
    __shared__ int32_t m_split[2];
    static_assert(256 * MP_PER == SEG_TILE, "one workgroup = one tile-sized slice");
    int32_t lo = 0, hi = a.n_cat;
    while (hi - lo > 1) {
        const int32_t mid = (lo + hi) >> 1;
        if (a.tile_off[mid] <= (int32_t)blockIdx.x) lo = mid; else hi = mid;
    }
    const int32_t k = lo;
    const int32_t sb = a.cat_off[k], se = a.cat_off[k + 1];
    if (se - sb <= longer_than) return;     // one tile, or the splitter kernels' category
    const int32_t tb = sb + ((int32_t)blockIdx.x - a.tile_off[k]) * SEG_TILE;
    const int32_t len = min(SEG_TILE, se - tb);
    if (len <= 0) return;
    const int s = pass & 1;
    const uint64_t *__restrict__ kin = a.key[s];
    const int32_t *__restrict__ iin = a.idx[s];
    const int64_t L = (int64_t)SEG_TILE << pass;
    const int64_t base = sb + ((tb - sb) / (2 * L)) * (2 * L);
    const int32_t a0 = (int32_t)base, a1 = (int32_t)min(base + L, (int64_t)se);
    const int32_t b1 = (int32_t)min((int64_t)a1 + L, (int64_t)se);
    const int32_t na = a1 - a0, nb = b1 - a1;
    const int32_t o0 = tb - a0;
    auto emit = [&](int32_t p, uint64_t kx, int32_t ix) {
        if (last) {
            if (a.order) a.order[p] = ix;
            if (a.dst) a.dst[ix] = p;
        } else {
            a.key[s ^ 1][p] = kx;
            a.idx[s ^ 1][p] = ix;
        }
    };
    if (nb <= 0) {      // no partner run: the slice passes through
        for (int32_t i = threadIdx.x; i < len; i += 256) emit(tb + i, kin[tb + i], iin[tb + i]);
        return;
    }
    // ---- which pieces of the two runs make this slice
    if (threadIdx.x == 0) m_split[0] = merge_split(kin + a0, na, kin + a1, nb, o0);
    if (threadIdx.x == 64) m_split[1] = merge_split(kin + a0, na, kin + a1, nb, o0 + len);
    __syncthreads();
    const int32_t i0 = m_split[0], i1 = m_split[1];
    const int32_t j0 = o0 - i0;
    const int32_t pa = i1 - i0, pb = len - pa;      // piece lengths
    for (int32_t i = threadIdx.x; i < len; i += 256) {
        const int32_t src = i < pa ? a0 + i0 + i : a1 + j0 + (i - pa);
        m_key[i] = kin[src];
        m_idx[i] = iin[src];
    }
    __syncthreads();
    // ---- my MP_PER outputs
    const int32_t q0 = min(len, (int32_t)threadIdx.x * MP_PER);
    const int32_t q1 = min(len, q0 + MP_PER);
    if (q0 >= q1) return;
    const uint64_t *A = m_key, *B = m_key + pa;
    int32_t ia = merge_split(A, pa, B, pb, q0);
    int32_t ib = q0 - ia;
    for (int32_t q = q0; q < q1; q++) {
        const bool takeA = ib >= pb || (ia < pa && A[ia] <= B[ib]);
        const int32_t from = takeA ? ia : pa + ib;
        emit(tb + q, m_key[from], m_idx[from]);
        ia += takeA ? 1 : 0;
        ib += takeA ? 0 : 1;
    }
}

extern "C" size_t taoamd_sort_segments_workspace(int64_t n)
{
    if (n < 1) n = 1;
    return 2 * align256((size_t)n * 8) + 2 * align256((size_t)n * 4) +
           align256(((size_t)n / SEG_TILE + 2) * SEG_BND_PER_SLOT * 4) + 4096;
}

extern "C" int taoamd_sort_segments(int64_t n, int32_t n_cat,
                                    const int32_t *cat_off,
                                    const int32_t *tile_off, int32_t n_tiles,
                                    int32_t max_segment, const int32_t *dt_cat,
                                    const double *dt_score, int32_t *order,
                                    int32_t *dst, void *workspace,
                                    size_t workspace_bytes, void *stream)
{
    if (n == 0 || n_cat == 0 || n_tiles == 0) return TAOAMD_OK;
    if (n > 0x7fffffff) return TAOAMD_ERR_TOO_LARGE;
    if (!cat_off || !tile_off || !dt_cat || !dt_score || !workspace) return TAOAMD_ERR_ARG;
    if (workspace_bytes < taoamd_sort_segments_workspace(n)) return TAOAMD_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    unsigned char *w = (unsigned char *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    SegArgs a;
    a.cat_off = cat_off; a.tile_off = tile_off; a.cat = dt_cat; a.score = dt_score;
    a.order = order; a.dst = dst; a.n = n; a.n_cat = n_cat; a.n_tiles = n_tiles;
    a.key[0] = (uint64_t *)w; w += align256((size_t)n * 8);
    a.key[1] = (uint64_t *)w; w += align256((size_t)n * 8);
    a.idx[0] = (int32_t *)w;  w += align256((size_t)n * 4);
    a.idx[1] = (int32_t *)w;  w += align256((size_t)n * 4);
    a.bnd = (int32_t *)w;
    TAO_TIMED("seg_tile_kernel", s, seg_tile_kernel<<<(unsigned)n_tiles, SEG_THREADS, 0, s>>>(a));
    if (max_segment > SEG_TILE && max_segment <= (int64_t)SEG_KMERGE_TILES * SEG_TILE) {
        TAO_TIMED("seg_kmerge_kernel", s, seg_kmerge_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(a));
    } else if (max_segment > SEG_TILE) {
        // categories of up to SEG_SPLIT_TILES tiles: splitter buckets; longer
        // ones (if any): pairwise merge-path passes, their last one final
        TAO_TIMED("seg_split_kernel", s, seg_split_kernel<<<(unsigned)n_cat, SEG_THREADS, 0, s>>>(a));
        TAO_TIMED("seg_bucket_kernel", s, seg_bucket_kernel<<<2u * (unsigned)n_tiles, SEG_THREADS, 0, s>>>(a));
        const int32_t longer_than = SEG_SPLIT_TILES * SEG_TILE;
        if (max_segment > longer_than) {
            int passes = 0;
            for (int64_t L = SEG_TILE; L < max_segment; L <<= 1) passes++;
            for (int p = 0; p < passes; p++)
                TAO_TIMED("seg_mpass_kernel", s, seg_mpass_kernel<<<(unsigned)n_tiles, 256, 0, s>>>(a, p, p == passes - 1, longer_than));
        }
    }
    TAO_LAUNCH_CHECK();
    return TAOAMD_OK;
}

// ===========================================================================
// SAMPLE SORT of the categories (round 3): one pass moves an element next to
// its final neighbours, one wavefront finishes a bucket in registers.
//
// The tile + bucket path above sorts every element of a long category twice
// in LDS (its tile, then its splitter bucket): 0.67 + 0.11 + 0.48 ms of a
// 2.28 ms step at 2000 videos, the tile sort at 6 % of HBM speed.  Here:
//
//   ss_split_kernel    one workgroup per chunk (a category, or SS_CHUNK elements
//                      of a longer one): SS_OVER jittered samples per bucket,
//                      sorted in LDS (<= 1760 of them); every SS_OVER-th is a
//                      splitter -- an actual element (key, index), so equal
//                      scores split by their input order;
//   ss_scatter_kernel  one workgroup per 2816 elements: bucket of an element =
//                      number of splitters before it (binary search in LDS), its
//                      slot claimed with an LDS counter per tile and one global
//                      cursor add per (tile, bucket) -- no order is kept, none
//                      is needed: (key, index) is a total order;
//   ss_sort_kernel     one wavefront per bucket (<= SS_CAP = 1024 elements, 416
//                      on average): bitonic network over 1..16 elements per lane
//                      held in registers, partners fetched with DPP-class lane
//                      exchanges; writes order[] / dst[] at the bucket's base =
//                      the category's start + the counts of the buckets before it.
//
// A chunk of <= SS_DIRECT elements is ONE bucket read straight from the scores
// (the whole track level: ~30 .. 300 tracks per category).  Categories longer
// than SS_CHUNK = 8 tiles are sorted chunk by chunk into the merge buffers and
// finished by the merge-path passes above (seg_mpass_kernel from pass 3 on).
//
// Bucket sizes: SS_OVER = 32 uniformly placed samples per bucket make a
// bucket's size Gamma(32)-distributed around SS_TARGET; SS_CAP is 8 sigma out
// (~1e-11 per bucket).  A bucket that does overflow is not lost: every
// wavefront of its chunk sees the cursor beyond the limit and ranks its slice
// of the chunk by counting, straight from the scores (slow, correct).
// ===========================================================================
#define SS_CAP 1024
#define SS_FAST 512                    // most elements the one-word network takes
#ifndef SS_TARGET                      // (build.sh -DSS_TARGET=.. -DSS_CHUNK_TILES=..: experiments)
#define SS_TARGET 352
#define SS_CHUNK_TILES 16
#define SS_FIRST_MERGE_PASS 4          // log2(SS_CHUNK / SEG_TILE)
#endif
#define SS_OVER 32
#define SS_DIRECT 1024
#define SS_CHUNK (SS_CHUNK_TILES * SEG_TILE)   // 45056 = SS_MAXB * SS_TARGET
#define SS_MAXB 128
#define SS_HALF_CHUNK (SS_CHUNK / 2)   // chunks up to here: SS_OVER samples per bucket, beyond: half

struct SsChunk {                       // 32 bytes
    int32_t begin, n;                  // elements [begin, begin + n)
    int32_t bucket0, n_buckets;
    int32_t stile0;                    // first scatter tile (split chunks)
    int32_t final;                     // 1: order / dst, 0: merge buffers
    int32_t cat, pad;
};

struct SsArgs {
    const double *score;
    const SsChunk *chunks;
    const int32_t *split_list, *stile_chunk, *bucket_chunk;
    int32_t *cursor;                   // [n_buckets] elements claimed
    uint64_t *spl_key;                 // [n_buckets] splitter ahead of bucket b (b >= 1 in its chunk)
    int32_t *spl_idx;
    uint64_t *slot_key;                // [n_buckets][SS_CAP]
    int32_t *slot_idx;
    uint64_t *key_out;                 // merge buffers of the chunked categories
    int32_t *idx_out;
    int32_t *order, *dst;
    int32_t *redo;                     // buckets left to ss_redo_kernel, redo[-1] = their number
    int32_t n_buckets, n_stiles, cap_limit;
    int32_t dbg;                       // ablation switches (TAOAMD_SS_DBG, timing experiments)
};
// Ablation switches for timing experiments (results are wrong with any bit
// set): compiled in only by `bash build.sh -DTAOAMD_ABLATE`, constant 0 in the
// library that ships.
#ifdef TAOAMD_ABLATE
#define SS_DBG(a, bit) ((a).dbg & (bit))
#else
#define SS_DBG(a, bit) 0
#endif

static int g_ss_cap_limit = SS_CAP;

// samples per bucket: SS_OVER, half of it in the long chunks (<= 2048 samples
// either way: one wavefront sorts them, 32 per lane)
__host__ __device__ inline int ss_over(int32_t n) { return n > SS_HALF_CHUNK ? SS_OVER / 2 : SS_OVER; }

__device__ __forceinline__ uint32_t ss_mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

__device__ __forceinline__ uint64_t ss_shfl_xor64(uint64_t v, int j)
{
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, j, WAVE);
    const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), j, WAVE);
    return ((uint64_t)hi << 32) | lo;
}

// ---- one-word bitonic network ------------------------------------------------
// What is sorted is ONE 64-bit word per element,
//     1 << 61 | ((key - lo) >> shift) << 16 | (index - chunk begin)
// (lo = the bucket's smallest key, shift so that the key part fits 45 bits, a
// chunk has < 2^16 elements; ss_sort_bucket), which read as fp64 are positive normal numbers in
// [2^-511, 2^-510): v_min_f64 / v_max_f64 return one of their operands bit for
// bit, so a compare-exchange is two VALU instructions and no select -- against
// ~11 for a (64-bit key, 32-bit index) pair with its three conditional moves.
// Equal keys are ordered by the index bits: exactly the stable order.  Only
// when shift > 0 can two different keys share a key part; the sorted bucket is
// then checked against the full keys and, if a pair is out of order, ranked
// again with full comparisons (ss_rank_bucket).
//
// Layout: element e = lane * R + r.  "Flip" form of the network -- every merge
// starts with the mirror partner e ^ (size - 1), then e ^ j for j = size / 4
// .. 1, all ascending -- so partners below R are register pairs of one lane
// with compile-time roles, above that lane exchanges (xor / mirror masks).
__device__ __forceinline__ double ss_fmin(double a, double b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double ss_fmax(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// The partner lane's word.  Exchanges inside a row of 16 lanes are DPP moves
// (VALU: no LDS crossbar, no s_waitcnt) -- xor 1 / 2 / 3 are quad permutations,
// xor 7 / 15 the half-row / row mirrors, xor 8 a row rotation, xor 4 the two
// mirrors 7 and 3 in a row; the masks that cross rows (16, 31, 32, 63) go
// through ds_bpermute.  `m` is a compile-time constant wherever the unrolled
// network calls this.
template <int CTRL>
__device__ __forceinline__ double ss_dpp_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double ss_shfl_xor_f64(double v, int m)
{
#ifndef SS_NO_DPP
    switch (m) {
    case 1: return ss_dpp_f64<0xB1>(v);           // quad_perm [1, 0, 3, 2]
    case 2: return ss_dpp_f64<0x4E>(v);           // quad_perm [2, 3, 0, 1]
    case 3: return ss_dpp_f64<0x1B>(v);           // quad_perm [3, 2, 1, 0]
    case 4: return ss_dpp_f64<0x1B>(ss_dpp_f64<0x141>(v));
    case 7: return ss_dpp_f64<0x141>(v);          // row_half_mirror
    case 8: return ss_dpp_f64<0x128>(v);          // row_ror:8
    case 15: return ss_dpp_f64<0x140>(v);         // row_mirror
    default: break;
    }
#endif
    const int lo = __shfl_xor(__double2loint(v), m, WAVE);
    const int hi = __shfl_xor(__double2hiint(v), m, WAVE);
    return __hiloint2double(hi, lo);
}

template <int R>
__device__ __forceinline__ void ss_bitonic_packed(double (&p)[R], int lane)
{
    // (loops over the exponents: trip counts the unroller can see)
    constexpr int LOG_N = 6 + (R == 1 ? 0 : R == 2 ? 1 : R == 4 ? 2 : R == 8 ? 3 : R == 16 ? 4 : 5);
#pragma unroll
    for (int ls = 1; ls <= LOG_N; ls++) {
        const int size = 1 << ls;
        if (size <= R) {
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int r2 = r ^ (size - 1);
                if (r < r2) {
                    const double lo = ss_fmin(p[r], p[r2]), hi = ss_fmax(p[r], p[r2]);
                    p[r] = lo;
                    p[r2] = hi;
                }
            }
        } else {
            const int m = size / R - 1;
            const bool lower = (lane & (size / R / 2)) == 0;
            if (R == 1) {
                const double o = ss_shfl_xor_f64(p[0], m);
                p[0] = ((o < p[0]) == lower) ? o : p[0];
            }
            // my register r meets the partner lane's register R - 1 - r.
            // Round 6: the lower lanes keep the minimum, the upper lanes the
            // maximum -- v_min_f64 / v_max_f64 under the two halves of the exec
            // mask, ONE instruction per register and lane (the words are
            // distinct positive normal numbers: an operand comes back bit for
            // bit) where compare + two selects were three; the kernel is bound
            // by VALU issue (1424 instructions per bucket, 88 % of its cycles).
            // Four registers at a time: the partners' words are live together.
            if (R > 1) {
                constexpr int G = R / 2 < 2 ? 1 : 2;
#pragma unroll
                for (int r0 = 0; r0 < R / 2; r0 += G) {
                    double o1[G], o2[G];
#pragma unroll
                    for (int g = 0; g < G; g++) {
                        o1[g] = ss_shfl_xor_f64(p[R - 1 - (r0 + g)], m);
                        o2[g] = ss_shfl_xor_f64(p[r0 + g], m);
                    }
                    if (lower) {
#pragma unroll
                        for (int g = 0; g < G; g++) {
                            p[r0 + g] = ss_fmin(p[r0 + g], o1[g]);
                            p[R - 1 - (r0 + g)] = ss_fmin(p[R - 1 - (r0 + g)], o2[g]);
                        }
                    } else {
#pragma unroll
                        for (int g = 0; g < G; g++) {
                            p[r0 + g] = ss_fmax(p[r0 + g], o1[g]);
                            p[R - 1 - (r0 + g)] = ss_fmax(p[R - 1 - (r0 + g)], o2[g]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int lj = ls - 2; lj >= 0; lj--) {
            const int j = 1 << lj;
            if (j >= R) {
                const int m = j / R;
                const bool lower = (lane & m) == 0;
                constexpr int G = R < 4 ? R : 4;
#pragma unroll
                for (int r0 = 0; r0 < R; r0 += G) {
                    double o[G];
#pragma unroll
                    for (int g = 0; g < G; g++) o[g] = ss_shfl_xor_f64(p[r0 + g], m);
                    if (lower) {
#pragma unroll
                        for (int g = 0; g < G; g++) p[r0 + g] = ss_fmin(p[r0 + g], o[g]);
                    } else {
#pragma unroll
                        for (int g = 0; g < G; g++) p[r0 + g] = ss_fmax(p[r0 + g], o[g]);
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < R; r++) {
                    if ((r & j) == 0) {
                        const double lo = ss_fmin(p[r], p[r | j]), hi = ss_fmax(p[r], p[r | j]);
                        p[r] = lo;
                        p[r | j] = hi;
                    }
                }
            }
        }
    }
}

__device__ __forceinline__ uint64_t ss_shfl_up64(uint64_t v)
{
    const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)v, 1, WAVE);
    const uint32_t hi = (uint32_t)__shfl_up((int)(uint32_t)(v >> 32), 1, WAVE);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t ss_shfl_down64(uint64_t v)
{
    const uint32_t lo = (uint32_t)__shfl_down((int)(uint32_t)v, 1, WAVE);
    const uint32_t hi = (uint32_t)__shfl_down((int)(uint32_t)(v >> 32), 1, WAVE);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t ss_wave_min_u64(uint64_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint64_t o = ss_shfl_xor64(v, off);
        v = o < v ? o : v;
    }
    return v;
}
__device__ __forceinline__ uint64_t ss_wave_max_u64(uint64_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint64_t o = ss_shfl_xor64(v, off);
        v = o > v ? o : v;
    }
    return v;
}

#define SS_SPLIT_KEY_BITS 45              // ss_split_chunk: key part | position in the chunk
#define SS_IDX_BITS 16
#define SS_PAD 0x3fffffffffffffffull       // sorts last; still a normal fp64
__device__ __forceinline__ double ss_pack(uint64_t key, uint64_t lo, int shift, int32_t rel)
{
    return __longlong_as_double((long long)((1ull << 61) | (((key - lo) >> shift) << SS_IDX_BITS) |
                                            (uint64_t)rel));
}

// Splitters of one chunk: its samples sorted by ONE wavefront in registers.
// The samples only have to be ordered well enough to cut the chunk evenly: key
// parts that tie after the shift are not repaired here (the scatter kernel
// checks that the splitters ascend and hands the chunk to the counting path if
// not -- scores that differ in their last 19 bits only).
template <int R>
__device__ __forceinline__ void ss_split_chunk(const SsArgs &a, const SsChunk &c, int lane)
{
    const int32_t S = ss_over(c.n), B = c.n_buckets, m = S * B;
    const int32_t stride = c.n / m;
    // (the top 45 key bits -- sign, exponent, 33 bits of mantissa -- are plenty
    // to cut a chunk evenly; no pass over the samples for their range)
    double p[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int j = lane * R + r;               // samples in ascending position
        p[r] = __longlong_as_double((long long)SS_PAD);
        if (j < m) {
            const int32_t rel = j * stride +
                (int32_t)(ss_mix((uint32_t)j * 0x9e3779b9u ^ (uint32_t)c.begin) % (uint32_t)stride);
            p[r] = ss_pack(desc_key(a.score[c.begin + rel]), 0, 64 - SS_SPLIT_KEY_BITS, rel);
        }
    }
    ss_bitonic_packed<R>(p, lane);
    // splitter b = the sample of rank b * S - 1 (b = 1 .. B - 1)
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int q = lane * R + r + 1;
        if (q % S == 0 && q / S < B) {
            const int32_t at = c.begin + (int32_t)((uint64_t)__double_as_longlong(p[r]) &
                                                   ((1u << SS_IDX_BITS) - 1));
            a.spl_key[c.bucket0 + q / S] = desc_key(a.score[at]);
            a.spl_idx[c.bucket0 + q / S] = at;
        }
    }
}

// Round 4: the splitters of one chunk by a WORKGROUP.  One wavefront sorting a
// chunk's 1000-2000 samples in 16-32 registers per lane was 0.09 ms of pure
// latency at the head of the step's critical chain (1300 wavefronts on 1024
// SIMDs, nothing to hide behind).  Here the four wavefronts of a workgroup each
// sort a quarter of the samples (every fourth one: four runs of the same
// distribution; 8 registers per lane, a fifth of the network), park their
// sorted runs in LDS, and an element's rank among all samples is its place in
// its own run plus, by binary search, the elements of the other three runs
// below it -- the packed words are distinct (they end in the sample's position
// in the chunk), so there are no ties to break.
template <int R>
__device__ __forceinline__ void ss_split_chunk4(const SsArgs &a, const SsChunk &c, int lane,
                                                int wave, uint64_t (*runs)[WAVE * 8])
{
    const int32_t S = ss_over(c.n), B = c.n_buckets, m = S * B;
    const int32_t stride = c.n / m;
    double p[R];
    // every lane loads R samples whether it needs them or not (a lane past the
    // m-th sample reads element 0): the R loads are in flight together -- under
    // `if (j < m)` the compiler gave each load its own branch and waited for it,
    // R round trips to HBM in a row (what the one-wavefront kernel's 0.09 ms
    // mostly were: 32 of them)
    int32_t rel[R];
    double sc[R];
#ifndef SS_SAMPLE_SCATTERED
    // Round 5: the samples are taken as WHOLE 64-byte lines -- eight consecutive
    // scores at a jittered, line-aligned place of every window of 8 * stride
    // elements, eight neighbouring lanes a line -- instead of one score out of
    // every `stride` (~11): at one score per 88 bytes every 64-byte sector of
    // the score array was fetched to look at an eighth of it (VERDICT r4 #6:
    // 153 MB for ~1 MB of samples at 21 M rows).  Which elements are sampled
    // only steers how evenly the buckets come out, never the order: the
    // elements of a line are neighbours in the input order (one or two cells,
    // a cell's scores independent draws), and a bucket that does come out too
    // large takes the 16-register network or the counting path as before.
    const int32_t win = 8 * stride;                   // elements per sampled line
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int j = (r * 4 + wave) * WAVE + lane;   // four interleaved quarters of 64-sample blocks
        const int g = j >> 3, e = j & 7;
        // one of the lines that lie INSIDE window g (aligned in memory; a
        // window too short to hold one: its first eight elements), so no two
        // windows ever sample the same element
        const int32_t w0 = c.begin + g * win;
        const int32_t a0 = (w0 + 7) & ~7;
        const int32_t lines = (w0 + win - a0) >> 3;
        const int32_t at = lines > 0
            ? a0 + 8 * (int32_t)(ss_mix((uint32_t)g * 0x9e3779b9u ^ (uint32_t)c.begin) %
                                 (uint32_t)lines) - c.begin
            : g * win;
        rel[r] = j < m ? at + e : 0;
    }
#else
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int j = (lane * R + r) * 4 + wave;  // my quarter: samples j = wave (mod 4)
        rel[r] = j < m ? j * stride + (int32_t)(ss_mix((uint32_t)j * 0x9e3779b9u ^
                                                       (uint32_t)c.begin) % (uint32_t)stride)
                       : 0;
    }
#endif
#pragma unroll
    for (int r = 0; r < R; r++) sc[r] = a.score[c.begin + rel[r]];
#pragma unroll
    for (int r = 0; r < R; r++) {
#ifndef SS_SAMPLE_SCATTERED
        const int j = (r * 4 + wave) * WAVE + lane;
#else
        const int j = (lane * R + r) * 4 + wave;
#endif
        p[r] = j < m ? ss_pack(desc_key(sc[r]), 0, 64 - SS_SPLIT_KEY_BITS, rel[r])
                     : __longlong_as_double((long long)SS_PAD);
    }
    ss_bitonic_packed<R>(p, lane);
#pragma unroll
    for (int r = 0; r < R; r++) runs[wave][lane * R + r] = (uint64_t)__double_as_longlong(p[r]);
    // (slots past R * 64 of a run are never read: every search is bounded by R * 64)
    __syncthreads();
    // ranks of my R elements: the searches of all of them step together
    // (fixed depth log2(R * 64): R x 3 independent LDS reads per step)
    int32_t rank[R];
    uint64_t v[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        v[r] = (uint64_t)__double_as_longlong(p[r]);
        rank[r] = lane * R + r;
    }
#pragma unroll
    for (int o = 1; o < 4; o++) {
        const uint64_t *__restrict__ run = runs[(wave + o) & 3];
        int32_t lo[R];
#pragma unroll
        for (int r = 0; r < R; r++) lo[r] = 0;
#pragma unroll
        for (int half = R * WAVE / 2; half > 0; half >>= 1) {
#pragma unroll
            for (int r = 0; r < R; r++)            // (lo + half - 1 < R * 64 always)
                lo[r] += run[lo[r] + half - 1] < v[r] ? half : 0;
        }
#pragma unroll
        for (int r = 0; r < R; r++) rank[r] += lo[r] + (run[lo[r]] < v[r] ? 1 : 0);
    }
    // splitter b = the sample of rank b * S - 1: its full key from the scores
    // (all of a lane's candidate loads in flight together, as above)
    int32_t at[R];
    double ks[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        at[r] = c.begin + (int32_t)(v[r] & ((1u << SS_IDX_BITS) - 1));
        const int q = rank[r] + 1;
        if (v[r] == SS_PAD || q % S != 0 || q / S >= B) at[r] = -1;
    }
#pragma unroll
    for (int r = 0; r < R; r++) ks[r] = a.score[at[r] >= 0 ? at[r] : c.begin];
#pragma unroll
    for (int r = 0; r < R; r++) {
        if (at[r] >= 0) {
            const int q = rank[r] + 1;
            a.spl_key[c.bucket0 + q / S] = desc_key(ks[r]);
            a.spl_idx[c.bucket0 + q / S] = at[r];
        }
    }
}

__global__ __launch_bounds__(256) void ss_split4_kernel(SsArgs a, int32_t n_split)
{
    __shared__ uint64_t runs[4][WAVE * 8];
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const SsChunk c = a.chunks[a.split_list[blockIdx.x]];
    const int32_t m = ss_over(c.n) * c.n_buckets;  // <= 2048 samples, <= 512 a wavefront
    if (m <= 256) ss_split_chunk4<1>(a, c, lane, wave, runs);
    else if (m <= 512) ss_split_chunk4<2>(a, c, lane, wave, runs);
    else if (m <= 1024) ss_split_chunk4<4>(a, c, lane, wave, runs);
    else ss_split_chunk4<8>(a, c, lane, wave, runs);
}

__global__ __launch_bounds__(256) void ss_split_kernel(SsArgs a, int32_t n_split)
{
    const int lane = lane_id();
    const int32_t w = (int32_t)blockIdx.x * 4 + (int32_t)(threadIdx.x >> 6);
    if (w >= n_split) return;
    const SsChunk c = a.chunks[a.split_list[w]];
    const int32_t m = ss_over(c.n) * c.n_buckets;  // <= 2048 samples
    if (m <= 128) ss_split_chunk<2>(a, c, lane);
    else if (m <= 256) ss_split_chunk<4>(a, c, lane);
    else if (m <= 512) ss_split_chunk<8>(a, c, lane);
    else if (m <= 1024) ss_split_chunk<16>(a, c, lane);
    else ss_split_chunk<32>(a, c, lane);
}

__global__ __launch_bounds__(SEG_THREADS) void ss_scatter_kernel(SsArgs a)
{
    __shared__ uint64_t s_key[SS_MAXB];
    __shared__ int32_t s_idx[SS_MAXB];
    __shared__ int32_t s_cnt[SS_MAXB], s_base[SS_MAXB];
    // a chunk's tiles behind one L2: the buckets they fill are shared
    const uint32_t blk = xcd_block(blockIdx.x, gridDim.x);
    const SsChunk c = a.chunks[a.stile_chunk[blk]];
    const int32_t B = c.n_buckets;
    const int32_t t0 = c.begin + ((int32_t)blk - c.stile0) * SEG_TILE;
    const int32_t t1 = min(t0 + SEG_TILE, c.begin + c.n);
    if (threadIdx.x < SS_MAXB) {
        const int b = threadIdx.x;
        s_cnt[b] = 0;
        // entries past the last splitter precede nothing
        const bool real = b >= 1 && b < B;
        s_key[b] = real ? a.spl_key[c.bucket0 + b] : ~0ull;
        s_idx[b] = real ? a.spl_idx[c.bucket0 + b] : INT32_MAX;
    }
    __syncthreads();
    // the splitters must ascend (ss_split_chunk sorts the samples by a shortened
    // key): if two are out of order the chunk goes to the counting path
    if (threadIdx.x >= 1 && (int)threadIdx.x + 1 < B) {
        const int b = threadIdx.x;
        const bool ok = s_key[b] < s_key[b + 1] ||
                        (s_key[b] == s_key[b + 1] && s_idx[b] < s_idx[b + 1]);
        if (!ok) atomicMax(&a.cursor[c.bucket0], 1 << 30);
    }
    uint64_t kr[SEG_ROUNDS];
    int32_t br[SEG_ROUNDS], rr[SEG_ROUNDS];
    // all of a thread's scores first: the eleven loads in flight together
    // (round 4: loaded inside `if (i < t1)` the compiler gave every load its
    // own branch and s_waitcnt vmcnt(0) -- eleven round trips to HBM in a row)
#pragma unroll
    for (int r = 0; r < SEG_ROUNDS; r++) {
        const int32_t i = t0 + r * SEG_THREADS + (int32_t)threadIdx.x;
        kr[r] = desc_key(a.score[i < t1 ? i : t0]);
    }
#pragma unroll
    for (int r = 0; r < SEG_ROUNDS; r++) {
        const int32_t i = t0 + r * SEG_THREADS + (int32_t)threadIdx.x;
        br[r] = -1;
        if (i < t1) {
            const uint64_t k = kr[r];
            // splitters 1 .. B-1 that precede (k, i); an element equal to a
            // splitter closes the lower bucket.  (A fixed-depth search with all
            // rounds of a thread in step -- their LDS reads in flight together --
            // took 79 instead of 61 VGPRs and was no faster: 0.139 vs 0.143 ms.)
            int lo = 1, hi = B;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                const uint64_t sk = s_key[mid];
                const bool before = sk < k || (sk == k && s_idx[mid] < i);
                if (before) lo = mid + 1; else hi = mid;
            }
            br[r] = lo - 1;
            rr[r] = SS_DBG(a, 32) ? (int32_t)threadIdx.x >> 4 : atomicAdd(&s_cnt[lo - 1], 1);
        }
    }
    __syncthreads();
    if (threadIdx.x < B) {
        const int32_t n = s_cnt[threadIdx.x];
        s_base[threadIdx.x] = n ? atomicAdd(&a.cursor[c.bucket0 + threadIdx.x], n) : 0;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SEG_ROUNDS; r++) {
        if (br[r] >= 0) {
            const int32_t at = s_base[br[r]] + rr[r];
            if (at < SS_CAP && !SS_DBG(a, 16)) {
                const int64_t s = (int64_t)(c.bucket0 + br[r]) * SS_CAP + at;
                a.slot_key[s] = kr[r];
                a.slot_idx[s] = t0 + r * SEG_THREADS + (int32_t)threadIdx.x;
            }
        }
    }
}

// One bucket, one wavefront: the words
//     1 << 61 | ((key - lo) >> shift) << 26 | (index - chunk begin) << 10 | slot
// (35 bits of key part; a chunk has < 2^16 elements; slot = the place the
// element was read from, < 1024) through the network.  Equal key parts are
// ordered by their index bits: the stable order for equal keys.  A bucket's
// keys usually span more than 2^35 (one binade of fp64 scores holds 2^52 bit
// patterns), so bits are shifted out and two DIFFERENT keys may share a key
// part: every wavefront keeps the full keys of its bucket in LDS under the
// slot number, reads them back in sorted order and checks that they ascend --
// a bucket with a pair out of order is left to ss_rank_bucket.  (Round 3 first
// fetched the full keys of tied neighbours from the score array: tracks that
// give all their boxes one score make ties the common case, and the dependent
// gather was 0.10 of the kernel's 0.35 ms.)
// The slots are read lane-contiguous (slot = r * 64 + lane: any arrangement is
// as good as another BEFORE the network); the sorted place of register r of a
// lane is lane * R + r, so a lane stores R consecutive order[] entries at once.
#define SS_KEY_BITS 35
#define SS_E_BITS 10
typedef int32_t ss_i4 __attribute__((ext_vector_type(4), aligned(4)));
typedef int32_t ss_i2 __attribute__((ext_vector_type(2), aligned(4)));

__device__ __forceinline__ int ss_shift_for(uint64_t range)
{
    const int bits = range ? 64 - __clzll((long long)range) : 0;
    return bits > SS_KEY_BITS ? bits - SS_KEY_BITS : 0;
}

template <int R>
__device__ __forceinline__ bool ss_sort_bucket(const SsArgs &a, const SsChunk &c,
                                               int32_t count, int32_t out0, int lane,
                                               const uint64_t *__restrict__ sk,
                                               const int32_t *__restrict__ si,
                                               uint64_t *__restrict__ full)
{
    static_assert(WAVE * R <= (1 << SS_E_BITS), "slot bits");
    const double pad = __longlong_as_double((long long)SS_PAD);
    uint64_t k[R];
    int32_t x[R];
    uint64_t kmin = ~0ull, kmax = 0;
    // every slot's loads issued before the first is looked at (slots past the
    // bucket's count read slot 0): see ss_scatter_kernel
    if (sk == nullptr) {                 // a direct chunk: read from the scores
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int e = r * WAVE + lane;
            k[r] = desc_key(a.score[c.begin + (e < count ? e : 0)]);
            x[r] = c.begin + e;
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int e = r * WAVE + lane;
            k[r] = sk[e < count ? e : 0];
            x[r] = si[e < count ? e : 0];
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int e = r * WAVE + lane;
        if (e < count) {
            kmin = k[r] < kmin ? k[r] : kmin;
            kmax = k[r] > kmax ? k[r] : kmax;
        } else {
            k[r] = ~0ull;
            x[r] = INT32_MAX;
        }
        full[e] = k[r];
    }
    kmin = ss_wave_min_u64(kmin);
    kmax = ss_wave_max_u64(kmax);
    const int shift = ss_shift_for(kmax - kmin);
    double p[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int e = r * WAVE + lane;
        p[r] = e < count
            ? __longlong_as_double((long long)((1ull << 61) |
                  (((k[r] - kmin) >> shift) << (SS_IDX_BITS + SS_E_BITS)) |
                  ((uint64_t)(x[r] - c.begin) << SS_E_BITS) | (uint64_t)e))
            : pad;
    }
    if (!SS_DBG(a, 1)) ss_bitonic_packed<R>(p, lane);
    // (the wavefront's own LDS writes are complete before its reads: one
    // wavefront, program order; the waitcnt is the compiler's)
    bool bad = false;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint64_t bits = (uint64_t)__double_as_longlong(p[r]);
        const bool ok = lane * R + r < count;
        x[r] = c.begin + (int32_t)((bits >> SS_E_BITS) & ((1u << SS_IDX_BITS) - 1));
        k[r] = ok ? full[bits & ((1u << SS_E_BITS) - 1) & (WAVE * R - 1)] : ~0ull;
        if (r > 0) bad |= k[r] < k[r - 1];
    }
    if (!SS_DBG(a, 2)) {
        const uint64_t up = ss_shfl_up64(k[R - 1]);
        bad |= lane > 0 && k[0] < up;
        if (__ballot(bad) != 0) return false;
    }
    if (lane * R + R <= count && c.final && a.order && !SS_DBG(a, 8)) {
        int32_t *o = a.order + out0 + lane * R;
        if (R == 2) *(ss_i2 *)o = ss_i2{x[0], x[1]};
#pragma unroll
        for (int q = 0; q + 4 <= R; q += 4)
            *(ss_i4 *)(o + q) = ss_i4{x[q], x[q + 1], x[q + 2], x[q + 3]};
        if (R == 1) o[0] = x[0];
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int e = lane * R + r;
        if (e < count) {
            const int32_t pp = out0 + e;
            if (c.final) {
                if (a.order && !SS_DBG(a, 8) && lane * R + R > count) a.order[pp] = x[r];
                if (a.dst && !SS_DBG(a, 4)) a.dst[x[r]] = pp;
            } else {
                a.key_out[pp] = k[r];
                a.idx_out[pp] = x[r];
            }
        }
    }
    return true;
}

// A bucket whose shortened key parts tied between different keys: every
// element is ranked by counting, with full (key, index) comparisons (scores
// that agree in 45 leading bits of their range and differ below: rare, slow,
// correct).  Rounds of 64 elements, the bucket read through the caches.
__device__ __forceinline__ void ss_rank_bucket(const SsArgs &a, const SsChunk &c,
                                               int32_t count, int32_t out0, int lane,
                                               const uint64_t *__restrict__ sk,
                                               const int32_t *__restrict__ si)
{
    auto get = [&](int32_t e, uint64_t &k, int32_t &x) {
        if (sk == nullptr) {
            k = desc_key(a.score[c.begin + e]);
            x = c.begin + e;
        } else {
            k = sk[e];
            x = si[e];
        }
    };
    for (int32_t base = 0; base < count; base += WAVE) {
        const int32_t e = base + lane;
        uint64_t k = 0;
        int32_t x = 0, rank = 0;
        if (e < count) get(e, k, x);
        for (int32_t j = 0; j < count; j++) {
            uint64_t kj;
            int32_t xj;
            get(j, kj, xj);
            rank += (kj < k || (kj == k && xj < x)) ? 1 : 0;
        }
        if (e < count) {
            const int32_t pp = out0 + rank;
            if (c.final) {
                if (a.order) a.order[pp] = x;
                if (a.dst) a.dst[x] = pp;
            } else {
                a.key_out[pp] = k;
                a.idx_out[pp] = x;
            }
        }
    }
}

// bucket gb of the plan: its chunk, element count and first output place
// (false: a bucket of the chunk overflowed -- handled by ss_sort_kernel)
__device__ __forceinline__ bool ss_bucket_geom(const SsArgs &a, int64_t gb, int lane,
                                               SsChunk &c, int32_t &count, int32_t &before,
                                               bool &overflow)
{
    c = a.chunks[a.bucket_chunk[gb]];
    const int32_t b = (int32_t)gb - c.bucket0;
    count = c.n;
    before = 0;
    if (c.n_buckets > 1) {
        // (<= SS_MAXB = 128 buckets: two per lane)
        const int32_t m0 = lane < c.n_buckets ? a.cursor[c.bucket0 + lane] : 0;
        const int32_t m1 = lane + WAVE < c.n_buckets ? a.cursor[c.bucket0 + WAVE + lane] : 0;
        overflow = __ballot(m0 > a.cap_limit || m1 > a.cap_limit) != 0;
        count = b < WAVE ? __shfl(m0, b, WAVE) : __shfl(m1, b - WAVE, WAVE);
        int32_t pre = (lane < b ? m0 : 0) + (lane + WAVE < b ? m1 : 0);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) pre += __shfl_xor(pre, off, WAVE);
        before = pre;
    } else {
        overflow = count > a.cap_limit;
    }
    return !overflow;
}

// Buckets ss_sort_kernel did not finish: more than SS_FAST elements (a few
// per cent: the one-word network over 16 registers) or key parts that tied
// between different keys (ranking by counting).  A fixed grid walks the list whose length lives on
// the device.
__global__ __launch_bounds__(256) void ss_redo_kernel(SsArgs a)
{
    __shared__ uint64_t s_full[4][WAVE * 16];
    uint64_t *full = s_full[threadIdx.x >> 6];
    const int lane = lane_id();
    const int32_t n = a.redo[-1];
    for (int32_t i = (int32_t)blockIdx.x * 4 + (int32_t)(threadIdx.x >> 6); i < n;
         i += (int32_t)gridDim.x * 4) {
        const int64_t gb = a.redo[i];
        SsChunk c;
        int32_t count, before;
        bool overflow;
        if (!ss_bucket_geom(a, gb, lane, c, count, before, overflow) || count <= 0) continue;
        const uint64_t *sk = nullptr;
        const int32_t *si = nullptr;
        if (c.n_buckets > 1) {
            sk = a.slot_key + gb * SS_CAP;
            si = a.slot_idx + gb * SS_CAP;
        }
        const int32_t out0 = c.begin + before;
        if (count > SS_FAST && ss_sort_bucket<16>(a, c, count, out0, lane, sk, si, full)) continue;
        ss_rank_bucket(a, c, count, out0, lane, sk, si);
    }
}

__global__ __launch_bounds__(256) void ss_sort_kernel(SsArgs a)
{
    __shared__ uint64_t s_full[4][WAVE * 8];
    uint64_t *full = s_full[threadIdx.x >> 6];
    const int lane = lane_id();
    // a category's buckets behind one L2: dst[] of the category is scattered
    const int64_t gb = (int64_t)xcd_block(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);
    if (gb >= a.n_buckets) return;
    SsChunk c;
    int32_t count, before;
    bool overflow;
    ss_bucket_geom(a, gb, lane, c, count, before, overflow);
    const int32_t b = (int32_t)gb - c.bucket0;
    if (overflow) {
        // a bucket of this chunk outgrew its slots: wavefront b ranks the b-th
        // slice of the chunk by counting, straight from the scores
        const int32_t s0 = (int32_t)((int64_t)c.n * b / c.n_buckets);
        const int32_t s1 = (int32_t)((int64_t)c.n * (b + 1) / c.n_buckets);
        for (int32_t i = s0 + lane; i < s1; i += WAVE) {
            const uint64_t ki = desc_key(a.score[c.begin + i]);
            int32_t rank = 0;
            for (int32_t j = 0; j < c.n; j++) {
                const uint64_t kj = desc_key(a.score[c.begin + j]);
                rank += (kj < ki || (kj == ki && j < i)) ? 1 : 0;
            }
            const int32_t p = c.begin + rank, d = c.begin + i;
            if (c.final) {
                if (a.order) a.order[p] = d;
                if (a.dst) a.dst[d] = p;
            } else {
                a.key_out[p] = ki;
                a.idx_out[p] = d;
            }
        }
        return;
    }
    if (count <= 0) return;
    const int32_t out0 = c.begin + before;
    const uint64_t *sk = nullptr;
    const int32_t *si = nullptr;
    if (c.n_buckets > 1) {
        sk = a.slot_key + gb * SS_CAP;
        si = a.slot_idx + gb * SS_CAP;
    }
    bool done = false;
    if (count <= 64) done = ss_sort_bucket<1>(a, c, count, out0, lane, sk, si, full);
    else if (count <= 128) done = ss_sort_bucket<2>(a, c, count, out0, lane, sk, si, full);
    else if (count <= 256) done = ss_sort_bucket<4>(a, c, count, out0, lane, sk, si, full);
    else if (count <= SS_FAST) done = ss_sort_bucket<8>(a, c, count, out0, lane, sk, si, full);
    if (!done && lane == 0) a.redo[atomicAdd(&a.redo[-1], 1)] = (int32_t)gb;
}

// ---- the plan: chunks, buckets and scatter tiles from the category offsets
extern "C" int taoamd_sort_plan_host(int32_t n_cat, const int32_t *cat_off_host,
                                     int64_t *sizes, int32_t *chunks,
                                     int32_t *split_list, int32_t *stile_chunk,
                                     int32_t *bucket_chunk)
{
    if (n_cat < 0 || !cat_off_host || !sizes) return TAOAMD_ERR_ARG;
    const bool fill = chunks != nullptr;
    if (fill && (!split_list || !stile_chunk || !bucket_chunk)) return TAOAMD_ERR_ARG;
    int64_t nc = 0, ns = 0, nt = 0, nb = 0, merge = 0;
    for (int32_t k = 0; k < n_cat; k++) {
        const int64_t sb = cat_off_host[k], len = cat_off_host[k + 1] - sb;
        if (len < 0) return TAOAMD_ERR_ARG;
        if (len > SS_CHUNK) merge = 1;
        for (int64_t o = 0; o < len; o += SS_CHUNK) {
            const int32_t n = (int32_t)std::min<int64_t>(SS_CHUNK, len - o);
            const int32_t B = n <= SS_DIRECT ? 1 : (n + SS_TARGET - 1) / SS_TARGET;
            const int32_t tiles = B > 1 ? (n + SEG_TILE - 1) / SEG_TILE : 0;
            if (fill) {
                SsChunk c;
                c.begin = (int32_t)(sb + o); c.n = n; c.bucket0 = (int32_t)nb;
                c.n_buckets = B; c.stile0 = (int32_t)nt; c.final = len <= SS_CHUNK;
                c.cat = k; c.pad = 0;
                memcpy(chunks + 8 * nc, &c, sizeof c);
                if (B > 1) split_list[ns] = (int32_t)nc;
                for (int32_t t = 0; t < tiles; t++) stile_chunk[nt + t] = (int32_t)nc;
                for (int32_t b = 0; b < B; b++) bucket_chunk[nb + b] = (int32_t)nc;
            }
            nc++;
            ns += B > 1;
            nt += tiles;
            nb += B;
        }
    }
    if (nb >= INT32_MAX / 2) return TAOAMD_ERR_TOO_LARGE;
    sizes[0] = nc; sizes[1] = ns; sizes[2] = nt; sizes[3] = nb; sizes[4] = merge;
    return TAOAMD_OK;
}

extern "C" size_t taoamd_sort_sampled_workspace(int64_t n, int64_t n_buckets, int32_t merge)
{
    if (n < 1) n = 1;
    if (n_buckets < 1) n_buckets = 1;
    size_t b = align256((size_t)n_buckets * 4 + 256)           // cursor, redo count
               + align256((size_t)n_buckets * 4)                // redo list
               + align256((size_t)n_buckets * 8) + align256((size_t)n_buckets * 4)
               + align256((size_t)n_buckets * SS_CAP * 8) + align256((size_t)n_buckets * SS_CAP * 4);
    if (merge) b += 2 * align256((size_t)n * 8) + 2 * align256((size_t)n * 4);
    return b + 4096;
}

// The splitter kernel is 0.06 ms of latency at the head of the image level's
// chain; a caller that starts other work beside the sort (the track level's 3D
// IoU: 11 k workgroups that take every wave slot they find) wants that work to
// wait for the splitters, not for the whole sort -- in a kernel trace of the
// step the splitters' workgroups, 139 VGPRs each, took 347 us to get their turn
// beside it.  `event` (a hipEvent_t) is recorded on the sort's stream right
// behind the splitter kernel of the calling thread's NEXT taoamd_sort_sampled;
// NULL = none.
static thread_local hipEvent_t g_ss_notify = nullptr;
extern "C" int taoamd_sort_sampled_notify(void *event)
{
    g_ss_notify = (hipEvent_t)event;
    return TAOAMD_OK;
}

extern "C" int taoamd_sort_sampled_cap_limit(int32_t limit)
{
    g_ss_cap_limit = limit > 0 && limit < SS_CAP ? limit : SS_CAP;
    return TAOAMD_OK;
}

extern "C" int taoamd_sort_sampled(int64_t n, int32_t n_cat, const int32_t *cat_off,
                                   const int32_t *tile_off, int32_t n_tiles,
                                   int32_t max_segment, const double *dt_score,
                                   int32_t n_chunks, const int32_t *chunks,
                                   int32_t n_split, const int32_t *split_list,
                                   int32_t n_stiles, const int32_t *stile_chunk,
                                   int32_t n_buckets, const int32_t *bucket_chunk,
                                   int32_t *order, int32_t *dst, void *workspace,
                                   size_t workspace_bytes, void *stream)
{
    if (n == 0 || n_cat == 0 || n_chunks == 0) return TAOAMD_OK;
    if (n > 0x7fffffff) return TAOAMD_ERR_TOO_LARGE;
    if (!cat_off || !dt_score || !chunks || !bucket_chunk || !workspace ||
        (n_split > 0 && (!split_list || !stile_chunk)))
        return TAOAMD_ERR_ARG;
    const int merge = max_segment > SS_CHUNK;
    if (merge && (!tile_off || n_tiles <= 0)) return TAOAMD_ERR_ARG;
    if (workspace_bytes < taoamd_sort_sampled_workspace(n, n_buckets, merge))
        return TAOAMD_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    unsigned char *w = (unsigned char *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    SsArgs a;
    a.score = dt_score; a.chunks = (const SsChunk *)chunks; a.split_list = split_list;
    a.stile_chunk = stile_chunk; a.bucket_chunk = bucket_chunk;
    a.order = order; a.dst = dst; a.n_buckets = n_buckets; a.n_stiles = n_stiles;
    a.cap_limit = g_ss_cap_limit;
#ifdef TAOAMD_ABLATE
    static const int dbg_env = getenv("TAOAMD_SS_DBG") ? atoi(getenv("TAOAMD_SS_DBG")) : 0;
    a.dbg = dbg_env;
#else
    a.dbg = 0;
#endif
    a.cursor = (int32_t *)w;    w += align256((size_t)n_buckets * 4 + 256);
    a.redo = (int32_t *)w;      w += align256((size_t)n_buckets * 4);
    // (the redo count sits in the last int of the cursor block, right ahead of the list)
    a.spl_key = (uint64_t *)w;  w += align256((size_t)n_buckets * 8);
    a.spl_idx = (int32_t *)w;   w += align256((size_t)n_buckets * 4);
    a.slot_key = (uint64_t *)w; w += align256((size_t)n_buckets * SS_CAP * 8);
    a.slot_idx = (int32_t *)w;  w += align256((size_t)n_buckets * SS_CAP * 4);
    SegArgs m;
    m.cat_off = cat_off; m.tile_off = tile_off; m.cat = nullptr; m.score = dt_score;
    m.order = order; m.dst = dst; m.n = n; m.n_cat = n_cat; m.n_tiles = n_tiles;
    m.key[0] = m.key[1] = nullptr; m.idx[0] = m.idx[1] = nullptr; m.bnd = nullptr;
    a.key_out = nullptr; a.idx_out = nullptr;
    if (merge) {
        m.key[0] = (uint64_t *)w; w += align256((size_t)n * 8);
        m.key[1] = (uint64_t *)w; w += align256((size_t)n * 8);
        m.idx[0] = (int32_t *)w;  w += align256((size_t)n * 4);
        m.idx[1] = (int32_t *)w;  w += align256((size_t)n * 4);
        a.key_out = m.key[SS_FIRST_MERGE_PASS & 1];
        a.idx_out = m.idx[SS_FIRST_MERGE_PASS & 1];
    }
    TAO_HIP(hipMemsetAsync(a.cursor, 0, align256((size_t)n_buckets * 4 + 256), s));
    if (n_split > 0) {
        // (TAOAMD_SS_SPLIT=1: one wavefront per chunk, the round-3 kernel, for A/B timing)
        static const bool one_wave = getenv("TAOAMD_SS_SPLIT") && atoi(getenv("TAOAMD_SS_SPLIT")) == 1;
        if (one_wave)
            TAO_TIMED("ss_split_kernel", s, ss_split_kernel<<<(unsigned)((n_split + 3) / 4), 256, 0, s>>>(a, n_split));
        else
            TAO_TIMED("ss_split_kernel", s, ss_split4_kernel<<<(unsigned)n_split, 256, 0, s>>>(a, n_split));
        if (g_ss_notify) TAO_HIP(hipEventRecord(g_ss_notify, s));
        TAO_TIMED("ss_scatter_kernel", s, ss_scatter_kernel<<<(unsigned)n_stiles, SEG_THREADS, 0, s>>>(a));
    } else if (g_ss_notify) {
        TAO_HIP(hipEventRecord(g_ss_notify, s));
    }
    g_ss_notify = nullptr;
    TAO_TIMED("ss_sort_kernel", s, ss_sort_kernel<<<(unsigned)((n_buckets + 3) / 4), 256, 0, s>>>(a));
    TAO_TIMED("ss_redo_kernel", s, ss_redo_kernel<<<(unsigned)std::min<int64_t>(1024, (n_buckets + 3) / 4), 256, 0, s>>>(a));
    if (merge) {
        int passes = 0;
        for (int64_t L = SEG_TILE; L < max_segment; L <<= 1) passes++;
        for (int p = SS_FIRST_MERGE_PASS; p < passes; p++)
            TAO_TIMED("seg_mpass_kernel", s, seg_mpass_kernel<<<(unsigned)n_tiles, 256, 0, s>>>(m, p, p == passes - 1, SS_CHUNK));
    }
    TAO_LAUNCH_CHECK();
    return TAOAMD_OK;
}
