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
