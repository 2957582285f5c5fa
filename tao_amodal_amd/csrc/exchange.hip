// Result exchange of the category-partitioned multi-GPU evaluation (gfx950).
//
// A rank that swept the categories of its block holds, per (category, range)
// row, the table val[t][j] = precision at recall threshold j (reference
// lvis_amodal/eval.py:398-417 == tao_amodal/eval.py:555-573).  Shipping the
// 101 columns of every row to every rank would move 58 MB (image level) +
// 194 MB (track level) per pass, most of it redundant: column j depends on j
// only through cj[j] = the TP count at which recall first reaches
// rec_thrs[j] (recall_crossing), a function of num_gt alone.  A row with n
// ground truths has at most min(n, 100) + 1 distinct crossings, so its 101
// columns hold at most that many distinct values ("levels"), in runs.  Track
// level rows typically have a handful of ground-truth tracks.
//
// The exchange therefore ships, per rank, ONE chunk
//
//     [ num_gt of the block rows | rec of the block rows | levels ]
//
// (the header also carries each row's offset into `levels`) where `levels`
// stores for every row with num_gt > 0 and every IoU threshold only the first
// column of each run.  Receivers rebuild the run map from the
// num_gt in the chunk header with the very fp64 comparison the sweep used
// (recall_crossing), expand, and write the reference layout
// precision[T][R][K][A] / recall[T][K][A] directly (the transpose that
// acc_finalize_kernel does on a single GPU), so expansion costs no extra pass.
// Copying runs is exact by construction: no arithmetic touches the values --
// they travel as the sweep's (tp, n) records (common.hpp) and become
// tp / (n + eps) when the receiver writes the reference layout.
//
// One chunk per rank and equal chunk sizes make the whole exchange a single
// in-place all_gather_into_tensor per evaluator.  The chunk capacity is fixed
// when the plan is built (taoamd_exchange_sizes on the gathered num_gt; the
// ground truth of a plan does not change between passes); an overflow flag
// guards against misuse.
#include "common.hpp"

using namespace taoamd;

namespace {

__host__ __device__ inline size_t ex_align(size_t x) { return (x + 255) & ~(size_t)255; }

struct NumSrc {             // where the num_gt of a global row lives
    const unsigned char *base;
    int64_t block_stride;   // bytes between the tables of consecutive blocks
    int64_t valid_rows;     // rows at or beyond are padding (num_gt = 0)
    int32_t block_rows;
};

__device__ __forceinline__ int32_t num_of(const NumSrc &s, int64_t row)
{
    if (row >= s.valid_rows) return 0;
    const int64_t b = row / s.block_rows, i = row - b * s.block_rows;
    return *(const int32_t *)(s.base + b * s.block_stride + i * 4);
}

struct ExWs {
    uint8_t *dmap;    // [rows][N_REC] run index of column j
    int32_t *nd;      // [rows] runs of the row (0: no ground truth)
    int32_t *off;     // [rows] first level of the row inside its block's levels
    int64_t *totals;  // [world] levels per block
};

// one wave per row: run index of each recall column
__global__ __launch_bounds__(256) void ex_levels_kernel(NumSrc src, int64_t row0,
                                                         int64_t row1, ExWs w,
                                                         RecThr rec)
{
    const int lane = lane_id();
    const int64_t row = row0 + (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= row1) return;
    const int32_t ng = num_of(src, row);
    if (ng <= 0) {
        if (lane == 0) w.nd[row] = 0;
        return;
    }
    const int j0 = lane, j1 = lane + WAVE;
    const bool has1 = j1 < N_REC;
    const int32_t c0 = recall_crossing(rec.v[j0], ng);
    const int32_t p0 = j0 > 0 ? recall_crossing(rec.v[j0 - 1], ng) : c0;
    const int32_t c1 = has1 ? recall_crossing(rec.v[j1], ng) : 0;
    const int32_t p1 = has1 ? recall_crossing(rec.v[j1 - 1], ng) : 0;
    const uint64_t f0 = __ballot(j0 > 0 && c0 != p0);
    const uint64_t f1 = __ballot(has1 && c1 != p1);
    const uint64_t incl = lane == 63 ? ~0ull : ((2ull << lane) - 1);
    const int d0 = __popcll(f0 & incl);
    const int d1 = __popcll(f0) + __popcll(f1 & incl);
    w.dmap[row * N_REC + j0] = (uint8_t)d0;
    if (has1) w.dmap[row * N_REC + j1] = (uint8_t)d1;
    if (lane == 0) w.nd[row] = __popcll(f0) + __popcll(f1) + 1;
}

// one workgroup per block of rows: exclusive scan of nd * T.  Each of the 16
// waves scans a contiguous share of the rows 64 at a time (coalesced, carry in
// a register), the wave totals meet once in LDS, a second sweep adds the base.
__global__ __launch_bounds__(1024) void ex_offsets_kernel(int32_t block_first,
                                                           int32_t block_rows,
                                                           int64_t capacity, ExWs w,
                                                           int32_t *overflow)
{
    __shared__ int64_t s_wave[16];
    const int b = block_first + blockIdx.x;
    const int64_t r0 = (int64_t)b * block_rows;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int32_t per = ((block_rows + 15) / 16 + WAVE - 1) / WAVE * WAVE;
    const int32_t i0 = min(block_rows, wave * per), i1 = min(block_rows, i0 + per);
    int64_t carry = 0;
    for (int32_t i = i0 + lane; i - lane < i1; i += WAVE) {
        const int64_t v = i < i1 ? (int64_t)w.nd[r0 + i] * N_THR : 0;
        int64_t x = v;
        for (int d = 1; d < WAVE; d <<= 1) {
            const int64_t y = __shfl_up(x, d);
            if (lane >= d) x += y;
        }
        if (i < i1) w.off[r0 + i] = (int32_t)(carry + x - v);
        carry += __shfl(x, WAVE - 1);
    }
    if (lane == 0) s_wave[wave] = carry;
    __syncthreads();
    int64_t base = 0;
    for (int q = 0; q < wave; q++) base += s_wave[q];
    if (base != 0)
        for (int32_t i = i0 + lane; i < i1; i += WAVE) w.off[r0 + i] += (int32_t)base;
    if (threadIdx.x == 1023) {
        const int64_t total = base + carry;
        w.totals[b] = total;
        if (total > capacity && overflow) atomicOr(overflow, 1);
    }
}

struct Chunk {
    size_t hdr_bytes, rec_bytes, bytes;
};

inline Chunk chunk_layout(int32_t block_cats, int32_t n_rng, int64_t capacity)
{
    const size_t rows = (size_t)block_cats * n_rng;
    Chunk c;
    c.hdr_bytes = ex_align(rows * 8);     // num_gt[rows], off[rows]
    c.rec_bytes = ex_align(rows * N_THR * 8);
    c.bytes = c.hdr_bytes + c.rec_bytes + ex_align((size_t)capacity * 8);
    return c;
}

struct PackArgs {
    int64_t row0;          // first global row of the block
    int32_t block_rows;
    int64_t valid_rows;    // n_cat * n_rng
    const int32_t *num_gt; // local table [n_cat][n_rng]
    const uint64_t *val;
    const double *rec;
    int32_t *hdr;
    double *rec_out;
    uint64_t *levels;
    int64_t capacity;
    ExWs w;
};

// thread = (row of the block, recall column j): if j starts a run, its value
// at every IoU threshold is copied to the run's level slot (lanes = consecutive
// columns: the reads of a threshold's row and the writes of its levels are
// contiguous).  Before (round 1): a thread per (row, threshold, column), ten
// times the threads for the same copies (52 us in the by-video step).
__global__ __launch_bounds__(256) void ex_pack_kernel(PackArgs a)
{
    const int64_t COLS = (int64_t)N_THR * N_REC;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i = idx / 128;                   // 128 threads per row: 101 columns
    if (i >= a.block_rows) return;
    const int j = (int)(idx - i * 128);
    const int64_t row = a.row0 + i;
    const int32_t ng = row < a.valid_rows ? a.num_gt[row] : 0;
    if (j == 0) {
        a.hdr[i] = ng;
        a.hdr[a.block_rows + i] = ng > 0 ? a.w.off[row] : 0;
    }
    if (j < N_THR) a.rec_out[i * N_THR + j] = ng > 0 ? a.rec[row * N_THR + j] : -1.0;
    if (ng <= 0 || j >= N_REC) return;
    const uint8_t *dm = a.w.dmap + row * N_REC;
    const int d = dm[j];
    if (j > 0 && dm[j - 1] == d) return;
    const int32_t nd = a.w.nd[row];
    const int64_t slot0 = (int64_t)a.w.off[row] + d;
#pragma unroll
    for (int t = 0; t < N_THR; t++) {
        const int64_t slot = slot0 + (int64_t)t * nd;
        if (slot < a.capacity) a.levels[slot] = a.val[row * COLS + (int64_t)t * N_REC + j];
    }
}

struct UnpackArgs {
    int32_t n_cat, n_rng, block_rows;
    const unsigned char *chunks;
    size_t chunk_bytes, hdr_bytes, rec_bytes;
    int64_t capacity;
    int32_t *num_gt_out;
    double *precision, *recall;
    int32_t *overflow;
    ExWs w;
};

// levels -> precision[T][R][K][A], rec -> recall[T][K][A]: the layout pass of
// acc_finalize_kernel (64 rows x 64 columns per workgroup, rows without
// evaluated ground truth never read, every column stored as full-wavefront
// 512-byte runs) with the run expansion on its load side.  The run maps come
// from ex_levels_kernel on the received headers, the row offsets from the
// headers themselves (no scan on the receiving side).  (Round 2; the 32 x 32
// tiling it replaces took 56 us where acc_finalize_kernel takes 21.)
__global__ __launch_bounds__(256) void ex_unpack_kernel(UnpackArgs a)
{
    __shared__ double tile[64][65];
    const int64_t KR = (int64_t)a.n_cat * a.n_rng;
    const int64_t COLS = (int64_t)N_THR * N_REC;
    const int64_t row0 = (int64_t)blockIdx.x * 64;
    const int64_t col0 = (int64_t)blockIdx.y * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ncol = (int)min((int64_t)64, COLS - col0);
    const int64_t orow = row0 + lane;                    // lane = row of the tile
    const bool olive = orow < KR && a.w.nd[orow] > 0;
    const uint64_t live_rows = __ballot(olive);          // same in every wavefront
    if (live_rows != 0) {
        const int64_t col = min(col0 + lane, COLS - 1);
        const int t = (int)(col / N_REC), j = (int)(col - (int64_t)t * N_REC);
        // rows i = wave (mod 4) are mine to fetch
        for (uint64_t m = live_rows & (0x1111111111111111ull << wave); m != 0; m &= m - 1) {
            const int idx = __builtin_ctzll(m);
            const int64_t row = row0 + idx;
            const int32_t nd = a.w.nd[row];
            const int64_t b = row / a.block_rows, li = row - b * a.block_rows;
            const unsigned char *ch = a.chunks + b * a.chunk_bytes;
            const uint64_t *lv = (const uint64_t *)(ch + a.hdr_bytes + a.rec_bytes);
            const int32_t off = ((const int32_t *)ch)[a.block_rows + li];
            const int64_t slot = (int64_t)off + (int64_t)t * nd + a.w.dmap[row * N_REC + j];
            double v = 0.0;
            if (slot < a.capacity) v = pr_value(lv[slot]);
            else if (a.overflow) atomicOr(a.overflow, 1);
            tile[idx][lane] = v;
        }
        __syncthreads();
    }
    if (orow < KR)
        for (int c = wave; c < ncol; c += 4)
            a.precision[(col0 + c) * KR + orow] = olive ? tile[lane][c] : -1.0;
    if (blockIdx.y == 0) {
        for (int i = threadIdx.x; i < 64 * N_THR; i += 256) {
            const int64_t row = row0 + i / N_THR;
            const int t = i % N_THR;
            if (row >= KR) continue;
            const int64_t b = row / a.block_rows, li = row - b * a.block_rows;
            const unsigned char *ch = a.chunks + b * a.chunk_bytes;
            a.recall[(int64_t)t * KR + row] =
                ((const double *)(ch + a.hdr_bytes))[li * N_THR + t];
            if (t == 0 && a.num_gt_out)
                a.num_gt_out[row] = ((const int32_t *)ch)[li];
        }
    }
}

size_t ws_bytes(int64_t rows, int32_t world)
{
    return ex_align((size_t)rows * N_REC) + 2 * ex_align((size_t)rows * 4) +
           ex_align((size_t)world * 8) + 256;
}

ExWs carve(void *workspace, int64_t rows, int32_t world)
{
    unsigned char *p = (unsigned char *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    ExWs w;
    w.dmap = p; p += ex_align((size_t)rows * N_REC);
    w.nd = (int32_t *)p; p += ex_align((size_t)rows * 4);
    w.off = (int32_t *)p; p += ex_align((size_t)rows * 4);
    w.totals = (int64_t *)p;
    return w;
}

bool bad_shape(int32_t block_cats, int32_t n_rng, int32_t world)
{
    return block_cats <= 0 || n_rng < 1 || n_rng > 32 || world < 1 ||
           (int64_t)block_cats * n_rng * N_THR * N_REC > 0x7fffffffll;
}

}  // namespace

extern "C" size_t taoamd_exchange_chunk_bytes(int32_t block_cats, int32_t n_rng,
                                              int64_t capacity)
{
    return chunk_layout(block_cats, n_rng, capacity).bytes;
}

extern "C" size_t taoamd_exchange_workspace(int32_t block_cats, int32_t n_rng,
                                            int32_t world)
{
    return ws_bytes((int64_t)block_cats * n_rng * world, world);
}

extern "C" int taoamd_exchange_sizes(int32_t block_cats, int32_t n_rng,
                                     int32_t world, const int32_t *num_gt,
                                     int64_t *totals, void *workspace,
                                     size_t workspace_bytes, void *stream)
{
    if (bad_shape(block_cats, n_rng, world) || !num_gt || !totals || !workspace)
        return TAOAMD_ERR_ARG;
    const int32_t BR = block_cats * n_rng;
    const int64_t rows = (int64_t)BR * world;
    if (workspace_bytes < ws_bytes(rows, world)) return TAOAMD_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    ExWs w = carve(workspace, rows, world);
    NumSrc src{(const unsigned char *)num_gt, (int64_t)BR * 4, rows, BR};
    TAO_TIMED("ex_levels_kernel", s, ex_levels_kernel<<<(unsigned)((rows + 3) / 4), 256, 0, s>>>(src, 0, rows, w,
                                                                rec_thr()));
    TAO_TIMED("ex_offsets_kernel", s, ex_offsets_kernel<<<world, 1024, 0, s>>>(0, BR, INT64_MAX, w, nullptr));
    TAO_LAUNCH_CHECK();
    TAO_HIP(hipMemcpyAsync(totals, w.totals, (size_t)world * 8,
                           hipMemcpyDeviceToDevice, s));
    return TAOAMD_OK;
}

extern "C" int taoamd_exchange_pack(int32_t n_cat, int32_t n_rng,
                                    int32_t block_cats, int32_t world,
                                    int32_t rank, const int32_t *num_gt,
                                    const double *val, const double *rec,
                                    void *chunk, int64_t capacity,
                                    int32_t *overflow, void *workspace,
                                    size_t workspace_bytes, int32_t maps_ready,
                                    void *stream)
{
    if (bad_shape(block_cats, n_rng, world) || n_cat <= 0 || rank < 0 ||
        rank >= world || (int64_t)block_cats * world < n_cat || capacity < 0)
        return TAOAMD_ERR_ARG;
    if (!num_gt || !val || !rec || !chunk || !workspace) return TAOAMD_ERR_ARG;
    const int32_t BR = block_cats * n_rng;
    const int64_t rows = (int64_t)BR * world;
    if (workspace_bytes < ws_bytes(rows, world)) return TAOAMD_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    ExWs w = carve(workspace, rows, world);
    const int64_t valid = (int64_t)n_cat * n_rng;
    const int64_t r0 = (int64_t)rank * BR;
    // the local table is addressed by global row: one "block" spanning it all
    NumSrc src{(const unsigned char *)num_gt, 0, valid, (int32_t)0x7fffffff};
    if (!maps_ready) {
        TAO_TIMED("ex_levels_kernel", s, ex_levels_kernel<<<(unsigned)((BR + 3) / 4), 256, 0, s>>>(src, r0, r0 + BR, w,
                                                                  rec_thr()));
        TAO_TIMED("ex_offsets_kernel", s, ex_offsets_kernel<<<1, 1024, 0, s>>>(rank, BR, capacity, w, overflow));
    }
    const Chunk c = chunk_layout(block_cats, n_rng, capacity);
    PackArgs a;
    a.row0 = r0; a.block_rows = BR; a.valid_rows = valid;
    a.num_gt = num_gt; a.val = (const uint64_t *)val; a.rec = rec;
    a.hdr = (int32_t *)chunk;
    a.rec_out = (double *)((unsigned char *)chunk + c.hdr_bytes);
    a.levels = (uint64_t *)((unsigned char *)chunk + c.hdr_bytes + c.rec_bytes);
    a.capacity = capacity; a.w = w;
    const int64_t threads = (int64_t)BR * 128;
    TAO_TIMED("ex_pack_kernel", s, ex_pack_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, s>>>(a));
    TAO_LAUNCH_CHECK();
    return TAOAMD_OK;
}

extern "C" int taoamd_exchange_unpack(int32_t n_cat, int32_t n_rng,
                                      int32_t block_cats, int32_t world,
                                      const void *chunks, int64_t capacity,
                                      int32_t *num_gt_out, double *precision,
                                      double *recall, int32_t *overflow,
                                      void *workspace, size_t workspace_bytes,
                                      int32_t maps_ready, void *stream)
{
    if (bad_shape(block_cats, n_rng, world) || n_cat <= 0 ||
        (int64_t)block_cats * world < n_cat || capacity < 0)
        return TAOAMD_ERR_ARG;
    if (!chunks || !precision || !recall || !workspace) return TAOAMD_ERR_ARG;
    const int32_t BR = block_cats * n_rng;
    const int64_t rows = (int64_t)BR * world;
    if (workspace_bytes < ws_bytes(rows, world)) return TAOAMD_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const Chunk c = chunk_layout(block_cats, n_rng, capacity);
    UnpackArgs a;
    a.w = carve(workspace, rows, world);
    NumSrc src{(const unsigned char *)chunks, (int64_t)c.bytes, rows, BR};
    if (!maps_ready)
        TAO_TIMED("ex_levels_kernel", s, ex_levels_kernel<<<(unsigned)((rows + 3) / 4), 256, 0, s>>>(src, 0, rows, a.w,
                                                                    rec_thr()));
    a.n_cat = n_cat; a.n_rng = n_rng; a.block_rows = BR;
    a.chunks = (const unsigned char *)chunks;
    a.chunk_bytes = c.bytes; a.hdr_bytes = c.hdr_bytes; a.rec_bytes = c.rec_bytes;
    a.capacity = capacity; a.num_gt_out = num_gt_out;
    a.precision = precision; a.recall = recall; a.overflow = overflow;
    const int64_t KR = (int64_t)n_cat * n_rng;
    dim3 grid((unsigned)((KR + 63) / 64), (unsigned)((N_THR * N_REC + 63) / 64));
    TAO_TIMED("ex_unpack_kernel", s, ex_unpack_kernel<<<grid, 256, 0, s>>>(a));
    TAO_LAUNCH_CHECK();
    return TAOAMD_OK;
}

// ---------------------------------------------------------------------------
// By-video partition, owner side: k-way merge of the received runs -- in two
// messages (round 5).
//
// Every source rank holds its records sorted by (category, -score): the sort
// gives a detection's place, the match writes its (matched, ignored) pairs
// there.  The reference's order of a category = stable sort by -score of the
// concatenation of the sources in rank order (tao_amodal/eval.py:508-518,
// lvis_amodal/eval.py:353-361); inside a source's run that order is already
// there, so a record's final row = its place in its own run + the number of
// records of the OTHER sources' runs of the category that precede it (better
// score; on a tie the lower rank) -- one binary search per other source.
//
// That row depends on the SCORES alone, which a rank knows as soon as its
// local sort is done.  So the scores travel first (8 bytes a record, while the
// 3D IoU and the match still run), the owner works out every record's row
// beside the match (ex_positions_kernel), and what is left behind the match
// is the exchange of the rows themselves (16 bytes a record and combo word, in
// category blocks: dist.ShardedEval) and one scatter of 16-byte pairs
// (ex_place_kernel) straight into the paired layout the sweep streams.
// Round 4 shipped {score, matched, ignored} records after the match and
// searched, read and placed them in one kernel on the critical chain.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t ex_desc_key(int64_t score_bits)
{
    double s = __longlong_as_double(score_bits) + 0.0;   // -0.0 -> +0.0
    const uint64_t u = (uint64_t)__double_as_longlong(s);
    const uint64_t asc = (u >> 63) ? ~u : (u | 0x8000000000000000ull);
    return ~asc;                                          // ascending = score descending
}

// scores at their sorted place: the first message of the exchange
__global__ void ex_scores_kernel(int64_t n, const int32_t *__restrict__ dst,
                                 const int64_t *__restrict__ score,
                                 int64_t *__restrict__ out)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) out[dst[i]] = score[i];
}

extern "C" int taoamd_exchange_scores(int64_t n, const int32_t *dst, const double *score,
                                      int64_t *out, void *stream)
{
    if (n < 0) return TAOAMD_ERR_ARG;
    if (n == 0) return TAOAMD_OK;
    if (!dst || !score || !out) return TAOAMD_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    TAO_TIMED("ex_scores_kernel", s, ex_scores_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(
        n, dst, (const int64_t *)score, out));
    TAO_LAUNCH_CHECK();
    return TAOAMD_OK;
}

struct ExSources {           // record i of the merged input -> where it lies
    int64_t n_recv;
    int32_t world, own;
    const int64_t *src_base; // [world + 1] rows of source s: [src_base[s], src_base[s + 1])
};

// element index (in units of one record) of record `at` of source o inside
// the wire buffer / the own buffer
__device__ __forceinline__ int64_t ex_wire_at(const ExSources &x, int o, int64_t at)
{
    const int64_t own_rows = x.own >= 0 ? x.src_base[x.own + 1] - x.src_base[x.own] : 0;
    return x.src_base[o] - (o > x.own ? own_rows : 0) + at;
}

__global__ void ex_positions_kernel(ExSources x, int32_t block_cats,
                                    const int64_t *__restrict__ wire,
                                    const int64_t *__restrict__ own_scores,
                                    const int64_t *__restrict__ run_off,
                                    const int64_t *__restrict__ cat_base,
                                    int32_t *__restrict__ pos_out)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= x.n_recv) return;
    int s = 0;
    while (s + 1 < x.world && x.src_base[s + 1] <= i) s++;
    const int64_t at = i - x.src_base[s];
    auto score_of = [&](int o, int64_t a) -> int64_t {
        return o == x.own ? own_scores[a] : wire[ex_wire_at(x, o, a)];
    };
    const uint64_t key = ex_desc_key(score_of(s, at));
    const int64_t *ro = run_off + (int64_t)s * (block_cats + 1);
    // the record's category is the run it lies in (a record carries none: the
    // sender lays its records out category by category): last kb with
    // ro[kb] <= place inside the source's rows
    int32_t kb = 0;
    for (int32_t lo = 0, hi = block_cats; lo < hi;) {
        const int32_t mid = (lo + hi + 1) >> 1;
        if (ro[mid] <= at) { lo = mid; kb = mid; } else hi = mid - 1;
    }
    int64_t pos = cat_base[kb] + (at - ro[kb]);
    for (int o = 0; o < x.world; o++) {
        if (o == s) continue;
        const int64_t *oo = run_off + (int64_t)o * (block_cats + 1);
        const int64_t b = oo[kb], e = oo[kb + 1];
        // records of run o that come first: key' < key, or key' == key from a lower rank
        int64_t lo = b, hi = e;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            const uint64_t km = ex_desc_key(score_of(o, mid));
            if (km < key || (km == key && o < s)) lo = mid + 1; else hi = mid;
        }
        pos += lo - b;
    }
    pos_out[i] = (int32_t)pos;
}

// one 16-byte (matched, ignored) pair per thread, at its row of the paired table
__global__ void ex_place_kernel(ExSources x, int32_t n_words,
                                const ulonglong2 *__restrict__ wire,
                                const ulonglong2 *__restrict__ own_rows,
                                const int32_t *__restrict__ pos,
                                ulonglong2 *__restrict__ out)
{
    // consecutive records lie in one category's segment of the output: the
    // blocks of a category behind one L2 (xcd_block, common.hpp)
    const int64_t t = (int64_t)xcd_block(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
    const int64_t i = t / n_words;
    if (i >= x.n_recv) return;
    const int w = (int)(t - i * n_words);
    int s = 0;
    while (s + 1 < x.world && x.src_base[s + 1] <= i) s++;
    const int64_t at = i - x.src_base[s];
    const ulonglong2 v = s == x.own ? own_rows[at * n_words + w]
                                    : wire[ex_wire_at(x, s, at) * n_words + w];
    out[(int64_t)pos[i] * n_words + w] = v;
}

static int ex_sources(ExSources &x, int64_t n_recv, int32_t world, const int64_t *src_base,
                      int32_t own_rank)
{
    if (n_recv < 0 || world < 1 || own_rank >= world || !src_base) return TAOAMD_ERR_ARG;
    x.n_recv = n_recv; x.world = world; x.own = own_rank < 0 ? -1 : own_rank;
    x.src_base = src_base;
    return TAOAMD_OK;
}

extern "C" int taoamd_exchange_positions(int64_t n_recv, int32_t world, int32_t block_cats,
                                         const int64_t *scores, const int64_t *own_scores,
                                         int32_t own_rank, const int64_t *src_base,
                                         const int64_t *run_off, const int64_t *cat_base,
                                         int32_t *pos, void *stream)
{
    ExSources x;
    if (ex_sources(x, n_recv, world, src_base, own_rank) != TAOAMD_OK || block_cats < 1)
        return TAOAMD_ERR_ARG;
    if (n_recv == 0) return TAOAMD_OK;
    if (!run_off || !cat_base || !pos) return TAOAMD_ERR_ARG;
    // (`scores` may be null when every row is the rank's own: world == 1)
    if ((x.own < 0 || world > 1) && !scores) return TAOAMD_ERR_ARG;
    if (x.own >= 0 && !own_scores) return TAOAMD_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    TAO_TIMED("ex_positions_kernel", s, ex_positions_kernel<<<(unsigned)((n_recv + 255) / 256), 256, 0, s>>>(
        x, block_cats, scores, own_scores, run_off, cat_base, pos));
    TAO_LAUNCH_CHECK();
    return TAOAMD_OK;
}

extern "C" int taoamd_exchange_place(int64_t n_recv, int32_t world, int32_t n_words,
                                     const uint64_t *rows, const uint64_t *own_rows,
                                     int32_t own_rank, const int64_t *src_base,
                                     const int32_t *pos, uint64_t *out, void *stream)
{
    ExSources x;
    if (ex_sources(x, n_recv, world, src_base, own_rank) != TAOAMD_OK || n_words < 1)
        return TAOAMD_ERR_ARG;
    if (n_recv == 0) return TAOAMD_OK;
    if (!pos || !out) return TAOAMD_ERR_ARG;
    if ((x.own < 0 || world > 1) && !rows) return TAOAMD_ERR_ARG;
    if (x.own >= 0 && !own_rows) return TAOAMD_ERR_ARG;
    // tables of 16-byte pairs
    if ((((uintptr_t)rows | (uintptr_t)own_rows | (uintptr_t)out) & 15) != 0)
        return TAOAMD_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int64_t threads = n_recv * n_words;
    TAO_TIMED("ex_place_kernel", s, ex_place_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, s>>>(
        x, n_words, (const ulonglong2 *)rows, (const ulonglong2 *)own_rows, pos,
        (ulonglong2 *)out));
    TAO_LAUNCH_CHECK();
    return TAOAMD_OK;
}
