// What HBM3E delivers on an MI355X for the access shapes of this library's
// kernels -- the practical ceiling the roofline fractions in DESIGN.md should be
// read against (the 8 TB/s of MI355X_MICROARCH.md is the pin rate).
//
//   /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/hbm_rates tools/hbm_rates.hip
//   /tmp/hbm_rates           (on an MI355X box, e.g. through gpurun)
//
// Every case moves a 2 GiB working set (far beyond the 256 MB Infinity Cache):
//   read   a wavefront reads pieces of P bytes (64 lanes x 16 B per request,
//          P / 1024 requests a piece, U pieces requested before the first is
//          waited for), pieces taken in sequence or from shuffled places
//   write  the same with stores
//   copy   read + write of a piece
// TB/s = bytes named by the loads / stores per second (decimal).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef uint32_t u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint64_t place(uint64_t i, uint64_t n, int shuffled)
{
    // a bijection of [0, n) for n a power of two: odd multiplier, xor-shift
    if (!shuffled) return i;
    uint64_t x = (i * 0x9E3779B97F4A7C15ull) & (n - 1);
    x ^= x >> 7;
    return (x * 0xD6E8FEB86659FD93ull) & (n - 1);
}

// MODE 0: read, 1: write, 2: copy.  REQ = 1 KB requests per piece.
template <int MODE, int REQ, int U>
__global__ __launch_bounds__(256) void k(const u4 *__restrict__ src, u4 *__restrict__ dst,
                                         uint64_t n_pieces, int shuffled, uint32_t *sink)
{
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = (gridDim.x * (uint64_t)blockDim.x) >> 6;
    u4 acc = {0, 0, 0, 0};
    for (uint64_t i = wave * U; i < n_pieces; i += n_waves * U) {
        u4 v[U][REQ];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t p = place(i + u < n_pieces ? i + u : i, n_pieces, shuffled);
            const uint64_t at = p * (REQ * 64) + lane;
#pragma unroll
            for (int r = 0; r < REQ; r++) {
                if (MODE != 1) v[u][r] = src[at + r * 64];
                else v[u][r] = u4{(uint32_t)at, (uint32_t)r, 0u, 0u};
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t p = place(i + u < n_pieces ? i + u : i, n_pieces, shuffled);
            const uint64_t at = p * (REQ * 64) + lane;
#pragma unroll
            for (int r = 0; r < REQ; r++) {
                if (MODE == 0) acc += v[u][r];
                else dst[at + r * 64] = v[u][r];
            }
        }
    }
    if (MODE == 0 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = 1;
}

template <int MODE, int REQ, int U>
void run(const char *name, const u4 *src, u4 *dst, uint64_t bytes, int shuffled, int blocks,
         uint32_t *sink)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const uint64_t n_pieces = bytes / (REQ * 1024);
    k<MODE, REQ, U><<<blocks, 256>>>(src, dst, n_pieces, shuffled, sink);
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
        (void)hipEventRecord(e0);
        k<MODE, REQ, U><<<blocks, 256>>>(src, dst, n_pieces, shuffled, sink);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double moved = (double)bytes * (MODE == 2 ? 2 : 1);
    printf("%-46s %5d blocks  %7.3f ms  %6.2f TB/s\n", name, blocks, best, moved / best * 1e-9);
}

int main()
{
    const uint64_t bytes = 2ull << 30;
    u4 *a, *b;
    uint32_t *sink;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) return 1;
    (void)hipMalloc(&sink, 64);
    (void)hipMemset(a, 1, bytes);
    (void)hipMemset(b, 2, bytes);
    const int grids[] = {256 * 8, 256 * 32};
    for (int blocks : grids) {
        run<0, 1, 4>("read  1 KB pieces, 4 ahead, sequential", a, b, bytes, 0, blocks, sink);
        run<0, 1, 8>("read  1 KB pieces, 8 ahead, sequential", a, b, bytes, 0, blocks, sink);
        run<0, 2, 4>("read  2 KB pieces, 4 ahead, sequential", a, b, bytes, 0, blocks, sink);
        run<0, 1, 4>("read  1 KB pieces, 4 ahead, shuffled", a, b, bytes, 1, blocks, sink);
        run<0, 2, 4>("read  2 KB pieces, 4 ahead, shuffled", a, b, bytes, 1, blocks, sink);
        run<0, 8, 1>("read  8 KB pieces, shuffled", a, b, bytes, 1, blocks, sink);
        run<1, 1, 4>("write 1 KB pieces, sequential", a, b, bytes, 0, blocks, sink);
        run<1, 1, 4>("write 1 KB pieces, shuffled", a, b, bytes, 1, blocks, sink);
        run<2, 1, 4>("copy  1 KB pieces, sequential", a, b, bytes, 0, blocks, sink);
        run<2, 2, 4>("copy  2 KB pieces, shuffled", a, b, bytes, 1, blocks, sink);
    }
    return 0;
}
