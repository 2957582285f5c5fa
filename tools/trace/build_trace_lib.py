"""Wave-lifetime traces: builds a copy of the C-ABI library in
which acc_emit_kernel, match_group_kernel and seg_tile_kernel stamp wall_clock64()
(100 MHz) at their phase boundaries into a device array, one record per wavefront
/ workgroup.  The product sources are not touched: the instrumented copies are
written to a scratch directory, compiled there, and the library goes to
tools/trace/_lib_trace.so (git-ignored).  tools/trace/run_trace.py prints the summaries kept
under profiles/r02_trace_*.txt.

    python tools/trace/build_trace_lib.py && gpurun -- python tools/trace/run_trace.py
"""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "tao_amodal_amd", "csrc")


def rep(text, old, new):
    assert old in text, old[:60]
    return text.replace(old, new, 1)


def instrument_accumulate(src):
    src = rep(src, "#define EMIT_RMAX 8   // ranges that can overlap one 64-combo word", """#define EMIT_RMAX 8
__device__ unsigned long long g_trace_emit[8 * 16384];
extern "C" int taoamd_trace_emit(unsigned long long *host)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_trace_emit), sizeof(g_trace_emit));
}
#define STAMP(i) tr##i = wall_clock64()""")
    a = src.index("template <bool INLINE>\n__global__ __launch_bounds__(256) void acc_emit_kernel(AccArgs a)")
    b = src.index("// Fused sweep for problems whose categories all fit one workgroup")
    k = src[a:b]
    k = rep(k, """    const ChunkInfo ci = chunk_info(a);
    if (!ci.valid) return;""", """    unsigned long long tr0 = wall_clock64(), tr1 = 0, tr2 = 0, tr3 = 0, tr4 = 0, tr5 = 0;
    const ChunkInfo ci = chunk_info(a);
    if (!ci.valid) return;
    const long long tr_idx = (long long)ci.c * a.n_words + ci.word;
    STAMP(1);""")
    k = rep(k, """    uint64_t *__restrict__ out =
        a.val + (((int64_t)ci.k * a.n_rng + r) * N_THR + t) * N_REC;""",
            """    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    STAMP(2);
    uint64_t *__restrict__ out =
        a.val + (((int64_t)ci.k * a.n_rng + r) * N_THR + t) * N_REC;""")
    k = rep(k, """    if (ci.last) {
        // thresholds the category never reaches get precision 0.""", """    STAMP(3);
    if (ci.last) {
        // thresholds the category never reaches get precision 0.""")
    k = rep(k, """    int32_t cnext = jcur > 0 ? cj[jcur - 1] : -1;
#pragma unroll
    for (int blk = ACC_BLK - 1; blk >= 0; blk--) {
        if (blk * WAVE >= ci.len) continue;""", """    int32_t cnext = jcur > 0 ? cj[jcur - 1] : -1;
    STAMP(4);
#pragma unroll
    for (int blk = ACC_BLK - 1; blk >= 0; blk--) {
        if (blk * WAVE >= ci.len) continue;""")
    k = rep(k, """        while (jcur > 0) {
            out[jcur - 1] = v;
            jcur--;
        }
    }
}""", """        while (jcur > 0) {
            out[jcur - 1] = v;
            jcur--;
        }
    }
    STAMP(5);
    if ((threadIdx.x & 63) == 0 && tr_idx < 16384) {
        unsigned long long *g = g_trace_emit + tr_idx * 8;
        g[0] = tr0; g[1] = tr1; g[2] = tr2; g[3] = tr3; g[4] = tr4; g[5] = tr5;
        g[6] = ci.c - a.cat_chunk_off[ci.k]; g[7] = 1;
    }
}""")
    return src[:a] + k + src[b:]


def instrument_match(src):
    a = src.index("template <bool FUSED>\n__global__ __launch_bounds__(256) void match_group_kernel")
    b = src.index("// Slow path: cells with more than 64 ground truths.")
    k = src[a:b]
    pre = """__device__ unsigned long long g_trace_match[8 * 65536];
extern "C" int taoamd_trace_match(unsigned long long *host)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_trace_match), sizeof(g_trace_match));
}
#define MSTAMP(i) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); tr##i = wall_clock64()
"""
    k = rep(k, """    if (item >= (int64_t)a.n_groups * a.n_words) return;""",
            """    if (item >= (int64_t)a.n_groups * a.n_words) return;
    unsigned long long tr0 = wall_clock64(), tr1 = 0, tr2 = 0, tr3 = 0, tr4 = 0, tr5 = 0, tr6 = 0, tr7 = 0;""")
    k = rep(k, "    // ---- lane = detection of the run\n    int32_t t_flags = 0,",
            "    MSTAMP(1);\n    // ---- lane = detection of the run\n    int32_t t_flags = 0,")
    k = rep(k, "    uint64_t IG = 0;\n    for (int q = 0; q < a.n_rng; q++) {",
            "    MSTAMP(2);\n    uint64_t IG = 0;\n    for (int q = 0; q < a.n_rng; q++) {")
    k = rep(k, "    uint64_t my_m = 0, my_i = 0;\n    const double tmin = fmin(thr.v[0], 1 - 1e-10);",
            "    MSTAMP(3);\n    uint64_t my_m = 0, my_i = 0;\n    const double tmin = fmin(thr.v[0], 1 - 1e-10);")
    k = rep(k, "        const bool consumes = !(t_flags & TAOAMD_DT_NO_CONSUME);",
            "        MSTAMP(4);\n        const bool consumes = !(t_flags & TAOAMD_DT_NO_CONSUME);")
    k = rep(k, "        if (simple) {\n            const uint32_t all10 = (1u << N_THR) - 1;",
            "        MSTAMP(5);\n        if (simple) {\n            const uint32_t all10 = (1u << N_THR) - 1;")
    k = rep(k, "    const uint64_t todo = __ballot(lane < nD && !simple);",
            "    MSTAMP(6);\n    const uint64_t todo = __ballot(lane < nD && !simple);")
    k = rep(k, """    if (lane < nD) {
        a.matched[t_row * a.out_stride + word] = my_m;
        a.ignored[t_row * a.out_stride + word] = my_i;
    }
}""", """    MSTAMP(7);
    if (lane < nD) {
        a.matched[t_row * a.out_stride + word] = my_m;
        a.ignored[t_row * a.out_stride + word] = my_i;
    }
    if (lane == 0 && item < 65536) {
        unsigned long long *g = g_trace_match + item * 8;
        g[0] = tr0; g[1] = tr1; g[2] = tr2; g[3] = tr3; g[4] = tr4; g[5] = tr5;
        g[6] = tr6; g[7] = tr7;
    }
}""")
    return src[:a] + pre + k + src[b:]


def instrument_sort(src):
    pre = """__device__ unsigned long long g_trace_tile[8 * 16384];
extern "C" int taoamd_trace_tile(unsigned long long *host)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_trace_tile), sizeof(g_trace_tile));
}
__shared__ unsigned long long s_tr[8];
#define SSTAMP(i) if (threadIdx.x == 0) s_tr[i] = wall_clock64()
"""
    a = src.index("template <class Load>\n__device__ __forceinline__ bool seg_sort_lds(SegLds &L, int n, Load load)")
    src = src[:a] + pre + src[a:]
    src = rep(src, """        load();
        if (threadIdx.x == 0) L.n_long = 0;
        __syncthreads();
        seg_passes(L, 0, n, attempt == 0 ? 4 : 0, 8);""", """        load();
        if (threadIdx.x == 0) L.n_long = 0;
        __syncthreads();
        SSTAMP(1);
        seg_passes(L, 0, n, attempt == 0 ? 4 : 0, 8);
        SSTAMP(2);""")
    src = rep(src, "        repaired = L.flag != 0;\n        if (L.flag != 2) break;",
              "        repaired = L.flag != 0;\n        SSTAMP(3);\n        if (L.flag != 2) break;")
    src = rep(src, """__global__ __launch_bounds__(SEG_THREADS, 5) void seg_tile_kernel(SegArgs a)
{
    __shared__ SegLds L;""", """__global__ __launch_bounds__(SEG_THREADS, 5) void seg_tile_kernel(SegArgs a)
{
    __shared__ SegLds L;
    if (threadIdx.x < 8) s_tr[threadIdx.x] = 0;
    __syncthreads();
    SSTAMP(0);""")
    src = rep(src, """            L.key[i] = desc_key(a.score[b + i]);
            L.pos[i] = (uint16_t)i;
        }
    });
""", """            L.key[i] = desc_key(a.score[b + i]);
            L.pos[i] = (uint16_t)i;
        }
    });
    SSTAMP(4);
""")
    src = rep(src, """        } else {
            a.key[0][b + to] = mine;
            a.idx[0][b + to] = d;
        }
    }
}
""", """        } else {
            a.key[0][b + to] = mine;
            a.idx[0][b + to] = d;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x < 16384) {
        unsigned long long *g = g_trace_tile + (size_t)blockIdx.x * 8;
        g[0] = s_tr[0]; g[1] = s_tr[1]; g[2] = s_tr[2]; g[3] = s_tr[3]; g[4] = s_tr[4];
        g[5] = wall_clock64(); g[6] = n; g[7] = 1;
    }
}
""")
    return src


def main():
    tmp = tempfile.mkdtemp(prefix="taoamd_trace_")
    work = os.path.join(tmp, "a", "b")
    os.makedirs(work)
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
    for f in os.listdir(CSRC):
        if f.endswith((".hip", ".hpp")):
            shutil.copy(os.path.join(CSRC, f), work)
    for name, fn in (("accumulate.hip", instrument_accumulate),
                     ("iou_match.hip", instrument_match), ("sort.hip", instrument_sort)):
        p = os.path.join(work, name)
        text = fn(open(p).read())
        open(p, "w").write(text)
    out = os.path.join(ROOT, "tools", "trace", "_lib_trace.so")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
           "-shared", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-result",
           "-mllvm", "-pragma-unroll-threshold=262144",
           "api.hip", "iou_match.hip", "track_iou.hip", "flatten.hip", "sort.hip",
           "accumulate.hip", "exchange.hip", "rle_iou.hip", "-o", out]
    subprocess.check_call(cmd, cwd=work)
    shutil.rmtree(tmp)
    print("built", out)


if __name__ == "__main__":
    sys.exit(main())
