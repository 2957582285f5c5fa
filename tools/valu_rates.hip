// VALU issue rates on gfx950: dependent-free streams of ONE instruction, eight
// wavefronts per SIMD, timed with HIP events.
//
//   /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rates tools/valu_rates.hip
//   /tmp/valu_rates          (on an MI355X box, e.g. through gpurun)
//
// Prints, per instruction, the cycles one wave-instruction occupies a SIMD at
// the nominal 2.4 GHz.  Round 5's reading: 32-bit add / sub / and / or / xor /
// mov / lshrrev / bitop3 and v_mul_f32 2.3-2.6 (double rate); everything else
// measured 4.1-4.9, the 32-bit integer multiplies and v_mad_u64_u32 included
// (not quarter rate); v_permlane32_swap 8.2.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define REP 64

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters, uint32_t seed)
{
    uint32_t a[8], b[8];
    double d[8], e[8];
    uint64_t q[8];
    uint64_t msk = 0x5555555555555555ull + seed;
    uint32_t sg = 0;
    for (int i = 0; i < 8; i++) {
        a[i] = seed + threadIdx.x * 7 + i;
        b[i] = seed * 3 + i + threadIdx.x;
        d[i] = (double)(a[i] & 1023) + 1.5;
        e[i] = (double)(b[i] & 1023) + 0.5;
        q[i] = a[i];
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP / 8; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 3) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(a[i]), "v"(b[i]) : "vcc");
                if (OP == 4) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 5) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 6) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 7) asm volatile("v_ffbh_u32 %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 8) asm volatile("v_cvt_f64_u32 %0, %1" : "+v"(d[i]) : "v"(b[i]));
                if (OP == 9) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(e[i]));
                if (OP == 10) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(e[i]));
                if (OP == 11) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(e[i]));
                if (OP == 12) asm volatile("v_cmp_gt_u64 vcc, %0, %1" : : "v"(q[i]), "v"(q[(i + 1) & 7]) : "vcc");
                if (OP == 13) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "s"(msk));
                if (OP == 14) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 15) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 16) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 17) asm volatile("v_max_f64 %0, %0, %1" : "+v"(d[i]) : "v"(e[i]));
                if (OP == 18) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 19) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 20) asm volatile("v_alignbit_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 21) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q[i]) : "v"(q[(i + 1) & 7]));
                if (OP == 22) asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(a[i]), "v"(b[i]) : "vcc");
                if (OP == 23) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 24) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 25) asm volatile("v_bfi_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 26) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 27) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 28) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 29) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 30) asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x30" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 31) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 32) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(b[i]));
                if (OP == 33) asm volatile("v_mov_b64 %0, %1" : "+v"(q[i]) : "v"(q[(i + 1) & 7]));
                if (OP == 34) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sg) : "v"(a[i]));
                if (OP == 35) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            }
        }
    }
    uint32_t s = sg;
    for (int i = 0; i < 8; i++) s += a[i] + b[i] + (uint32_t)q[i] + (uint32_t)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
void run(const char *name, uint32_t *out)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int iters = 2000, blocks = 256 * 8;        // eight wavefronts per SIMD
    k<OP><<<blocks, 256>>>(out, 10, 1);
    (void)hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(out, iters, 1);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double winst = (double)blocks * 4 * iters * REP;       // wave-instructions
    const double cyc = ms * 1e-3 * 2.4e9;                        // nominal 2.4 GHz
    printf("%-22s %8.3f ms  %6.2f cycles per wave-instruction and SIMD\n", name, ms,
           cyc / (winst / 1024.0));
}

int main()
{
    uint32_t *out;
    (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    run<0>("v_add_u32", out); run<27>("v_sub_u32", out); run<23>("v_and_b32", out);
    run<35>("v_or_b32", out); run<24>("v_xor_b32", out); run<26>("v_mov_b32", out);
    run<15>("v_lshrrev_b32", out); run<14>("v_lshlrev_b32", out); run<30>("v_bitop3_b32", out);
    run<16>("v_mul_f32", out); run<25>("v_bfi_b32", out); run<19>("v_perm_b32", out);
    run<20>("v_alignbit_b32", out); run<6>("v_bcnt_u32_b32", out); run<7>("v_ffbh_u32", out);
    run<28>("v_min_u32", out); run<29>("v_add3_u32", out); run<13>("v_cndmask_b32", out);
    run<22>("v_cmp_gt_u32", out); run<12>("v_cmp_gt_u64", out); run<18>("v_mov_b32_dpp", out);
    run<31>("v_mbcnt_lo_u32_b32", out); run<34>("v_readlane_b32", out);
    run<1>("v_mul_lo_u32", out); run<2>("v_mul_hi_u32", out); run<3>("v_mad_u64_u32", out);
    run<4>("v_mul_u32_u24", out); run<5>("v_mul_hi_u32_u24", out);
    run<21>("v_lshl_add_u64", out); run<33>("v_mov_b64", out); run<8>("v_cvt_f64_u32", out);
    run<9>("v_mul_f64", out); run<10>("v_fma_f64", out); run<11>("v_add_f64", out);
    run<17>("v_max_f64", out); run<32>("v_permlane32_swap", out);
    return 0;
}
