// valu_cost.hip -- issue cost (SIMD cycles per wave64 instruction) of the instruction kinds the sweep kernel is made of,
// with W waves resident per SIMD (W = 1, 2, 4, 8), every CU busy.  Each wave runs a long unrolled stream of INDEPENDENT
// instructions of one kind (8 accumulators), so the figure is throughput, not latency.
//   hipcc -O3 --offload-arch=gfx950 valu_cost.hip -o valu_cost && ./valu_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP 64
template <int KIND>
__global__ void __launch_bounds__(256) k(double* out, int iters, double seed, unsigned long long* clk)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double m = 1.0000001, c = 1e-9;
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
            if constexpr (KIND == 0) {          // v_fma_f64
                asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                             "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            } else if constexpr (KIND == 1) {   // v_mul_f64
                asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n"
                             "v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
            } else if constexpr (KIND == 2) {   // v_add_f64
                asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                             "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
            } else if constexpr (KIND == 3) {   // v_rcp_f64
                asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n"
                             "v_rcp_f64 %4, %4\n v_rcp_f64 %5, %5\n v_rcp_f64 %6, %6\n v_rcp_f64 %7, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if constexpr (KIND == 4) {   // v_cndmask_b32 (vcc)
                asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                             "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc"
                             : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(it) : "vcc");
            } else if constexpr (KIND == 5) {   // v_add_u32
                asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                             "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8"
                             : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(it));
            } else if constexpr (KIND == 6) {   // v_mov_b32 dpp (row_mirror)
                asm volatile("v_mov_b32_dpp %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %2, %2 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 row_mirror row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %4, %4 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %5 row_mirror row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %6, %6 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %7 row_mirror row_mask:0xf bank_mask:0xf"
                             : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7));
            } else if constexpr (KIND == 7) {   // v_permlane32_swap
                asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                             "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7"
                             : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7));
            } else if constexpr (KIND == 8) {   // v_mov_b64
                asm volatile("v_mov_b64 %0, %8\n v_mov_b64 %1, %8\n v_mov_b64 %2, %8\n v_mov_b64 %3, %8\n"
                             "v_mov_b64 %4, %8\n v_mov_b64 %5, %8\n v_mov_b64 %6, %8\n v_mov_b64 %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
            } else if constexpr (KIND == 9) {   // mixed: fma_f64 interleaved with cndmask (1:1)
                asm volatile("v_fma_f64 %0, %0, %8, %9\n v_cndmask_b32 %4, %4, %10, vcc\n v_fma_f64 %1, %1, %8, %9\n v_cndmask_b32 %5, %5, %10, vcc\n"
                             "v_fma_f64 %2, %2, %8, %9\n v_cndmask_b32 %6, %6, %10, vcc\n v_fma_f64 %3, %3, %8, %9\n v_cndmask_b32 %7, %7, %10, vcc"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(m), "v"(c), "v"(it) : "vcc");
            } else if constexpr (KIND == 10) {  // s_mul_i32 (SALU)
                int s;
                asm volatile("s_mul_i32 %0, %1, %1\n s_mul_i32 %0, %1, %1\n s_mul_i32 %0, %1, %1\n s_mul_i32 %0, %1, %1\n"
                             "s_mul_i32 %0, %1, %1\n s_mul_i32 %0, %1, %1\n s_mul_i32 %0, %1, %1\n s_mul_i32 %0, %1, %1" : "=s"(s) : "s"(it));
                i0 += s & 0;
            } else if constexpr (KIND == 12) {  // v_cndmask_b32_e64 with an SGPR-pair mask, as the sweep uses it
                asm volatile("v_cndmask_b32_e64 %0, %0, 0, %8\n v_cndmask_b32_e64 %1, %1, 0, %8\n v_cndmask_b32_e64 %2, %2, 0, %8\n v_cndmask_b32_e64 %3, %3, 0, %8\n"
                             "v_cndmask_b32_e64 %4, %4, 0, %8\n v_cndmask_b32_e64 %5, %5, 0, %8\n v_cndmask_b32_e64 %6, %6, 0, %8\n v_cndmask_b32_e64 %7, %7, 0, %8"
                             : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "s"(0x5555555555555555ull));
            } else if constexpr (KIND == 13) {  // v_ldexp_f64 (the exponential's last step)
                asm volatile("v_ldexp_f64 %0, %0, %8\n v_ldexp_f64 %1, %1, %8\n v_ldexp_f64 %2, %2, %8\n v_ldexp_f64 %3, %3, %8\n"
                             "v_ldexp_f64 %4, %4, %8\n v_ldexp_f64 %5, %5, %8\n v_ldexp_f64 %6, %6, %8\n v_ldexp_f64 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(it & 1));
            } else if constexpr (KIND == 14) {  // v_rndne_f64
                asm volatile("v_rndne_f64 %0, %0\n v_rndne_f64 %1, %1\n v_rndne_f64 %2, %2\n v_rndne_f64 %3, %3\n"
                             "v_rndne_f64 %4, %4\n v_rndne_f64 %5, %5\n v_rndne_f64 %6, %6\n v_rndne_f64 %7, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if constexpr (KIND == 15) {  // v_cvt_i32_f64
                asm volatile("v_cvt_i32_f64 %0, %8\n v_cvt_i32_f64 %1, %9\n v_cvt_i32_f64 %2, %10\n v_cvt_i32_f64 %3, %11\n"
                             "v_cvt_i32_f64 %4, %8\n v_cvt_i32_f64 %5, %9\n v_cvt_i32_f64 %6, %10\n v_cvt_i32_f64 %7, %11"
                             : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
            } else if constexpr (KIND == 16) {  // v_cmp_lt_f64 into an SGPR pair
                unsigned long long s0, s1;
                asm volatile("v_cmp_lt_f64 %0, %2, %3\n v_cmp_lt_f64 %1, %3, %4\n v_cmp_lt_f64 %0, %4, %5\n v_cmp_lt_f64 %1, %5, %2\n"
                             "v_cmp_lt_f64 %0, %2, %4\n v_cmp_lt_f64 %1, %3, %5\n v_cmp_lt_f64 %0, %4, %2\n v_cmp_lt_f64 %1, %5, %3"
                             : "=s"(s0), "=s"(s1) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
                i0 += (int)(s0 & s1 & 0);
            } else if constexpr (KIND == 17) {  // v_min_f64
                asm volatile("v_min_f64 %0, %0, %8\n v_min_f64 %1, %1, %8\n v_min_f64 %2, %2, %8\n v_min_f64 %3, %3, %8\n"
                             "v_min_f64 %4, %4, %8\n v_min_f64 %5, %5, %8\n v_min_f64 %6, %6, %8\n v_min_f64 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
            } else if constexpr (KIND == 18) {  // v_accvgpr_write_b32 + v_accvgpr_read_b32 (the register-file spill moves)
                asm volatile("v_accvgpr_write_b32 a0, %0\n v_accvgpr_read_b32 %1, a0\n v_accvgpr_write_b32 a1, %2\n v_accvgpr_read_b32 %3, a1\n"
                             "v_accvgpr_write_b32 a2, %4\n v_accvgpr_read_b32 %5, a2\n v_accvgpr_write_b32 a3, %6\n v_accvgpr_read_b32 %7, a3"
                             : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : : "a0", "a1", "a2", "a3");
            } else if constexpr (KIND == 19) {  // a double's select as the compiler writes it: TWO vcc cndmasks in a row, then two fmas
                asm volatile("v_cndmask_b32 %4, %4, %10, vcc\n v_cndmask_b32 %5, %5, %10, vcc\n v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n"
                             "v_cndmask_b32 %6, %6, %10, vcc\n v_cndmask_b32 %7, %7, %10, vcc\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(m), "v"(c), "v"(it) : "vcc");
            } else if constexpr (KIND == 20) {  // the same with the mask in an SGPR pair (VOP3 form)
                asm volatile("v_cndmask_b32_e64 %4, %4, %10, %11\n v_cndmask_b32_e64 %5, %5, %10, %11\n v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n"
                             "v_cndmask_b32_e64 %6, %6, %10, %11\n v_cndmask_b32_e64 %7, %7, %10, %11\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(m), "v"(c), "v"(it), "s"(0x5555555555555555ull));
            } else if constexpr (KIND == 21) {  // six vcc cndmasks in a row (three doubles selected on one condition), then two fmas
                asm volatile("v_cndmask_b32 %2, %2, %10, vcc\n v_cndmask_b32 %3, %3, %10, vcc\n v_cndmask_b32 %4, %4, %10, vcc\n v_cndmask_b32 %5, %5, %10, vcc\n"
                             "v_cndmask_b32 %6, %6, %10, vcc\n v_cndmask_b32 %7, %7, %10, vcc\n v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9"
                             : "+v"(a0), "+v"(a1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(m), "v"(c), "v"(it) : "vcc");
            } else if constexpr (KIND == 22) {  // a compare into vcc followed by the two cndmasks that use it, then an fma (the compiler's select of a double)
                asm volatile("v_cmp_lt_f64 vcc, %0, %1\n v_cndmask_b32 %4, %4, %10, vcc\n v_cndmask_b32 %5, %5, %10, vcc\n v_fma_f64 %2, %2, %8, %9\n"
                             "v_cmp_lt_f64 vcc, %1, %0\n v_cndmask_b32 %6, %6, %10, vcc\n v_cndmask_b32 %7, %7, %10, vcc\n v_fma_f64 %3, %3, %8, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(m), "v"(c), "v"(it) : "vcc");
            } else if constexpr (KIND == 23) {  // FOUR vcc cndmasks in a row (two doubles on one condition: the linear rule's w2 series branch), then four fmas
                asm volatile("v_cndmask_b32 %4, %4, %10, vcc\n v_cndmask_b32 %5, %5, %10, vcc\n v_cndmask_b32 %6, %6, %10, vcc\n v_cndmask_b32 %7, %7, %10, vcc\n"
                             "v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(m), "v"(c), "v"(it) : "vcc");
            } else if constexpr (KIND == 24) {  // six SGPR-mask cndmasks in a row, then two fmas
                asm volatile("v_cndmask_b32_e64 %2, %2, %10, %11\n v_cndmask_b32_e64 %3, %3, %10, %11\n v_cndmask_b32_e64 %4, %4, %10, %11\n v_cndmask_b32_e64 %5, %5, %10, %11\n"
                             "v_cndmask_b32_e64 %6, %6, %10, %11\n v_cndmask_b32_e64 %7, %7, %10, %11\n v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9"
                             : "+v"(a0), "+v"(a1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(m), "v"(c), "v"(it), "s"(0x5555555555555555ull));
            } else if constexpr (KIND == 11) {  // v_fma_f64 with one wave-uniform SGPR operand... same as 0 but literal 0.5
                asm volatile("v_fma_f64 %0, %0, 0.5, %8\n v_fma_f64 %1, %1, 0.5, %8\n v_fma_f64 %2, %2, 0.5, %8\n v_fma_f64 %3, %3, 0.5, %8\n"
                             "v_fma_f64 %4, %4, 0.5, %8\n v_fma_f64 %5, %5, 0.5, %8\n v_fma_f64 %6, %6, 0.5, %8\n v_fma_f64 %7, %7, 0.5, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0 && clk) { const int w = blockIdx.x * 4 + (threadIdx.x >> 6); clk[2 * w] = t1 - t0; clk[2 * w + 1] = r1 - r0; }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7;
}

static unsigned long long* d_clk;
static double g_shader_cyc, g_ghz, g_min, g_max;
template <int KIND>
double run(int waves_per_simd, double* d_out)
{
    const int iters = 2000;
    const int nblk = 256 * waves_per_simd;           // 256-thread blocks: one wave on each SIMD of a CU; W blocks per CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(nblk), dim3(256), 0, 0, d_out, 10, 1.0, (unsigned long long*)nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(nblk), dim3(256), 0, 0, d_out, iters, 1.0, d_clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // cycles per instruction per SIMD at 2.4 GHz nominal: time * f / (instructions issued per SIMD)
    const double inst_per_simd = (double)iters * REP * waves_per_simd;
    std::vector<unsigned long long> h(2 * (size_t)nblk * 4);
    hipMemcpy(h.data(), d_clk, h.size() * 8, hipMemcpyDeviceToHost);
    double sc = 0, rc = 0, mx = 0, mn = 1e30;
    for (int b = 0; b < nblk * 4; ++b) { sc += (double)h[2 * b]; rc += (double)h[2 * b + 1]; mx = std::max(mx, (double)h[2 * b]); mn = std::min(mn, (double)h[2 * b]); }
    g_min = mn / ((double)iters * REP); g_max = mx / ((double)iters * REP);
    g_shader_cyc = sc / (nblk * 4) / ((double)iters * REP);          // shader cycles a WAVE spends per instruction of its own
    g_ghz = sc / rc * 0.1;                                     // s_memrealtime ticks at 100 MHz
    return ms * 1e-3 * 2.4e9 / inst_per_simd;
}

int main()
{
    double* d_out;
    hipMalloc(&d_out, sizeof(double) * 256 * 256 * 8);
    hipMalloc(&d_clk, 16 * 256 * 4 * 8 * 4);
    const char* names[] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_rcp_f64", "v_cndmask_b32", "v_add_u32", "v_mov_b32_dpp",
                           "v_permlane32_swap", "v_mov_b64", "fma_f64+cndmask (per instr)", "s_mul_i32", "v_fma_f64 (inline const)", "v_cndmask_b32_e64 (sgpr mask)",
                           "v_ldexp_f64", "v_rndne_f64", "v_cvt_i32_f64", "v_cmp_lt_f64 (sgpr)", "v_min_f64", "v_accvgpr write+read",
                           "2 cndmask(vcc) + 2 fma", "2 cndmask(sgpr) + 2 fma", "6 cndmask(vcc) + 2 fma", "cmp->vcc, 2 cndmask(vcc), fma", "4 cndmask(vcc) + 4 fma", "6 cndmask(sgpr) + 2 fma"};
    printf("cycles (at a nominal 2.4 GHz) per wave64 instruction per SIMD; waves per SIMD = 1, 2, 4, 8\n");
    for (int w : {1, 2, 4}) {
        double r[25], sc[25], gh[25], lo[25], hi[25];
#define RUN(i) r[i] = run<i>(w, d_out); sc[i] = g_shader_cyc; gh[i] = g_ghz; lo[i] = g_min; hi[i] = g_max;
        RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14) RUN(15) RUN(16) RUN(17) RUN(18) RUN(19) RUN(20) RUN(21) RUN(22) RUN(23) RUN(24)
        for (int i = 0; i < 25; ++i)
            printf("W=%d %-32s wall@2.4GHz %.2f | per wave: shader cycles per own instr avg %.2f (min %.2f max %.2f) -> per SIMD %.2f | clock %.2f GHz\n", w, names[i], r[i],
                   sc[i], lo[i], hi[i], sc[i] / w, gh[i]);
    }
    return 0;
}
