// mfma_rowsum.hip -- can v_mfma_f64_4x4x4 (4 blocks of 4x4x4, one double per lane for A, B and D) sum the 16 lanes of
// each DPP row?  Two chained MFMAs with a ones operand; the lane layouts decide which operand the intermediate goes
// into, so both arrangements are tried and checked against the exact row sums (integers: no rounding).
// Also times the pair against the DPP row reduction (4 x (2 v_mov_dpp + v_add_f64)).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ double mfma4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }

__global__ void k_check(const double* in, double* outA, double* outB, double* mid)
{
    const double v = in[threadIdx.x];
    const double s1 = mfma4(v, 1.0, 0.0);        // D[i][j] = sum_k A[i][k]
    mid[threadIdx.x] = s1;
    outA[threadIdx.x] = mfma4(s1, 1.0, 0.0);     // intermediate as A
    outB[threadIdx.x] = mfma4(1.0, s1, 0.0);     // intermediate as B
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, ROW_MASK, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, ROW_MASK, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_sums(double v)
{
    v += dpp_f64<0xB1, 0xf>(v);
    v += dpp_f64<0x4E, 0xf>(v);
    v += dpp_f64<0x141, 0xf>(v);
    v += dpp_f64<0x140, 0xf>(v);
    return v;
}

template <int KIND>
__global__ void __launch_bounds__(256) k_time(double* out, int iters, unsigned long long* clk)
{
    double a0 = threadIdx.x + 1.0, a1 = a0 * 1.5, a2 = a0 * 0.25, a3 = a0 + 7;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if constexpr (KIND == 0) { a0 = row_sums(a0) * 0.0625; a1 = row_sums(a1) * 0.0625; a2 = row_sums(a2) * 0.0625; a3 = row_sums(a3) * 0.0625; }
            else if constexpr (KIND == 1) {
                a0 = mfma4(mfma4(a0, 1.0, 0.0), 1.0, 0.0) * 0.0625; a1 = mfma4(mfma4(a1, 1.0, 0.0), 1.0, 0.0) * 0.0625;
                a2 = mfma4(mfma4(a2, 1.0, 0.0), 1.0, 0.0) * 0.0625; a3 = mfma4(mfma4(a3, 1.0, 0.0), 1.0, 0.0) * 0.0625;
            } else {    // the multiplies alone (baseline)
                a0 *= 0.0625; a1 *= 0.0625; a2 *= 0.0625; a3 *= 0.0625;
                asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3;
}

int main()
{
    std::vector<double> h(64), oa(64), ob(64), mid(64);
    for (int i = 0; i < 64; ++i) h[i] = (double)(1 + i * i % 37);
    double *din, *dA, *dB, *dM;
    hipMalloc(&din, 512); hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dM, 512);
    hipMemcpy(din, h.data(), 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, din, dA, dB, dM);
    hipMemcpy(oa.data(), dA, 512, hipMemcpyDeviceToHost);
    hipMemcpy(ob.data(), dB, 512, hipMemcpyDeviceToHost);
    hipMemcpy(mid.data(), dM, 512, hipMemcpyDeviceToHost);
    int okA = 1, okB = 1;
    for (int r = 0; r < 4; ++r) {
        double s = 0;
        for (int i = 0; i < 16; ++i) s += h[16 * r + i];
        for (int i = 0; i < 16; ++i) { okA &= oa[16 * r + i] == s; okB &= ob[16 * r + i] == s; }
        printf("row %d: exact %.0f  A-variant lane0 %.0f lane5 %.0f  B-variant lane0 %.0f lane5 %.0f  mid lanes 0..7:", r, s, oa[16 * r], oa[16 * r + 5], ob[16 * r], ob[16 * r + 5]);
        for (int i = 0; i < 8; ++i) printf(" %.0f", mid[16 * r + i]);
        printf("\n");
    }
    printf("ROWSUM via 2 MFMA: intermediate-as-A %s, intermediate-as-B %s\n", okA ? "OK" : "wrong", okB ? "OK" : "wrong");
    // timing: 8 blocks of 256 threads per CU
    double* dout; unsigned long long* dclk;
    const int nblk = 256 * 4, iters = 2000;
    hipMalloc(&dout, (size_t)nblk * 256 * 8); hipMalloc(&dclk, (size_t)nblk * 4 * 8);
    std::vector<unsigned long long> hc((size_t)nblk * 4);
    const char* nm[] = {"4 x row_sums (DPP) + mul", "4 x (2 MFMA) + mul", "4 x mul only"};
    for (int kind = 0; kind < 3; ++kind) {
        for (int rep = 0; rep < 2; ++rep) {
            if (kind == 0) hipLaunchKernelGGL(k_time<0>, dim3(nblk), dim3(256), 0, 0, dout, iters, dclk);
            if (kind == 1) hipLaunchKernelGGL(k_time<1>, dim3(nblk), dim3(256), 0, 0, dout, iters, dclk);
            if (kind == 2) hipLaunchKernelGGL(k_time<2>, dim3(nblk), dim3(256), 0, 0, dout, iters, dclk);
            hipDeviceSynchronize();
        }
        hipMemcpy(hc.data(), dclk, hc.size() * 8, hipMemcpyDeviceToHost);
        double mx = 0, sum = 0;
        for (auto c : hc) { mx = c > mx ? (double)c : mx; sum += (double)c; }
        printf("%-28s wave cycles per group of 4 reductions: avg %.1f max %.1f  (4 waves/SIMD resident)\n", nm[kind], sum / hc.size() / (iters * 4.0), mx / (iters * 4.0));
    }
    return 0;
}
