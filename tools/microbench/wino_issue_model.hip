// Round 4: what the Winograd-on-split-operands tower (dualnet_fwd_wsplit_kernel) stands on.
//   1. f16 SUBNORMAL operands of v_mfma_f32_16x16x32_f16: kept or flushed?  (decides whether the low operand piece may
//      stay unscaled, which makes one accumulator set enough)
//   2. accuracy of  acc = sum (ah wl + al wh) ; acc += sum ah wh  in ONE accumulator (unscaled low pieces, cross terms
//      first) against the two-accumulator scheme of net_forward_split.hip, both against an fp64 dot product, K = 64
//   3. issue model, one / two waves per SIMD: a body of 48 MFMAs with NV single-issue VALU instructions (the input
//      transform's v_add_f32 / v_sub_f32, v_cvt_pk_f16_f32, v_fma_mix_f32) and NL ds_read_b128 placed between them.
//   hipcc --offload-arch=gfx950 -O3 -o wino_issue_model tools/microbench/wino_issue_model.hip && ./wino_issue_model
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// ---------------------------------------------------------------------------------------------- 1 + 2
// one wave: D[16 x 16] = A[16 x K] B[K x 16], K = 64 (two MFMAs per product).  a: [16][64] fp32 (rows = m), b: [16][64]
// fp32 (rows = n).  mode 0: two accumulators, low pieces scaled by 2^11; mode 1: one accumulator, low pieces unscaled,
// cross terms first; mode 2: subnormal probe (A = 2^-20 everywhere, B = 2^10: D = 64 x 2^-10 unless flushed)
__global__ void k_acc(const float *a, const float *b, float *d, int mode) {
    const int lane = threadIdx.x, li = lane & 15, lg = lane >> 4;
    f16x8 ah[2], al[2], bh[2], bl[2];
    for (int kc = 0; kc < 2; ++kc)
        for (int e = 0; e < 8; ++e) {
            const int k = kc * 32 + lg * 8 + e;
            float av = a[li * 64 + k], bv = b[li * 64 + k];
            if (mode == 2) { av = ldexpf(1.f, -20); bv = 1024.f; }
            const _Float16 h1 = (_Float16)av, h2 = (_Float16)bv;
            const float s = mode == 0 ? 2048.f : 1.f;
            ah[kc][e] = h1; al[kc][e] = (_Float16)((av - (float)h1) * s);
            bh[kc][e] = h2; bl[kc][e] = (_Float16)((bv - (float)h2) * s);
        }
    f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    if (mode == 0) {
        for (int kc = 0; kc < 2; ++kc) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[kc], bh[kc], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[kc], bh[kc], c1, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[kc], bl[kc], c1, 0, 0, 0);
        }
        for (int j = 0; j < 4; ++j) c0[j] = fmaf(c1[j], 1.f / 2048.f, c0[j]);
    } else {
        for (int kc = 0; kc < 2; ++kc) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[kc], bh[kc], c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[kc], bl[kc], c0, 0, 0, 0);
        }
        for (int kc = 0; kc < 2; ++kc) c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[kc], bh[kc], c0, 0, 0, 0);
    }
    for (int j = 0; j < 4; ++j) d[(lg * 4 + j) * 16 + li] = c0[j];      // D[m = 4 lg + j][n = li]
}

// ---------------------------------------------------------------------------------------------- 3
#define MFMA(acc, a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define VADD(d, x, y) asm volatile("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y))
#define VSUB(d, x, y) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y))
#define VCVT(d, x, y) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y))
#define VMIX(d, h, x) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h), "v"(x))
#define VPKADD(d, x, y) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y))
#define LDS(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))

// MIX 0: v_add_f32 only; 1: the transform's mix (2 add : 1 cvt_pk : 1 fma_mix ... per 4); 2: v_pk_add_f32 (half as many)
template <int NV, int NL, int MIX>
__global__ __launch_bounds__(512) void k_issue(float *out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) reinterpret_cast<unsigned *>(smem)[i] = 0x3c003c00u + (i * 2654435761u >> 20);
    __syncthreads();
    const int addr = (threadIdx.x & 63) * 16;
    f32x4 acc[16];
    i32x4 fa[8], fb[4], fn[4];
    float v[32];
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 pv[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 8; ++i) fa[i] = i32x4{0x3c003c00, 0x3c003800, 0x38003c00, 0x3c003400};
    for (int i = 0; i < 4; ++i) { fb[i] = i32x4{0x3c003800, 0x38003c00, 0x3c003c00, 0x34003c00}; fn[i] = fb[i]; }
    for (int i = 0; i < 32; ++i) v[i] = threadIdx.x * 1e-3f + i;
    for (int i = 0; i < 16; ++i) pv[i] = f32x2{threadIdx.x * 1e-3f + i, 1.f};
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 48; ++m) {
            MFMA(acc[m % 16], fa[m % 8], fb[m % 4]);
#pragma unroll
            for (int q = m * NV / 48; q < (m + 1) * NV / 48; ++q) {
                const int d = q % 32, x = (q + 7) % 32, y = (q + 13) % 32;
                if (MIX == 0) VADD(v[d], v[x], v[y]);
                if (MIX == 1) {
                    if ((q & 3) == 0) VADD(v[d], v[x], v[y]);
                    if ((q & 3) == 1) VSUB(v[d], v[x], v[y]);
                    if ((q & 3) == 2) VCVT(v[d], v[x], v[y]);
                    if ((q & 3) == 3) VMIX(v[d], v[x], v[y]);
                }
                if (MIX == 2 && (q & 1) == 0) VPKADD(pv[d % 16], pv[x % 16], pv[y % 16]);
            }
#pragma unroll
            for (int q = m * NL / 48; q < (m + 1) * NL / 48; ++q) {
                if ((q & 3) == 0) LDS(fn[0], addr, 0);
                if ((q & 3) == 1) LDS(fn[1], addr, 1024);
                if ((q & 3) == 2) LDS(fn[2], addr, 2048);
                if ((q & 3) == 3) LDS(fn[3], addr, 3072);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 4; ++i) fb[i] = fn[i];
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
    for (int i = 0; i < 32; ++i) s += v[i];
    for (int i = 0; i < 16; ++i) s += pv[i][0];
    for (int i = 0; i < 4; ++i) s += (float)fn[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// PHASED: the body's MFMAs first (back to back), then its VALU instructions (no MFMA in between); NM = 0: VALU only.
// MIX 0: v_add_f32, 2: v_pk_add_f32 (NV / 2 of them), 3: v_pk_fma_f32 (NV / 2)
template <int NM, int NV, int MIX>
__global__ __launch_bounds__(512) void k_phased(float *out, int iters) {
    f32x4 acc[16];
    i32x4 fa[8], fb[4];
    float v[32];
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 pv[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 8; ++i) fa[i] = i32x4{0x3c003c00, 0x3c003800, 0x38003c00, 0x3c003400};
    for (int i = 0; i < 4; ++i) fb[i] = i32x4{0x3c003800, 0x38003c00, 0x3c003c00, 0x34003c00};
    for (int i = 0; i < 32; ++i) v[i] = threadIdx.x * 1e-3f + i;
    for (int i = 0; i < 16; ++i) pv[i] = f32x2{threadIdx.x * 1e-3f + i, 1.f};
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < NM; ++m) MFMA(acc[m % 16], fa[m % 8], fb[m % 4]);
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const int d = q % 32, x = (q + 7) % 32, y = (q + 13) % 32;
            if (MIX == 0) VADD(v[d], v[x], v[y]);
            if (MIX == 2 && (q & 1) == 0) VPKADD(pv[d % 16], pv[x % 16], pv[y % 16]);
            if (MIX == 3 && (q & 1) == 0)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(pv[d % 16]) : "v"(pv[x % 16]), "v"(pv[y % 16]), "v"(pv[(d + 3) % 16]));
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3] + pv[i][0] + pv[i][1];
    for (int i = 0; i < 32; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NM, int NV, int MIX>
void run_phased(int threads, float *out) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k_phased<NM, NV, MIX>), dim3(256), dim3(threads), 0, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_phased<NM, NV, MIX>), dim3(256), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const int wps = threads / 256, ninst = MIX >= 2 ? NV / 2 : NV;
    const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * wps);
    printf("phased  waves/SIMD=%d  %2d MFMAs then %3d x %-12s: %7.1f cycles per body; beyond the MFMAs (%d x 16.4): %6.1f = %4.1f per VALU instruction\n",
           wps, NM, ninst, MIX == 0 ? "v_add_f32" : (MIX == 2 ? "v_pk_add_f32" : "v_pk_fma_f32"), cyc, NM, cyc - NM * 16.4,
           (cyc - NM * 16.4) / ninst);
}

template <int NV, int NL, int MIX>
void run(int threads, float *out) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k_issue<NV, NL, MIX>), dim3(256), dim3(threads), 32768, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_issue<NV, NL, MIX>), dim3(256), dim3(threads), 32768, 0, out, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const int wps = threads / 256;
    // cycles (at a nominal 2.4 GHz) one SIMD spends per body of ONE wave
    const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * wps);
    printf("waves/SIMD=%d  %-8s VALU %3d  ds_read_b128 %2d per 48 MFMAs: %7.1f cycles per body = %5.2f per MFMA (%.0f TFLOP/s); beyond 48 x 16.4: %6.1f = %4.1f per filler\n",
           wps, MIX == 0 ? "add" : (MIX == 1 ? "mix" : "pk_add"), MIX == 2 ? NV / 2 : NV, NL, cyc, cyc / 48.0,
           256.0 * 4 * iters * wps * 48 * 16384.0 / ms / 1e9, cyc - 48 * 16.4, (NV + NL) ? (cyc - 48 * 16.4) / ((MIX == 2 ? NV / 2 : NV) + NL) : 0.0);
}

int main() {
    // ---- 1 + 2 ----
    std::vector<float> a(16 * 64), b(16 * 64), d(256);
    srand(7);
    for (auto &x : a) x = ((rand() % 20001) - 10000) * 1e-4f * (rand() % 7 == 0 ? 30.f : 1.f);
    for (auto &x : b) x = ((rand() % 20001) - 10000) * 1e-4f * 700.f;
    float *da, *db, *dd;
    hipMalloc(&da, 4096); hipMalloc(&db, 4096); hipMalloc(&dd, 1024);
    hipMemcpy(da, a.data(), 4096, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), 4096, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(k_acc, dim3(1), dim3(64), 0, 0, da, db, dd, mode);
        hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
        if (mode == 2) {
            printf("subnormal probe: D = %g (kept: %g, flushed: 0)\n", d[0], 64.0 * ldexp(1.0, -10));
            continue;
        }
        double worst = 0, worst32 = 0;
        for (int m = 0; m < 16; ++m)
            for (int n = 0; n < 16; ++n) {
                double ref = 0, mag = 0;
                float f32 = 0.f;
                for (int k = 0; k < 64; ++k) {
                    ref += (double)a[m * 64 + k] * b[n * 64 + k];
                    mag += fabs((double)a[m * 64 + k] * b[n * 64 + k]);
                    f32 = fmaf(a[m * 64 + k], b[n * 64 + k], f32);
                }
                worst = fmax(worst, fabs(d[m * 16 + n] - ref) / mag);
                worst32 = fmax(worst32, fabs((double)f32 - ref) / mag);
            }
        printf("%s: max |D - fp64| / sum|a b| = %.3e   (fp32 fma chain: %.3e; 2^-24 = %.3e)\n",
               mode == 0 ? "two accumulators, scaled low pieces " : "one accumulator, unscaled low pieces", worst, worst32, ldexp(1.0, -24));
    }
    // ---- 3 ----
    float *out;
    hipMalloc(&out, 1 << 22);
    for (int threads : {256, 512}) {
        run_phased<0, 288, 0>(threads, out);
        run_phased<0, 288, 2>(threads, out);
        run_phased<0, 288, 3>(threads, out);
        run_phased<48, 288, 0>(threads, out);
        run_phased<48, 288, 2>(threads, out);
        run_phased<48, 288, 3>(threads, out);
        run_phased<8, 48, 2>(threads, out);
    }
    for (int threads : {256, 512}) {
        run<0, 0, 0>(threads, out);
        run<48, 0, 0>(threads, out);
        run<96, 0, 0>(threads, out);
        run<144, 0, 0>(threads, out);
        run<192, 0, 0>(threads, out);
        run<288, 0, 0>(threads, out);
        run<96, 0, 1>(threads, out);
        run<144, 0, 1>(threads, out);
        run<192, 0, 1>(threads, out);
        run<288, 0, 1>(threads, out);
        run<0, 16, 0>(threads, out);
        run<0, 32, 0>(threads, out);
        run<144, 16, 1>(threads, out);
        run<192, 16, 1>(threads, out);
        run<288, 24, 1>(threads, out);
        run<192, 0, 2>(threads, out);
        run<288, 0, 2>(threads, out);
    }
    return 0;
}
