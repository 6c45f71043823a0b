// Two wavefronts per SIMD, each executing per "unit" 64 v_mfma_f32_16x16x4_f32 + NV packed VALU ops +
// NL ds_read_b128: either in phases (all VALU + LDS first, then 64 MFMAs back to back - the shape of
// dualnet_fwd_wino8_kernel) or finely interleaved (after every MFMA its share of the other work).
// Prints wall-clock cycles per MFMA per SIMD (32 = matrix pipe saturated).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <bool INTERLEAVED, int NV, int NL>
__global__ void k(float *out, long long *cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x2 v[16];
    for (int i = 0; i < 16; ++i) v[i] = f32x2{threadIdx.x * 1e-3f + i, 1.f};
    f32x4 l[4];
    for (int i = 0; i < 4; ++i) l[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    const int laddr = (threadIdx.x & 63) * 16;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) reinterpret_cast<float *>(smem)[i] = 1.f;
    __syncthreads();
    const f32x2 one = {1.0001f, 0.9999f};
    long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if (!INTERLEAVED) {
#pragma unroll
            for (int q = 0; q < NL; ++q) l[q % 4] = *reinterpret_cast<const f32x4 *>(smem + laddr + (q % 4) * 1024);
#pragma unroll
            for (int q = 0; q < NV; ++q) v[q % 16] = v[q % 16] * one + v[(q + 5) % 16];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 64; ++m) {
                acc[m % 16] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b + v[m % 16][0] * 0.f, acc[m % 16], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
            for (int m = 0; m < 64; ++m) {
                acc[m % 16] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b + v[m % 16][0] * 0.f, acc[m % 16], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = m * NV / 64; q < (m + 1) * NV / 64; ++q) v[(q + 8) % 16] = v[(q + 8) % 16] * one + v[(q + 13) % 16];
#pragma unroll
                for (int q = m * NL / 64; q < (m + 1) * NL / 64; ++q) l[q % 4] = *reinterpret_cast<const f32x4 *>(smem + laddr + (q % 4) * 1024);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3] + v[i][0] + v[i][1];
    for (int i = 0; i < 4; ++i) s += l[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <bool I, int NV, int NL>
void run(int threads, float *out, long long *cyc) {
    const int iters = 500;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<I, NV, NL>), dim3(256), dim3(threads), 16384, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<I, NV, NL>), dim3(256), dim3(threads), 16384, 0, out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // wall clock (HIP events), not s_memtime: the s_memtime delta of one wave is not a reliable cycle
    // count when several waves share a SIMD
    const double mfma_per_simd = (double)iters * 64.0 * (threads / 256);
    printf("%-11s NV=%3d NL=%2d waves/SIMD=%d: %.3f ms -> %.1f cycles (2.4 GHz) per MFMA per SIMD, %.1f TFLOP/s\n",
           I ? "interleaved" : "phased", NV, NL, threads / 256, ms, ms * 1e-3 * 2.4e9 / mfma_per_simd,
           256.0 * 4 * mfma_per_simd * 2048 / ms / 1e9);
}
int main() {
    float *out; long long *cyc;
    hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 64);
    for (int threads : {256, 512, 1024}) {
        run<false, 0, 0>(threads, out, cyc);
        run<false, 96, 16>(threads, out, cyc);
        run<true, 96, 16>(threads, out, cyc);
        run<false, 64, 16>(threads, out, cyc);
        run<true, 64, 16>(threads, out, cyc);
        run<false, 128, 32>(threads, out, cyc);
        run<true, 128, 32>(threads, out, cyc);
    }
    return 0;
}
