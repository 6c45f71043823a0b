#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
// NACC independent accumulators round-robin; NV independent v_fma_f32 between MFMAs; NL ds_read_b128 between MFMAs
template <int NACC, int NV, int NL>
__global__ void k(float *out, long long *cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    f32x4 l[4];
    for (int i = 0; i < 4; ++i) l[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int laddr = (threadIdx.x & 63) * 16;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) reinterpret_cast<float *>(smem)[i] = 1.f;
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            acc[m % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m % NACC], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < NV; ++q) v[(m * NV + q) % 8] = __builtin_fmaf(v[(m * NV + q) % 8], b, a);
#pragma unroll
            for (int q = 0; q < NL; ++q) l[(m + q) % 4] = *reinterpret_cast<const f32x4 *>(smem + laddr + ((m + q) % 4) * 1024);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) s += l[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NACC, int NV, int NL>
void run(int threads, int blocks, float *out, long long *cyc) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<NACC, NV, NL>), dim3(blocks), dim3(threads), 16384, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((k<NACC, NV, NL>), dim3(blocks), dim3(threads), 16384, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long h;
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const int waves_per_simd = threads / 256;
    printf("NACC=%d NV=%d NL=%d waves/SIMD=%d blocks=%d: %.1f cycles per MFMA per wave, %.1f per MFMA per SIMD\n", NACC, NV, NL, waves_per_simd, blocks,
           (double)h / (iters * 16.0), (double)h / (iters * 16.0) / waves_per_simd);
}
int main() {
    float *out; long long *cyc;
    hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 64);
    for (int blocks : {1, 256}) {
        for (int threads : {256, 512}) {
            run<1, 0, 0>(threads, blocks, out, cyc);
            run<2, 0, 0>(threads, blocks, out, cyc);
            run<3, 0, 0>(threads, blocks, out, cyc);
            run<4, 0, 0>(threads, blocks, out, cyc);
            run<4, 2, 0>(threads, blocks, out, cyc);
            run<4, 4, 0>(threads, blocks, out, cyc);
            run<4, 6, 0>(threads, blocks, out, cyc);
            run<4, 8, 0>(threads, blocks, out, cyc);
            run<2, 4, 0>(threads, blocks, out, cyc);
            run<4, 0, 1>(threads, blocks, out, cyc);
            run<4, 3, 1>(threads, blocks, out, cyc);
            run<2, 3, 1>(threads, blocks, out, cyc);
        }
    }
    return 0;
}
