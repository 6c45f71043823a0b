// What the f16 matrix pipe SUSTAINS under the board's power cap (MI355X: 1 400 W): back-to-back MFMAs on operands
// held in registers - no LDS, no memory - for a second or more per configuration, so that the power management has
// settled.  Data: zeros (no toggling), random f16 of O(1) magnitude, and the "split" operand mix of the DualNet
// forward kernels (high pieces O(1), low pieces uniformly random mantissas scaled by 2^11).  Reports TFLOP/s from the
// wall clock and the effective shader clock (s_memtime ticks of one wave / wall clock).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_power tools/microbench/mfma_power.hip && ./mfma_power
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ inline unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
// mode 0: zeros, 1: random in [-2, 2), 2: random in [-1024, 1024) (scaled low pieces), 3: a mix (even index: 1, odd: 2)
__device__ inline f16x8 make_frag(int mode, unsigned seed) {
    f16x8 v;
    for (int e = 0; e < 8; ++e) {
        const unsigned h = hash32(seed * 8 + e);
        const float u = (float)(h & 0xFFFFFF) / (float)0x1000000 * 2.f - 1.f;   // [-1, 1)
        const int m = mode == 3 ? 1 + (seed & 1) : mode;
        v[e] = (_Float16)(m == 0 ? 0.f : (m == 1 ? 2.f * u : 1024.f * u));
    }
    return v;
}

template <int SHAPE>   // 16 or 32
__global__ __launch_bounds__(512) void k(float *out, long long *ticks, int iters, int mode) {
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    long long t0 = 0, t1 = 0;
    if constexpr (SHAPE == 16) {
        f16x8 a[4], b[4];
        f32x4 acc[4][4];
        for (int i = 0; i < 4; ++i) { a[i] = make_frag(mode, gid * 16 + i); b[i] = make_frag(mode, gid * 16 + 8 + i); }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[(j + u) & 3], acc[i][j], 0, 0, 0);
            // keep the accumulators bounded without stalling the pipe: nothing (fp32 range is ample for the run length)
        }
        t1 = __builtin_amdgcn_s_memtime();
        float s = 0.f;
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
        out[gid] = s;
    } else {
        f16x8 a[2], b[2];
        f32x16 acc[2][2];
        for (int i = 0; i < 2; ++i) { a[i] = make_frag(mode, gid * 16 + i); b[i] = make_frag(mode, gid * 16 + 8 + i); }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[(j + u) & 1], acc[i][j], 0, 0, 0);
        }
        t1 = __builtin_amdgcn_s_memtime();
        float s = 0.f;
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][15];
        out[gid] = s;
    }
    if (gid == 0) ticks[0] = t1 - t0;
}

int main(int argc, char **argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 1.5;
    float *out;
    long long *ticks;
    (void)hipMalloc(&out, 256 * 512 * sizeof(float));
    (void)hipMalloc(&ticks, sizeof(long long));
    const char *mode_name[4] = {"zeros", "random O(1)", "random O(1000)", "split mix (hi O(1) / lo O(1000))"};
    printf("# tools/microbench/mfma_power.hip on MI355X: 256 workgroups (one per CU), MFMAs back to back on register operands,\n"
           "# %.1f s per row (power management settled).  Peak at 2.4 GHz: 2 516 TFLOP/s dense f16.\n", seconds);
    for (int shape : {16, 32})
        for (int wps : {1, 2})
            for (int mode : {0, 1, 3}) {
                const int threads = 256 * wps;
                const int iters = 20000;
                const double flop_per_launch = 256.0 * threads / 64 * iters * 64.0 * (shape == 16 ? 16384.0 : 32768.0) / (shape == 16 ? 1 : 2);
                auto launch = [&] {
                    if (shape == 16) hipLaunchKernelGGL(k<16>, dim3(256), dim3(threads), 0, 0, out, ticks, iters, mode);
                    else hipLaunchKernelGGL(k<32>, dim3(256), dim3(threads), 0, 0, out, ticks, iters, mode);
                };
                launch();
                (void)hipDeviceSynchronize();
                // run for `seconds`, time the last half
                int n = 0;
                auto t0 = std::chrono::steady_clock::now();
                while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds / 2) { launch(); (void)hipDeviceSynchronize(); }
                auto t1 = std::chrono::steady_clock::now();
                while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count() < seconds / 2) { launch(); (void)hipDeviceSynchronize(); ++n; }
                (void)hipDeviceSynchronize();
                const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
                long long tk = 0;
                (void)hipMemcpy(&tk, ticks, sizeof(tk), hipMemcpyDeviceToHost);
                const double per_launch = dt / n;
                printf("v_mfma_f32_%s_f16  waves/SIMD=%d  %-34s %8.1f TFLOP/s  clock %.3f GHz  (%.1f cycles per MFMA per SIMD)\n",
                       shape == 16 ? "16x16x32" : "32x32x16", wps, mode_name[mode], flop_per_launch / per_launch / 1e12,
                       (double)tk / per_launch / 1e9, (double)tk / (iters * 64.0 / (shape == 16 ? 1 : 2)) / wps);
                fflush(stdout);
            }
    return 0;
}
