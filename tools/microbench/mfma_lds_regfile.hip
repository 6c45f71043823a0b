// What does a ds_read_b128 cost a SIMD that is issuing v_mfma_f32_16x16x32_f16 back to back - and does it depend on
// WHICH half of the unified register file the fragment lands in and the accumulators live in?
// One loop body = 24 MFMAs (2 weight x 4 activation fragments x 3 products, four-register accumulators: the k-chunk of
// dualnet_fwd_w2_kernel) with NL ds_read_b128 spread between them.  Everything is inline asm so that the register
// classes are what the variant says: "v" = architectural VGPRs, "a" = accumulation VGPRs.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_lds_regfile tools/microbench/mfma_lds_regfile.hip && ./mfma_lds_regfile
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define MFMA_VV(acc, a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MFMA_AV(acc, a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
#define MFMA_VA(acc, a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "a"(b))
#define MFMA_AA(acc, a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "a"(a), "a"(b))
#define LDS_V(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
#define LDS_A(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=a"(dst) : "v"(addr))

// VAR: 0 acc v / frag v, 1 acc a / frag v, 2 acc v / frag a, 3 acc a / frag a.  NL: reads per body (0, 6 or 12).
template <int VAR, int NL>
__global__ __launch_bounds__(512) void k(float *out, long long *ticks, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<unsigned *>(smem)[i] = 0x3c003c00u + (i * 2654435761u >> 20);
    __syncthreads();
    const int addr = (threadIdx.x & 63) * 16;
    f32x4 acc[8];
    i32x4 fa[2], fb[4], fn[4];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 2; ++i) fa[i] = i32x4{0x3c003c00, 0x3c003c00, 0x3c003c00, 0x3c003c00};
    for (int i = 0; i < 4; ++i) { fb[i] = i32x4{0x3c003800, 0x38003c00, 0x3c003c00, 0x34003c00}; fn[i] = fb[i]; }
    // move the start values into the register class of the variant (the asm constraints below then keep them there)
    long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = q * 8 + c * 4 + r;
                    if constexpr (VAR == 0) MFMA_VV(acc[c * 4 + r], fa[c], fb[r]);
                    if constexpr (VAR == 1) MFMA_AV(acc[c * 4 + r], fa[c], fb[r]);
                    if constexpr (VAR == 2) MFMA_VA(acc[c * 4 + r], fa[c], fb[r]);
                    if constexpr (VAR == 3) MFMA_AA(acc[c * 4 + r], fa[c], fb[r]);
                    if (NL > 0 && m % (24 / (NL ? NL : 1)) == 0 && m / (24 / (NL ? NL : 1)) < NL) {
                        const int j = (m / (24 / (NL ? NL : 1))) & 3;
                        if constexpr (VAR < 2) {
                            if (j == 0) LDS_V(fn[0], addr, 0);
                            if (j == 1) LDS_V(fn[1], addr, 1024);
                            if (j == 2) LDS_V(fn[2], addr, 2048);
                            if (j == 3) LDS_V(fn[3], addr, 3072);
                        } else {
                            if (j == 0) LDS_A(fn[0], addr, 0);
                            if (j == 1) LDS_A(fn[1], addr, 1024);
                            if (j == 2) LDS_A(fn[2], addr, 2048);
                            if (j == 3) LDS_A(fn[3], addr, 3072);
                        }
                    }
                }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) {
        f32x4 v = acc[i];
        s += v[0] + v[3];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
}

template <int VAR, int NL>
void run(int wps, float *out, long long *ticks) {
    const int iters = 4000;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<VAR, NL>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<VAR, NL>), dim3(256), dim3(256 * wps), 65536, 0, out, ticks, iters);
        (void)hipDeviceSynchronize();
    }
    long long tk = 0;
    (void)hipMemcpy(&tk, ticks, sizeof(tk), hipMemcpyDeviceToHost);
    const char *names[4] = {"acc VGPR, fragments VGPR", "acc AGPR, fragments VGPR", "acc VGPR, fragments AGPR", "acc AGPR, fragments AGPR"};
    const double per_body = (double)tk / iters;
    printf("%-26s waves/SIMD=%d  ds_read_b128 per 24 MFMAs: %2d   %7.1f cycles per body per wave = %5.1f per MFMA per SIMD "
           "(pure MFMA: 16.4)\n", names[VAR], wps, NL, per_body, per_body / 24.0 / wps);
    fflush(stdout);
}

int main() {
    float *out;
    long long *ticks;
    (void)hipMalloc(&out, 256 * 512 * sizeof(float));
    (void)hipMalloc(&ticks, sizeof(long long));
    printf("# tools/microbench/mfma_lds_regfile.hip on MI355X: 256 workgroups, body = 24 x v_mfma_f32_16x16x32_f16 + NL x ds_read_b128\n"
           "# (conflict-free 1 KB reads).  With two waves per SIMD a body's cycles are shared by both waves' 48 MFMAs.\n");
    for (int wps : {1, 2}) {
        run<0, 0>(wps, out, ticks); run<0, 6>(wps, out, ticks); run<0, 12>(wps, out, ticks);
        run<1, 0>(wps, out, ticks); run<1, 6>(wps, out, ticks); run<1, 12>(wps, out, ticks);
        run<2, 0>(wps, out, ticks); run<2, 6>(wps, out, ticks); run<2, 12>(wps, out, ticks);
        run<3, 0>(wps, out, ticks); run<3, 6>(wps, out, ticks); run<3, 12>(wps, out, ticks);
    }
    return 0;
}
