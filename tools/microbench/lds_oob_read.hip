#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned *addrs, float *out, int n, int fill_floats) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < fill_floats; i += blockDim.x) reinterpret_cast<float *>(smem)[i] = 1.0f + i;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        f32x4 v = *reinterpret_cast<const f32x4 *>(smem + addrs[i]);
        out[4 * i] = v[0]; out[4 * i + 1] = v[1]; out[4 * i + 2] = v[2]; out[4 * i + 3] = v[3];
    }
}
int main() {
    const int lds = 152416;   // like WinoCfg<9,3>::LDS_BYTES
    unsigned h[] = {0u, 16u, 152400u, 152416u, 152432u, 160000u, 163840u, 163840u + 16u, 200000u, 0x80000000u, 0x80000010u, 0xFFFFFF00u, 0xFFFFFFF0u, (unsigned)-272, 65536u*4};
    const int n = sizeof(h) / 4;
    unsigned *d; float *o; float ho[4 * 32];
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(ho));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), lds, 0, d, o, n, lds / 4);
    hipError_t e = hipDeviceSynchronize();
    printf("sync: %s\n", hipGetErrorString(e));
    hipMemcpy(ho, o, n * 16, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("addr %10u (0x%08x): %g %g %g %g\n", h[i], h[i], ho[4*i], ho[4*i+1], ho[4*i+2], ho[4*i+3]);
    return 0;
}
