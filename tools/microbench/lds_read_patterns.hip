// LDS cycles of ds_read_b128 for the address patterns of the DualNet forward kernels, measured: every wave of a
// workgroup issues the same pattern back to back (independent destination registers), cycles per read per CU from
// s_memtime; SQ_LDS_BANK_CONFLICT can be collected on this binary as a cross-check.
//   pattern 0: linear (lane * 16) - conflict-free by construction
//   pattern 1: activation fragment of the 16x16x32 kernels: row = li + shift, 16-byte slot = lg ^ ((row >> 1) & 3), 64 B rows
//   pattern 2: the same without the swizzle (slot = lg)
//   pattern 3: 32x32x16 fragment: row = l & 31 + shift, slot = (l >> 5) ^ ((row >> 2) & 3)
//   pattern 4: pattern 1 with a third of the lanes redirected to ONE zero row (padding taps, old scheme)
//   pattern 5: pattern 1 with the same lanes redirected to zero block + (address mod 256) (new scheme)
//   hipcc --offload-arch=gfx950 -O3 -o lds_read_patterns tools/microbench/lds_read_patterns.hip && ./lds_read_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k(int pattern, int shift, int iters, long long *ticks, int *sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<unsigned *>(smem)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    int addr;
    const int row = li + shift;
    const int nat = row * 64 + ((lg ^ ((row >> 1) & 3)) << 4);
    const bool pad = (li % 3) == 0;
    switch (pattern) {
    case 0: addr = lane * 16; break;
    case 1: addr = nat; break;
    case 2: addr = row * 64 + (lg << 4); break;
    case 3: { const int r2 = (lane & 31) + shift; addr = r2 * 64 + (((lane >> 5) ^ ((r2 >> 2) & 3)) << 4); } break;
    case 4: addr = pad ? 200 * 64 + (lg << 4) : nat; break;
    default: addr = pad ? 200 * 64 + (nat & 255) : nat; break;
    }
    i32x4 d0, d1, d2, d3, d4, d5, d6, d7;
    long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:4096\n ds_read_b128 %2, %8 offset:8192\n ds_read_b128 %3, %8 offset:12288\n"
                     "ds_read_b128 %4, %8 offset:16384\n ds_read_b128 %5, %8 offset:20480\n ds_read_b128 %6, %8 offset:24576\n ds_read_b128 %7, %8 offset:28672\n"
                     "s_waitcnt lgkmcnt(0)"
                     : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3), "=v"(d4), "=v"(d5), "=v"(d6), "=v"(d7) : "v"(addr) : "memory");
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
    if (d0[0] + d1[1] + d2[2] + d3[3] + d4[0] + d5[1] + d6[2] + d7[3] == 0x12345) sink[0] = 1;
}

int main() {
    long long *ticks;
    int *sink;
    (void)hipMalloc(&ticks, 8);
    (void)hipMalloc(&sink, 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    const int iters = 2000;
    printf("# tools/microbench/lds_read_patterns.hip on MI355X: cycles per ds_read_b128 wave-instruction per CU (4 = the LDS peak,\n"
           "# 256 B/clk); WAVES waves per workgroup issue 8 reads + s_waitcnt in a loop\n");
    for (int waves : {4, 8})
        for (int pattern = 0; pattern < 6; ++pattern)
            for (int shift : {0, 1, 9, 10, 11}) {
                if (pattern == 0 && shift) continue;
                for (int rep = 0; rep < 2; ++rep) {
                    hipLaunchKernelGGL(k, dim3(256), dim3(64 * waves), 65536, 0, pattern, shift, iters, ticks, sink);
                    (void)hipDeviceSynchronize();
                }
                long long tk = 0;
                (void)hipMemcpy(&tk, ticks, 8, hipMemcpyDeviceToHost);
                printf("waves %d  pattern %d  shift %2d: %6.2f cycles per read per CU\n", waves, pattern, shift,
                       (double)tk / (iters * 8.0 * waves));
                fflush(stdout);
            }
    return 0;
}
