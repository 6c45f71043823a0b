// Device helpers shared by the one-axis-Winograd forward kernels (net_forward_w1d.hip: 9x9; net_forward_w1dband.hip: 19x19 over two
// workgroups): absolute LDS addressing, the per-phase lane id, the unscaled f16 operand split, weight-fragment requests
// into AGPRs / VGPRs by inline asm.
#pragma once
#include "split_common.h"

namespace {

constexpr int kWsRangeLimit = 16000;                       // |V| <= 2 |d| must stay below 65504 (f16): a layer output beyond this raises the range flag

// LDS accesses by ABSOLUTE LDS byte address (the kernel has no static LDS: the dynamic array starts at 0, checked at
// kernel start).  Through `smem + addr` every access costs a v_add_u32 with the array's (relocatable) base.
typedef __attribute__((address_space(3))) f32x4 lds_f32x4_t;
template <int OFF>
__device__ __forceinline__ f32x4 lds_f32x4_at(int addr) {
    return *reinterpret_cast<const lds_f32x4_t *>(static_cast<unsigned>(addr + OFF));
}
template <int OFF>
__device__ __forceinline__ void lds_f32x4_put(int addr, f32x4 v) {
    *reinterpret_cast<lds_f32x4_t *>(static_cast<unsigned>(addr + OFF)) = v;
}

// The lane id, computed where it is asked for: hipcc treats the mbcnt pair as a pure value, computes it once at the top of a
// persistent kernel and - with the register file full of weight fragments - keeps it in scratch, one exposed reload per use.
// A volatile asm is neither hoisted nor merged: two instructions per phase instead.
__device__ __forceinline__ int fresh_lane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

// "These values are used here": hipcc waits for a load it tracks where the value is first used, and with vmcnt(0) - it does
// not see the weight requests of the inline asm, so the wait must sit where none of them is in flight.  (A free function:
// clang rejects asm operands that name captured variables inside a generic lambda.)
__device__ __forceinline__ void use_here(f32x4 &v, float &s) {
    asm volatile("" : "+v"(v), "+v"(s));
}

// Low pieces of two values whose high pieces are packed in h: f16(v0 - h.lo) | f16(v1 - h.hi) << 16, i.e. v_fma_mixlo_f16 /
// v_fma_mixhi_f16 with the f16 halves of h as source 0, -1.0 as source 1 and the fp32 value as source 2: the difference is
// exact in fp32 and rounded once.  (hipcc does not form these from C: it emits v_cvt_f32_f16 + v_sub_f32 + v_cvt_pk_f16_f32.)
__device__ __forceinline__ unsigned low_pieces(float v0, float v1, unsigned h) {
    unsigned r;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(v0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(h), "v"(v1));
    return r;
}
// four fp32 values -> two packed registers of high pieces, two of UNSCALED low pieces (6 VALU instructions)
__device__ __forceinline__ void split4_unscaled(const f32x4 v, unsigned (&hi)[2], unsigned (&lo)[2]) {
    hi[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{v[0], v[1]}, f16x2));
    hi[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{v[2], v[3]}, f16x2));
    lo[0] = low_pieces(v[0], v[1], hi[0]);
    lo[1] = low_pieces(v[2], v[3], hi[1]);
}

// dualnet_fwd_w1d_kernel: fragment F = 8 kc + 4 piece + ct of a tap block ([kc 2][piece 2][ct 4][lane][16 B]) into AGPR slot SLOT
template <int SLOT, int F>
__device__ __forceinline__ void w1_request(i32x4v (&ua)[4][2][2][4], const unsigned char *tapbase, int wlane, std::integral_constant<int, F>) {
    constexpr int kc = F >> 3, p = (F >> 2) & 1, ct = F & 3;
    const unsigned char *base = tapbase + kc * 8192 + p * 4096;
    if constexpr (ct == 0) asm volatile("global_load_dwordx4 %0, %1, %2" : "=a"(ua[SLOT][kc][p][0]) : "v"(wlane), "s"(base) : "memory");
    else if constexpr (ct == 1) asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=a"(ua[SLOT][kc][p][1]) : "v"(wlane), "s"(base) : "memory");
    else if constexpr (ct == 2) asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "=a"(ua[SLOT][kc][p][2]) : "v"(wlane), "s"(base) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, %2 offset:3072" : "=a"(ua[SLOT][kc][p][3]) : "v"(wlane), "s"(base) : "memory");
}
// ... the same fragment into a VGPR destination (the three-board variant keeps tap 2 / k-chunk 1 of odd layers there)
template <int F>
__device__ __forceinline__ void w1_request_v(i32x4v &dst, const unsigned char *tapbase, int wlane, std::integral_constant<int, F>) {
    constexpr int kc = F >> 3, p = (F >> 2) & 1, ct = F & 3;
    const unsigned char *base = tapbase + kc * 8192 + p * 4096;
    if constexpr (ct == 0) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(wlane), "s"(base) : "memory");
    else if constexpr (ct == 1) asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(dst) : "v"(wlane), "s"(base) : "memory");
    else if constexpr (ct == 2) asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "=v"(dst) : "v"(wlane), "s"(base) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, %2 offset:3072" : "=v"(dst) : "v"(wlane), "s"(base) : "memory");
}

template <int SLOT>
__device__ __forceinline__ void w1_request_tap(i32x4v (&ua)[4][2][2][4], const unsigned char *tapbase, int wlane) {
    static_for<16>([&](auto F_) { w1_request<SLOT>(ua, tapbase, wlane, F_); });
}


}  // namespace
