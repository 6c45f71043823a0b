// What does it cost to hand 4 864 bytes (one 19-cell edge row of fp32 [cell][64 channels]) from one workgroup to another on
// MI355X?  Two workgroups (blocks 0 and `dist`: the same XCD when dist % 8 == 0) play ping-pong: A publishes, B consumes and
// checks every word, B publishes, A consumes ...  One-way latency = time / (2 x rounds).  Variants:
//   0  flag protocol, agent scope throughout: 16-byte `sc1` stores, s_waitcnt vmcnt(0), barrier, flag store (relaxed, agent);
//      consumer: one lane polls the flag (relaxed, agent), barrier, 16-byte `sc1` loads        (net_forward_band.hip's exchange)
//   1  flag protocol through the XCD's L2: plain stores (the vector L1 writes through), vmcnt(0), barrier, flag = plain store;
//      consumer polls with an atomic add of 0 (executed AT the L2), loads the data with `nt` (misses the L1)   [same XCD only]
//   2  tagged chunks, agent scope: every 16-byte chunk = 3 payload words + the round number; consumer loads the chunks with
//      `sc1` until every tag matches (no flag, no store acknowledgement)
//   3  tagged chunks through the L2: plain stores, `nt` loads                                                  [same XCD only]
//   4  as 1, data loads `sc1` instead of `nt`
// Every consumed word is compared with what the producer wrote for that round: a stale read is counted, not tolerated.
//   hipcc --offload-arch=gfx950 -O3 -o xwg_pingpong tools/microbench/xwg_pingpong.hip && ./xwg_pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int kWords = 19 * 64;          // payload words (4 864 bytes)
constexpr int kChunks = (kWords + 2) / 3;   // tagged: 3 payload words per 16-byte chunk
constexpr int kSpin = 1 << 22;

__device__ __forceinline__ unsigned payload(int round, int side, int w) { return 0x9E3779B9u * (unsigned)(round * 2 + side + 1) + (unsigned)w * 2654435761u; }

template <int V>
__global__ __launch_bounds__(256) void pingpong(int *buf, int *flags, int rounds, int dist, int *bad, long long *ticks) {
    const int tid = threadIdx.x;
    int side;
    if (blockIdx.x == 0) side = 0;
    else if ((int)blockIdx.x == dist) side = 1;
    else return;
    __shared__ int dead;
    if (tid == 0) dead = 0;
    __syncthreads();
    int *area[2] = {buf, buf + 4096};                 // what side s publishes (4 096 ints = 16 KB apart)
    int *flag[2] = {flags, flags + 64};
    int errors = 0;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 1; r <= rounds; ++r) {
        for (int turn = 0; turn < 2; ++turn) {
            if (turn == side) {
                // ---- publish ----
                if constexpr (V == 0 || V == 1 || V == 4) {
                    for (int c = tid; c < kWords / 4; c += 256) {
                        i32x4 v;
                        for (int j = 0; j < 4; ++j) v[j] = (int)payload(r, side, c * 4 + j);
                        int *p = area[side] + c * 4;
                        if constexpr (V == 0) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
                        else asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    if (tid == 0) {
                        if constexpr (V == 0) __hip_atomic_store(flag[side], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        else asm volatile("global_store_dword %0, %1, off" ::"v"(flag[side]), "v"(r) : "memory");
                    }
                } else {
                    for (int c = tid; c < kChunks; c += 256) {
                        i32x4 v;
                        for (int j = 0; j < 3; ++j) v[j] = c * 3 + j < kWords ? (int)payload(r, side, c * 3 + j) : 0;
                        v[3] = r;
                        int *p = area[side] + c * 4;
                        if constexpr (V == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
                        else asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
                    }
                }
            } else {
                // ---- consume ----
                const int other = 1 - side;
                if constexpr (V == 0 || V == 1 || V == 4) {
                    if (tid == 0) {
                        int n = 0;
                        for (;;) {
                            int f;
                            if constexpr (V == 0) f = __hip_atomic_load(flag[other], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            else f = atomicAdd(flag[other], 0);
                            if (f >= r) break;
                            if (++n > kSpin) { dead = 1; break; }
                        }
                    }
                    __syncthreads();
                    if (dead) { if (tid == 0) atomicAdd(bad, 1 << 20); return; }
                    for (int c = tid; c < kWords / 4; c += 256) {
                        i32x4 v;
                        const int *p = area[other] + c * 4;
                        if constexpr (V == 1) asm volatile("global_load_dwordx4 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
                        else asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
                        for (int j = 0; j < 4; ++j) errors += (unsigned)v[j] != payload(r, other, c * 4 + j);
                    }
                } else {
                    for (int c = tid; c < kChunks; c += 256) {
                        i32x4 v;
                        const int *p = area[other] + c * 4;
                        int n = 0;
                        for (;;) {
                            if constexpr (V == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
                            else asm volatile("global_load_dwordx4 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
                            if (v[3] >= r) break;
                            if (++n > kSpin) { dead = 1; break; }
                        }
                        for (int j = 0; j < 3; ++j)
                            if (c * 3 + j < kWords) errors += v[3] == r && (unsigned)v[j] != payload(r, other, c * 3 + j);
                    }
                    __syncthreads();
                    if (dead) { if (tid == 0) atomicAdd(bad, 1 << 20); return; }
                }
            }
            __syncthreads();
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (errors) atomicAdd(bad, errors);
    if (tid == 0) ticks[side] = t1 - t0;
}

template <int V>
void run(const char *name, int dist, int rounds, int *buf, int *flags, int *bad, long long *ticks) {
    (void)hipMemset(buf, 0, 8192 * 4);
    (void)hipMemset(flags, 0, 128 * 4);
    (void)hipMemset(bad, 0, 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(pingpong<V>, dim3(dist + 1), dim3(256), 0, 0, buf, flags, rounds, dist, bad, ticks);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    int hbad = 0;
    (void)hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost);
    printf("variant %d  %-44s blocks 0 / %-3d (%s XCD)  one-way %7.3f us   wrong words %d%s\n", V, name, dist, dist % 8 == 0 ? "same " : "other",
           ms * 1e3 / (2.0 * rounds), hbad & ((1 << 20) - 1), hbad >> 20 ? "   GAVE UP WAITING" : "");
    fflush(stdout);
}

int main() {
    int *buf, *flags, *bad;
    long long *ticks;
    (void)hipMalloc(&buf, 8192 * 4);
    (void)hipMalloc(&flags, 128 * 4);
    (void)hipMalloc(&bad, 4);
    (void)hipMalloc(&ticks, 16);
    const int rounds = 2000;
    printf("# tools/microbench/xwg_pingpong.hip on MI355X: 4 864-byte hand-off between two workgroups, %d rounds\n", rounds);
    for (int rep = 0; rep < 2; ++rep)
        for (int dist : {8, 128, 1, 129}) {
            run<0>("flag, sc1 stores / loads (agent scope)", dist, rounds, buf, flags, bad, ticks);
            run<2>("tagged chunks, sc1", dist, rounds, buf, flags, bad, ticks);
            if (dist % 8 == 0) {
                run<1>("flag at L2: plain stores, atomic poll, nt loads", dist, rounds, buf, flags, bad, ticks);
                run<4>("flag at L2: plain stores, atomic poll, sc1 loads", dist, rounds, buf, flags, bad, ticks);
                run<3>("tagged chunks at L2: plain stores, nt loads", dist, rounds, buf, flags, bad, ticks);
            }
        }
    return 0;
}
