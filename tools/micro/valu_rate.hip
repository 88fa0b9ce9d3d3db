// micro-benchmark (gfx950): how long does a wave64 integer VALU instruction occupy its SIMD, and what does a
// binary-search probe on an LDS window cost?  Settles DESIGN.md section 3 ("4 cycles per wave64 VALU instruction",
// from SQ_ACTIVE_INST_VALU) against MI355X_MICROARCH.md:52-54 ("SIMD-32, 2 cycles").
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
//
// Method: workgroups of 256 threads (one wave per SIMD of a CU), k workgroups resident per CU (dynamic LDS sized to
// 160 KB / k forces exactly that), grid = 256 CUs * k: every SIMD holds k waves for the whole kernel.  Each wave runs
// the same instruction stream; cycles per instruction = s_memrealtime-independent: wall time (HIP events) * f / (instr
// per wave * k), reported for f = 2.4 GHz and as "ns per wave-instruction per SIMD".
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <hip/hip_runtime.h>
typedef unsigned long long u64; typedef unsigned int u32;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

extern __shared__ int dyn_lds[];

#define REP8(X) X X X X X X X X
#define REP16(X) REP8(X) REP8(X)

// ---- A: independent v_add_u32 chains (8 chains x 16 = 128 instructions per iteration) -------------------------
__global__ void __launch_bounds__(256) k_add(int iters, int* out)
{
    int a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, b = blockIdx.x | 1;
    for (int it = 0; it < iters; ++it) {
        REP16(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                           "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
    }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x7fffffff) out[0] = 1;
    if (dyn_lds[threadIdx.x] == 12345) out[1] = 1;
}
// ---- A2: ONE dependent chain (latency of a back-to-back dependent v_add_u32) --------------------------------------
__global__ void __launch_bounds__(256) k_add_dep(int iters, int* out)
{
    int a0 = threadIdx.x, b = blockIdx.x | 1;
    for (int it = 0; it < iters; ++it) {
        REP16(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n"
                           "v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n"
                           : "+v"(a0) : "v"(b));)
    }
    if (a0 == 0x7fffffff) out[0] = 1;
    if (dyn_lds[threadIdx.x] == 12345) out[1] = 1;
}
// ---- B: v_cmp_lt_i32 + v_cndmask_b32 pairs (8 independent pairs x 16 per iteration = 256 instructions) ------------
__global__ void __launch_bounds__(256) k_cmp32(int iters, int* out)
{
    int a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, b = blockIdx.x | 1;
    for (int it = 0; it < iters; ++it) {
        REP16(asm volatile("v_cmp_lt_i32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %8, vcc\n v_cmp_lt_i32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %8, vcc\n"
                           "v_cmp_lt_i32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %8, vcc\n v_cmp_lt_i32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %8, vcc\n"
                           "v_cmp_lt_i32 vcc, %4, %8\n v_cndmask_b32 %4, %4, %8, vcc\n v_cmp_lt_i32 vcc, %5, %8\n v_cndmask_b32 %5, %5, %8, vcc\n"
                           "v_cmp_lt_i32 vcc, %6, %8\n v_cndmask_b32 %6, %6, %8, vcc\n v_cmp_lt_i32 vcc, %7, %8\n v_cndmask_b32 %7, %7, %8, vcc\n"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");)
    }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x7fffffff) out[0] = 1;
    if (dyn_lds[threadIdx.x] == 12345) out[1] = 1;
}
// ---- C: v_cmp_lt_u64 + v_cndmask_b32 pairs --------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_cmp64(int iters, int* out)
{
    int a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, b = blockIdx.x | 1;
    u64 k0 = threadIdx.x * 77ull, k1 = k0 + 1, k2 = k0 + 2, k3 = k0 + 3, lim = (u64)blockIdx.x << 20;
    for (int it = 0; it < iters; ++it) {
        REP16(asm volatile("v_cmp_lt_u64 vcc, %4, %8\n v_cndmask_b32 %0, %0, %9, vcc\n v_cmp_lt_u64 vcc, %5, %8\n v_cndmask_b32 %1, %1, %9, vcc\n"
                           "v_cmp_lt_u64 vcc, %6, %8\n v_cndmask_b32 %2, %2, %9, vcc\n v_cmp_lt_u64 vcc, %7, %8\n v_cndmask_b32 %3, %3, %9, vcc\n"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k0), "v"(k1), "v"(k2), "v"(k3), "v"(lim), "v"(b) : "vcc");)
    }
    if (a0 + a1 + a2 + a3 == 0x7fffffff) out[0] = 1;
    if (dyn_lds[threadIdx.x] == 12345) out[1] = 1;
}
// ---- D: dependent ds_read_b64 chain (LDS round-trip latency at 1 wave / SIMD, throughput at more) -----------------
__global__ void __launch_bounds__(256) k_lds_chase(int iters, int* out)
{
    int2* w = reinterpret_cast<int2*>(dyn_lds);
    for (int k = threadIdx.x; k < 2048; k += 256) w[k] = make_int2((k * 37 + 11) & 2047, k);
    __syncthreads();
    int p = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) p = w[p].x;
    }
    if (p == 0x7fffffff) out[0] = 1;
}

// ---- E: the probe loops themselves ---------------------------------------------------------------------------------
// window of 2048 (q, sp) pairs sorted by (strip, q): 8 strips of 256 entries, q ascending inside a strip
template <int K, typename P>
__device__ __forceinline__ int first_true(const int2* __restrict__ w, int pos, P&& pred)
{
#pragma unroll
    for (int step = 1 << (K - 1); step >= 1; step >>= 1) {
        const int2 v = w[pos + step - 1];
        pos = pred(v) ? pos : pos + step;
    }
    return pos;
}
template <int K>
__device__ __forceinline__ int first_ge64(const u64* __restrict__ w, int pos, u64 key)
{
#pragma unroll
    for (int step = 1 << (K - 1); step >= 1; step >>= 1) {
        const u64 v = w[pos + step - 1];
        pos = (v >= key) ? pos : pos + step;
    }
    return pos;
}
template <int K>
__device__ __forceinline__ int first_ge32(const int* __restrict__ w, int pos, int key)
{
#pragma unroll
    for (int step = 1 << (K - 1); step >= 1; step >>= 1) {
        const int v = w[pos + step - 1];
        pos = (v >= key) ? pos : pos + step;
    }
    return pos;
}
#define RB 12
__device__ __forceinline__ void fill_window(int2* w)
{
    for (int k = threadIdx.x; k < 2048 + 512; k += 256) {
        const int strip = k >> 8, r = k & 255;
        w[k] = k < 2048 ? make_int2(r * 40 + ((k * 2654435761u) >> 28), (strip << RB) | ((k * 40503u) & 4095)) : make_int2(INT_MAX, INT_MAX);
    }
}
// MODE 0: pair predicates, CH independent searches per round (the current k_region_core probe)
template <int CH>
__global__ void __launch_bounds__(256) k_probe_pair(int iters, int* out)
{
    int2* w = reinterpret_cast<int2*>(dyn_lds);
    fill_window(w);
    __syncthreads();
    int acc = 0;
    int li = 256 * 3 + threadIdx.x;                // a PET of strip 3
    for (int it = 0; it < iters; ++it) {
        const int2 me = w[li];
        const int qlo = me.x - 900, pbeg = me.y & ~4095;
        int r[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int ql = qlo + c * 7, pb = pbeg + ((c & 1) << (RB + 1));
            r[c] = first_true<8>(w, 256 * 2 + ((c & 1) << 9), [&](int2 v) { return (v.y >= pb) | (v.x >= ql); });
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) acc += r[c];
        li = 256 * 3 + ((li + acc) & 255);
    }
    if (acc == 0x7fffffff) out[0] = 1;
    out[2 + blockIdx.x * 256 + threadIdx.x] = acc;
}
// MODE 1: the window as ONE sorted u64 key (strip << 32 | q): a probe is ds_read_b64 + v_cmp_u64 + select
template <int CH>
__global__ void __launch_bounds__(256) k_probe_u64(int iters, int* out)
{
    int2* w = reinterpret_cast<int2*>(dyn_lds);
    fill_window(w);
    __syncthreads();
    u64* w64 = reinterpret_cast<u64*>(dyn_lds);
    for (int k = threadIdx.x; k < 2048 + 512; k += 256) {
        const int2 v = w[k];
        __syncthreads();
        w64[k] = k < 2048 ? ((u64)(v.y >> RB) << 32) | (u32)v.x : ~0ull;
    }
    __syncthreads();
    int acc = 0;
    int li = 256 * 3 + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        const u64 me = w64[li];
        const u32 s = (u32)(me >> 32), q = (u32)me;
        int r[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const u64 key = ((u64)(s - 1 + ((c & 1) << 1)) << 32) | (q - 900 + c * 7);
            r[c] = first_ge64<8>(w64, 256 * 2 + ((c & 1) << 9), key);
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) acc += r[c];
        li = 256 * 3 + ((li + acc) & 255);
    }
    if (acc == 0x7fffffff) out[0] = 1;
    out[2 + blockIdx.x * 256 + threadIdx.x] = acc;
}
// MODE 2: q alone as a 32-bit array, single compare (the floor of any search on this hardware)
template <int CH>
__global__ void __launch_bounds__(256) k_probe_u32(int iters, int* out)
{
    int2* w = reinterpret_cast<int2*>(dyn_lds);
    fill_window(w);
    __syncthreads();
    int* w32 = dyn_lds + 2 * (2048 + 512);
    for (int k = threadIdx.x; k < 2048 + 512; k += 256) w32[k] = k < 2048 ? ((k >> 8) << 16) + w[k].x : INT_MAX;
    __syncthreads();
    int acc = 0;
    int li = 256 * 3 + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        const int me = w32[li];
        int r[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int key = me - (1 << 16) + ((c & 1) << 17) - 900 + c * 7;
            r[c] = first_ge32<8>(w32, 256 * 2 + ((c & 1) << 9), key);
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) acc += r[c];
        li = 256 * 3 + ((li + acc) & 255);
    }
    if (acc == 0x7fffffff) out[0] = 1;
    out[2 + blockIdx.x * 256 + threadIdx.x] = acc;
}


// MODE 3: u64 keys in a ROW-SKEWED window (row = 32 entries, pitch 33): the power-of-two strides of a binary search from
// a common base land on distinct banks.  Bases are row aligned, so every probe offset is an immediate.
#define SKEW(j) ((j) + ((j) >> 5))
template <int K>
__device__ __forceinline__ int first_ge64_skew(const u64* __restrict__ w, int pphys, u64 key)   // returns the PHYSICAL index
{
#pragma unroll
    for (int step = 1 << (K - 1); step >= 32; step >>= 1) {
        const int pstep = step + (step >> 5);
        const u64 v = w[pphys + pstep - 2];
        pphys = (v >= key) ? pphys : pphys + pstep;
    }
#pragma unroll
    for (int step = 16; step >= 1; step >>= 1) {
        const u64 v = w[pphys + step - 1];
        pphys = (v >= key) ? pphys : pphys + step;
    }
    return pphys;
}
template <int CH>
__global__ void __launch_bounds__(256) k_probe_u64_skew(int iters, int* out)
{
    int2* w = reinterpret_cast<int2*>(dyn_lds);
    fill_window(w);
    __syncthreads();
    u64* w64 = reinterpret_cast<u64*>(dyn_lds);
    u64 tmp[10];
    for (int u = 0; u < 10; ++u) { const int k = threadIdx.x + u * 256; const int2 v = w[k]; tmp[u] = k < 2048 ? ((u64)(v.y >> RB) << 32) | (u32)v.x : ~0ull; }
    __syncthreads();
    for (int u = 0; u < 10; ++u) { const int k = threadIdx.x + u * 256; w64[SKEW(k)] = tmp[u]; }
    __syncthreads();
    int acc = 0;
    int li = 256 * 3 + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        const u64 me = w64[SKEW(li)];
        const u32 s = (u32)(me >> 32), q = (u32)me;
        int r[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const u64 key = ((u64)(s - 1 + ((c & 1) << 1)) << 32) | (q - 900 + c * 7);
            r[c] = first_ge64_skew<8>(w64, SKEW(256 * 2 + ((c & 1) << 9)), key);
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) { const int ph = r[c]; acc += ph - ((ph * 1986) >> 16); }      // physical -> logical: ph - ph / 33
        li = 256 * 3 + ((li + acc) & 255);
    }
    if (acc == 0x7fffffff) out[0] = 1;
    out[2 + blockIdx.x * 256 + threadIdx.x] = acc;
}
// MODE 4: pair predicates in the skewed window, physical address computed per probe (arbitrary bases)
template <int K, typename P>
__device__ __forceinline__ int first_true_skew(const int2* __restrict__ w, int pos, P&& pred)
{
#pragma unroll
    for (int step = 1 << (K - 1); step >= 1; step >>= 1) {
        const int idx = pos + step - 1;
        const int2 v = w[idx + (idx >> 5)];
        pos = pred(v) ? pos : pos + step;
    }
    return pos;
}
template <int CH>
__global__ void __launch_bounds__(256) k_probe_pair_skew(int iters, int* out)
{
    int2* w = reinterpret_cast<int2*>(dyn_lds);
    fill_window(w);
    __syncthreads();
    int2 tmp[10];
    for (int u = 0; u < 10; ++u) tmp[u] = w[threadIdx.x + u * 256];
    __syncthreads();
    for (int u = 0; u < 10; ++u) { const int k = threadIdx.x + u * 256; w[SKEW(k)] = tmp[u]; }
    __syncthreads();
    int acc = 0;
    int li = 256 * 3 + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        const int2 me = w[SKEW(li)];
        const int qlo = me.x - 900, pbeg = me.y & ~4095;
        int r[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int ql = qlo + c * 7, pb = pbeg + ((c & 1) << (RB + 1));
            r[c] = first_true_skew<8>(w, 256 * 2 + ((c & 1) << 9), [&](int2 v) { return (v.y >= pb) | (v.x >= ql); });
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) acc += r[c];
        li = 256 * 3 + ((li + acc) & 255);
    }
    if (acc == 0x7fffffff) out[0] = 1;
    out[2 + blockIdx.x * 256 + threadIdx.x] = acc;
}
// ---- F: raw LDS read throughput: 8 independent reads in flight per lane, addresses (a) consecutive, (b) random, (c) power-of-two strided
template <int MODE, int WIDE>
__global__ void __launch_bounds__(256) k_lds_tp(int iters, int* out)
{
    int* w = dyn_lds;
    for (int k = threadIdx.x; k < 8192; k += 256) w[k] = k * 7;
    __syncthreads();
    unsigned a = threadIdx.x;
    if (MODE == 1) a = (threadIdx.x * 2654435761u) >> 20;            // pseudo-random over 4096 entries
    if (MODE == 2) a = (threadIdx.x & 31) * 64;                      // 32 distinct addresses, stride 64 entries (binary-search level)
    if (MODE == 3) a = (threadIdx.x & 7) * 256;                      // 8 distinct addresses, stride 256
    int acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned idx = (a + u * 9 + it) & 2047;
            if (WIDE) { const int2 v = reinterpret_cast<const int2*>(w)[idx]; acc += v.x ^ v.y; }
            else acc += w[idx];
        }
    }
    if (acc == 0x7fffffff) out[0] = 1;
    out[2 + blockIdx.x * 256 + threadIdx.x] = acc;
}

template <typename F>
static double time_ms(F&& launch, int reps = 5)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main(int argc, char** argv)
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("device %s, %d CUs, clockRate %d kHz, LDS/block max %zu\n", prop.gcnArchName, ncu, prop.clockRate, prop.sharedMemPerBlock);
    int* out; CK(hipMalloc(&out, (size_t)(2 + 8 * ncu * 256) * 4 + 64));
    const double ghz = 2.4;
    const int iters = 4000;
    struct { const char* name; void (*k)(int, int*); double instr_per_iter; } valu[] = {
        {"v_add_u32 x8 independent", k_add, 128.0}, {"v_add_u32 dependent chain", k_add_dep, 128.0},
        {"v_cmp_lt_i32 + v_cndmask", k_cmp32, 256.0}, {"v_cmp_lt_u64 + v_cndmask", k_cmp64, 128.0},
    };
    const int lds_max = 160 * 1024;
    auto set_lds = [&](const void* f) { CK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max)); };
    set_lds((const void*)k_add); set_lds((const void*)k_add_dep); set_lds((const void*)k_cmp32); set_lds((const void*)k_cmp64); set_lds((const void*)k_lds_chase);
    printf("\n== VALU issue: ns and cycles@2.4GHz per wave64 instruction per SIMD, k waves resident per SIMD ==\n");
    for (auto& v : valu) {
        printf("%-28s", v.name);
        for (int k = 1; k <= 8; ++k) {
            const int lds = (lds_max / k) & ~1023;
            const double ms = time_ms([&] { hipLaunchKernelGGL(v.k, dim3(ncu * k), dim3(256), lds, 0, iters, out); });
            const double ns = ms * 1e6 / (v.instr_per_iter * iters * k);
            printf("  k=%d %.3f ns (%.2f cyc)", k, ns, ns * ghz);
        }
        printf("\n");
    }
    printf("\n== dependent ds_read_b64 chain: ns per read per wave (k = 1: latency), per SIMD aggregate below ==\n");
    for (int k = 1; k <= 8; ++k) {
        const int lds = (lds_max / k) & ~1023;
        const double ms = time_ms([&] { hipLaunchKernelGGL(k_lds_chase, dim3(ncu * k), dim3(256), lds, 0, iters, out); });
        const double ns = ms * 1e6 / (16.0 * iters);
        printf("  k=%d: %.1f ns per dependent read (%.0f cyc); %.2f ns per wave-read per SIMD\n", k, ns, ns * ghz, ns / k);
    }
    printf("\n== search probes: 8-step searches on a 2048-entry LDS window; ns per probe per wave-slot per SIMD (lower = better) ==\n");
    struct { const char* name; void (*k)(int, int*); int ch; } pr[] = {
        {"pair predicate, 1 chain", k_probe_pair<1>, 1}, {"pair predicate, 2 chains", k_probe_pair<2>, 2}, {"pair predicate, 4 chains", k_probe_pair<4>, 4},
        {"u64 key, 1 chain", k_probe_u64<1>, 1}, {"u64 key, 2 chains", k_probe_u64<2>, 2}, {"u64 key, 4 chains", k_probe_u64<4>, 4},
        {"u32 key, 1 chain", k_probe_u32<1>, 1}, {"u32 key, 2 chains", k_probe_u32<2>, 2}, {"u32 key, 4 chains", k_probe_u32<4>, 4},
        {"u64 key skewed, 1 chain", k_probe_u64_skew<1>, 1}, {"u64 key skewed, 2 chains", k_probe_u64_skew<2>, 2}, {"u64 key skewed, 4 chains", k_probe_u64_skew<4>, 4},
        {"pair skewed, 1 chain", k_probe_pair_skew<1>, 1}, {"pair skewed, 2 chains", k_probe_pair_skew<2>, 2}, {"pair skewed, 4 chains", k_probe_pair_skew<4>, 4},
    };
    const int piters = 2000;
    for (auto& p : pr) {
        set_lds((const void*)p.k);
        printf("%-26s", p.name);
        for (int k = 1; k <= 8; ++k) {
            const int lds = std::max((lds_max / k) & ~1023, 32 * 1024);
            if (lds * k > lds_max) { printf("  k=%d n/a", k); continue; }
            const double ms = time_ms([&] { hipLaunchKernelGGL(p.k, dim3(ncu * k), dim3(256), lds, 0, piters, out); });
            const double ns = ms * 1e6 / (8.0 * p.ch * piters * k);
            printf("  k=%d %.2f", k, ns);
        }
        printf("\n");
    }
    printf("\n== raw LDS reads, 8 in flight per lane: LDS cycles@2.4GHz per wave-instruction per CU (k = 4 waves per SIMD) ==\n");
    struct { const char* name; void (*k)(int, int*); } tp[] = {
        {"b32 consecutive", k_lds_tp<0, 0>}, {"b32 random", k_lds_tp<1, 0>}, {"b32 stride 64 x32", k_lds_tp<2, 0>}, {"b32 stride 256 x8", k_lds_tp<3, 0>},
        {"b64 consecutive", k_lds_tp<0, 1>}, {"b64 random", k_lds_tp<1, 1>}, {"b64 stride 64 x32", k_lds_tp<2, 1>}, {"b64 stride 256 x8", k_lds_tp<3, 1>},
    };
    for (auto& t : tp) {
        set_lds((const void*)t.k);
        const int k = 4, lds = 40 * 1024;
        const double ms = time_ms([&] { hipLaunchKernelGGL(t.k, dim3(ncu * k), dim3(256), lds, 0, piters, out); });
        const double ns = ms * 1e6 / (8.0 * piters * k * 4);
        printf("%-22s %.2f ns = %.1f cycles per wave-read per CU\n", t.name, ns, ns * ghz);
    }
    CK(hipFree(out));
    return 0;
}
