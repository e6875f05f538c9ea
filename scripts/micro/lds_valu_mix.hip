// lds_valu_mix.hip -- does a wave's LDS gather work OVERLAP with its VALU work on gfx950, or do the two pipes take turns?
//
// Round 5 modelled k_scan_region's time as (LDS-array cycles) + (VALU cycles at 2 per wave64 instruction), within 5 %, and
// concluded that the two pipes take turns (profiles/r05_region_model.md).  This microbenchmark isolates the question with the
// kernel's own inner loop -- reg_walk of pigo_kernels.hip.inc: per tree level and window ONE ds_read_b64 (the packed offset pairs
// of both children), TWO ds_read_u8 (the pixels, pigo.go:126-135), a compare, two selects and the next node's address, N windows
// per lane side by side -- over a synthetic region and a synthetic tree, and turns knobs the kernel cannot turn by itself:
//   K      extra VALU instructions (v_xad_u32, on the walk's dependent chain) per window and level on top of the walk's own ~7;
//   PAT    the pixel reads' address pattern: coherent (every lane reads base + lane * step + the SAME offset whatever node it
//          stands on: conflict-free) or divergent (each node its own offsets, uniform in +-12 rows / columns; the lanes spread
//          over the nodes level by level as in a real walk);
//   MODE   the schedule inside a wave: 0 lock step (all loads of a level, then all arithmetic -- rounds 1-4), 1 two half-batches
//          half a level apart (round 5's reg_walk), 2 a rotation window by window (retire window n, issue its next level at once),
//          3 the child's 4-byte entry loaded AFTER the compare (the bit enters the node index through v_addc: 5 instead of 6
//          VALU instructions per window-level, two LDS round trips instead of one);
//   waves  16 / 12 / 8 per CU (4, 3, 2 per SIMD) and N = 8 / 4 windows per lane;
//   and the same arithmetic WITHOUT the loads (register moves in their place): the VALU side alone.
// Output: cycles per (window-level of one wave) per CU -- wall time x clock / (levels x windows x waves per CU).
// What it showed (profiles/r06_lds_valu_mix.txt, r06_experiments.md section 1):
//   * an extra VALU instruction per window-level costs 1.05 cycles per CU = 4.2 cycles per SIMD at every occupancy: a wave64 VALU
//     instruction of this class takes a SIMD 4 cycles, not the 2 that MI355X_MICROARCH.md quotes for v_fma_f32 (valu_rate.hip has
//     the table per opcode).  Round 5's "VALU 37 % busy" was therefore ~75 % busy, and its sum model only fitted by coincidence;
//   * the pipes DO overlap here: with the LDS side at ~11.9 cycles per window-level (divergent) and the VALU side at 11.1 (K = 4)
//     the mix runs at 12.6 -- max(), not sum(); coherent: LDS ~7.5, VALU 7.2, mix 7.7.  The schedule inside the wave (modes 0-2) is
//     worth nothing, the dependent entry load (mode 3) loses;
//   * so what k_scan_region loses against max() is not in the walk's steady state: it is in what surrounds it (decode, compaction,
//     queue traffic, chunk claims, the stage boundaries' drains) and in the latency-bound small batches of the later stages.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/lds_valu_mix.hip -o scripts/micro/lds_valu_mix && scripts/micro/lds_valu_mix
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <type_traits>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kPitch = 332;              // bytes (83 dwords: odd, like the regions of k_scan_region)
constexpr int kRows = 300;
constexpr int kPixBytes = kPitch * kRows;     // 99,600 B of "pixels"
constexpr int kTabOff = 100 << 10;       // the tree's 64 packed entries {int16 d1, int16 d2}
constexpr int kLdsBytes = (100 << 10) + 1024;

typedef __attribute__((address_space(3))) const uint8_t *lds_u8;
typedef __attribute__((address_space(3))) const uint32_t *lds_u32;
typedef __attribute__((address_space(3))) const unsigned long long *lds_u64;

__device__ __forceinline__ uint32_t pick_half(unsigned long long ch, bool bit)
{
    uint32_t lo = (uint32_t)ch, hi = (uint32_t)(ch >> 32);
    asm("" : "+v"(lo), "+v"(hi));
    return bit ? hi : lo;
}

template <int K>
__device__ __forceinline__ void extra_valu(uint32_t &x, uint32_t e)
{
#pragma unroll
    for (int k = 0; k < K; ++k) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(x) : "v"(e), "v"(x));  // x = (x ^ e) + x: full-rate, on the chain
}

// N windows per lane, `iters` tree walks of 6 levels each.  LOADS = false: the same arithmetic with the three loads of a
// window-level replaced by register moves (the VALU side alone).
template <int N, int K, int MODE, bool LOADS>
__global__ __launch_bounds__(1024) void k_mix(const uint8_t *__restrict__ pix, const uint32_t *__restrict__ tab, int iters, int step, uint32_t *out)
{
    extern __shared__ __align__(16) uint8_t smem[];
    for (int i = threadIdx.x; i < kPixBytes / 4; i += blockDim.x) reinterpret_cast<uint32_t *>(smem)[i] = reinterpret_cast<const uint32_t *>(pix)[i];
    for (int i = threadIdx.x; i < 64; i += blockDim.x) reinterpret_cast<uint32_t *>(smem + kTabOff)[i] = tab[i];
    __syncthreads();
    const uint32_t sb = (uint32_t)(size_t)(__attribute__((address_space(3))) uint8_t *)smem;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t tb = sb + kTabOff;
    uint32_t base[N], x[N], pp[N], e[N];
    // 64 neighbouring windows of a row per wave and batch (step px apart), a wave's N batches on N rows, the waves further down
#pragma unroll
    for (int n = 0; n < N; ++n) {
        base[n] = sb + (uint32_t)((40 + (int)((wave * N + n) % 16u) * 13) * kPitch + 40 + (int)lane * step);
        x[n] = lane + n;
    }
    const uint32_t k0 = 0u - tb, k1 = 8u - tb;
    bool bit0 = false;
    unsigned long long ch[N];
    uint32_t p1[N], p2[N];
    auto issue = [&](auto lc, auto loc, auto hic) {
        constexpr int l = decltype(lc)::value, lo = decltype(loc)::value, hi = decltype(hic)::value;
#pragma unroll
        for (int n = lo; n < hi; ++n) {
            const int d1 = (int)(short)(e[n] & 0xffffu), d2 = ((int)e[n]) >> 16;
            if constexpr (LOADS) {
                if constexpr (l < 5) ch[n] = *(lds_u64)(size_t)pp[n];
                p1[n] = *(lds_u8)(size_t)(uint32_t)((int)base[n] + d1);
                p2[n] = *(lds_u8)(size_t)(uint32_t)((int)base[n] + d2);
            } else {
                uint32_t a = (uint32_t)((int)base[n] + d1), b = (uint32_t)((int)base[n] + d2), c = pp[n];
                asm volatile("" : "+v"(a), "+v"(b), "+v"(c));
                ch[n] = ((unsigned long long)(c * 0u + 0x00050003u) << 32) | 0xfffb0002u;
                p1[n] = a & 0xffu;
                p2[n] = b & 0xffu;
            }
        }
    };
    auto retire = [&](auto lc, auto loc, auto hic) {
        constexpr int l = decltype(lc)::value, lo = decltype(loc)::value, hi = decltype(hic)::value;
#pragma unroll
        for (int n = lo; n < hi; ++n) {
            const bool bit = p1[n] <= p2[n];
            if constexpr (l < 5) {
                pp[n] = (pp[n] << 1) + (bit ? k1 : k0);
                e[n] = pick_half(ch[n], bit);
            } else {
                x[n] += pp[n] + (bit ? 1u : 0u);
                bit0 = bit0 != bit;
            }
            extra_valu<K>(x[n], e[n]);
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using IN = std::integral_constant<int, N>;
    using IH = std::integral_constant<int, N / 2>;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < N; ++n) {
            pp[n] = tb + 8u;
            e[n] = *(lds_u32)(size_t)(tb + 4u);
        }
        if constexpr (MODE == 0) {
            auto level = [&](auto lc) {
                issue(lc, I0{}, IN{});
                __builtin_amdgcn_sched_barrier(0);
                retire(lc, I0{}, IN{});
            };
            level(std::integral_constant<int, 0>{});
            level(std::integral_constant<int, 1>{});
            level(std::integral_constant<int, 2>{});
            level(std::integral_constant<int, 3>{});
            level(std::integral_constant<int, 4>{});
            level(std::integral_constant<int, 5>{});
        } else if constexpr (MODE == 1) {
            issue(I0{}, I0{}, IH{});
            auto level = [&](auto lc) {
                constexpr int l = decltype(lc)::value;
                issue(lc, IH{}, IN{});
                __builtin_amdgcn_sched_barrier(0);
                retire(lc, I0{}, IH{});
                if constexpr (l < 5) issue(std::integral_constant<int, l + 1>{}, I0{}, IH{});
                __builtin_amdgcn_sched_barrier(0);
                retire(lc, IH{}, IN{});
            };
            level(std::integral_constant<int, 0>{});
            level(std::integral_constant<int, 1>{});
            level(std::integral_constant<int, 2>{});
            level(std::integral_constant<int, 3>{});
            level(std::integral_constant<int, 4>{});
            level(std::integral_constant<int, 5>{});
        } else if constexpr (MODE == 3) {
            // dependent entry load: the compare's bit goes into the node index through the carry (idx = 2 idx + bit, ONE v_addc), the
            // child's 4-byte entry is loaded AFTER the compare -- two LDS round trips per level, but 5 instead of 6 VALU instructions
            // per window-level (no select of the pair's halves) and a 4-byte instead of an 8-byte table read
            uint32_t idx[N];
#pragma unroll
            for (int n = 0; n < N; ++n) idx[n] = 1u;
#pragma unroll
            for (int l = 0; l < 6; ++l) {
#pragma unroll
                for (int n = 0; n < N; ++n) {
                    const int d1 = (int)(short)(e[n] & 0xffffu), d2 = ((int)e[n]) >> 16;
                    p1[n] = *(lds_u8)(size_t)(uint32_t)((int)base[n] + d1);
                    p2[n] = *(lds_u8)(size_t)(uint32_t)((int)base[n] + d2);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int n = 0; n < N; ++n) {
                    // idx = 2 idx + (p1 <= p2): the compare's bit enters through the carry (the compiler selects 0 / 1 and adds)
                    asm("v_cmp_le_u32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(idx[n]) : "v"(p1[n]), "v"(p2[n]) : "vcc");
                    if (l < 5) e[n] = *(lds_u32)(size_t)(tb + idx[n] * 4u);
                    else x[n] += idx[n];
                    extra_valu<K>(x[n], e[n]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // rotation: window n's level l is retired and its level l + 1 issued at once -- the other N - 1 windows' loads are in flight
            issue(I0{}, I0{}, IN{});
            auto level = [&](auto lc) {
                constexpr int l = decltype(lc)::value;
                auto one = [&](auto nc) {
                    constexpr int n = decltype(nc)::value;
                    __builtin_amdgcn_sched_barrier(0);
                    retire(lc, std::integral_constant<int, n>{}, std::integral_constant<int, n + 1>{});
                    if constexpr (l < 5) issue(std::integral_constant<int, l + 1>{}, std::integral_constant<int, n>{}, std::integral_constant<int, n + 1>{});
                };
                one(std::integral_constant<int, 0>{});
                if constexpr (N > 1) one(std::integral_constant<int, 1 % N>{});
                if constexpr (N > 2) one(std::integral_constant<int, 2 % N>{});
                if constexpr (N > 3) one(std::integral_constant<int, 3 % N>{});
                if constexpr (N > 4) one(std::integral_constant<int, 4 % N>{});
                if constexpr (N > 5) one(std::integral_constant<int, 5 % N>{});
                if constexpr (N > 6) one(std::integral_constant<int, 6 % N>{});
                if constexpr (N > 7) one(std::integral_constant<int, 7 % N>{});
            };
            level(std::integral_constant<int, 0>{});
            level(std::integral_constant<int, 1>{});
            level(std::integral_constant<int, 2>{});
            level(std::integral_constant<int, 3>{});
            level(std::integral_constant<int, 4>{});
            level(std::integral_constant<int, 5>{});
        }
    }
    uint32_t acc = bit0 ? 1u : 0u;
#pragma unroll
    for (int n = 0; n < N; ++n) acc += x[n];
    if (acc == 0xdeadbeefu) out[0] = acc;
}

struct Runner {
    uint8_t *d_pix = nullptr;
    uint32_t *d_tab = nullptr, *d_out = nullptr;
    hipEvent_t e0, e1;
    int cus = 0;
    double ghz = 2.4;
    int iters = 300;
    template <class Kern>
    double run(Kern kern, int threads, int nwin)
    {
        CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
        float ms[2] = {0, 0};
        for (int k = 0; k < 2; ++k) {
            const int it = k == 0 ? iters : 3 * iters;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(kern, dim3(cus), dim3(threads), kLdsBytes, 0, d_pix, d_tab, it, 2, d_out);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float t = 0;
                CHECK(hipEventElapsedTime(&t, e0, e1));
                ms[k] = rep == 0 ? t : std::min(ms[k], t);
            }
        }
        // slope between the two iteration counts: launch, LDS fill and drain cancel
        const double wl = 2.0 * iters * 6.0 * nwin * (threads / 64);  // window-levels of a CU
        return (ms[1] - ms[0]) * 1e6 * ghz / wl;
    }
};

template <int N, int MODE>
void row(Runner &r, const char *pat, int threads)
{
    printf("%-9s mode %d  %2d waves  N=%d |", pat, MODE, threads / 64, N);
    printf(" %6.2f", r.run(k_mix<N, 0, MODE, true>, threads, N));
    printf(" %6.2f", r.run(k_mix<N, 4, MODE, true>, threads, N));
    printf(" %6.2f", r.run(k_mix<N, 8, MODE, true>, threads, N));
    printf(" %6.2f", r.run(k_mix<N, 16, MODE, true>, threads, N));
    printf(" %6.2f", r.run(k_mix<N, 24, MODE, true>, threads, N));
    printf(" | VALU alone:");
    printf(" %6.2f", r.run(k_mix<N, 0, MODE, false>, threads, N));
    printf(" %6.2f", r.run(k_mix<N, 8, MODE, false>, threads, N));
    printf(" %6.2f", r.run(k_mix<N, 24, MODE, false>, threads, N));
    printf("\n");
    fflush(stdout);
}

int main(int argc, char **argv)
{
    Runner r;
    r.iters = argc > 1 ? atoi(argv[1]) : 300;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    r.cus = prop.multiProcessorCount;
    r.ghz = prop.clockRate / 1e6;
    printf("device %s, %d CUs, nominal %.0f MHz.  cycles per (window x level) of one wave, per CU; one workgroup per CU\n", prop.name, r.cus, prop.clockRate / 1000.0);
    printf("columns: K = 0 4 8 16 24 extra VALU instructions per window-level (the walk's own: ~8); then the same arithmetic WITHOUT the loads (K = 0 8 24)\n");
    printf("a window-level is 1 ds_read_b64 + 2 ds_read_u8 and ~7 VALU instructions; mode 3 has no variant without loads (its last three columns repeat the mix)\n");
    std::vector<uint8_t> pix(kPixBytes);
    std::mt19937 rng(777);
    for (auto &p : pix) p = (uint8_t)(rng() & 0xff);
    CHECK(hipMalloc(&r.d_pix, kPixBytes));
    CHECK(hipMalloc(&r.d_tab, 256));
    CHECK(hipMalloc(&r.d_out, 64));
    CHECK(hipMemcpy(r.d_pix, pix.data(), kPixBytes, hipMemcpyHostToDevice));
    CHECK(hipEventCreate(&r.e0));
    CHECK(hipEventCreate(&r.e1));
    for (int pat = 0; pat < 2; ++pat) {
        // the tree's 63 nodes: {d1, d2} byte offsets into the region.  coherent: every node the SAME pair (all lanes of a wave read
        // base + lane * step + one offset whatever node they stand on: conflict-free); divergent: uniform in +-12 rows / columns
        std::vector<uint32_t> tab(64, 0);
        for (int i = 1; i < 64; ++i) {
            auto off = [&]() { return pat == 0 ? 3 * kPitch + 5 : ((int)(rng() % 25u) - 12) * kPitch + (int)(rng() % 25u) - 12; };
            const int d1 = off(), d2 = pat == 0 ? -2 * kPitch - 7 : off();
            tab[i] = (uint32_t)(d1 & 0xffff) | ((uint32_t)d2 << 16);
        }
        CHECK(hipMemcpy(r.d_tab, tab.data(), 256, hipMemcpyHostToDevice));
        const char *name = pat == 0 ? "coherent" : "divergent";
        row<8, 0>(r, name, 1024);
        row<8, 1>(r, name, 1024);
        row<8, 2>(r, name, 1024);
        row<8, 3>(r, name, 1024);
        row<4, 0>(r, name, 1024);
        row<4, 2>(r, name, 1024);
        row<8, 1>(r, name, 768);
        row<8, 2>(r, name, 768);
        row<8, 1>(r, name, 512);
        row<8, 2>(r, name, 512);
        row<4, 2>(r, name, 512);
    }
    return 0;
}
