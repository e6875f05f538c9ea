// lds_gather.hip -- microbenchmark of the LDS pipe on gfx950 for the byte gathers of the cascade scan (k_scan_region): what does
// ONE wave-wide LDS read cost, per CU, as a function of the lanes' address pattern -- and does the bank model of
// MI355X_MICROARCH.md (section LDS: a wave64 access is served in two groups of 32 lanes, one LDS cycle per group plus one per extra
// distinct dword on a busy bank, bank = (a / 4) mod 32) predict it?  Every case prints the measured cycles per wave-instruction per
// CU next to the model's (host-side count of the same addresses), at 16 waves per CU (4 per SIMD: the region kernel's shape).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/lds_gather.hip -o scripts/micro/lds_gather && scripts/micro/lds_gather
// Patterns (the scan's: pigo.go:123-135 reads two bytes per tree level at window centre + a per-node offset):
//   linear4      lane * 4                                   -- conflict-free reference (2 cycles)
//   same         one address                                -- broadcast
//   step S       base + lane * S                            -- 64 neighbouring windows of a rung at the SAME node (level 0 of a tree)
//   adj S R      base + lane * S + dy * pitch + dx          -- neighbouring windows, every lane at its own node: (dy, dx) uniform in
//                                                              [-R, R]^2 (R = half a window: 12 for s = 24, 25 for s = 51)
//   compact R    random centre in the region + (dy, dx)     -- the survivors of a stage, compacted: lanes are unrelated windows
//   random       anywhere in the region's 128 KiB
//   rows         base + lane * pitch                        -- one column, 64 rows (odd dword pitch)
// Widths: u8 (ds_read_u8), u16, b32 (aligned dword + shift: the same LDS work, one VALU more), b64.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kThreads = 1024;           // 16 waves per CU
constexpr int kLdsBytes = 128 << 10;
constexpr int kSets = 8;                 // address sets per lane (a wave cycles through them: different conflict draws)
constexpr int kUnroll = 16;              // DS operations in flight per s_waitcnt

typedef __attribute__((address_space(3))) const uint8_t *lds_u8;
typedef __attribute__((address_space(3))) const uint16_t *lds_u16;
typedef __attribute__((address_space(3))) const uint32_t *lds_u32;
typedef __attribute__((address_space(3))) const uint64_t *lds_u64;

// addr: [kSets][blocks? no: waves 16][64] byte addresses, identical for every workgroup
template <int WIDTH>
__global__ __launch_bounds__(kThreads) void k_lds(const uint32_t *__restrict__ addr, int iters, uint32_t *out, unsigned long long *cyc)
{
    extern __shared__ __align__(16) uint8_t smem[];
    for (int i = threadIdx.x; i < kLdsBytes / 4; i += kThreads) reinterpret_cast<uint32_t *>(smem)[i] = (uint32_t)i * 2654435761u;
    __syncthreads();
    const uint32_t base = (uint32_t)(size_t)(__attribute__((address_space(3))) uint8_t *)smem;
    uint32_t a[kSets];
#pragma unroll
    for (int k = 0; k < kSets; ++k) a[k] = base + addr[(k * 16 + (threadIdx.x >> 6)) * 64 + (threadIdx.x & 63)];
    uint32_t acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            uint32_t ad = a[u % kSets];
            asm volatile("" : "+v"(ad));  // (opaque: the loads stay in the loop)
            if constexpr (WIDTH == 1) acc += *(lds_u8)(size_t)ad;
            else if constexpr (WIDTH == 2) acc += *(lds_u16)(size_t)(ad & ~1u);
            else if constexpr (WIDTH == 4) acc += (*(lds_u32)(size_t)(ad & ~3u) >> ((ad & 3u) * 8u)) & 0xffu;
            else acc += (uint32_t)*(lds_u64)(size_t)(ad & ~7u);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (acc == 0xdeadbeefu) out[0] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

struct Pattern {
    std::string name;
    std::vector<uint32_t> addr;  // [kSets][16][64]
};

// the guide's model: two groups of 32 lanes; per group the cycles are the largest number of DISTINCT dwords on one bank
static double model_cycles(const std::vector<uint32_t> &addr, int nbanks, int width)
{
    double total = 0;
    const int ninst = kSets * 16;
    for (int i = 0; i < ninst; ++i) {
        for (int g = 0; g < 2; ++g) {
            std::vector<std::vector<uint32_t>> bank(nbanks);
            for (int l = 0; l < 32; ++l) {
                uint32_t a = addr[i * 64 + g * 32 + l];
                a &= ~(uint32_t)(width >= 4 ? width - 1 : 3);
                const uint32_t dw = a / 4;
                for (int w = 0; w < std::max(1, width / 4); ++w) {
                    auto &b = bank[(dw + w) % nbanks];
                    if (std::find(b.begin(), b.end(), dw + w) == b.end()) b.push_back(dw + w);
                }
            }
            size_t mx = 1;
            for (auto &b : bank) mx = std::max(mx, b.size());
            total += (double)mx;
        }
    }
    return total / ninst;
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 400;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, nominal %.0f MHz; %d waves per CU, %d DS ops per wave in flight\n", prop.name, cus, prop.clockRate / 1000.0, kThreads / 64, kUnroll);
    std::mt19937 rng(12345);
    const int pitch = 332;  // 83 dwords: odd (k_scan_region's regions have an odd dword pitch)
    const int rows = kLdsBytes / pitch;
    auto clampa = [&](long long a) { return (uint32_t)std::min<long long>(std::max<long long>(a, 0), kLdsBytes - 8); };
    std::vector<Pattern> pats;
    auto make = [&](const std::string &name, auto f) {
        Pattern p;
        p.name = name;
        p.addr.resize((size_t)kSets * 16 * 64);
        for (int k = 0; k < kSets; ++k)
            for (int w = 0; w < 16; ++w) {
                const int brow = 30 + (int)(rng() % (unsigned)(rows - 60)), bcol = 30 + (int)(rng() % 100u);
                for (int l = 0; l < 64; ++l) p.addr[((size_t)k * 16 + w) * 64 + l] = clampa(f(brow, bcol, l));
            }
        pats.push_back(std::move(p));
    };
    auto rnd = [&](int r) { return (int)(rng() % (unsigned)(2 * r + 1)) - r; };
    make("linear4", [&](int, int, int l) { return (long long)l * 4; });
    make("same", [&](int br, int bc, int) { return (long long)br * pitch + bc; });
    for (int s : {1, 2, 3, 5}) make("step " + std::to_string(s), [&](int br, int bc, int l) { return (long long)br * pitch + bc + l * s; });
    for (int s : {2, 5})
        for (int r : {12, 25}) make("adj step " + std::to_string(s) + " R " + std::to_string(r), [&](int br, int bc, int l) { return (long long)(br + rnd(r)) * pitch + bc + l * s + rnd(r); });
    for (int r : {12, 25}) make("compact R " + std::to_string(r), [&](int, int, int) { return (long long)(30 + (int)(rng() % (unsigned)(rows - 60)) + rnd(r)) * pitch + 30 + (int)(rng() % 270u) + rnd(r); });
    make("random", [&](int, int, int) { return (long long)(rng() % (unsigned)(kLdsBytes - 8)); });
    make("rows", [&](int, int bc, int l) { return (long long)(30 + l) * pitch + bc; });
    // the scan's second read of a level is (dy2, dx2) away from the first: same lanes, another random offset -- same statistics as adj / compact

    uint32_t *d_addr = nullptr, *d_out = nullptr;
    unsigned long long *d_cyc = nullptr;
    CHECK(hipMalloc(&d_addr, (size_t)kSets * 16 * 64 * 4));
    CHECK(hipMalloc(&d_out, 64));
    CHECK(hipMalloc(&d_cyc, 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    // cycles per wave-instruction per CU from the SLOPE of the wall time between `iters` and 4 x `iters` (launch, LDS fill and drain
    // cancel), at the nominal clock; wave 0's own cycle counter is not used -- the waves of a CU do not progress evenly
    const double ghz = prop.clockRate / 1e6;
    auto run = [&](auto kern, const Pattern &p, int width, const char *wname) {
        CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
        CHECK(hipMemcpy(d_addr, p.addr.data(), p.addr.size() * 4, hipMemcpyHostToDevice));
        float ms[2] = {0, 0};
        for (int k = 0; k < 2; ++k) {
            const int it = k == 0 ? iters : 4 * iters;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(kern, dim3(cus), dim3(kThreads), kLdsBytes, 0, d_addr, it, d_out, d_cyc);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float t = 0;
                CHECK(hipEventElapsedTime(&t, e0, e1));
                ms[k] = rep == 0 ? t : std::min(ms[k], t);
            }
        }
        const double inst_per_cu = 3.0 * iters * kUnroll * (kThreads / 64);
        const double cyc = (ms[1] - ms[0]) * 1e6 * ghz / inst_per_cu;
        printf("%-4s %-22s %8.3f ms  %6.2f cycles / wave-instruction / CU (%.1f GHz)   model: %5.2f (32 banks) %5.2f (64 banks)\n", wname, p.name.c_str(), ms[1], cyc, ghz,
               model_cycles(p.addr, 32, width), model_cycles(p.addr, 64, width));
    };
    for (const Pattern &p : pats) run(k_lds<1>, p, 1, "u8");
    for (const Pattern &p : pats)
        if (p.name == "linear4" || p.name.rfind("adj step 2", 0) == 0 || p.name.rfind("compact", 0) == 0 || p.name == "random") run(k_lds<2>, p, 2, "u16");
    for (const Pattern &p : pats)
        if (p.name == "linear4" || p.name.rfind("adj step 2", 0) == 0 || p.name.rfind("compact", 0) == 0 || p.name == "random") run(k_lds<4>, p, 4, "b32");
    for (const Pattern &p : pats)
        if (p.name == "linear4" || p.name.rfind("adj step 2", 0) == 0 || p.name.rfind("compact", 0) == 0 || p.name == "random") run(k_lds<8>, p, 8, "b64");
    return 0;
}
