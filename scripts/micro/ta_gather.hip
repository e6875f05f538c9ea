// ta_gather.hip -- microbenchmark of the vector-memory (texture-address) path on gfx950 for the access patterns of the
// cascade scan: what does ONE wave-wide byte gather cost as a function of distinct cache lines, active lanes and width?
// Also the calibration run for FETCH_SIZE on byte gathers (rocprofv3 --pmc FETCH_SIZE -- ./ta_gather): every case prints the
// number of distinct 128-byte lines it touches per launch.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/ta_gather.hip -o scripts/micro/ta_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// mode: 0 = 64 distinct lines per gather, 1 = all lanes one line (different bytes), 2 = quads share a line (16 lines),
//       3 = 64 distinct lines, odd lanes masked off, 4 = 64 distinct lines, only lanes < 16 active, 5 = all lanes the SAME byte
//       6 = half-waves share a line (2 lines), 7 = 8 lanes share a line (8 lines)
template <int WIDTH>
__global__ __launch_bounds__(256) void k_gather(const uint8_t *__restrict__ buf, uint32_t span_mask, int iters, int mode, uint32_t *out)
{
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint32_t acc = 0;
    uint32_t h = wave * 2654435761u + 12345u;
    const bool active = mode == 3 ? (lane & 1u) == 0 : mode == 4 ? lane < 16u : true;
    for (int i = 0; i < iters; i += 4) {
        uint32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            h = h * 1664525u + 1013904223u;                 // wave-uniform pseudo-random base
            const uint32_t base = (h >> 4) & span_mask & ~127u;
            uint32_t a;
            switch (mode) {
            case 1: a = base + lane; break;                                   // one line
            case 2: a = base + (lane >> 2) * 128u + (lane & 3u) * 17u; break;  // quads share a line
            case 5: a = base; break;
            case 6: a = base + (lane >> 5) * 128u + (lane & 31u); break;
            case 7: a = base + (lane >> 3) * 128u + (lane & 7u) * 9u; break;
            default: a = base + lane * 128u + ((lane * 37u) & 127u); break;    // every lane its own line
            }
            a &= span_mask;
            if (active) {
                if constexpr (WIDTH == 1) v[k] = buf[a];
                else v[k] = *reinterpret_cast<const uint32_t *>(buf + (a & ~3u));
            } else {
                v[k] = 0;
            }
        }
        acc += v[0] + v[1] + v[2] + v[3];
    }
    if (acc == 0xdeadbeefu) out[0] = acc;
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const double mhz = prop.clockRate / 1000.0;
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, %.0f MHz\n", prop.name, cus, mhz);
    const size_t bytes = 512u << 20;
    uint8_t *buf;
    uint32_t *out;
    CHECK(hipMalloc((void **)&buf, bytes));
    CHECK(hipMalloc((void **)&out, 64));
    std::vector<uint8_t> h(bytes);
    for (size_t i = 0; i < bytes; ++i) h[i] = (uint8_t)(i * 2654435761u >> 13);
    CHECK(hipMemcpy(buf, h.data(), bytes, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    struct Span { const char *name; uint32_t mask; } spans[] = {{"16KiB(L1)", (16u << 10) - 1}, {"2MiB(L2)", (2u << 20) - 1}, {"64MiB(MALL)", (64u << 20) - 1}, {"512MiB(HBM)", (512u << 20) - 1}};
    const char *mnames[] = {"64 lines", "1 line", "16 lines (quads)", "64 lines, 32 lanes", "64 lines, 16 lanes", "same byte", "2 lines", "8 lines"};
    for (int waves_per_cu : {4, 8, 16}) {
        for (const Span &sp : spans) {
            for (int width : {1, 4}) {
                for (int mode = 0; mode < 8; ++mode) {
                    if (sp.mask > (2u << 20) && !(mode == 0 || mode == 2)) continue;
                    if (waves_per_cu != 8 && !(mode == 0 || mode == 2 || mode == 4)) continue;
                    const int blocks = cus * waves_per_cu / 4;
                    for (int rep = 0; rep < 2; ++rep) {
                        CHECK(hipEventRecord(e0));
                        if (width == 1) k_gather<1><<<blocks, 256>>>(buf, sp.mask, iters, mode, out);
                        else k_gather<4><<<blocks, 256>>>(buf, sp.mask, iters, mode, out);
                        CHECK(hipEventRecord(e1));
                        CHECK(hipEventSynchronize(e1));
                    }
                    float ms;
                    CHECK(hipEventElapsedTime(&ms, e0, e1));
                    const double per_cu = (double)waves_per_cu * iters;
                    printf("waves/CU %2d  %-12s  %s  %-20s  %8.3f ms  %7.1f cycles per gather per CU\n", waves_per_cu, sp.name, width == 1 ? "u8 " : "u32", mnames[mode], ms,
                           ms * 1e-3 * mhz * 1e6 / per_cu);
                }
            }
        }
    }
    return 0;
}
