// ta_policy.hip -- does a cache-policy bit make a 64-line byte gather cheaper on gfx950?  ta_gather.hip measured 2.2 cycles
// per distinct 128-byte line and CU for L2-resident data: the L1 fill path (64 B/clk) moving a whole line per byte.  This run
// repeats the 64-line case through raw buffer loads with every combination of the sc0 / nt / sc1 bits, for bytes and dwords,
// on an L2-resident (2 MiB) and an L1-resident (16 KiB) span.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/ta_policy.hip -o scripts/micro/ta_policy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), (short)0, (int)bytes, 0x00020000);
}

template <int WIDTH, int AUX>
__global__ __launch_bounds__(256) void k_gather(const uint8_t *__restrict__ buf, uint32_t span_mask, int iters, uint32_t *out)
{
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(buf, span_mask + 1u);
    uint32_t acc = 0;
    uint32_t h = wave * 2654435761u + 12345u;
    for (int i = 0; i < iters; i += 4) {
        uint32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            h = h * 1664525u + 1013904223u;
            const uint32_t base = (h >> 4) & span_mask & ~127u;
            uint32_t a = (base + lane * 128u + ((lane * 37u) & 127u)) & span_mask;
            if constexpr (WIDTH == 1) v[k] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(rs, (int)a, 0, AUX);
            else v[k] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(a & ~3u), 0, AUX);
        }
        acc += v[0] + v[1] + v[2] + v[3];
    }
    if (acc == 0xdeadbeefu) out[0] = acc;
}

template <int WIDTH, int AUX>
static void run(const uint8_t *buf, uint32_t *out, const char *span, uint32_t mask, int cus, double mhz, int iters, int waves_per_cu)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int blocks = cus * waves_per_cu / 4;
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0));
        k_gather<WIDTH, AUX><<<blocks, 256>>>(buf, mask, iters, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
    }
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("waves/CU %2d  %-10s %s  aux=%2d (%s%s%s)  %8.3f ms  %7.1f cycles per 64-line gather per CU\n", waves_per_cu, span, WIDTH == 1 ? "u8 " : "u32", AUX,
           (AUX & 1) ? "sc0 " : "", (AUX & 2) ? "nt " : "", (AUX & 16) ? "sc1" : "", ms, ms * 1e-3 * mhz * 1e6 / ((double)waves_per_cu * iters));
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
}

template <int WIDTH>
static void all_aux(const uint8_t *buf, uint32_t *out, const char *span, uint32_t mask, int cus, double mhz, int iters, int w)
{
    run<WIDTH, 0>(buf, out, span, mask, cus, mhz, iters, w);
    run<WIDTH, 1>(buf, out, span, mask, cus, mhz, iters, w);
    run<WIDTH, 2>(buf, out, span, mask, cus, mhz, iters, w);
    run<WIDTH, 3>(buf, out, span, mask, cus, mhz, iters, w);
    run<WIDTH, 16>(buf, out, span, mask, cus, mhz, iters, w);
    run<WIDTH, 17>(buf, out, span, mask, cus, mhz, iters, w);
    run<WIDTH, 18>(buf, out, span, mask, cus, mhz, iters, w);
    run<WIDTH, 19>(buf, out, span, mask, cus, mhz, iters, w);
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const double mhz = prop.clockRate / 1000.0;
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, %.0f MHz\n", prop.name, cus, mhz);
    const size_t bytes = 64u << 20;
    uint8_t *buf;
    uint32_t *out;
    CHECK(hipMalloc((void **)&buf, bytes));
    CHECK(hipMalloc((void **)&out, 64));
    std::vector<uint8_t> h(bytes);
    for (size_t i = 0; i < bytes; ++i) h[i] = (uint8_t)(i * 2654435761u >> 13);
    CHECK(hipMemcpy(buf, h.data(), bytes, hipMemcpyHostToDevice));
    for (int w : {8, 16}) {
        all_aux<1>(buf, out, "16KiB(L1)", (16u << 10) - 1, cus, mhz, iters, w);
        all_aux<1>(buf, out, "2MiB(L2)", (2u << 20) - 1, cus, mhz, iters, w);
        all_aux<4>(buf, out, "2MiB(L2)", (2u << 20) - 1, cus, mhz, iters, w);
        all_aux<1>(buf, out, "64MiB", (64u << 20) - 1, cus, mhz, iters / 4, w);
    }
    return 0;
}
