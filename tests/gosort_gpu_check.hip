// GPU check of k_sort_by_q + k_gosort_ties (the wave-parallel Go-order sort, pigo_kernels.hip.inc) against the host restatement of
// Go's pdqsort (gosort::Data) on random lists with ties of every density, sorted / nearly sorted / descending inputs and lengths
// on both sides of the LDS limit.  Built and run by tests/test_gpu_parity.py::test_go_order_sort_program; also the debugging
// tool: argv[1] = threads per workgroup (64 = one wave, no sharing of parts), argv[2] = trials.  core/pigo.go:264-266.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <unistd.h>

#include "../include/pigo_hip.h"
#include "../pigo_amd/csrc/pigo_kernels.hip.inc"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 3; } } while (0)

// LIFO part stack of the host replay (the order one device wave works in) with the trace PIGO_GOSORT_DBG & 131072 records
struct HostParts {
    struct P { int a, b, limit, fl; };
    std::vector<P> parts;
    std::vector<int> trace;
    bool pop(int &a, int &b, int &limit, int &fl)
    {
        if (parts.empty()) return false;
        const P p = parts.back();
        parts.pop_back();
        a = p.a; b = p.b; limit = p.limit; fl = p.fl;
        return true;
    }
    void push(int a, int b, int limit, int fl) { parts.push_back(P{a, b, limit, fl}); }
    void done() {}
    bool defer(int, int, int, int) { return false; }
    void note(int code, int v0, int v1, int v2, int v3) { trace.insert(trace.end(), {code, v0, v1, v2, v3}); }
};

static uint32_t rng_state = 2463534242u;
static uint32_t rnd() { return rng_state = rng_state * 1664525u + 1013904223u; }

int main(int argc, char **argv)
{
    const int threads = argc > 1 ? std::atoi(argv[1]) : gosort::kSortThreads;
    const int trials = argc > 2 ? std::atoi(argv[2]) : 40;
    const int maxn = argc > 3 ? std::atoi(argv[3]) : 1 << 30;  // (debugging: keep every list below a length)
    const int cap = 20000, nfr = 8;
    pigo_det *d_in, *d_out;
    int32_t *d_counts, *d_ties;
    uint8_t *d_ws;
    CHECK(hipMalloc((void **)&d_in, sizeof(pigo_det) * (size_t)cap * nfr));
    CHECK(hipMalloc((void **)&d_out, sizeof(pigo_det) * (size_t)cap * nfr));
    CHECK(hipMalloc((void **)&d_counts, 4 * nfr));
    CHECK(hipMalloc((void **)&d_ties, 4 * nfr));
#if PIGO_GOSORT_DBG & 131072
    CHECK(hipHostMalloc((void **)&d_ws, (size_t)cap * nfr * 12, hipHostMallocMapped));  // the trace is readable while the kernel runs
#else
    CHECK(hipMalloc((void **)&d_ws, (size_t)cap * nfr * 12));
#endif
    const int lds_keys = std::min(cap, kGoSortKeys);
    const size_t lds_fixed = sizeof(gosort::PartList) + (size_t)(gosort::kSortThreads / 64) * 2 * gosort::kWaveFifo * 4;
    CHECK(hipFuncSetAttribute((const void *)k_gosort_ties, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds_fixed + (size_t)kGoSortKeys * 10)));
    std::vector<pigo_det> host((size_t)cap * nfr), got((size_t)cap * nfr);
    double gpu_ms = 0.0, cpu_ms = 0.0;
    long long total = 0;
    int bad = 0;
    for (int trial = 0; trial < trials; ++trial) {
        int32_t counts[nfr];
        std::vector<std::vector<pigo_det>> want((size_t)nfr);
        for (int f = 0; f < nfr; ++f) {
            const uint32_t size_mode = rnd() % 6u;
            int n = size_mode == 0 ? 13 + (int)(rnd() % 50u) : size_mode == 1 ? 100 + (int)(rnd() % 900u) : size_mode == 2 ? 2000 + (int)(rnd() % 6000u)
                        : size_mode == 3 ? 12000 + (int)(rnd() % 2336u) : size_mode == 4 ? 14337 + (int)(rnd() % 5000u) : (int)(rnd() % 16u);
            n = std::min(n, maxn);
            const uint32_t mode = rnd() % 6u;
            const uint32_t nvals = mode == 0 ? 1u + rnd() % 4u : mode == 1 ? 2u + rnd() % 32u : mode == 2 ? (uint32_t)n / 2u + 1u : (uint32_t)n * 16u + 1u;
            std::vector<pigo_det> list((size_t)n);
            for (int i = 0; i < n; ++i) list[(size_t)i] = pigo_det{i, 7 * i + 1, 20 + i % 50, 0.25f * (float)(rnd() % nvals) + 0.5f};
            if (mode == 4 && n > 20) list[(size_t)(rnd() % (uint32_t)n)].q = list[(size_t)(rnd() % (uint32_t)n)].q;  // one tie, somewhere
            const uint32_t order = rnd() % 5u;  // 0, 1: as drawn; 2: ascending; 3: ascending with a few exchanges; 4: descending
            if (order >= 2) {
                std::vector<float> qs((size_t)n);
                for (int i = 0; i < n; ++i) qs[(size_t)i] = list[(size_t)i].q;
                std::sort(qs.begin(), qs.end());
                if (order == 4) std::reverse(qs.begin(), qs.end());
                if (order == 3 && n > 4)
                    for (int k = 0; k < 3; ++k) std::swap(qs[(size_t)(rnd() % (uint32_t)n)], qs[(size_t)(rnd() % (uint32_t)n)]);
                for (int i = 0; i < n; ++i) list[(size_t)i].q = qs[(size_t)i];
            }
            counts[f] = n;
            std::copy(list.begin(), list.end(), host.begin() + (size_t)f * cap);
            want[(size_t)f] = list;
            const auto t0 = std::chrono::steady_clock::now();
            if (n > 1) gosort::pdqsort(gosort::Data{want[(size_t)f].data()}, 0, n, gosort::bits_len((unsigned long long)n));
            cpu_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            total += n;
        }
        CHECK(hipMemcpy(d_in, host.data(), sizeof(pigo_det) * host.size(), hipMemcpyHostToDevice));
        CHECK(hipMemcpy(d_counts, counts, sizeof(counts), hipMemcpyHostToDevice));
        CHECK(hipMemset(d_ties, 0, 4 * nfr));
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        dim3 grid((unsigned)((cap + kThreads - 1) / kThreads), (unsigned)nfr);
        k_sort_by_q<<<grid, kThreads>>>(d_in, d_counts, cap, d_out, d_ties);
        CHECK(hipEventRecord(e0));
        k_gosort_ties<<<nfr, threads, lds_fixed + (size_t)lds_keys * 10>>>(d_in, d_counts, cap, d_ties, d_out, lds_keys, d_ws);
        CHECK(hipEventRecord(e1));
#if PIGO_GOSORT_DBG & 131072
        {   // a kernel that does not end within 3 s: print where every frame's trace stands and leave
            const auto t0 = std::chrono::steady_clock::now();
            while (hipEventQuery(e1) == hipErrorNotReady) {
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 3.0) {
                    for (int f = 0; f < nfr; ++f) {
                        const volatile int *t = reinterpret_cast<const volatile int *>(d_ws + (size_t)f * cap * 12);
                        const int dn = t[0];
                        std::printf("STUCK frame %d n %d: %d records\n", f, counts[f], dn);
                        for (int j = std::max(0, dn - 8); j < dn && j < 4000; ++j)
                            std::printf("   %d: %d (%d %d %d %d)\n", j, t[1 + 5 * j], t[2 + 5 * j], t[3 + 5 * j], t[4 + 5 * j], t[5 + 5 * j]);
                    }
                    std::fflush(stdout);
                    _exit(5);
                }
            }
        }
#endif
        CHECK(hipEventSynchronize(e1));
        CHECK(hipGetLastError());
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        gpu_ms += ms;
        CHECK(hipMemcpy(got.data(), d_out, sizeof(pigo_det) * got.size(), hipMemcpyDeviceToHost));
#if PIGO_GOSORT_DBG & 131072
        {   // replay every LDS-sized list on the host (64 emulated lanes, LIFO parts) and compare the traces
            std::vector<int> dtrace((size_t)cap * 3);
            for (int f = 0; f < nfr; ++f) {
                const int n = counts[f];
                if (n <= 12 || n > lds_keys) continue;
                const pigo_det *list = host.data() + (size_t)f * cap;
                std::vector<int> order((size_t)n);
                for (int i = 0; i < n; ++i) order[(size_t)i] = i;
                std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return list[a].q < list[b].q; });
                std::vector<uint2> keys((size_t)n);
                std::vector<uint16_t> tie_pre((size_t)n);
                uint32_t run = 0;
                for (int i = 0; i < n; ++i) {
                    uint32_t bits;
                    std::memcpy(&bits, &list[i].q, 4);
                    keys[(size_t)i] = make_uint2(bits, (uint32_t)i);
                    tie_pre[(size_t)i] = (uint16_t)run;
                    if (i + 1 < n && list[order[(size_t)i]].q == list[order[(size_t)i + 1]].q) ++run;
                }
                if (run == 0) continue;
                std::vector<uint32_t> fifo(2 * gosort::kWaveFifo);
                HostParts hq;
                hq.push(0, n, gosort::bits_len((unsigned long long)n), 3);
                const gosort::WaveKeys<unsigned long long *, const uint16_t *, uint32_t *> xw{reinterpret_cast<unsigned long long *>(keys.data()), tie_pre.data(), fifo.data(), gosort::Wave{0}};
                gosort::pdqsort_wave(xw, hq);
                CHECK(hipMemcpy(dtrace.data(), d_ws + (size_t)f * cap * 12, (size_t)cap * 12, hipMemcpyDeviceToHost));
                const int dn = dtrace[0], hn = (int)hq.trace.size() / 5;
                int k = 0;
                for (; k < std::min(dn, hn); ++k)
                    if (std::memcmp(&dtrace[1 + 5 * (size_t)k], &hq.trace[5 * (size_t)k], 20) != 0) break;
                if (k < std::min(dn, hn) || dn != hn) {
                    std::printf("TRACE trial %d frame %d n %d: device %d records, host %d; first difference at %d\n", trial, f, n, dn, hn, k);
                    for (int j = std::max(0, k - 3); j < std::min(std::max(dn, hn), k + 3); ++j) {
                        if (j < dn) std::printf("   dev  %d: %d (%d %d %d %d)\n", j, dtrace[1 + 5 * (size_t)j], dtrace[2 + 5 * (size_t)j], dtrace[3 + 5 * (size_t)j], dtrace[4 + 5 * (size_t)j], dtrace[5 + 5 * (size_t)j]);
                        if (j < hn) std::printf("   host %d: %d (%d %d %d %d)\n", j, hq.trace[5 * (size_t)j], hq.trace[5 * (size_t)j + 1], hq.trace[5 * (size_t)j + 2], hq.trace[5 * (size_t)j + 3], hq.trace[5 * (size_t)j + 4]);
                    }
                }
            }
        }
#endif
        for (int f = 0; f < nfr; ++f)
            for (int i = 0; i < counts[f]; ++i) {
                const pigo_det &a = got[(size_t)f * cap + (size_t)i], &b = want[(size_t)f][(size_t)i];
                if (a.row != b.row || a.col != b.col || a.scale != b.scale || std::memcmp(&a.q, &b.q, 4) != 0) {
                    std::printf("MISMATCH threads %d trial %d frame %d n %d position %d: got (%d,%d,%d,%g) want (%d,%d,%d,%g)\n", threads, trial, f, counts[f], i,
                                a.row, a.col, a.scale, (double)a.q, b.row, b.col, b.scale, (double)b.q);
                    ++bad;
                    break;
                }
            }
    }
    if (bad) {
        std::printf("FAILED threads=%d: %d of %d lists differ\n", threads, bad, trials * nfr);
        return 1;
    }
    std::printf("ok threads=%d trials=%d lists=%d elements=%lld  k_gosort_ties %.3f ms total (host serial sort of the same lists: %.3f ms)\n", threads, trials,
                trials * nfr, total, gpu_ms, cpu_ms);
    return 0;
}
