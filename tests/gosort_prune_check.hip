// Host-only check of the pruned Go-order sort (gosort::KeyData::prune, pigo_kernels.hip.inc) against the plain restatement of
// Go's pdqsort (gosort::Data), on random lists with tied Q values of every density.  Built by tests/test_abi_cpu.py with
// `hipcc --cuda-host-only`: no device code is generated or run.  What it replays is k_gosort_ties' flow: keys {Q bits, index}
// in RunCascade's order, the stable (Q, index) order k_sort_by_q leaves in `out`, tie counts from that order, pdqsort with
// pruning, then every position either keeps the stable record or takes the record its key points at.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

#include "../include/pigo_hip.h"
#include "../pigo_amd/csrc/pigo_kernels.hip.inc"

static uint32_t rng_state = 12345u;
static uint32_t rnd() { return rng_state = rng_state * 1664525u + 1013904223u; }

// The wave-parallel sort's part queue on the host: one emulated wave at a time, which takes the waiting parts in RANDOM order --
// any interleaving of the device's waves at part granularity (parts are disjoint ranges of the key array).
struct HostParts {
    struct P { int a, b, limit, fl; };
    std::vector<P> parts;
    size_t max_waiting = 0;
    bool pop(int &a, int &b, int &limit, int &fl)
    {
        if (parts.empty()) return false;
        const size_t i = (size_t)(rnd() >> 8) % parts.size();
        const P p = parts[i];
        parts[i] = parts.back();
        parts.pop_back();
        a = p.a; b = p.b; limit = p.limit; fl = p.fl;
        return true;
    }
    void push(int a, int b, int limit, int fl)
    {
        parts.push_back(P{a, b, limit, fl});
        max_waiting = std::max(max_waiting, parts.size());
    }
    void done() {}
    void note(int, int, int, int, int) {}
    // the device hands short parts over to other waves in the state the loop is in: here, now and then, at random
    bool defer(int a, int b, int limit, int fl)
    {
        if (b - a > 60 || (rnd() & 0x300u) != 0) return false;
        push(a, b, limit, fl);
        return true;
    }
};

int main()
{
    long long lists = 0, pruned_positions = 0, positions = 0;
    for (int trial = 0; trial < 600; ++trial) {
        const int n = 13 + (int)(rnd() % (trial % 10 == 0 ? 4000u : 400u));
        // number of distinct Q values: from "all tied" to "one tied pair at most"
        const uint32_t mode = rnd() % 5u;
        const uint32_t nvals = mode == 0 ? 1u + rnd() % 4u : mode == 1 ? 2u + rnd() % 32u : mode == 2 ? (uint32_t)n / 2u + 1u : (uint32_t)n * 16u;
        std::vector<pigo_det> list((size_t)n);
        for (int i = 0; i < n; ++i) list[(size_t)i] = pigo_det{i, 7 * i + 1, 20 + i % 50, 0.25f * (float)(rnd() % nvals) + 0.5f};
        if (mode == 4 && n > 20) list[(size_t)(rnd() % (uint32_t)n)].q = list[(size_t)(rnd() % (uint32_t)n)].q;  // one tie, somewhere
        // reference: Go's sort on the records themselves
        std::vector<pigo_det> want = list;
        gosort::pdqsort(gosort::Data{want.data()}, 0, n, gosort::bits_len((unsigned long long)n));
        // the device flow
        std::vector<int> order((size_t)n);
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return list[(size_t)a].q < list[(size_t)b].q; });
        std::vector<pigo_det> out((size_t)n);
        for (int i = 0; i < n; ++i) out[(size_t)i] = list[(size_t)order[(size_t)i]];
        std::vector<uint2> keys((size_t)n);
        std::vector<uint16_t> tie_pre((size_t)n);
        uint32_t run = 0;
        for (int i = 0; i < n; ++i) {
            uint32_t bits;
            std::memcpy(&bits, &list[(size_t)i].q, 4);
            keys[(size_t)i] = make_uint2(bits, (uint32_t)i);
            tie_pre[(size_t)i] = (uint16_t)run;
            if (i + 1 < n && out[(size_t)i].q == out[(size_t)i + 1].q) ++run;
        }
        gosort::pdqsort(gosort::KeyData{keys.data(), tie_pre.data()}, 0, n, gosort::bits_len((unsigned long long)n));
        for (int i = 0; i < n; ++i) {
            const uint32_t e = keys[(size_t)i].y;
            if (e & gosort::KeyData::kKeep) ++pruned_positions;
            else out[(size_t)i] = list[(size_t)e];
        }
        positions += n;
        ++lists;
        // the same flow through the wave-parallel sort (k_gosort_ties as it runs on the device), 64 emulated lanes
        {
            std::vector<pigo_det> out2((size_t)n);
            for (int i = 0; i < n; ++i) out2[(size_t)i] = list[(size_t)order[(size_t)i]];
            std::vector<uint2> keys2((size_t)n);
            for (int i = 0; i < n; ++i) {
                uint32_t bits;
                std::memcpy(&bits, &list[(size_t)i].q, 4);
                keys2[(size_t)i] = make_uint2(bits, (uint32_t)i);
            }
            std::vector<uint32_t> fifo(2 * gosort::kWaveFifo);
            HostParts hq;
            hq.push(0, n, gosort::bits_len((unsigned long long)n), 3);
            const gosort::WaveKeys<unsigned long long *, const uint16_t *, uint32_t *> xw{reinterpret_cast<unsigned long long *>(keys2.data()), tie_pre.data(), fifo.data(), gosort::Wave{0}};
            gosort::pdqsort_wave(xw, hq);
            for (int i = 0; i < n; ++i) {
                const uint32_t e = keys2[(size_t)i].y;
                if (!(e & gosort::KeyData::kKeep)) out2[(size_t)i] = list[(size_t)e];
            }
            for (int i = 0; i < n; ++i) {
                const pigo_det &a = out2[(size_t)i], &b = want[(size_t)i];
                if (a.row != b.row || a.col != b.col || a.scale != b.scale || std::memcmp(&a.q, &b.q, 4) != 0) {
                    std::printf("WAVE MISMATCH trial %d n %d mode %u position %d: got (%d,%d,%d,%g) want (%d,%d,%d,%g)\n", trial, n, mode, i, a.row,
                                a.col, a.scale, (double)a.q, b.row, b.col, b.scale, (double)b.q);
                    return 1;
                }
            }
        }
        for (int i = 0; i < n; ++i) {
            const pigo_det &a = out[(size_t)i], &b = want[(size_t)i];
            if (a.row != b.row || a.col != b.col || a.scale != b.scale || std::memcmp(&a.q, &b.q, 4) != 0) {
                std::printf("MISMATCH trial %d n %d mode %u position %d: got (%d,%d,%d,%g) want (%d,%d,%d,%g)\n", trial, n, mode, i, a.row, a.col,
                            a.scale, (double)a.q, b.row, b.col, b.scale, (double)b.q);
                return 1;
            }
        }
    }
    std::printf("ok lists=%lld positions=%lld kept_from_stable=%lld\n", lists, positions, pruned_positions);
    return pruned_positions > 0 ? 0 : 2;
}
