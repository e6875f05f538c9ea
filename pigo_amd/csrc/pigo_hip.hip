// pigo_hip.hip -- host side of libpigo_hip.so: the C ABI declared in include/pigo_hip.h.
//
// Built for gfx950 only (see pigo_amd/build.py):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -shared -fPIC pigo_hip.hip -o libpigo_hip.so
//
// Host responsibilities (everything else is in pigo_kernels.hip.inc):
//   - Unpack (core/pigo.go:51-110): parse the cascade file, keep host copies, upload the tables
//   - RunCascade's scale ladder (core/pigo.go:226-231,255) evaluated in float64 exactly as the Go code
//     does, turned into the flattened (scale,row,col) index space, the tile list and the per-scale
//     offset tables (built on the GPU by k_build_tab)
//   - Go's sort.Slice (pdqsort) restated for ClusterDetections' in-place sort (core/pigo.go:264-266)
//   - plan / workspace management, launches, the status flags and the overflow fallback
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <list>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/pigo_hip.h"
#include "pigo_kernels.hip.inc"

namespace {

thread_local std::string g_last_error;

pigo_status fail(pigo_status st, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return st;
}

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(PIGO_ERR_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

int env_int(const char *name, int dflt);
const char *tune_env(const char *name);

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    hipError_t alloc(size_t count)
    {
        release();
        if (count == 0) count = 1;
        hipError_t e = hipMalloc((void **)&p, count * sizeof(T));
        if (e == hipSuccess) n = count;
        return e;
    }
    size_t bytes() const { return n * sizeof(T); }
};

uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

// pigo.go:156-157
const int kQCos[33] = {256, 251, 236, 212, 181, 142, 97, 49, 0, -49, -97, -142, -181, -212, -236, -251, -256,
                       -251, -236, -212, -181, -142, -97, -49, 0, 49, 97, 142, 181, 212, 236, 251, 256};
const int kQSin[33] = {0, 49, 97, 142, 181, 212, 236, 251, 256, 251, 236, 212, 181, 142, 97, 49, 0,
                       -49, -97, -142, -181, -212, -236, -251, -256, -251, -236, -212, -181, -142, -97, -49, 0};

}  // namespace

// ---- handles ----------------------------------------------------------------------------------------------

struct pigo_plan;

struct pigo_cascade {
    int device = 0;
    uint32_t depth = 0, ntrees = 0;
    int nodes = 0;                       // 2^depth (what RunCascade calls treeDepth, pigo.go:216)
    std::vector<int8_t> codes;           // treeCodes     [ntrees][4*nodes]
    std::vector<float> pred;             // treePred      [ntrees][nodes]
    std::vector<float> thr;              // treeThreshold [ntrees]
    DevBuf<int8_t> d_codes;
    DevBuf<float> d_leaf, d_thr;
    DevBuf<int16_t> d_pass_end;          // k_tail_deep: the lane=tree pass that starts at tree t covers trees [t, d_pass_end[t])
    DevBuf<uint2> d_codes_t;             // depth-6 cascades: child-pair table, node-major [32][ntrees] (pigo_cascade_create)
    std::mutex mu;                       // guards the two slot lists below -- never held across a launch or a synchronisation
    // RunCascade slots: everything one call needs (plan, device + pinned host buffers, a stream, the captured graph of
    // "upload, scan, download").  A call takes a free slot with its parameters (or makes one), so goroutines calling RunCascade on
    // one *Pigo concurrently -- the reference is re-entrant, examples/web/main.go:71,141 -- run next to each other on the GPU.
    struct RunSlot;
    std::list<std::unique_ptr<RunSlot>> slots;   // most recently used first
    // ClusterDetections slots: the reference's ClusterDetections is a pure function that goroutines call concurrently on one
    // *Pigo (examples/web/main.go:141-144), so a call owns its scratch -- pinned staging, a stream, the long lists' device
    // buffers -- for its duration and N callers run next to each other (and next to RunCascade's slot lookup)
    struct ClusterSlot {
        bool busy = false;
        pigo_det *h_cl = nullptr;        // pinned host staging of the short lists: [in | out | counts]
        hipStream_t stream = nullptr;
        DevBuf<pigo_det> d_sorted, d_clusters, d_cl_tmp;
        DevBuf<int32_t> d_small;         // [0] n, [1] clusters, [2] seeds
        DevBuf<int32_t> d_cl_seeds, d_cl_tmpn;   // long lists: k_cluster_seeds / _members / _compact
        DevBuf<float> d_mq;
        ~ClusterSlot()
        {
            if (stream) (void)hipStreamDestroy(stream);
            if (h_cl) (void)hipHostFree(h_cl);
        }
    };
    std::list<std::unique_ptr<ClusterSlot>> cl_slots;
};

struct PlanKey {
    int rows, cols, dim, min_size, max_size;
    double shift, scale, angle;
    bool operator==(const PlanKey &o) const
    {
        return rows == o.rows && cols == o.cols && dim == o.dim && min_size == o.min_size && max_size == o.max_size &&
               shift == o.shift && scale == o.scale && angle == o.angle;
    }
};

struct pigo_cascade::RunSlot {
    PlanKey key{};
    std::unique_ptr<pigo_plan> plan;
    bool busy = false;
    hipStream_t stream = nullptr;
    DevBuf<uint8_t> d_frame;
    DevBuf<pigo_det> d_dets;
    DevBuf<int32_t> d_count;
    uint8_t *h_frame = nullptr;          // pinned staging: the caller's pixels go through it (cgo memory is pageable)
    pigo_det *h_dets = nullptr;
    int32_t *h_small = nullptr;          // [0] detection count, [1..4] the plan's status flags
    size_t fbytes = 0;
    hipGraphExec_t exec = nullptr;       // H2D + pigo_plan_run + D2H of count, flags and detections
    ~RunSlot();
};

struct pigo_plan {
    pigo_cascade *c = nullptr;
    PlanKey key{};
    int max_frames = 0, det_cap = 0;
    bool rot = false, guard = false;
    int angle_idx = 0;
    int variant = 0;
    std::vector<ScaleDesc> scales;
    std::vector<uint32_t> tiles;
    int n_ladder = 0;
    long long windows = 0;
    ScanArgs args{};                     // template for launches (frame pointers filled per run)
    DevBuf<ScaleDesc> d_scales;
    DevBuf<uint32_t> d_tiles;
    DevBuf<int2> d_tab;
    // variant 2 (k_scan_tile): tile classes = (geometry, LDS footprint bucket), one launch per class
    struct TileClass {
        int tw_log2, th, nwin;
        int qb_div;                      // LDS queue B holds nwin / qb_div entries
        bool lds;
        uint32_t tile0, ntiles;
        size_t dyn_lds;
        uint32_t v3_skip;                // variant 3: leading tiles of this class that belong to rungs the region kernel scans
    };
    std::vector<TileClass> classes;
    std::vector<uint2> tiles2;
    DevBuf<uint2> d_tiles2;
    // variant 3 (k_scan_region): scale groups, each scanned by one launch of 1024-thread workgroups that own an LDS-resident
    // region of a frame; rungs beyond the last group keep k_scan_tile's global-gather class
    struct RegionGroup {
        RegionArgs args;
        size_t dyn_lds;
    };
    std::vector<RegionGroup> regions;
    bool region_ok = false;
    // plans of a few frames: the whole scan as ONE persistent launch (k_scan_one) -- items, global queues, in-launch consumers
    bool one_ok = false;
    OneArgs one{};                       // template (item counts are filled per run: they depend on nframes)
    BigArgs one_big{};                   // the big rungs' chunk stages as k_scan_one runs them (hand-over at the pooling tree)
    size_t one_lds = 0;
    int one_grid = 256;                  // workgroups of the launch: one per CU
    DevBuf<uint4> d_oneq;                // [8][one.qcap]
    DevBuf<uint32_t> d_onecnt;           // OneArgs::cnt
    mutable pigo_det *one_dets = nullptr; // (per run) the caller's detection buffer, for the launch's own order restore
    int32_t *one_host_flags = nullptr;   // (per run, pigo_run_cascade's slots) OneArgs::host_flags
    // variant 3, the rungs beyond the region groups: k_scan_big (persistent, one small workgroup per CU NEXT to a region
    // workgroup) + a chain of k_tail_deep launches whose code windows fit the LDS the region groups leave free
    bool big_ok = false;
    BigArgs big{};
    DevBuf<uint2> d_big_items;
    DevBuf<uint4> d_big_midq;            // k_scan_big -> k_big_pool: [8][big_midcap]
    uint32_t big_midcap = 0;
    std::vector<uint2> big_items;
    size_t big_lds = 0;
    size_t side_lds = 0;                 // LDS a region workgroup of group 0 leaves to a co-resident side workgroup (0: none reserved)
    bool big_ct = false;                 // the side chain's k_tail_deep reads its codes from global memory (no LDS table)
    bool big_side_first = false;         // PIGO_BIG_FIRST=1: the side chain is launched before the region groups (default: after the first)
    int big_skip = 0;                    // PIGO_BIG_SKIP, timing experiments only (results incomplete): 1 = no tail, 2 = no k_scan_big
    std::vector<int> side_splits;        // code windows of the side chain's k_tail_deep launches: [splits[i], splits[i+1])
    bool tile_patch = true;              // variant 3: do the tile classes' survivors include scales <= kPatchMaxS?
    DevBuf<uint32_t> d_tabr;
    DevBuf<uint32_t> d_tabp;
    bool tile_ok = false;
    int tab_lds = 0, tab_glb = 0;        // LDS table capacity (trees) of the LDS-pixel / global-pixel classes
    bool rot_lds = false;                // rotated scan out of LDS tiles (landscape frames: the column clamp nrows-1 stays inside a row)
    size_t deep_lds = 0, deep_lds2 = 0;  // dynamic LDS of the two k_tail_deep launches
    int deep_mid = 0;                    // first launch walks [deep_lo, deep_mid), second [deep_mid, ntrees)
    int nh_reg_mid = 0;                  // variant 3: the mid scale group's hand-over tree (the small group's is nh_lds)
    DevBuf<QEntry> d_queue2;
    long long qcap2 = 0;
    DevBuf<QEntry> d_queue;
    DevBuf<uint32_t> d_qcount;
    DevBuf<int32_t> d_ties;              // per-frame tie counts when the caller does not ask for them
    DevBuf<uint8_t> d_gosort_ws;         // k_gosort_ties: keys + tie counts of lists longer than kGoSortKeys
    // ClusterDetections for long lists (k_cluster_seeds / _members / _compact): seed lists, per-seed clusters before compaction
    DevBuf<int32_t> d_cl_seeds, d_cl_nseeds, d_cl_tmpn;
    DevBuf<pigo_det> d_cl_tmp;
    int cluster_mode = -1;               // -1: by det_cap, 0: k_cluster (one workgroup per frame), 1: the seeds/members kernels
    DevBuf<RawDet> d_raw;
    DevBuf<int32_t> d_flags;
    DevBuf<float> d_mq;
    DevBuf<unsigned long long> d_stats;
    // pigo_run_batch_sharded: this rank's lists and the wire rows it contributes to the all-gather
    DevBuf<pigo_det> sh_dets, sh_sorted, sh_clusters;
    DevBuf<int32_t> sh_counts, sh_wire;
    long long qcap = 0;
    // profiling
    bool profiling = false;
    std::vector<hipEvent_t> events;
    std::vector<const char *> ev_names;
    int n_timed = 0;
    bool gosort_attr = false;
    int last_nframes = 0;
    int32_t last_flags[3] = {0, 0, 0};   // what the most recent pigo_plan_status read: queue overflow, panic, det_cap overflow
    std::mutex mu;
    // the global-gather tile classes (vector-memory bound) run on a side stream next to the LDS-tile classes (LDS bound)
    hipStream_t side = nullptr;
    int side_mode = 1;
    int fork_min_frames = 8;             // batches of at least this many frames run their tile classes on separate streams
    bool small_ct = false;               // plans of a few frames: k_tail_deep as one table-free launch (launch_tail)
    int big_chunk = 128;                 // frames per pass of the side chain over the batch (0: the whole batch at once)
    bool no_fork = false;                // (while pigo_plan_run captures a graph) every launch stays on the one stream
    bool split_tail = false;             // plans of a few frames: the global-gather class and the LDS classes each with a queue set and a tail
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipStream_t grp_stream = nullptr;    // variant 3, small batches: the second region group runs next to the first
    hipEvent_t ev_gfork = nullptr, ev_gjoin = nullptr;
    // small batches: the launch sequence of pigo_plan_run is captured once per (buffers, batch) and replayed as a hipGraph
    struct GraphCache {
        const void *frames = nullptr;
        size_t stride = 0;
        int nframes = 0, variant = -1;
        void *dets = nullptr, *counts = nullptr;
        hipGraphExec_t exec = nullptr;
    } gc;
    hipStream_t cap_stream = nullptr;
    int graph_max_frames = 0;            // batches up to this size are replayed from a graph (0 = never)
    // chunked pipeline: the deep tail of chunk c runs on `tail_stream` next to the tile kernels of chunk c+1 (two queue sets)
    int pipe_chunks = 0;                 // 0 = automatic
    hipStream_t tail_stream = nullptr;
    hipEvent_t ev_tiles[2] = {nullptr, nullptr}, ev_tail[2] = {nullptr, nullptr};
    ~pigo_plan()
    {
        for (int i = 0; i < 2; ++i) {
            if (ev_tiles[i]) (void)hipEventDestroy(ev_tiles[i]);
            if (ev_tail[i]) (void)hipEventDestroy(ev_tail[i]);
        }
        if (tail_stream) (void)hipStreamDestroy(tail_stream);
        if (gc.exec) (void)hipGraphExecDestroy(gc.exec);
        if (cap_stream) (void)hipStreamDestroy(cap_stream);
        if (grp_stream) (void)hipStreamDestroy(grp_stream);
        if (ev_gfork) (void)hipEventDestroy(ev_gfork);
        if (ev_gjoin) (void)hipEventDestroy(ev_gjoin);
        for (hipEvent_t e : events) (void)hipEventDestroy(e);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        if (side) (void)hipStreamDestroy(side);
    }
    size_t workspace_bytes() const
    {
        return d_scales.bytes() + d_tiles.bytes() + d_tiles2.bytes() + d_tabp.bytes() + d_tabr.bytes() + d_tab.bytes() + d_big_items.bytes() + d_big_midq.bytes() + d_oneq.bytes() + d_onecnt.bytes() + d_queue.bytes() + d_queue2.bytes() +
               d_qcount.bytes() + d_raw.bytes() + d_flags.bytes() + d_mq.bytes() + d_ties.bytes() + d_cl_seeds.bytes() + d_cl_nseeds.bytes() + d_cl_tmpn.bytes() + d_cl_tmp.bytes() +
               d_gosort_ws.bytes();
    }
};

pigo_cascade::RunSlot::~RunSlot()
{
    if (exec) (void)hipGraphExecDestroy(exec);
    plan.reset();
    if (stream) (void)hipStreamDestroy(stream);
    if (h_frame) (void)hipHostFree(h_frame);
    if (h_dets) (void)hipHostFree(h_dets);
    if (h_small) (void)hipHostFree(h_small);
}

extern "C" const char *pigo_last_error(void) { return g_last_error.c_str(); }

extern "C" int pigo_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---- Unpack -------------------------------------------------------------------------------------------------

extern "C" pigo_status pigo_cascade_create(const uint8_t *packet, size_t len, int device, pigo_cascade **out)
{
    if (!out) return fail(PIGO_ERR_PARAM, "out is NULL");
    *out = nullptr;
    if (!packet) return fail(PIGO_ERR_PARAM, "packet is NULL");
    // pigo.go:61-75: skip 8 bytes, u32 depth, u32 tree count
    if (len < 16) return fail(PIGO_ERR_PACKET, "Unpack: packet of %zu bytes is shorter than the 16-byte header", len);
    const uint32_t depth = le32(packet + 8), ntrees = le32(packet + 12);
    if (depth > 12) return fail(PIGO_ERR_PARAM, "Unpack: tree depth %u not supported (max 12)", depth);
    if (ntrees > 32767) return fail(PIGO_ERR_PARAM, "Unpack: %u trees not supported (max 32767: tree indices travel as int16)", ntrees);
    const size_t nodes = (size_t)1 << depth;
    const size_t ncode = 4 * nodes - 4;  // pigo.go:81
    const size_t rec = ncode + 4 * nodes + 4;
    if ((len - 16) / rec < ntrees) return fail(PIGO_ERR_PACKET, "Unpack: packet too short for %u trees of depth %u", ntrees, depth);

    std::unique_ptr<pigo_cascade> c(new (std::nothrow) pigo_cascade);
    if (!c) return fail(PIGO_ERR_NOMEM, "out of memory");
    c->device = device;
    c->depth = depth;
    c->ntrees = ntrees;
    c->nodes = (int)nodes;
    c->codes.assign((size_t)ntrees * 4 * nodes, 0);
    c->pred.resize((size_t)ntrees * nodes);
    c->thr.resize(ntrees);
    size_t pos = 16;
    for (uint32_t t = 0; t < ntrees; ++t) {
        // pigo.go:79-86: four zero bytes, then the 4*2^d-4 code bytes reinterpreted as int8
        memcpy(c->codes.data() + (size_t)t * 4 * nodes + 4, packet + pos, ncode);
        pos += ncode;
        for (size_t i = 0; i < nodes; ++i) {  // pigo.go:89-95
            const uint32_t u = le32(packet + pos);
            memcpy(&c->pred[(size_t)t * nodes + i], &u, 4);
            pos += 4;
        }
        const uint32_t u = le32(packet + pos);  // pigo.go:96-100
        memcpy(&c->thr[t], &u, 4);
        pos += 4;
    }
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(PIGO_ERR_HIP, "device %d not available (%d visible)", device, ndev);
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(c->d_codes.alloc(c->codes.size()));
    HIP_TRY(c->d_leaf.alloc(c->pred.size()));
    HIP_TRY(c->d_thr.alloc(c->thr.size()));
    if (ntrees) {
        HIP_TRY(hipMemcpy(c->d_codes.p, c->codes.data(), c->codes.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->d_leaf.p, c->pred.data(), c->pred.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->d_thr.p, c->thr.data(), c->thr.size() * 4, hipMemcpyHostToDevice));
        // A window can only die at a tree whose threshold is above the cascade's floor value (facefinder: 24 of 468 trees,
        // 20..40 apart in the deep part), so a lane=tree pass that ends right after the next such tree evaluates no tree a dying
        // window would not have reached -- a 64-tree pass reads up to 44 trees' pixels (random 64-byte lines of a large
        // window: the deep tail is bound by exactly that fabric traffic) for nothing.  At least 16 trees per pass.
        const int nt = (int)ntrees;
        float lo = c->thr[0];
        for (int i = 0; i < nt; ++i) lo = std::min(lo, c->thr[i]);
        // Second table: plain 64-tree passes, for plans of a few frames -- there the tail is a handful of entries per wave and
        // the call's latency is the number of dependent passes, not the traffic (one 1080p frame: 0.203 ms vs 0.223 ms).
        // Third table: k_big_pool's steps (lane = window, up to kBigSeg trees of a window in flight): a step that starts at tree t
        // ends right behind the next tree a window can die at, so no tree is walked that the reference would not have reached.
        // Round 6: from tree PIGO_DEEP_LONG (default 88) on a pass runs to the LAST such tree within its 64 lanes instead of the
        // first: what is still alive there is a face with few exceptions (facefinder rejects 99.95 % of all windows by tree 100) and
        // will be walked to the end, so the number of dependent passes is what it costs -- 11 instead of 15 for a window that passes
        // all 468 trees (the 4K stress config keeps 13 k of them per frame: its deep tail IS its step).
        const int long_from = std::max(0, env_int("PIGO_DEEP_LONG", 88));
        std::vector<int16_t> pe((size_t)nt * 3);
        for (int t = 0; t < nt; ++t) {
            int e = std::min(t + 15, nt - 1);
            while (e < nt - 1 && !(c->thr[e] > lo)) ++e;
            if (t >= long_from) {
                int last = -1;
                for (int j = std::min(t + 63, nt - 1); j > e && last < 0; --j)
                    if (c->thr[j] > lo || j == nt - 1) last = j;
                if (last > e) e = last;
            }
            pe[(size_t)t] = (int16_t)std::min(e + 1, t + 64);
            pe[(size_t)nt + t] = (int16_t)std::min(nt, t + 64);
            int e2 = t;
            while (e2 < nt - 1 && !(c->thr[e2] > lo)) ++e2;
            pe[(size_t)2 * nt + t] = (int16_t)std::min(e2 + 1, t + kBigSeg);
        }
        HIP_TRY(c->d_pass_end.alloc(pe.size()));
        HIP_TRY(hipMemcpy(c->d_pass_end.p, pe.data(), pe.size() * 2, hipMemcpyHostToDevice));
        if (depth == 6) {
            // Child-pair table, node-major: codes_t[p * ntrees + t] = the code words of nodes 2p and 2p+1 of tree t (p = 0: {unused,
            // root}).  Where every lane walks its OWN tree (k_tail_deep: lane = tree; k_scan_big's pool) a level's fetch touches the
            // rows of the nodes the lanes stand on -- 2^level of them, each 8 bytes per tree apart -- instead of one 256-byte tree
            // per lane: a third of the cache lines, which is what lets those kernels read the codes from global memory and run
            // without an LDS code table next to a k_scan_region workgroup.
            std::vector<uint32_t> ct((size_t)32 * ntrees * 2);
            for (uint32_t t = 0; t < ntrees; ++t)
                for (int pp = 0; pp < 32; ++pp) {
                    memcpy(&ct[((size_t)pp * ntrees + t) * 2], c->codes.data() + ((size_t)t * 64 + 2 * pp) * 4, 8);
                }
            HIP_TRY(c->d_codes_t.alloc(ct.size() / 2));
            HIP_TRY(hipMemcpy(c->d_codes_t.p, ct.data(), ct.size() * 4, hipMemcpyHostToDevice));
        }
    }
    *out = c.release();
    return PIGO_OK;
}

extern "C" pigo_status pigo_cascade_info(const pigo_cascade *c, uint32_t *tree_depth, uint32_t *tree_num)
{
    if (!c) return fail(PIGO_ERR_PARAM, "cascade is NULL");
    if (tree_depth) *tree_depth = c->depth;
    if (tree_num) *tree_num = c->ntrees;
    return PIGO_OK;
}

extern "C" pigo_status pigo_cascade_tables(const pigo_cascade *c, int8_t *codes, size_t ncodes, float *pred, size_t npred, float *thr,
                                           size_t nthr)
{
    if (!c) return fail(PIGO_ERR_PARAM, "cascade is NULL");
    if (codes) {
        if (ncodes < c->codes.size()) return fail(PIGO_ERR_CAPACITY, "codes buffer too small");
        memcpy(codes, c->codes.data(), c->codes.size());
    }
    if (pred) {
        if (npred < c->pred.size()) return fail(PIGO_ERR_CAPACITY, "pred buffer too small");
        memcpy(pred, c->pred.data(), c->pred.size() * 4);
    }
    if (thr) {
        if (nthr < c->thr.size()) return fail(PIGO_ERR_CAPACITY, "thr buffer too small");
        memcpy(thr, c->thr.data(), c->thr.size() * 4);
    }
    return PIGO_OK;
}

extern "C" void pigo_cascade_destroy(pigo_cascade *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    c->cl_slots.clear();
    delete c;
}

// ---- plans ------------------------------------------------------------------------------------------------------

namespace {

// RunCascade's ladder, pigo.go:219-231,255, in the reference's own arithmetic (float64 products, int
// truncation).  Only rungs that classify at least one window are kept.
pigo_status build_ladder(pigo_plan &p)
{
    const PlanKey &k = p.key;
    const int rows = k.rows, cols = k.cols;
    long long scale = k.min_size;  // :219
    long long nwin = 0;
    p.scales.clear();
    p.tiles.clear();
    p.n_ladder = 0;
    while (scale <= k.max_size) {  // :226
        const double m = k.shift * (double)scale;
        const double mm = m > 1.0 ? m : 1.0;  // math.Max(ShiftFactor*scale, 1)
        const long long step = mm >= 2147483647.0 ? 2147483647LL : (long long)mm;  // :227 (any step >= the image size behaves the same)
        const long long off = scale / 2 + 1;  // :228
        long long nr = 0, nc = 0;
        if ((long long)rows - off >= off) nr = ((long long)rows - 2 * off) / step + 1;  // :230
        if ((long long)cols - off >= off) nc = ((long long)cols - 2 * off) / step + 1;  // :231
        ++p.n_ladder;
        if (nr > 0 && nc > 0) {
            if (p.scales.size() >= 2047) return fail(PIGO_ERR_PARAM, "more than 2047 non-empty scales");
            ScaleDesc sd{};
            sd.s = (int32_t)scale;
            sd.step = (int32_t)step;
            sd.off = (int32_t)off;
            sd.nr = (int32_t)nr;
            sd.nc = (int32_t)nc;
            if (nwin + nr * nc > 0xffffffffLL) return fail(PIGO_ERR_PARAM, "more than 2^32-1 windows per frame");
            sd.win_base = (uint32_t)nwin;
            nwin += nr * nc;
            const uint32_t sidx = (uint32_t)p.scales.size();
            const long long tys = (nr + kTH - 1) / kTH, txs = (nc + kTW - 1) / kTW;
            for (long long ty = 0; ty < tys; ++ty)
                for (long long tx = 0; tx < txs; ++tx) p.tiles.push_back((sidx << 21) | ((uint32_t)ty << 10) | (uint32_t)tx);
            p.scales.push_back(sd);
        } else if (scale > 0) {
            break;  // offsets only grow with the scale: every later rung is empty too
        }
        // :255 scale = int(float64(scale) + math.Max(2, float64(scale)*ScaleFactor - float64(scale)))
        const double grow = (double)scale * k.scale - (double)scale;
        const double next = (double)scale + (grow > 2.0 ? grow : 2.0);
        if (next > 4.0e9) break;
        scale = (long long)next;
    }
    p.windows = nwin;
    return PIGO_OK;
}

#ifdef PIGO_DEBUG_BUILD
// (variant 1) Compaction points of the cascade: trees whose threshold is a real one (facefinder: 24 of 468; the rest
// hold the -15 sentinel and never reject).  Correctness does not depend on this choice -- every tree's
// threshold is tested at every tree -- it only decides where survivors are re-packed.
void build_stages(pigo_plan &p, int head_stages_wanted)
{
    const pigo_cascade &c = *p.c;
    const int nt = (int)c.ntrees;
    std::vector<int> ends;
    float lo = nt ? c.thr[0] : 0.f;
    for (int i = 0; i < nt; ++i) lo = std::min(lo, c.thr[i]);
    for (int i = 0; i < nt; ++i)
        if (c.thr[i] > lo || i == nt - 1) ends.push_back(i);
    ScanArgs &a = p.args;
    a.n_head_stages = 0;
    a.n_tail_stages = 0;
    a.nh = 0;
    size_t e = 0;
    for (; e < ends.size() && a.n_head_stages < std::min(head_stages_wanted, kMaxHeadStages) && ends[e] < kNHMax; ++e) {
        a.head_end[a.n_head_stages++] = ends[e];
        a.nh = ends[e] + 1;
    }
    if (a.n_head_stages == 0 && nt > 0) {  // first real threshold is deeper than the LDS tables reach
        a.head_end[0] = std::min(nt, kNHMax) - 1;
        a.n_head_stages = 1;
        a.nh = a.head_end[0] + 1;
        while (e < ends.size() && ends[e] <= a.head_end[0]) ++e;
    }
    std::vector<int> tail(ends.begin() + e, ends.end());
    if (!tail.empty() || a.nh < nt) {
        if (tail.empty() || tail.back() != nt - 1) tail.push_back(nt - 1);
        while ((int)tail.size() > kMaxTailStages) {  // merge neighbours until they fit
            std::vector<int> m;
            for (size_t i = 1; i < tail.size(); i += 2) m.push_back(tail[i]);
            if (tail.size() % 2) m.push_back(tail.back());
            tail.swap(m);
        }
        for (int v : tail) a.tail_end[a.n_tail_stages++] = v;
    }
}
#endif  // PIGO_DEBUG_BUILD

int env_int(const char *name, int dflt);

// The two launches of k_tail_deep: the first three passes (192 trees from `lo`; measured plateau 160..256) with their codes in
// LDS; the few windows that survive them continue in a second launch holding the remaining codes.  false: the codes do not
// fit one CU's LDS.
bool set_deep_window(pigo_plan &p, int lo)
{
    const int nt = (int)p.c->ntrees;
    p.args.deep_lo = lo;
    p.deep_mid = std::min(nt, lo + std::max(64, env_int("PIGO_DEEP_SPLIT", 192)));
    p.deep_lds = (size_t)(p.deep_mid - lo) * kCodeStride * 4 + (size_t)kDeepWaves * kPatchBytes;
    p.deep_lds2 = (size_t)(nt - p.deep_mid) * kCodeStride * 4 + (size_t)kDeepWaves * kPatchBytes;
    return p.deep_lds <= (size_t)(160 << 10) - 1024 && p.deep_lds2 <= (size_t)(160 << 10) - 1024;
}

// Stages (compaction points) and LDS table windows of k_scan_tile.  A stage never spans more than kTabTrees
// trees, so the tables of the trees it walks are always resident.
bool build_tile_stages(pigo_plan &p)
{
    const pigo_cascade &c = *p.c;
    const int nt = (int)c.ntrees;
    if (nt == 0 || c.depth != 6) return false;
    float lo = c.thr[0];
    for (int i = 0; i < nt; ++i) lo = std::min(lo, c.thr[i]);
    ScanArgs &a = p.args;
    a.nh_lds = std::min(nt, std::max(1, env_int("PIGO_NH_LDS", 28)));
    // (plans that run variant 3 leave only the big scales -- 1 % of the windows, a quarter of the deep entries -- to the tile
    // kernel: walking them to the next real threshold, tree 47, before the hand-off costs little there and thins the tail)
    const bool v3_plan = (!p.rot || p.rot_lds) && p.key.dim % 4 == 0 && p.max_frames >= 8;
    a.nh_glb = std::min(nt, std::max(1, (p.rot && !p.rot_lds) ? env_int("PIGO_NH_ROT", 18) : env_int("PIGO_NH_GLB", v3_plan ? 48 : 28)));
    a.deep_lo = std::min(a.nh_lds, a.nh_glb);
    // (variant 3's mid scale group hands its survivors to the deep list earlier than the small group -- build_region_groups --
    // and a deep list that overflows spills into k_tail_deep's queue: its code table has to start there)
    p.nh_reg_mid = std::min(a.nh_lds, std::max(1, env_int("PIGO_NH_REG1", 13)));
    if (v3_plan) a.deep_lo = std::min(a.deep_lo, p.nh_reg_mid);
    // LDS table capacity per class: enough for the trees the class walks before handing off; the dense stages'
    // table windows are planned for the smaller of the two so that they fit either
    p.tab_lds = std::min(kTabTrees, std::max(a.nh_lds, 8));
    p.tab_glb = std::min(kTabTrees, std::max(a.nh_glb, 8));
    a.tab_trees = (p.rot && !p.rot_lds) ? p.tab_glb : std::min(p.tab_lds, p.tab_glb);
    const int cap = a.tab_trees;
    std::vector<int> ends;
    int begin = 0;
    for (int i = 0; i < nt; ++i) {
        if (c.thr[i] > lo || i == nt - 1) {
            while (i - begin + 1 > cap) {  // split over-long stages
                ends.push_back(begin + cap - 1);
                begin += cap;
            }
            ends.push_back(i);
            begin = i + 1;
        }
    }
    while ((int)ends.size() > kMaxStages) {  // too many compaction points: merge neighbours that still fit
        std::vector<int> m;
        int b = 0;
        bool merged = false;
        for (size_t i = 0; i < ends.size(); ++i) {
            if (i + 1 < ends.size() && ends[i + 1] - b + 1 <= cap) {
                merged = true;
                ++i;
            }
            m.push_back(ends[i]);
            b = ends[i] + 1;
        }
        if (!merged) return false;
        ends.swap(m);
    }
    a.n_stages = (int)ends.size();
    if (!set_deep_window(p, a.deep_lo)) return false;
    int hi = 0;  // trees [.., hi) are resident
    for (int st = 0; st < a.n_stages; ++st) {
        const int t0 = st == 0 ? 0 : ends[st - 1] + 1;
        a.st_end[st] = (int16_t)ends[st];
        a.st_load_hi[st] = 0;
        if (ends[st] >= hi) {
            int last = st;
            while (last + 1 < a.n_stages && ends[last + 1] - t0 + 1 <= cap) ++last;
            hi = ends[last] + 1;
            a.st_load_hi[st] = (int16_t)hi;
        }
    }
    return true;
}

struct TileRule {
    int tw_log2, th;
    long long max_pix;
};

// Per-rung tile geometry: the first rule whose LDS pixel footprint fits wins; rungs that fit none (or frames
// that cannot be copied as aligned dwords, or the rotated scan) read pixels from global memory.
void build_tile_classes(pigo_plan &p)
{
    std::vector<TileRule> rules;
    const char *env = tune_env("PIGO_TILE_RULES");
    // (plans of a few frames: ONE LDS class -- 64 x 16 tiles for every rung whose footprint fits 40 KiB -- so that the LDS classes
    // are one launch: a one-frame class is a single round of workgroups, and launches of it run one after the other)
    std::string spec = env && *env ? env : (p.max_frames < 8 ? "6,16,40960" : "6,32,16384;6,16,40960");
    {
        size_t pos = 0;
        while (pos < spec.size()) {
            size_t end = spec.find(';', pos);
            if (end == std::string::npos) end = spec.size();
            TileRule r{};
            if (sscanf(spec.substr(pos, end - pos).c_str(), "%d,%d,%lld", &r.tw_log2, &r.th, &r.max_pix) == 3 && r.tw_log2 >= 5 && r.tw_log2 <= 6 &&
                r.th >= 1 && ((1 << r.tw_log2) * r.th) % 512 == 0 && (1 << r.tw_log2) * r.th <= 4096)
                rules.push_back(r);
            pos = end + 1;
        }
    }
    const bool lds_allowed = (!p.rot || p.rot_lds) && (p.key.dim % 4 == 0) && env_int("PIGO_LDS_TILES", 1) != 0;
    const long long rqc = p.rot ? kQCos[p.angle_idx] : 0, rqs = p.rot ? kQSin[p.angle_idx] : 0;
    const int g_tw_log2 = 6, g_th = env_int("PIGO_GLOBAL_TH", 16);
    const size_t static_slack = 1024;
    const size_t buckets[] = {36u << 10, 48u << 10, 64u << 10, 80u << 10, 104u << 10, 128u << 10, (160u << 10) - static_slack};
    struct Pick {
        int tw_log2, th;
        bool lds;
        size_t dyn;
        int bucket;
        int qb_div;
    };
    std::vector<Pick> picks(p.scales.size());
    for (size_t k = 0; k < p.scales.size(); ++k) {
        ScaleDesc &sd = p.scales[k];
        int up = (sd.s + 1) / 2, down = (127 * sd.s) >> 8;
        if (p.rot) {
            // extent of the rotated sample offsets (dR, dC) of this rung over every node of the cascade (pix_le / k_build_tab):
            // the LDS tile must cover [centre - up, centre + down] in both directions
            const long long qc = (long long)sd.s * rqc, qs = (long long)sd.s * rqs;  // pigo.go:159-160
            long long lo = 0, hi = 0;
            const pigo_cascade &c = *p.c;
            for (size_t i = 0; i + 3 < c.codes.size(); i += 2) {  // byte pairs (cr, cc) of both points of every node
                const long long cr = c.codes[i], cc = c.codes[i + 1];
                const long long dr = (qc * cr - qs * cc) >> 16, dc = (qs * cr + qc * cc) >> 16;
                lo = std::min(lo, std::min(dr, dc));
                hi = std::max(hi, std::max(dr, dc));
            }
            up = (int)-lo;
            down = (int)hi;
        }
        sd.up = up;
        sd.down = down;
        sd.pitch = 0;
        // LDS queues: 1.5 tiles' worth of entries shared by the two ping-pong regions; an overflow (stage 0 and stage 1
        // survivors together exceeding that) is caught on the device
        const int qb_div = p.rot ? (env_int("PIGO_ROT_QB_DIV", 1) == 1 ? 1 : 2) : (env_int("PIGO_QB_DIV", 2) == 1 ? 1 : 2);
        Pick pk{g_tw_log2, g_th, false, 0, 0, qb_div};
        if (lds_allowed) {
            for (const TileRule &r : rules) {
                const int tw = 1 << r.tw_log2;
                long long w = (long long)(tw - 1) * sd.step + up + down + 1 + 3;
                long long pitch = (w + 3) / 4 * 4;
                if ((pitch / 4) % 2 == 0) pitch += 4;  // odd dword pitch: consecutive rows start on different banks
                const long long ph = (long long)(r.th - 1) * sd.step + up + down + 1;
                const long long pix = pitch * ph;
                if (pix > r.max_pix) continue;
                if ((long long)std::max(up, down) * pitch + std::max(up, down) > 32767) continue;  // offsets must fit the packed int16 table
                sd.pitch = (int32_t)pitch;
                pk = Pick{r.tw_log2, r.th, true, (size_t)((pix + 15) / 16 * 16), 0, qb_div};
                break;
            }
        }
        const size_t nwin = (size_t)(1 << pk.tw_log2) * pk.th;
        const size_t qbytes = 6 * (nwin + nwin / pk.qb_div);
        pk.dyn += (size_t)(pk.lds ? p.tab_lds : p.tab_glb) * 64 * (pk.lds ? 4 : 8) + qbytes + (qbytes >= (size_t)64 * (kLateWaves * kLateTrees + 1) * 4 ? 0 : (size_t)64 * (kLateWaves * kLateTrees + 1) * 4);
        pk.bucket = (int)(sizeof(buckets) / sizeof(buckets[0])) - 1;
        for (int b = 0; b < (int)(sizeof(buckets) / sizeof(buckets[0])); ++b)
            if (pk.dyn <= buckets[b]) {
                pk.bucket = b;
                break;
            }
        picks[k] = pk;
    }
    p.classes.clear();
    p.tiles2.clear();
    const bool merge_buckets = p.max_frames < 8 && env_int("PIGO_MERGE_BUCKETS", 1) != 0;
    std::vector<char> done(p.scales.size(), 0);
    for (size_t k = 0; k < p.scales.size(); ++k) {
        if (done[k]) continue;
        pigo_plan::TileClass cls{picks[k].tw_log2, picks[k].th, (1 << picks[k].tw_log2) * picks[k].th, picks[k].qb_div, picks[k].lds, (uint32_t)p.tiles2.size(), 0, 0, 0};
        for (size_t j = k; j < p.scales.size(); ++j) {
            const Pick &q = picks[j];
            // (plans of a few frames: one launch per tile geometry whatever the LDS footprint -- a one-frame class is a single round
            // of workgroups bound by a tile's own latency, and two launches of it run one after the other: 25 + 28 us instead of ~30)
            if (done[j] || q.tw_log2 != cls.tw_log2 || q.th != cls.th || q.lds != cls.lds || (!merge_buckets && q.bucket != picks[k].bucket) || q.qb_div != cls.qb_div) continue;
            done[j] = 1;
            cls.dyn_lds = std::max(cls.dyn_lds, q.dyn);
            const ScaleDesc &sd = p.scales[j];
            const int tw = 1 << cls.tw_log2;
            const int tys = (sd.nr + cls.th - 1) / cls.th, txs = (sd.nc + tw - 1) / tw;
            for (int ty = 0; ty < tys; ++ty)
                for (int tx = 0; tx < txs; ++tx) p.tiles2.push_back(make_uint2((unsigned)j, ((unsigned)ty << 16) | (unsigned)tx));
        }
        cls.ntiles = (uint32_t)p.tiles2.size() - cls.tile0;
        p.classes.push_back(cls);
    }
}

// The quad pass of the region kernel's deep list (k_scan_region), per scale group {small, mid}: PIGO_REG_QUAD0/1 override it under
// PIGO_TUNING=1 (profiles/r04_experiments.md section 15).
constexpr int kRegQuad[2] = {32, 16};

// Variant 3: cut the scale ladder into groups by footprint and the image into cells whose region (cell + halo) fits the LDS
// next to the tables and queues of k_scan_region.  Returns false when the plan is not eligible (rotated scan, stride not a
// multiple of 4, cascade without a usable stage list).
bool build_region_groups(pigo_plan &p)
{
    p.regions.clear();
    const ScanArgs &a = p.args;
    // (rotated scans: only where the clamp-free LDS form exists -- rot_lds: no Go panic possible, see k_scan_tile's loader)
    if (!p.tile_ok || (p.rot && !p.rot_lds) || p.key.dim % 4 != 0 || p.scales.empty()) return false;
    const int nh0 = std::min(a.nh_lds, a.nh_glb);
    if (nh0 < 1 || nh0 > kTabTrees || a.deep_lo > nh0) return false;
    // The hand-over tree per group: 28 for the small group (it loses with less: 3.81 -> 3.94 ms at 18).  The mid group's survivors
    // are mostly face windows that go a long way: lane = tree takes them over at tree 13 already and fifteen trees' leaves and
    // codes less stay resident -- mid group 1.85 -> 1.55 ms at the 1080p config (18: 1.73, 9: 1.60, 6: 1.66).  Not where a
    // region holds tens of thousands of mid-scale windows (the 4K stress config): there the earlier hand-over floods the deep
    // list (64.2 -> 61.2 Gwindows/s even with longer lists), so the group loop below decides by the windows per region unless
    // PIGO_NH_REG1 forces a tree.  It must sit right behind a stage end.
    int nh1 = nh0;
    for (int st = 0; st < a.n_stages; ++st)
        if (a.st_end[st] + 1 == p.nh_reg_mid && (p.nh_reg_mid >= a.deep_lo || (p.max_frames < 8 && env_int("PIGO_ONE", 1) != 0)) && p.nh_reg_mid <= nh0) nh1 = p.nh_reg_mid;
    const bool nh1_forced = tune_env("PIGO_NH_REG1") != nullptr;
    int nh = nh0;
    // chunk stages: the leading stages that end below the pooling tree
    const int t_pool_wanted = std::max(1, env_int("PIGO_REG_POOL_TREE", 4));
    int n_cs = 0;
    while (n_cs < a.n_stages && n_cs < 4 && a.st_end[n_cs] < t_pool_wanted && a.st_end[n_cs] < nh) ++n_cs;
    if (n_cs < 1) return false;
    int t_pool = a.st_end[n_cs - 1] + 1;
    int cs_end[4] = {0, 0, 0, 0};
    for (int i = 0; i < n_cs; ++i) cs_end[i] = a.st_end[i];
    // The mid group (few windows per chunk, two per lane in stage 0) runs the single-tree stages [1] and [2] as ONE two-tree
    // stage and pools from tree 3: some windows evaluate tree 2 that tree 1 would have stopped -- wasted work, not a different
    // result, every threshold is still tested in order -- for one compaction and one dependent stage less (mid group
    // 1.56 -> 1.45 ms; the small group, eight windows per lane in stage 0, loses with it: 3.81 -> 3.91).
    const bool mid_merge = env_int("PIGO_REG_MERGE1", 1) != 0 && n_cs == 4 && cs_end[0] == 0 && cs_end[1] == 1 && cs_end[2] == 2 && cs_end[3] == 3;
    // The small group merges the OTHER pair: [0] [1] [2-3], pool from tree 4 as before (3.82 -> 3.75 ms; [0] [1-2] loses there,
    // [0] [1-3] and a fifth stage lose badly: profiles/r03_experiments.md section 19).
    const bool small_merge = env_int("PIGO_REG_MERGE0", 1) != 0 && n_cs == 4 && cs_end[0] == 0 && cs_end[1] == 1 && cs_end[2] == 2 && cs_end[3] == 3;
    int n_cs_g[3] = {small_merge ? 3 : n_cs, mid_merge ? 2 : n_cs, mid_merge ? 2 : n_cs};
    int t_pool_g[3] = {t_pool, mid_merge ? 3 : t_pool, mid_merge ? 3 : t_pool};
    int cs_end_g[3][4] = {{cs_end[0], cs_end[1], small_merge ? 3 : cs_end[2], small_merge ? 0 : cs_end[3]},
                          {mid_merge ? 0 : cs_end[0], mid_merge ? 2 : cs_end[1], mid_merge ? 0 : cs_end[2], mid_merge ? 0 : cs_end[3]},
                          {mid_merge ? 0 : cs_end[0], mid_merge ? 2 : cs_end[1], mid_merge ? 0 : cs_end[2], mid_merge ? 0 : cs_end[3]}};
    // (experiments: PIGO_REG_CS0 / PIGO_REG_CS1 = "e0,e1,..": the last tree of every chunk stage of the small / mid group; every
    // end must be a stage end of the cascade, at most four, ascending, below the hand-over tree)
    for (int g = 0; g < 2; ++g) {
        const char *e = tune_env(g == 0 ? "PIGO_REG_CS0" : "PIGO_REG_CS1");
        if (!e || !*e) continue;
        int v[4], nv = 0;
        for (const char *q = e; *q && nv < 4;) {
            v[nv++] = atoi(q);
            while (*q && *q != ',') ++q;
            if (*q == ',') ++q;
        }
        bool ok = nv >= 1;
        for (int i = 0; i < nv && ok; ++i) {
            bool is_end = false;
            for (int st = 0; st < a.n_stages; ++st) is_end = is_end || a.st_end[st] == v[i];
            ok = is_end && (i == 0 || v[i] > v[i - 1]) && v[i] + 1 < nh0 && v[i] + 1 <= kTabTrees;
        }
        if (!ok) continue;
        n_cs_g[g] = nv;
        t_pool_g[g] = v[nv - 1] + 1;
        for (int i = 0; i < 4; ++i) cs_end_g[g][i] = i < nv ? v[i] : 0;
        if (g == 1) {
            n_cs_g[2] = n_cs_g[1];
            t_pool_g[2] = t_pool_g[1];
            for (int i = 0; i < 4; ++i) cs_end_g[2][i] = cs_end_g[1][i];
        }
    }
    const int pool_cap = kRegWavePool;
    // leaves + raw codes of the nh trees, per wave a queue of kRegWaveChunk 6-byte and a pool of kRegWavePool 8-byte entries;
    // plus, per group, the offset tables of the chunk-stage trees of every scale of the group
    constexpr int NG = 3;  // scale groups: small, mid (a third one was measured and loses: profiles/r02_experiments.md; its slot stays empty)
    const int chunkg[NG] = {std::min(kRegWaveChunk, std::max(64, env_int("PIGO_REG_CHUNK0", 512) & ~63)),
                            std::min(kRegWaveChunk, std::max(64, env_int("PIGO_REG_CHUNK1", 128) & ~63)),
                            std::min(kRegWaveChunk, std::max(64, env_int("PIGO_REG_CHUNK2", 64) & ~63))};
    // (deep lists of 1024 / 512 entries: 768 / 1024 / 1536 for the small group measure 65.3 / 65.5 / 64.7 k -- the LDS goes to
    // the cell instead --, 256 for the mid group starts to spill on the benchmark frames)
    const int deepg[NG] = {std::max(64, env_int("PIGO_REG_DEEP0", 1024)), std::max(64, env_int("PIGO_REG_DEEP1", 512)), std::max(64, env_int("PIGO_REG_DEEP2", 256))};
    // static LDS of k_scan_region (per-scale geometry, counters, thresholds): 3072.  Plans that have rungs beyond the groups give
    // the big scales' kernels (k_scan_big, k_tail_deep: bound by the vector-memory path the region kernel leaves idle) room for
    // one small workgroup on the same CU: the first group -- the launch they run next to -- takes that much less (PIGO_REG_RESERVE0_KB,
    // PIGO_REG_RESERVE1_KB for the second group; 0 = the whole CU).
    const bool has_big = p.scales.back().s > env_int("PIGO_REG_S1", 148) && env_int("PIGO_BIG", 1) != 0;
    const bool one_mode = p.max_frames < 8 && env_int("PIGO_ONE", 1) != 0;  // k_scan_one: nothing runs next to its workgroups
    const size_t reserve_g[3] = {(size_t)std::max(0, std::min(96, env_int("PIGO_REG_RESERVE0_KB", (has_big && !one_mode) ? 8 : 0))) << 10,
                                 (size_t)std::max(0, std::min(96, env_int("PIGO_REG_RESERVE1_KB", 0))) << 10, 0};
    p.side_lds = reserve_g[0];
    const size_t max_dyn_all = (size_t)(160 << 10) - 3072;
    // (group limits: 51 / 148 measured best after the deep list got cheaper -- 42…51 / 148 within 0.3 %, 62 / 135 3 % slower)
    // (no rung beyond s = 255 in a region group: the pool and the deep lists compute a window's offsets in packed 16-bit arithmetic,
    // |code * s| <= 128 * 255 -- fly_points_pk)
    const int smax[NG] = {std::min(255, env_int("PIGO_REG_S0", 51)), std::min(255, env_int("PIGO_REG_S1", 148)), 0};
    const int cwmax[NG] = {env_int("PIGO_REG_CW0", 320), env_int("PIGO_REG_CW1", 192), env_int("PIGO_REG_CW2", 128)};
    int k = 0;
    const int nscales = (int)p.scales.size();
    // k_scan_one (plans of a few frames): how many regions each group gets.  A frame's share of the chip is 256 / max_frames
    // items; the big rungs take ceil(chunks / 16) of them (one chunk per wave), the groups split the rest by their windows -- a
    // mid-group window costs about three small-group ones (fixed costs per region, two windows per lane in stage 0).
    int one_target[NG] = {0, 0, 0};
    if (one_mode) {
        long long w[2] = {0, 0}, chunks = 0;
        for (int j = 0; j < nscales; ++j) {
            const long long nw = (long long)p.scales[j].nr * p.scales[j].nc;
            if (p.scales[j].s <= smax[0]) w[0] += nw;
            else if (p.scales[j].s <= smax[1]) w[1] += nw;
            else if (has_big) chunks += (nw + kBigChunk - 1) / kBigChunk;
        }
        const int slots = std::max(2, env_int("PIGO_ONE_SLOTS", 256) / p.max_frames);
        // (PIGO_ONE_BIGMIX, the default: the big-scale chunks are no items of their own -- they ride on the last waves of the region
        // items, whose scan phase leaves the texture path idle; see k_scan_one)
        const bool bigmix = env_int("PIGO_ONE_BIGMIX", 1) != 0;
        const int nbig = bigmix ? 0 : (int)std::min<long long>((chunks + kOneBigWaves - 1) / kOneBigWaves, slots / 2);
        // (a mid-group window costs about three small-group ones alone; with the big-scale chunks riding on the regions -- the mid
        // group's first -- 4.5 balances faces, rotated scans and small frames best: r05_experiments.md section 5)
        const double cost0 = (double)w[0], cost1 = (double)w[1] * (env_int("PIGO_ONE_W1_X10", bigmix ? 45 : 30) / 10.0);
        const int rest = std::max(2, slots - nbig);
        int t1 = w[1] > 0 ? std::max(1, (int)(rest * cost1 / std::max(1.0, cost0 + cost1) + 0.5)) : 0;
        if (w[0] > 0) t1 = std::min(t1, rest - 1);
        one_target[0] = w[0] > 0 ? std::max(1, rest - t1) : 0;
        one_target[1] = t1;
    }
    // (a group that does not fit is fatal for the first two -- the plan then runs variant 2 -- and simply dropped for the third:
    // its rungs stay with the tile classes)
#define REG_BAIL          \
    {                     \
        if (env_int("PIGO_SYNC_DEBUG", 0) != 0) fprintf(stderr, "[pigo] region group %d does not fit (pigo_hip.hip:%d)\n", g, __LINE__); \
        if (g < 2) return false; \
        k = k_lo;         \
        dropped = true;   \
        break;            \
    }
    bool dropped = false;
    for (int g = 0; g < NG && k < nscales && !dropped; ++g) {
        nh = (g >= 1 && nh1_forced) ? nh1 : nh0;
        n_cs = n_cs_g[g];
        t_pool = t_pool_g[g];
        const int k_lo = k;
        int up = 0, dn = 0;
        while (k < nscales && p.scales[k].s <= smax[g]) {
            // footprint of a window above/left and below/right of its centre (build_tile_classes: upright ceil(s/2) and
            // (127*s) >> 8, rotated the extreme rotated offsets of the rung)
            up = std::max(up, p.scales[k].up);
            dn = std::max(dn, p.scales[k].down);
            ++k;
        }
        if (k == k_lo) continue;
        if (k - k_lo > kRegMaxScales) REG_BAIL;
        // The deep list is sized for the windows a region holds: 1024 / 512 entries are right for the 1080p config's ~72 k / ~5.5 k
        // windows per region, but a dense ladder (the 4K stress config: shift 0.05, scale 1.05 -- several hundred thousand windows
        // per region) overflowed them into k_tail_deep's queue: 2 ms of a 15 ms step.  Pass 0 sizes the cells with the default,
        // counts the region's windows and, if the list is short for them, pass 1 sizes again with a longer one (<= 2048 entries:
        // the LDS it takes comes out of the cell).
        const size_t max_dyn = max_dyn_all - reserve_g[g];
        // (plans of a few frames are latency-bound: a face keeps ~100 windows of ONE region alive through all 468 trees, and the 16
        // waves of that region's workgroup would finish them in three rounds of ~12 dependent passes while the rest of the chip
        // idles -- there every window that passes the hand-over tree goes to k_tail_deep's queue instead, one wave per window
        // over the whole chip: no deep list)
        // (k_scan_one keeps the lists: their quad pass and first passes out of LDS thin what reaches its global queues)
        const bool no_deep_list = p.max_frames < 8 && !one_mode && env_int("PIGO_REG_DEEP_SMALL", 0) == 0;
        int deep_cap_g = no_deep_list ? 0 : one_mode ? std::max(64, env_int(g == 0 ? "PIGO_ONE_DEEP0" : "PIGO_ONE_DEEP1", g == 0 ? 512 : 256)) : deepg[g];
        const double deep_per_window[NG] = {0.0125, 0.026, 0.026};  // (the 1080p config: 81 k windows -> 1024, 6 k -> the 512 minimum)
        const bool deep_fixed = tune_env(g == 0 ? "PIGO_REG_DEEP0" : g == 1 ? "PIGO_REG_DEEP1" : "PIGO_REG_DEEP2") != nullptr;
        size_t fixed = 0;
        RegionArgs r{};
        const int halo = up + dn;
        // wave queues: with stage 0 = tree 0 alone its survivors go to a 2-byte queue (window index within the chunk | leaf index),
        // and the {index, sum} queue behind it only holds wq entries at a time (reg_wave_batch): 28 instead of 48 KiB for 512-window chunks
        const bool compress = env_int("PIGO_REG_COMPRESS", 1) != 0 && cs_end_g[g][0] == 0 && chunkg[g] <= 512;
        const int wq = compress ? std::min(chunkg[g], std::max(64, env_int("PIGO_REG_WQ", 256) & ~63)) : chunkg[g];
        const size_t wave_bytes = (compress ? (size_t)chunkg[g] * 2 : 0) + (size_t)wq * 6 + kRegWavePool * 8;
        for (int pass = 0; pass < 2 && !dropped; ++pass) {
        fixed = (size_t)nh * 64 * 8 + (size_t)(kRegThreads / 64) * wave_bytes + (size_t)deep_cap_g * 8 +
                (size_t)(k - k_lo) * t_pool * 256;
        if (fixed + 16384 > max_dyn) REG_BAIL;
        const size_t budget = max_dyn - fixed;
        // the largest cell whose region fits; then as many equal cells as the image needs; more, smaller cells when the
        // plan's batch is too small to give every CU a workgroup
        double shrink = 1.0;
        r = RegionArgs{};
        if (one_target[g] > 0) {
            // k_scan_one: a frame's items (big bundles + regions of both groups) are about one per CU and frame of the plan -- the
            // grid with the most regions within the group's share wins (ties: the smaller region)
            long long best_n = -1, best_bytes = 0;
            for (int ncx = 1; ncx <= std::max(1, p.key.cols / 16); ++ncx) {  // (beyond the share too: wide cells may not fit the packed offsets)
                const int cell_w = (((p.key.cols + ncx - 1) / ncx) + 3) & ~3;
                int pitch = (cell_w + halo + 3 + 3) & ~3;
                if ((pitch / 4) % 2 == 0) pitch += 4;  // odd dword pitch: consecutive rows start on different banks
                if ((long long)std::max(up, dn) * pitch + std::max(up, dn) > 32767) continue;  // packed int16 offsets (checked again below)
                const int ch_max = (int)(budget / (size_t)pitch) - halo;
                if (ch_max < 8) continue;
                const int ncy_min = (p.key.rows + ch_max - 1) / ch_max;
                const int ncy = std::min(std::max(ncy_min, one_target[g] / ncx), std::max(1, p.key.rows / 8));
                if (ncy < ncy_min) continue;
                const int cell_h = (p.key.rows + ncy - 1) / ncy;
                const long long nreg = (long long)ncx * ncy, bytes = (long long)pitch * (cell_h + halo);
                const bool within = nreg <= one_target[g], best_within = best_n >= 0 && best_n <= one_target[g];
                // grids within the share beat grids beyond it; within: more regions, then fewer bytes; beyond: fewer regions
                bool better = best_n < 0;
                if (!better && within && !best_within) better = true;
                if (!better && within && best_within) better = nreg > best_n || (nreg == best_n && bytes < best_bytes);
                if (!better && !within && !best_within) better = nreg < best_n || (nreg == best_n && bytes < best_bytes);
                if (better) {
                    best_n = nreg;
                    best_bytes = bytes;
                    r.ncx = ncx;
                    r.cell_w = cell_w;
                    r.ncy = ncy;
                    r.cell_h = cell_h;
                    r.pitch = pitch;
                    r.rows = cell_h + halo;
                }
            }
            if (best_n < 0) REG_BAIL;
        } else
        for (;;) {
            int cw_max = std::max(32, (int)(cwmax[g] * shrink)) & ~3;
            // the widest cells first, then up to six more columns of cells: the grid with the fewest regions wins (ties: the
            // smaller region), so a budget that does not divide well by the widest cell is not wasted on a thin one
            const int ncx0 = (p.key.cols + cw_max - 1) / cw_max;
            long long best_n = -1, best_bytes = 0;
            for (int ncx = ncx0; ncx <= ncx0 + (env_int("PIGO_REG_CELL_SEARCH", 1) != 0 ? 6 : 0); ++ncx) {
                const int cell_w = (((p.key.cols + ncx - 1) / ncx) + 3) & ~3;
                int pitch = (cell_w + halo + 3 + 3) & ~3;
                if ((pitch / 4) % 2 == 0) pitch += 4;  // odd dword pitch: consecutive rows start on different banks
                int ch_max = (int)(budget / (size_t)pitch) - halo;
                if (ch_max < 16 && ncx > ncx0) continue;
                ch_max = std::max(16, (int)(ch_max * shrink));
                const int ncy = (p.key.rows + ch_max - 1) / ch_max;
                const int cell_h = (p.key.rows + ncy - 1) / ncy;
                const long long nreg = (long long)ncx * ncy, bytes = (long long)pitch * (cell_h + halo);
                if (best_n < 0 || nreg < best_n || (nreg == best_n && bytes < best_bytes)) {
                    best_n = nreg;
                    best_bytes = bytes;
                    r.ncx = ncx;
                    r.cell_w = cell_w;
                    r.ncy = ncy;
                    r.cell_h = cell_h;
                    r.pitch = pitch;
                    r.rows = cell_h + halo;
                }
            }
            const int ch_max = r.cell_h;
            if ((long long)r.ncx * r.ncy * p.max_frames >= env_int("PIGO_REG_MIN_REGIONS", 256) || (cw_max <= 32 && ch_max <= 16) || shrink < 0.05) break;
            shrink *= 0.8;
        }
        if ((size_t)r.pitch * r.rows > budget) REG_BAIL;
        if (pass == 0) {
            long long wins = 0;
            for (int j = k_lo; j < k; ++j)
                wins += (long long)((r.cell_h + p.scales[j].step - 1) / p.scales[j].step) * ((r.cell_w + p.scales[j].step - 1) / p.scales[j].step);
            bool again = false;
            if (g >= 1 && !nh1_forced && nh != nh1 && wins <= 16384) {  // few mid-scale windows per region: hand over early
                nh = nh1;
                again = true;
            }
            const int want = std::min(2048, (int)((wins * deep_per_window[g] + 255) / 256) * 256);
            if (!deep_fixed && !no_deep_list && want > deep_cap_g) {
                deep_cap_g = want;
                again = true;
            }
            if (again) continue;
        }
        break;
        }
        if (dropped) break;
        if ((long long)std::max(up, dn) * r.pitch + std::max(up, dn) > 32767) REG_BAIL;  // packed int16 offsets
        if ((size_t)r.pitch * r.rows + fixed + 2048 >= (1u << 18)) REG_BAIL;  // pool entries hold an 18-bit LDS address
        for (int j = k_lo; j < k; ++j) {  // queue entries hold a window's index within (rung, cell) in 16 bits
            const long long ni = (r.cell_h + p.scales[j].step - 1) / p.scales[j].step + 1, nj = (r.cell_w + p.scales[j].step - 1) / p.scales[j].step + 1;
            if (ni * nj > 65535) REG_BAIL;
        }
        if (dropped) break;
        r.k_lo = k_lo;
        r.k_hi = k;
        r.halo_up = up;
        r.pool_cap = pool_cap;
        r.n_chunk_stages = n_cs;
        r.t_pool = t_pool;
        for (int i = 0; i < 4; ++i) r.cs_end[i] = cs_end_g[g][i];
        r.nh = nh;
        r.wave_chunk = chunkg[g];
        r.deep_cap = deep_cap_g;
        // the 64 x 65 dwords of the first deep pass's codes must fit the wave queues + pools (16.25 KiB)
        r.deep_lds_codes = (env_int("PIGO_REG_DEEP_LDS", 1) != 0 && (size_t)(kRegThreads / 64) * wave_bytes >= (size_t)64 * 65 * 4) ? 1 : 0;
        r.prio = one_mode ? 0 : std::max(0, std::min(3, env_int("PIGO_REG_PRIO", 1)));
        // quad pass in front of the deep list's one-window passes (k_scan_region): 16 = four windows x 16 trees, 32 = two x 32.  It needs
        // the staged codes of 64 + that many trees in the wave queues' LDS and a second list of deep_cap entries in the chunk-stage
        // tables' (both idle by then); a setting that does not fit falls back to the next smaller one.
        r.quad = 0;
        for (int want = env_int(g == 0 ? "PIGO_REG_QUAD0" : "PIGO_REG_QUAD1", kRegQuad[g > 0 ? 1 : 0]); want >= 16 && !r.quad; want -= 16)
            if ((want == 16 || want == 32) && r.deep_lds_codes && deep_cap_g > 0 && (size_t)(kRegThreads / 64) * wave_bytes >= (size_t)(64 + want) * 65 * 4 &&
                ((size_t)(k - k_lo) * t_pool * 64 + (size_t)nh * 128) * 4 >= (size_t)deep_cap_g * 8)
                r.quad = want;
        r.compress = compress ? 1 : 0;
        r.cut = env_int("PIGO_REG_CUT", 0);  // (read by the debug build only)
        r.one_local = one_mode ? std::max(0, std::min(8, env_int(g == 0 ? "PIGO_ONE_LOCAL0" : "PIGO_ONE_LOCAL1", g == 0 ? 1 : 2))) : 0;
        r.wave_q = wq;
        for (int j = k_lo; j < k; ++j)
            if (p.scales[j].s >= (1 << 14)) REG_BAIL;
        if (dropped) break;
        p.regions.push_back({r, fixed + (size_t)r.pitch * r.rows});
    }
#undef REG_BAIL
    if (p.regions.empty()) return false;
    const int kbig = p.regions.back().args.k_hi;
    for (pigo_plan::TileClass &cls : p.classes) {
        cls.v3_skip = 0;
        for (uint32_t t = 0; t < cls.ntiles; ++t)
            if ((int)p.tiles2[cls.tile0 + t].x < kbig) ++cls.v3_skip;  // tiles are stored in rung order within a class
    }
    // the tile classes keep rungs [kbig, ..): their survivors need k_tail_deep's LDS patch only if one of them is small enough
    p.tile_patch = kbig < nscales && p.scales[kbig].s <= kPatchMaxS;

    return true;
}

// Variant 3: the rungs beyond the last region group go through k_scan_big (+ a chain of k_tail_deep launches that fits the LDS
// the first region group leaves free).  Cuts those rungs' windows into chunks of kBigChunk and plans the stages.
pigo_status build_big(pigo_plan &p)
{
    p.big_ok = false;
    p.big_items.clear();
    p.side_splits.clear();
    const ScanArgs &a = p.args;
    const pigo_cascade &c = *p.c;
    const int nt = (int)c.ntrees, nscales = (int)p.scales.size();
    if (!p.region_ok || p.regions.empty() || env_int("PIGO_BIG", 1) == 0) return PIGO_OK;
    const int kbig = p.regions.back().args.k_hi;
    if (kbig >= nscales) return PIGO_OK;
    // hand-over tree: right behind a stage end, as for the tile classes (48 for the facefinder: the next real threshold after 27)
    // (plans of a few frames are latency-bound: a pooled window walks its trees one after the other, 18 dependent round trips to
    // memory per pool step, so there the pool only finishes the cascade's first stages and lane = tree takes over at tree 4)
    // Batches: the pool (lane = window, one wave per CU: its working set has to fit the XCD's L2) takes the windows through the
    // first real thresholds only and lane = tree takes over at tree 13 -- 2,048 waves with one window each hold a twentieth of the
    // pool's working set and keep the texture path busy where the pool waits for its dependent round trips (hand-over at
    // 6 / 13 / 28 / 48: the pool + tail launches take 2.48 / 2.65 / 2.79 / 3.09 ms per 128-frame batch, 6.05 / 6.29 / 6.76 / 7.28 ms
    // at the 4K config).
    int nh = std::min(env_int("PIGO_NH_BIG", p.max_frames >= 8 ? 13 : 4), std::min(nt, kTabTrees));
    {
        bool at_end = nh == nt;
        for (int st = 0; st < a.n_stages; ++st) at_end = at_end || a.st_end[st] + 1 == nh;
        if (!at_end) return PIGO_OK;
    }
    BigArgs &B = p.big;
    B = BigArgs{};
    // chunk stages: [0] [1] [2-3] where the cascade's first stages are single trees (as the small region group), else its own first stages
    int n_cs = 0;
    while (n_cs < a.n_stages && n_cs < 4 && a.st_end[n_cs] < std::max(1, env_int("PIGO_BIG_POOL_TREE", 4)) && a.st_end[n_cs] + 1 < nh) ++n_cs;
    if (n_cs < 1) return PIGO_OK;
    for (int i = 0; i < n_cs; ++i) B.cs_end[i] = a.st_end[i];
    if (env_int("PIGO_BIG_MERGE", 1) != 0 && n_cs == 4 && B.cs_end[0] == 0 && B.cs_end[1] == 1 && B.cs_end[2] == 2 && B.cs_end[3] == 3) {
        B.cs_end[2] = 3;
        B.cs_end[3] = 0;
        n_cs = 3;
    }
    B.n_cs = n_cs;
    B.t_pool = B.cs_end[n_cs - 1] + 1;
    B.nh = nh;
    for (int k = kbig; k < nscales; ++k) {
        const ScaleDesc &sd = p.scales[k];
        if (sd.s >= 65536 || sd.nc >= 65536 - kBigChunk) return PIGO_OK;  // (big_decode's 24-bit arithmetic; plan_build limits rows/cols to 65535 anyway)
        const long long nwin = (long long)sd.nr * sd.nc;
        for (long long f0 = 0; f0 < nwin; f0 += kBigChunk) p.big_items.push_back(make_uint2((unsigned)k, (unsigned)f0));
    }
    if (p.big_items.empty() || p.big_items.size() > (1u << 24)) return PIGO_OK;
    // (k_scan_big counts its work items -- chunks x the frames an XCD scans -- in 32 bits)
    if ((unsigned long long)p.big_items.size() * (unsigned long long)((p.max_frames + 7) / 8) >= (1ull << 32)) return PIGO_OK;
    B.cpf = (uint32_t)p.big_items.size();
    p.big_lds = (size_t)kBigWaves * (kBigChunk * 6);
    // k_big_pool's input queues, one per XCD: room for a quarter of the big-scale windows of the frames an XCD scans (8 % survive
    // the chunk stages on faces, 4 % on noise); an overflow raises the queue flag like every survivor queue
    {
        long long bigwin = 0;
        for (int k = kbig; k < nscales; ++k) bigwin += (long long)p.scales[k].nr * p.scales[k].nc;
        const long long per_xcd = (bigwin * ((p.max_frames + 7) / 8) + 3) / 4;
        p.big_midcap = (uint32_t)std::min<long long>(std::max<long long>(4096, per_xcd), 1LL << 28);
    }
    // the tail of the side chain: ONE k_tail_deep launch without an LDS code table (CT: codes from the node-major pair table in
    // global memory) over all the remaining trees; PIGO_BIG_CT=0: launches with LDS code windows of PIGO_BIG_DEEP_SPLIT trees
    p.big_ct = env_int("PIGO_BIG_CT", 1) != 0 && c.d_codes_t.p != nullptr;
    p.big_side_first = env_int("PIGO_BIG_FIRST", 0) != 0;
    p.big_skip = env_int("PIGO_BIG_SKIP", 0);
    p.big_chunk = std::max(0, env_int("PIGO_BIG_CHUNK_FRAMES", 128)) & ~7;
    if (nh < nt) {
        const int wmax = p.big_ct ? nt : std::max(64, std::min(600, env_int("PIGO_BIG_DEEP_SPLIT", 192)));
        for (int t = nh; t < nt; t += wmax) p.side_splits.push_back(t);
        p.side_splits.push_back(nt);
        if (p.side_splits.size() > 7) return PIGO_OK;  // (counters 8..15 of a queue set)
    }
    if (!p.big_ct) p.side_lds = 0;  // (LDS code windows do not fit next to a region workgroup)
    if (p.side_lds && p.big_lds + 1536 > p.side_lds) p.side_lds = 0;  // does not fit the reserve: runs, but not next to a region workgroup
    p.big_ok = true;
    return PIGO_OK;
}

// Plans of a few frames: the whole scan as one persistent launch (k_scan_one).  Needs the region groups (sized for it by
// build_region_groups: no deep lists, one item per CU and frame), the big rungs' chunk list (build_big) and the node-major code table.
void build_one(pigo_plan &p)
{
    p.one_ok = false;
    const ScanArgs &a = p.args;
    const pigo_cascade &c = *p.c;
    const int nscales = (int)p.scales.size();
    static const bool dbg = env_int("PIGO_SYNC_DEBUG", 0) != 0;
    auto why = [&](const char *reason) {
        if (dbg && p.max_frames < 8) fprintf(stderr, "[pigo] k_scan_one not used for this plan: %s\n", reason);
    };
    if (p.max_frames >= 8 || env_int("PIGO_ONE", 1) == 0) return;
    if (!p.region_ok || p.regions.empty() || p.regions.size() > 2) return why("no region groups (or more than two)");
    if (p.guard) return why("a frame shape whose rotated scan may panic (guard kernels)");
    if (!c.d_codes_t.p || c.ntrees > 511 || nscales > 2047) return why("cascade / ladder beyond the queue entry's fields");  // (tag A of a queue entry: 9 bits of tree, 11 of rung, 3 of frame)
    const int kbig = p.regions.back().args.k_hi;
    if (kbig < nscales && !p.big_ok) return why("rungs beyond the region groups without a chunk list");  // rungs beyond the groups that k_scan_big's chunk list does not cover
    OneArgs &o = p.one;
    o = OneArgs{};
    o.ngrp = (int)p.regions.size();
    o.grp[0] = p.regions[0].args;
    if (o.ngrp > 1) o.grp[1] = p.regions[1].args;
    size_t lds = 0;
    for (const pigo_plan::RegionGroup &g : p.regions) lds = std::max(lds, g.dyn_lds);
    // the big rungs: chunk stages [0] [1] [2-3] where the cascade's first stages are single trees, else its own leading stages;
    // lane = tree takes over right behind them
    p.one_big = BigArgs{};
    if (kbig < nscales) {
        BigArgs &B = p.one_big;
        B = p.big;
        // ONE chunk stage over the trees [0, 4) where a stage of the cascade ends at tree 3 (all four in flight per window: a
        // dependent chain of six round trips instead of 24), then [4, 13) five or nine at a time; lane = tree from 13.
        // (PIGO_ONE_BIG_CS = 3: the stages [0] [1] [2-3] of k_scan_big instead; PIGO_ONE_NH_BIG: the hand-over tree)
        int n_cs = 0;
        while (n_cs < a.n_stages && n_cs < 4 && a.st_end[n_cs] < 4) ++n_cs;
        if (n_cs < 1) return why("no single-tree stage in front of the cascade");
        for (int i = 0; i < 4; ++i) B.cs_end[i] = i < n_cs ? a.st_end[i] : 0;
        const int want_cs = env_int("PIGO_ONE_BIG_CS", 1);
        if (want_cs == 1) {
            B.cs_end[0] = B.cs_end[n_cs - 1];
            B.cs_end[1] = B.cs_end[2] = B.cs_end[3] = 0;
            n_cs = 1;
        } else if (n_cs == 4 && B.cs_end[0] == 0 && B.cs_end[1] == 1 && B.cs_end[2] == 2 && B.cs_end[3] == 3) {
            B.cs_end[2] = 3;
            B.cs_end[3] = 0;
            n_cs = 3;
        }
        B.n_cs = n_cs;
        B.t_pool = B.cs_end[n_cs - 1] + 1;
        int nh = std::min(env_int("PIGO_ONE_NH_BIG", 13), (int)c.ntrees);
        {
            bool at_end = nh == (int)c.ntrees;
            for (int st = 0; st < a.n_stages; ++st) at_end = at_end || a.st_end[st] + 1 == nh;
            if (!at_end || nh < B.t_pool) nh = B.t_pool;
        }
        B.nh = nh;
        B.one_pti = env_int("PIGO_ONE_PTI", 9);
        o.bigmix = std::max(0, std::min(2, env_int("PIGO_ONE_BIGMIX", 1)));
        o.regs_per_frame = (uint32_t)(o.grp[0].ncx * o.grp[0].ncy) + (o.ngrp > 1 ? (uint32_t)(o.grp[1].ncx * o.grp[1].ncy) : 0u);
        // (mixed in, a workgroup's k-th chunk goes to its k-th wave from the top: at most kOneBigWaves / 2 chunks per region item)
        if (o.bigmix && (uint64_t)o.regs_per_frame * (kOneBigWaves / 2) < B.cpf) o.bigmix = 0;
        for (int g = 0; g < o.ngrp; ++g)
            if (o.grp[g].wave_q < kBigChunk) o.bigmix = 0;  // (the chunk's windows live in the wave's own queue)
        o.nbig = o.bigmix ? 0u : (B.cpf + kOneBigWaves - 1) / kOneBigWaves;
        lds = std::max(lds, (size_t)kOneBigWaves * kBigChunk * 6);
    }
    p.one_lds = lds;
    o.nt = std::max(1, std::min(4, env_int("PIGO_ONE_NT", 1)));
    o.nt_late = std::max(1, std::min(4, env_int("PIGO_ONE_NT_LATE", 4)));
    o.late_items = std::max(0, env_int("PIGO_ONE_LATE_ITEMS", 4));  // (round 6: 4 and two local passes in the mid group -- 0.1013-0.1017 against 0.1030-0.1041 ms on a 1080p frame with faces, neutral elsewhere)
    o.plain_zero = env_int("PIGO_ONE_PLAINZERO", 0);
    // (a user switch, always honoured: how long a consumer of the one-launch scan waits for a claimed slot before the run is given up)
    o.wait_ticks = (long long)std::max(1, std::min(600000, env_int("PIGO_ONE_TIMEOUT_MS", 2000))) * 100000LL;
    o.restore = (p.det_cap <= kOneRestoreMax && (size_t)p.det_cap * 4 <= lds && env_int("PIGO_ONE_RESTORE", 1) != 0) ? 1 : 0;
    // queues: room for 1/32 of the windows of the plan's frames in each of the eight (what passes tree 4 of the big rungs, 13 / 28
    // of the groups: well below 1 % on faces and on noise); an overflow raises the queue flag like every survivor queue
    o.qcap = (uint32_t)std::min<long long>(std::max<long long>(2048, p.windows * p.max_frames / 32), 1LL << 24);
    if (const int forced = env_int("PIGO_ONE_QCAP", 0); forced > 0) o.qcap = (uint32_t)forced;  // (tests: a queue that overflows)
    p.one_ok = true;
    if (dbg) {
        fprintf(stderr, "[pigo] k_scan_one: %u big items (%u chunks; stages end %d, lane = tree from %d)", o.nbig, p.one_big.cpf, p.one_big.t_pool - 1, p.one_big.nh);
        for (int g = o.ngrp - 1; g >= 0; --g)
            fprintf(stderr, "; group %d: rungs [%d, %d) %d x %d cells of %d x %d px, region %d x %d, hand-over %d, deep list %d, quad %d", g, o.grp[g].k_lo, o.grp[g].k_hi,
                    o.grp[g].ncx, o.grp[g].ncy, o.grp[g].cell_w, o.grp[g].cell_h, o.grp[g].pitch, o.grp[g].rows, o.grp[g].nh, o.grp[g].deep_cap, o.grp[g].quad);
        fprintf(stderr, "; LDS %zu B\n", p.one_lds);
    }
}

// Environment switches.  A handful are for users and always honoured: PIGO_SCAN_VARIANT (force a scan implementation),
// PIGO_QUEUE_MIN (survivor-queue capacity), PIGO_GRAPH_FRAMES (graph replay of small batches / RunCascade slots),
// PIGO_ONE_TIMEOUT_MS (patience of the one-launch scan's hand-offs), PIGO_COMM_INIT_TIMEOUT_S, PIGO_RCCL_LIB, PIGO_SYNC_DEBUG, PIGO_DEBUG_STATS (debug build).  Everything else is a TUNING switch
// of the A/B scripts and the parity suite (schedule constants, forced code paths, timing experiments) and is ignored unless
// PIGO_TUNING=1 is set as well -- a production process cannot wander onto a path nobody benchmarks by inheriting a stray variable.
bool tuning_enabled()
{
    const char *t = getenv("PIGO_TUNING");
    return t && *t && atoi(t) != 0;
}

const char *tune_env(const char *name)
{
    return tuning_enabled() ? getenv(name) : nullptr;
}

int env_int(const char *name, int dflt)
{
    static const char *const user[] = {"PIGO_SCAN_VARIANT", "PIGO_QUEUE_MIN", "PIGO_GRAPH_FRAMES", "PIGO_COMM_INIT_TIMEOUT_S", "PIGO_SYNC_DEBUG", "PIGO_DEBUG_STATS",
                                       "PIGO_ONE_TIMEOUT_MS"};
    bool is_user = false;
    for (const char *u : user) is_user = is_user || strcmp(u, name) == 0;
    if (!is_user && !tuning_enabled()) return dflt;
    const char *v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

pigo_status plan_alloc_batch(pigo_plan &p, int max_frames, int det_cap)
{
    p.max_frames = max_frames;
    p.det_cap = det_cap;
    // survivor queue: room for 1/8 of a frame's windows (noise keeps 5.5 % after four trees); an overflow
    // is detected on the device and answered with the monolithic kernel (pigo_plan_run_sync).
    long long qcap = std::max<long long>(std::max(8, env_int("PIGO_QUEUE_MIN", 4096)), p.windows / env_int("PIGO_QUEUE_DIV", 8));
    qcap = std::min<long long>(qcap, std::max<long long>(p.windows, 1));
    qcap = (qcap + kTailChunk - 1) / kTailChunk * kTailChunk;
    p.qcap = qcap;
    HIP_TRY(p.d_queue.alloc((size_t)qcap * max_frames));
    HIP_TRY(p.d_qcount.alloc(std::max(max_frames, 64)));  // [0..7] per-XCD queues, [8] second-level queue; [16..24] the second set; [32..39] bucket lists
    p.qcap2 = std::max<long long>(4096, (qcap * (long long)max_frames) / 8);
    HIP_TRY(p.d_queue2.alloc((size_t)p.qcap2));
    HIP_TRY(p.d_raw.alloc((size_t)det_cap * max_frames));
    HIP_TRY(p.d_mq.alloc((size_t)det_cap * max_frames));
    HIP_TRY(p.d_ties.alloc(max_frames));
    // ClusterDetections workspaces, sized here so that pigo_plan_cluster never allocates (it may be called while the caller's
    // stream is being captured): the kernel family is fixed at plan creation -- PIGO_CLUSTER_V2 read once, else by det_cap
    {
        const int mode = env_int("PIGO_CLUSTER_V2", -1);
        // (seeds / members / compact for every plan since round 4: on the default batch -- ~130 detections per frame -- they take
        // 0.13 ms per 128 frames against k_cluster's 0.19, and the choice no longer hangs on det_cap instead of the lists' lengths)
        p.cluster_mode = mode >= 0 ? (mode != 0 ? 1 : 0) : 1;
        if (p.cluster_mode == 1) {
            const size_t need = (size_t)max_frames * det_cap;
            HIP_TRY(p.d_cl_seeds.alloc(need));
            HIP_TRY(p.d_cl_tmpn.alloc(need));
            HIP_TRY(p.d_cl_tmp.alloc(need));
            HIP_TRY(p.d_cl_nseeds.alloc(max_frames));
        }
        if (det_cap > kGoSortKeys) HIP_TRY(p.d_gosort_ws.alloc((size_t)max_frames * det_cap * 12));  // k_gosort_ties: keys + tie counts of lists beyond the LDS
    }
    return PIGO_OK;
}

pigo_status plan_build(pigo_cascade *c, const PlanKey &key, int max_frames, int det_cap, std::unique_ptr<pigo_plan> &out)
{
    if (!c) return fail(PIGO_ERR_PARAM, "cascade is NULL");
    if (key.rows < 1 || key.cols < 1 || key.rows >= 65536 || key.cols >= 65536 || key.dim >= 65536)
        return fail(PIGO_ERR_PARAM, "rows/cols/dim must be in [1, 65535]");
    if (key.dim < key.cols) return fail(PIGO_ERR_PARAM, "dim (%d) < cols (%d)", key.dim, key.cols);
    if ((long long)key.rows * key.dim >= 0x7fffffffLL) return fail(PIGO_ERR_PARAM, "frame larger than 2 GiB");
    if (key.min_size < 0) return fail(PIGO_ERR_PARAM, "min_size < 0 (the reference would index out of range)");
    if (!std::isfinite(key.shift) || !std::isfinite(key.scale) || !std::isfinite(key.angle))
        return fail(PIGO_ERR_PARAM, "non-finite shift/scale/angle");
    if (max_frames < 1 || det_cap < 1) return fail(PIGO_ERR_PARAM, "max_frames and det_cap must be >= 1");
    if (max_frames >= (1 << 21)) return fail(PIGO_ERR_PARAM, "max_frames must be below 2^21");
    if ((long long)max_frames * det_cap > (1LL << 31)) return fail(PIGO_ERR_PARAM, "max_frames * det_cap too large");

    std::unique_ptr<pigo_plan> p(new (std::nothrow) pigo_plan);
    if (!p) return fail(PIGO_ERR_NOMEM, "out of memory");
    p->c = c;
    p->key = key;
    p->max_frames = max_frames;  // (the region grid of variant 3 is sized for the plan's batch)
    p->rot = key.angle > 0.0;  // pigo.go:232
    if (p->rot) {
        const double a = key.angle > 1.0 ? 1.0 : key.angle;  // pigo.go:233-235
        p->angle_idx = (int)(32.0 * a);                      // pigo.go:159
        if (key.rows >= 32768 || key.cols >= 32768) return fail(PIGO_ERR_PARAM, "rotated scan supports images below 32768 pixels per side");
        // quirk Q1: columns are clamped with nrows-1, so the largest index is (rows-1)*dim + rows-1.  Go
        // panics when that leaves the pixel slice; the kernels then guard every load and raise the flag.
        p->guard = (long long)(key.rows - 1) * key.dim + (key.rows - 1) >= (long long)key.rows * key.dim;
    }
    pigo_status st = build_ladder(*p);
    if (st != PIGO_OK) return st;

    p->rot_lds = p->rot && !p->guard && key.dim % 4 == 0 && env_int("PIGO_ROT_LDS", 1) != 0 && env_int("PIGO_LDS_TILES", 1) != 0;
    p->tile_ok = build_tile_stages(*p);
    build_tile_classes(*p);  // also fills ScaleDesc::pitch / up

    HIP_TRY(hipSetDevice(c->device));
    // Everything a plan uploads or builds on the device goes through ONE private stream and is waited for with
    // hipStreamSynchronize: no device-wide call and nothing on the legacy null stream.  hipDeviceSynchronize() is refused by
    // the runtime ("operation not permitted when stream is capturing") while ANY thread of the process captures a stream --
    // and it invalidates that thread's capture on the way: that was round 2's abort, a RunCascade slot being captured
    // (PIGO_GRAPH_FRAMES) while another goroutine's call built its plan.
    struct BuildStream {
        hipStream_t s = nullptr;
        ~BuildStream()
        {
            if (s) {
                (void)hipStreamSynchronize(s);  // an early error return must not leave copies from host vectors in flight
                (void)hipStreamDestroy(s);
            }
        }
    } bs;
    HIP_TRY(hipStreamCreateWithFlags(&bs.s, hipStreamNonBlocking));
    const int nscales = (int)p->scales.size();
    HIP_TRY(p->d_scales.alloc(nscales));
    HIP_TRY(p->d_tiles.alloc(p->tiles.size()));
    HIP_TRY(p->d_flags.alloc(4));
    HIP_TRY(hipMemsetAsync(p->d_flags.p, 0, 16, bs.s));
    const size_t tab_n = (size_t)nscales * c->ntrees * c->nodes;
    HIP_TRY(p->d_tab.alloc(tab_n));
    if (nscales) {
        HIP_TRY(hipMemcpyAsync(p->d_scales.p, p->scales.data(), nscales * sizeof(ScaleDesc), hipMemcpyHostToDevice, bs.s));
        HIP_TRY(hipMemcpyAsync(p->d_tiles.p, p->tiles.data(), p->tiles.size() * 4, hipMemcpyHostToDevice, bs.s));
    }
    if (tab_n) {
        const int blocks = (int)std::min<size_t>((tab_n + 255) / 256, 4096);
        k_build_tab<<<blocks, 256, 0, bs.s>>>(c->d_codes.p, p->d_scales.p, p->d_tab.p, nscales, (int)c->ntrees, c->nodes, key.dim, p->rot ? 1 : 0,
                                              kQCos[p->angle_idx], kQSin[p->angle_idx]);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipStreamSynchronize(bs.s));
    if (p->tile_ok && nscales) {
        HIP_TRY(p->d_tiles2.alloc(p->tiles2.size()));
        HIP_TRY(hipMemcpyAsync(p->d_tiles2.p, p->tiles2.data(), p->tiles2.size() * sizeof(uint2), hipMemcpyHostToDevice, bs.s));
        HIP_TRY(p->d_tabp.alloc((size_t)nscales * c->ntrees * 64));
        const size_t n = (size_t)nscales * c->ntrees * 64;
        k_build_tabp<<<(int)std::min<size_t>((n + 255) / 256, 4096), 256, 0, bs.s>>>(c->d_codes.p, p->d_scales.p, p->d_tabp.p, nscales, (int)c->ntrees, p->rot ? 1 : 0,
                                                                                            kQCos[p->angle_idx], kQSin[p->angle_idx]);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(bs.s));
        const int max_dyn = (160 << 10) - 1024;
        HIP_TRY(hipFuncSetAttribute((const void *)k_scan_tile<false, false, true, 256, true>, hipFuncAttributeMaxDynamicSharedMemorySize, max_dyn));
        HIP_TRY(hipFuncSetAttribute((const void *)k_scan_tile<false, false, false, 256, true>, hipFuncAttributeMaxDynamicSharedMemorySize, max_dyn));
        HIP_TRY(hipFuncSetAttribute((const void *)k_scan_tile<true, false, false, 256, true>, hipFuncAttributeMaxDynamicSharedMemorySize, max_dyn));
        HIP_TRY(hipFuncSetAttribute((const void *)k_scan_tile<true, false, true, 256, true>, hipFuncAttributeMaxDynamicSharedMemorySize, max_dyn));
        HIP_TRY(hipFuncSetAttribute((const void *)k_scan_tile<true, true, false, 256, true>, hipFuncAttributeMaxDynamicSharedMemorySize, max_dyn));
        p->side_mode = env_int("PIGO_SIDE_STREAM", 1);
        // (a batch plan's side stream is a high-priority one, and a fork / join pair across priority levels costs ~0.24 ms: small
        // batches on such a plan stay on the caller's stream)
        p->fork_min_frames = std::max(1, env_int("PIGO_FORK_MIN_FRAMES", max_frames >= 8 ? 8 : 1));
        p->small_ct = max_frames < 8 && c->d_codes_t.p != nullptr && env_int("PIGO_SMALL_CT", 1) != 0;
        p->split_tail = max_frames < 8 && env_int("PIGO_SPLIT_TAIL", 1) != 0;
        p->pipe_chunks = std::max(0, std::min(16, env_int("PIGO_PIPE_CHUNKS", 0)));  // 0 = automatic: about 32 frames per chunk
        // (streams are only created where the plan can use them: a process's HIP streams share a handful of hardware queues, and two
        // of a plan's streams landing on ONE queue silently serialises what was forked -- the one-frame leg of bench.py ran 0.21
        // instead of 0.17 ms next to a batch plan that held six streams)
        if (p->pipe_chunks != 1 && max_frames >= 16) {
            HIP_TRY(hipStreamCreateWithFlags(&p->tail_stream, hipStreamNonBlocking));  // (a high-priority stream measured no different)
            for (int i = 0; i < 2; ++i) {
                HIP_TRY(hipEventCreateWithFlags(&p->ev_tiles[i], hipEventDisableTiming));
                HIP_TRY(hipEventCreateWithFlags(&p->ev_tail[i], hipEventDisableTiming));
            }
        }
        if (p->side_mode) {
            // The side stream is a HIGH-PRIORITY stream: the runtime keeps its hardware queues per priority level and hands a new stream
            // the least used queue of its level, so with a handful of normal-priority streams alive in the process (other plans, the
            // run slots of pigo_run_cascade, the host's own) a normal-priority side stream can land on the queue of the very stream
            // it is forked from -- and the side chain then runs BEHIND the region launches instead of next to them, silently (bench.py's
            // 1,024-frame leg, created after three other plans: 65 ms per step instead of 45).  A queue of the other level cannot
            // be the caller's.  Batch plans only: with the side stream of a ONE-frame plan on the other level a call took 0.38 instead of
            // 0.14 ms (its two fork / join pairs per call cross priority levels, and that is slow) -- there a collision costs 20 %, here
            // it costs 2.5 x.  (PIGO_SIDE_PRIO=0 / 1: a normal- / high-priority stream whatever the plan, for the A/B.)
            int prio_least = 0, prio_greatest = 0;
            HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
            if (env_int("PIGO_SIDE_PRIO", max_frames >= 8 ? 1 : 0) != 0 && prio_greatest != prio_least)
                HIP_TRY(hipStreamCreateWithPriority(&p->side, hipStreamNonBlocking, prio_greatest));
            else
                HIP_TRY(hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming));
        }
        if (max_frames < 8 && env_int("PIGO_SCAN_VARIANT", -1) == 3) {  // variant 3 forced on a small plan: its region groups side by side
            HIP_TRY(hipStreamCreateWithFlags(&p->grp_stream, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&p->ev_gfork, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&p->ev_gjoin, hipEventDisableTiming));
        }
        // graph replay of small batches: off by default -- on ROCm 7.2 a replayed graph of this sequence is no faster than the
        // eager launches (profiles/r02_experiments.md); PIGO_GRAPH_FRAMES=n turns it on for batches of up to n frames
        p->graph_max_frames = std::max(0, env_int("PIGO_GRAPH_FRAMES", 0));
        if (p->graph_max_frames > 0) HIP_TRY(hipStreamCreateWithFlags(&p->cap_stream, hipStreamNonBlocking));
        p->region_ok = build_region_groups(*p);
        {
            // k_tail_deep's code window starts at the earliest hand-over tree that really occurs: the tile classes' and, with region
            // groups, each group's (build_tile_stages assumed the mid group's early hand-over, which dense ladders and plans without
            // region groups do not use -- the window then held trees no queue entry can start at)
            int lo = std::min(p->args.nh_lds, p->args.nh_glb);
            if (p->region_ok)
                for (const pigo_plan::RegionGroup &g : p->regions) lo = std::min(lo, g.args.nh);
            if (lo > p->args.deep_lo && !set_deep_window(*p, lo)) return fail(PIGO_ERR_PARAM, "tree codes do not fit the LDS");
        }
        if (p->region_ok) {
            // the region groups' offset tables: k_build_tabp with every rung's pitch set to its group's region pitch
            std::vector<ScaleDesc> sreg(p->scales);
            for (ScaleDesc &sd : sreg) sd.pitch = 0;
            for (const pigo_plan::RegionGroup &g : p->regions)
                for (int j = g.args.k_lo; j < g.args.k_hi; ++j) sreg[j].pitch = g.args.pitch;
            DevBuf<ScaleDesc> d_sreg;
            HIP_TRY(d_sreg.alloc(nscales));
            HIP_TRY(hipMemcpyAsync(d_sreg.p, sreg.data(), nscales * sizeof(ScaleDesc), hipMemcpyHostToDevice, bs.s));
            HIP_TRY(p->d_tabr.alloc(n));
            k_build_tabp<<<(int)std::min<size_t>((n + 255) / 256, 4096), 256, 0, bs.s>>>(c->d_codes.p, d_sreg.p, p->d_tabr.p, nscales, (int)c->ntrees, p->rot ? 1 : 0,
                                                                                                kQCos[p->angle_idx], kQSin[p->angle_idx]);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(bs.s));  // (before sreg / d_sreg go out of scope)
            HIP_TRY(hipFuncSetAttribute((const void *)k_scan_region<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (160 << 10) - 3072));
            HIP_TRY(hipFuncSetAttribute((const void *)k_scan_region<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (160 << 10) - 3072));
            st = build_big(*p);
            if (st != PIGO_OK) return st;
            if (p->big_ok) {
                HIP_TRY(p->d_big_items.alloc(p->big_items.size()));
                HIP_TRY(hipMemcpyAsync(p->d_big_items.p, p->big_items.data(), p->big_items.size() * sizeof(uint2), hipMemcpyHostToDevice, bs.s));
                HIP_TRY(hipStreamSynchronize(bs.s));
                p->big.items = p->d_big_items.p;
                HIP_TRY(p->d_big_midq.alloc((size_t)8 * p->big_midcap));
                p->big.midq = p->d_big_midq.p;
                p->big.midcap = p->big_midcap;
                HIP_TRY(hipFuncSetAttribute((const void *)k_scan_big<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->big_lds));
                HIP_TRY(hipFuncSetAttribute((const void *)k_scan_big<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->big_lds));
            }
        }
        build_one(*p);
        if (p->one_ok) {
            HIP_TRY(p->d_oneq.alloc((size_t)8 * p->one.qcap));
            HIP_TRY(p->d_onecnt.alloc(kOneCntWords));
            HIP_TRY(hipMemsetAsync(p->d_onecnt.p, 0, (size_t)kOneCntWords * 4, bs.s));
            HIP_TRY(hipMemsetAsync(p->d_oneq.p, 0, (size_t)8 * p->one.qcap * sizeof(uint4), bs.s));  // tags: zero = empty
            HIP_TRY(hipStreamSynchronize(bs.s));
            p->one.q = p->d_oneq.p;
            HIP_TRY(hipFuncSetAttribute((const void *)k_scan_one<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (160 << 10) - 3072));
            HIP_TRY(hipFuncSetAttribute((const void *)k_scan_one<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (160 << 10) - 3072));
        }
        HIP_TRY(hipFuncSetAttribute((const void *)k_tail_deep<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, max_dyn));
        HIP_TRY(hipFuncSetAttribute((const void *)k_tail_deep<false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, max_dyn));
        HIP_TRY(hipFuncSetAttribute((const void *)k_tail_deep<true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, max_dyn));
        HIP_TRY(hipFuncSetAttribute((const void *)k_tail_deep<true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, max_dyn));
    }
    st = plan_alloc_batch(*p, max_frames, det_cap);
    if (st != PIGO_OK) return st;
    ScanArgs &a = p->args;
    a.scales = p->d_scales.p;
    a.tiles = p->d_tiles.p;
    a.tab = p->d_tab.p;
    a.tiles2 = p->d_tiles2.p;
    a.tabp = p->d_tabp.p;
    a.tabr = p->d_tabr.p;
    a.codes = c->d_codes.p;
    a.codes_t = c->d_codes_t.p;
    a.pass_end = c->d_pass_end.p + (env_int("PIGO_DEEP_PASS", p->max_frames >= 8 ? 1 : 0) != 0 ? 0 : (size_t)c->ntrees);
    a.seg_end = c->d_pass_end.p + (size_t)2 * c->ntrees;
    a.late_waves = std::max(1, std::min(kLateWaves, env_int("PIGO_LATE_WAVES", kLateWaves)));
    a.qb_div = 2;  // per class, see build_tile_classes
#ifdef PIGO_DEBUG_BUILD
    a.stats = nullptr;
    if (env_int("PIGO_DEBUG_STATS", 0)) {
        HIP_TRY(p->d_stats.alloc(65536 * 8 + 16 * 256 + 256 * 8));
        HIP_TRY(hipMemsetAsync(p->d_stats.p, 0, (65536 * 8 + 16 * 256 + 256 * 8) * 8, bs.s));
        HIP_TRY(hipStreamSynchronize(bs.s));
        a.stats = p->d_stats.p;
    }
#endif
    a.qcos = kQCos[p->angle_idx];
    a.qsin = kQSin[p->angle_idx];
    a.leaf = c->d_leaf.p;
    a.thr = c->d_thr.p;
    a.queue = p->d_queue.p;
    a.qcount = p->d_qcount.p;
    a.raw = p->d_raw.p;
    a.flags = p->d_flags.p;
    a.npix = (uint32_t)((long long)key.rows * key.dim);
    a.dim = key.dim;
    a.lim = key.rows - 1;
    a.ntiles = (int)p->tiles.size();
    a.ntrees = (int)c->ntrees;
    a.nodes = c->nodes;
    a.depth = (int)c->depth;
    a.qcap = (uint32_t)p->qcap;
    a.det_cap = det_cap;
    // variant 2 (LDS tiles, whole cascade per tile) for depth-6 cascades, variant 0 (monolithic lane-per-window kernel) for
    // everything else and as the overflow answer; variant 1 (the first head + tail design) exists in the debug build only
    // variant 3 (LDS regions) for batches; a plan for a handful of frames cannot fill 256 CUs with 1024-thread region workgroups
    // and is served faster by the tile kernel (measured: one 1080p frame 0.20 ms vs 0.29 ms, profiles/r02_experiments.md)
    // ... unless the whole scan fits ONE persistent launch (k_scan_one: variant 3 of such a plan)
    p->variant = (c->depth == 6 && c->ntrees > 0 && p->tile_ok) ? env_int("PIGO_SCAN_VARIANT", (p->region_ok && (max_frames >= 8 || p->one_ok)) ? 3 : 2) : 0;
    if (p->variant == 3 && !p->region_ok) p->variant = 2;
    if (p->variant == 2 && !p->tile_ok) p->variant = 0;
#ifdef PIGO_DEBUG_BUILD
    build_stages(*p, env_int("PIGO_HEAD_STAGES", kMaxHeadStages));
    if (p->variant == 1 && !(c->depth == 6 && c->ntrees > 0)) p->variant = 0;
#else
    if (p->variant == 1) p->variant = 0;
#endif
    if (p->variant < 0 || p->variant > 3) p->variant = 0;
    HIP_TRY(hipStreamSynchronize(bs.s));
    out = std::move(p);
    return PIGO_OK;
}

// variant 2, first half of a (chunk of a) batch: one k_scan_tile launch per tile class; `xcd_cap` = entries per XCD queue
template <bool ROT, bool GUARD, class Mark>
void launch_tiles(const pigo_plan &p, const ScanArgs &a, uint32_t xcd_cap, hipStream_t s, Mark &mark, bool v3 = false, int what = 7)
{
    // variant 3: the region kernel scans the rungs of its scale groups, k_scan_tile only what lies beyond them
    // (what & 1: the region launches, what & 2: the LDS-pixel tile classes, what & 4: the global-gather tile classes)
#ifdef PIGO_DEBUG_BUILD
    if (v3 && env_int("PIGO_REG_ONLY", -1) >= 0) what &= 1;
#endif
    if (v3 && (what & 1)) {
        // a small batch cannot fill the chip with one group's regions: the groups then run next to each other
        const bool par = p.grp_stream && !p.profiling && a.nframes < 8 && p.regions.size() > 1;
        for (const pigo_plan::RegionGroup &g : p.regions) {
            const bool first = &g == &p.regions.front();
#ifdef PIGO_DEBUG_BUILD
            {  // per-group phase timers: PIGO_REG_ONLY=g launches that group alone (and no tile class) -- results are incomplete
                static const int only = env_int("PIGO_REG_ONLY", -1);
                if (only >= 0 && &g != &p.regions[(size_t)std::min<int>(only, (int)p.regions.size() - 1)]) continue;
            }
#endif
            hipStream_t gs = (par && !first) ? p.grp_stream : s;
            if (par && first) {  // fork BEFORE the first group's launch, so that the others really run next to it
                (void)hipEventRecord(p.ev_gfork, s);
                (void)hipStreamWaitEvent(p.grp_stream, p.ev_gfork, 0);
            }
            ScanArgs ra = a;
            ra.qcap = xcd_cap;
            ra.reg = g.args;
            mark(first ? "scan_region_small" : &g == &p.regions[1] ? "scan_region_mid" : "scan_region_big");
            k_scan_region<ROT><<<(uint32_t)a.nframes * (uint32_t)(g.args.ncx * g.args.ncy), kRegThreads, g.dyn_lds, gs>>>(ra);
        }
        if (par) {  // one join after the last group
            (void)hipEventRecord(p.ev_gjoin, p.grp_stream);
            (void)hipStreamWaitEvent(s, p.ev_gjoin, 0);
        }
    }
    if (!(what & 6)) return;
    // fork: classes that gather from global memory go to the side stream when the plan also has LDS-tile classes
    bool has_lds = false, has_glb = false;
    for (const pigo_plan::TileClass &cls : p.classes)
        if ((what & (cls.lds ? 2 : 4)) && cls.ntiles - (v3 ? cls.v3_skip : 0u)) (cls.lds ? has_lds : has_glb) = true;
    // (a small batch is launch-bound: every fork / join costs more than the overlap buys -- one stream then)
    const bool fork = p.side && has_lds && has_glb && !p.profiling && !p.no_fork && a.nframes >= p.fork_min_frames;
    if (fork) {
        (void)hipEventRecord(p.ev_fork, s);
        (void)hipStreamWaitEvent(p.side, p.ev_fork, 0);
    }
    for (const pigo_plan::TileClass &cls : p.classes) {
        const uint32_t skip = v3 ? cls.v3_skip : 0u;
        if (cls.ntiles == skip || !(what & (cls.lds ? 2 : 4))) continue;
        hipStream_t cs = (fork && !cls.lds) ? p.side : s;
        ScanArgs ca = a;
        ca.qcap = xcd_cap;
        ca.cls_tile0 = cls.tile0 + skip;
        ca.cls_ntiles = cls.ntiles - skip;
        ca.tw_log2 = cls.tw_log2;
        ca.th = cls.th;
        ca.nwin = cls.nwin;
        ca.tab_trees = cls.lds ? p.tab_lds : p.tab_glb;
        ca.qb_div = cls.qb_div;
        const uint32_t grid = (uint32_t)a.nframes * (cls.ntiles - skip);
        mark(cls.lds ? "scan_tile_lds" : "scan_tile_glb");
        if constexpr (!ROT) {
            if (cls.lds) {
                k_scan_tile<false, false, true, 256, true><<<grid, 256, cls.dyn_lds, cs>>>(ca);
            } else {
                k_scan_tile<false, false, false, 256, true><<<grid, 256, cls.dyn_lds, cs>>>(ca);
            }
        } else {
            if (cls.lds)
                k_scan_tile<true, false, true, 256, true><<<grid, 256, cls.dyn_lds, cs>>>(ca);
            else
                k_scan_tile<true, GUARD, false, 256, true><<<grid, 256, cls.dyn_lds, cs>>>(ca);
        }
    }
    if (fork) {
        (void)hipEventRecord(p.ev_join, p.side);
        (void)hipStreamWaitEvent(s, p.ev_join, 0);
    }
}

// variant 2, second half: the two k_tail_deep launches over the queues `a.queue` / `a.qcount` (8 per-XCD queues of xcd_cap
// entries) with `queue2` (cap2 entries, counter a.qcount[8]) in between
// `patch` = false: every entry of the queues comes from a scale above kPatchMaxS (see k_tail_deep)
template <bool ROT, bool GUARD, class Mark>
void launch_tail(const pigo_plan &p, const ScanArgs &a, uint32_t xcd_cap, QEntry *queue2, uint32_t cap2, hipStream_t s, Mark &mark, bool patch = !ROT)
{
    static_assert(!(GUARD && !ROT), "only rotated scans are guarded");
    const size_t patch_lds = (size_t)kDeepWaves * kPatchBytes;
    const bool nop = ROT || !patch;
    const size_t lds1 = p.deep_lds - (nop ? patch_lds : 0), lds2 = p.deep_lds2 - (nop ? patch_lds : 0);
    // workgroups per CU the grid is sized for.  The patch variant's 130 VGPRs leave room for one resident workgroup (12 waves)
    // and a grid of two per CU has measured best since round 1; rotated scans (65 VGPRs) gain from two resident ones, the
    // upright no-patch launches (windows of the scales above the region groups only: few, large footprints) lose with two (58.0 vs 56.4 Gwindows/s).
    const int cu_max = nop ? env_int("PIGO_TAIL_PER_CU", ROT ? 2 : 1) : 2;
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(cu_max, (size_t)(158 << 10) / std::max<size_t>(lds1, 1)));
    const int per_cu2 = (int)std::max<size_t>(1, std::min<size_t>(nop ? cu_max : 1, (size_t)(158 << 10) / std::max<size_t>(lds2, 1)));
    if (a.deep_lo >= a.ntrees) return;
    if (p.small_ct) {
        // plans of a few frames: ONE launch over all the remaining trees, codes from the node-major pair table in global memory --
        // no 50 KiB code copy per workgroup, no second launch behind a queue round trip (the call's latency is what counts there)
        mark("tail_deep");
        ScanArgs ta = a;
        ta.qcap = xcd_cap;
        ta.nqueues = 8;
        ta.deep_hi = a.ntrees;
        ta.queue2 = nullptr;
        ta.qcount2 = nullptr;
        ta.qcap2 = 0;
        const int per = std::max(1, env_int("PIGO_SMALL_CT_PER_CU", 2));
        if constexpr (ROT)
            k_tail_deep<true, GUARD, false, true><<<256 * per, kDeepThreads, 0, s>>>(ta);
        else
            k_tail_deep<false, false, false, true><<<256 * per, kDeepThreads, 0, s>>>(ta);
        (void)queue2;
        (void)cap2;
        return;
    }
    mark("tail_deep");
    ScanArgs ta = a;
    ta.qcap = xcd_cap;
    ta.nqueues = 8;
    ta.deep_hi = p.deep_mid;
    ta.queue2 = queue2;
    ta.qcount2 = a.qcount + 8;
    ta.qcap2 = cap2;
    if constexpr (ROT) {
        k_tail_deep<true, GUARD, false><<<256 * per_cu, kDeepThreads, lds1, s>>>(ta);
    } else {
        if (patch) k_tail_deep<false, false, true><<<256 * per_cu, kDeepThreads, lds1, s>>>(ta);
        else k_tail_deep<false, false, false><<<256 * per_cu, kDeepThreads, lds1, s>>>(ta);
    }
    if (p.deep_mid < a.ntrees) {
        mark("tail_deep2");
        ScanArgs tb = a;
        tb.queue = queue2;
        tb.nqueues = 1;
        tb.qcount = a.qcount + 8;
        tb.qcap = cap2;
        tb.deep_lo = p.deep_mid;
        tb.deep_hi = a.ntrees;
        tb.queue2 = nullptr;
        tb.qcount2 = nullptr;
        tb.qcap2 = 0;
        if constexpr (ROT) {
            k_tail_deep<true, GUARD, false><<<256 * per_cu2, kDeepThreads, lds2, s>>>(tb);
        } else {
            if (patch) k_tail_deep<false, false, true><<<256, kDeepThreads, lds2, s>>>(tb);
            else k_tail_deep<false, false, false><<<256 * per_cu2, kDeepThreads, lds2, s>>>(tb);
        }
    }
}

// variant 3, the rungs beyond the region groups: k_scan_big fills the 8 per-XCD queues of `a`, then one k_tail_deep launch per
// code window of p.side_splits drains them -- each hands what is still alive to the next through the two halves of `queue2`
// (counters a.qcount[8 + i]).  With an LDS reserve in the region groups (p.side_lds) every launch is ONE workgroup of 8 waves
// per CU, <= 64 VGPRs and <= p.side_lds of LDS: it fits next to a resident k_scan_region workgroup (16 waves x 96 VGPRs).
template <bool ROT, bool GUARD, class Mark>
void launch_big(const pigo_plan &p, const ScanArgs &a, uint32_t xcd_cap, QEntry *queue2, uint32_t cap2, hipStream_t s, Mark &mark)
{
    ScanArgs ba = a;
    ba.qcap = xcd_cap;
    ba.big = p.big;
    ba.big.next = a.qcount + 40;
    ba.big.midcount = a.qcount + 48;
    ba.big.midclaim = a.qcount + 56;
    const int per_cu = std::max(1, env_int("PIGO_BIG_PER_CU", 1));
    const int skip = p.big_skip;
    mark("scan_big");
    if (!(skip & 2)) k_scan_big<ROT><<<256 * per_cu, kBigThreads, p.big_lds, s>>>(ba);
    mark("big_pool");
    // (two waves per CU: what an XCD's pool waves hold at any time -- 32 CUs x 2 x 64 windows -- is then about one frame's
    // survivors, the frame the gathers find in the XCD's L2; eight waves per CU hold five frames' worth and miss)
    const int pool_waves = std::max(1, std::min(kBigWaves, env_int("PIGO_BIG_POOL_WAVES", 1)));
    if (!(skip & 2)) k_big_pool<ROT><<<256 * std::max(1, env_int("PIGO_BIG_POOL_PER_CU", 1)), 64 * std::min(pool_waves, kBigPoolWaves), 0, s>>>(ba);
    static const char *names[] = {"tail_deep", "tail_deep2", "tail_deep3", "tail_deep4", "tail_deep5", "tail_deep6"};
    const int nl = (int)p.side_splits.size() - 1;
    const uint32_t capq = cap2 / 2;
    // (eight waves: with a region workgroup's 16 x 96 VGPRs on the CU two 64-register waves per SIMD are what is left; fewer waves
    // make the chain gentler on the region kernel's copy and deep-list phases and slower -- PIGO_BIG_TAIL_THREADS, r04_experiments.md section 17)
    const int threads = p.side_lds ? std::max(64, std::min(512, env_int("PIGO_BIG_TAIL_THREADS", 512) & ~63)) : kDeepThreads;
    const int tail_per_cu = std::max(1, env_int("PIGO_BIG_TAIL_PER_CU", 1));
    for (int i = 0; i < nl; ++i) {
        ScanArgs ta = a;
        ta.deep_lo = p.side_splits[i];
        ta.deep_hi = p.side_splits[i + 1];
        if (i == 0) {
            ta.qcap = xcd_cap;
            ta.nqueues = 8;
        } else {
            ta.queue = queue2 + (size_t)((i - 1) & 1) * capq;
            ta.qcount = a.qcount + 8 + (i - 1);
            ta.qcap = capq;
            ta.nqueues = 1;
        }
        const bool last = i == nl - 1;
        ta.queue2 = last ? nullptr : queue2 + (size_t)(i & 1) * capq;
        ta.qcount2 = last ? nullptr : a.qcount + 8 + i;
        ta.qcap2 = last ? 0 : capq;
        const size_t lds = p.big_ct ? 0 : (size_t)(ta.deep_hi - ta.deep_lo) * kCodeStride * 4;
        mark(names[std::min(i, 5)]);
        if (skip & 1) continue;
        if constexpr (ROT) {
            if (p.big_ct) k_tail_deep<true, GUARD, false, true><<<256 * tail_per_cu, threads, lds, s>>>(ta);
            else k_tail_deep<true, GUARD, false><<<256 * tail_per_cu, threads, lds, s>>>(ta);
        } else {
            if (p.big_ct) k_tail_deep<false, false, false, true><<<256 * tail_per_cu, threads, lds, s>>>(ta);
            else k_tail_deep<false, false, false><<<256 * tail_per_cu, threads, lds, s>>>(ta);
        }
    }
}

template <bool ROT, bool GUARD, class Mark>
void launch_scan(const pigo_plan &p, const ScanArgs &a, int variant, hipStream_t s, Mark &mark)
{
    const uint32_t nb = (uint32_t)a.nframes * (uint32_t)a.ntiles;
    if (variant == 3 && p.one_ok && !GUARD) {
        // plans of a few frames: ONE persistent launch, at most one workgroup per CU (k_scan_one)
        OneArgs o = p.one;
        const uint32_t nf = (uint32_t)a.nframes;
        o.item_grp1 = o.nbig * nf;
        o.item_grp0 = o.item_grp1 + (o.ngrp > 1 ? (uint32_t)(o.grp[1].ncx * o.grp[1].ncy) * nf : 0u);
        o.nitems = o.item_grp0 + (uint32_t)(o.grp[0].ncx * o.grp[0].ncy) * nf;
        o.cnt = p.d_onecnt.p;
        ScanArgs oa = a;
        oa.big = p.one_big;
        if (o.restore) {
            // the launch counts its detections itself, its last workgroup restores the reference's order, writes the caller's counts
            // and leaves the counters zeroed: ONE node on the stream per call (plan_run_variant skips its memsets and k_restore_order)
            o.dets = p.one_dets;
            o.counts = a.counts;
            o.host_flags = p.one_host_flags;
            oa.counts = reinterpret_cast<int32_t *>(p.d_onecnt.p + 26 * kOneLine);
        } else {
            (void)hipMemsetAsync(p.d_onecnt.p, 0, (size_t)kOneCntWords * 4, s);
        }
        mark("scan_one");
        // At least eight workgroups: producers spread their survivors over all eight queues (one_push) and a workgroup only drains
        // queue blockIdx & 7 (one_consume) -- a small frame has fewer than eight items (30 x 40 px: ~6 regions), and with a grid of
        // nitems nobody would take what lands in the queues beyond it.  A workgroup that finds no item consumes at once.
        k_scan_one<ROT><<<std::max<uint32_t>(8u, std::min<uint32_t>(o.nitems, (uint32_t)p.one_grid)), kRegThreads, p.one_lds, s>>>(oa, o);
    } else if (variant == 2 || variant == 3) {
        const bool v3 = variant == 3;
        const long long qtotal = p.qcap * (long long)p.max_frames;  // entries of d_queue
        const int want_chunks = p.pipe_chunks > 0 ? p.pipe_chunks : std::max(2, std::min(8, (a.nframes + 31) / 32));
        const int chunks = (p.tail_stream && !p.profiling && a.nframes >= 16 && a.deep_lo < a.ntrees && !v3) ? want_chunks : 1;
        if (v3 && a.deep_lo < a.ntrees) {
            // Variant 3: the region launches (LDS-bound, they keep every window of their scales to themselves) on `s`, the tile
            // classes of the big scales and their deep tail (vector-memory / latency bound) next to them on the side stream.
            // Two queue sets: A for the tile classes, B for the (rare) spill of the regions' deep lists.
            const long long half = qtotal / 2, half2 = p.qcap2 / 2;
            const uint32_t xcd_cap = (uint32_t)std::min<long long>(half / 8, 0xffffffffLL);
            // (per-kernel timing runs the same launches one after the other on `s`)
            const bool fork = p.side && !p.profiling && !p.no_fork;
            hipStream_t sa = fork ? p.side : s;
            if (fork) {
                (void)hipEventRecord(p.ev_fork, s);
                (void)hipStreamWaitEvent(p.side, p.ev_fork, 0);
            }
            ScanArgs aa = a, ab = a;
            aa.queue = p.d_queue.p;
            aa.qcount = p.d_qcount.p;
            ab.queue = p.d_queue.p + half;
            ab.qcount = p.d_qcount.p + 16;
            // which launch goes to the hardware first: with the region workgroups resident everywhere (one per CU, the reserve free)
            // a side kernel's 256 workgroups land one per CU; launched into an empty chip the dispatcher may stack them
            const bool reg_early = !p.big_side_first && p.big_ok;
            if (reg_early) launch_tiles<ROT, GUARD>(p, ab, xcd_cap, s, mark, true, 1);
            if (p.big_ok) {
                // The chain walks the batch in chunks of big_chunk frames (128; a multiple of 8: the XCD dealing), its three launches
                // per chunk: an XCD's k_tail_deep queue then holds at most 16 frames' entries instead of nframes / 8.  The tail's waves
                // take their entries with a static stride and drift apart over the queue -- one wave sits on a face window for a dozen
                // passes while its neighbours finish ten entries --, i.e. over the FRAMES it holds in order: with 64 frames per queue
                // (a 512-frame step) its L2 hit rate was 0.69 and the chain's fabric traffic 90 MB per frame, with 8 frames 0.91 and
                // 27 MB (profiles/r04_traffic.json, r05_experiments.md section 4).  Timing: the chain is hidden behind the region
                // launches either way (1,024 frames: tail 17.1 -> 14.9 ms, k_scan_big 4.1 -> 5.6 -- a persistent kernel's ramp per
                // chunk --, step 41.4 / 41.7 ms); chunks of 64 / 32 frames cost the step 1-3 % / 10 %.  Entries handed out in queue
                // order from a counter instead (one atomic per entry on the wave's critical path): tail 1.83 -> 3.00 ms, removed.
                const int per = p.big_chunk > 0 ? p.big_chunk : a.nframes;
                for (int f0 = 0; f0 < a.nframes; f0 += per) {
                    ScanArgs ac = aa;
                    ac.nframes = std::min(per, a.nframes - f0);
                    ac.frames = a.frames + (size_t)f0 * a.frame_stride;
                    ac.counts = a.counts + f0;
                    ac.raw = a.raw + (size_t)f0 * a.det_cap;
                    if (f0 > 0) {  // the chain's counters: queues [0, 16), k_scan_big / k_big_pool's [40, 64)
                        (void)hipMemsetAsync(ac.qcount, 0, 16 * sizeof(uint32_t), sa);
                        (void)hipMemsetAsync(ac.qcount + 40, 0, 24 * sizeof(uint32_t), sa);
                    }
                    launch_big<ROT, GUARD>(p, ac, xcd_cap, p.d_queue2.p, (uint32_t)half2, sa, mark);
                }
            } else {
                launch_tiles<ROT, GUARD>(p, aa, xcd_cap, sa, mark, true, 6);
                launch_tail<ROT, GUARD>(p, aa, xcd_cap, p.d_queue2.p, (uint32_t)half2, sa, mark, p.tile_patch);
            }
            if (fork) (void)hipEventRecord(p.ev_join, p.side);
            if (!reg_early) launch_tiles<ROT, GUARD>(p, ab, xcd_cap, s, mark, true, 1);
            launch_tail<ROT, GUARD>(p, ab, xcd_cap, p.d_queue2.p + half2, (uint32_t)half2, s, mark);
            if (fork) (void)hipStreamWaitEvent(s, p.ev_join, 0);
        } else if (chunks <= 1 && !v3 && p.split_tail && p.side && !p.profiling && !p.no_fork && a.nframes >= p.fork_min_frames && a.deep_lo < a.ntrees) {
            // Plans of a few frames (the drop-in single-frame call): two chains side by side, each with its own queue set and its
            // own tail -- the global-gather class and its tail on the side stream, the LDS classes and theirs on `s`.  The long
            // entries (face windows: seven dependent passes) come from the big scales of the global class; their tail now runs
            // while the LDS classes are still scanning instead of behind the last of them.
            const long long half = qtotal / 2, half2 = p.qcap2 / 2;
            const uint32_t xcd_cap = (uint32_t)std::min<long long>(half / 8, 0xffffffffLL);
            ScanArgs aa = a, ab = a;
            aa.queue = p.d_queue.p;
            aa.qcount = p.d_qcount.p;
            ab.queue = p.d_queue.p + half;
            ab.qcount = p.d_qcount.p + 16;
            (void)hipEventRecord(p.ev_fork, s);
            (void)hipStreamWaitEvent(p.side, p.ev_fork, 0);
            launch_tiles<ROT, GUARD>(p, aa, xcd_cap, p.side, mark, false, 4);
            launch_tail<ROT, GUARD>(p, aa, xcd_cap, p.d_queue2.p, (uint32_t)half2, p.side, mark);
            (void)hipEventRecord(p.ev_join, p.side);
            launch_tiles<ROT, GUARD>(p, ab, xcd_cap, s, mark, false, 2);
            launch_tail<ROT, GUARD>(p, ab, xcd_cap, p.d_queue2.p + half2, (uint32_t)half2, s, mark);
            (void)hipStreamWaitEvent(s, p.ev_join, 0);
        } else if (chunks <= 1) {
            const uint32_t xcd_cap = (uint32_t)std::min<long long>(qtotal / 8, 0xffffffffLL);
            launch_tiles<ROT, GUARD>(p, a, xcd_cap, s, mark, v3);
            launch_tail<ROT, GUARD>(p, a, xcd_cap, p.d_queue2.p, (uint32_t)p.qcap2, s, mark);
        } else {
            // Frames are independent, so the batch is cut into chunks (multiples of 8 frames: the XCD dealing) and the deep
            // tail of chunk c -- latency-bound, few waves -- runs on its own stream while the tile kernels of chunk c+1 --
            // issue-bound -- fill the rest of the machine.  Two queue sets alternate; everything joins `s` before returning.
            const int per = (((a.nframes + chunks - 1) / chunks) + 7) & ~7;
            const long long half = qtotal / 2, half2 = p.qcap2 / 2;
            const uint32_t xcd_cap = (uint32_t)std::min<long long>(half / 8, 0xffffffffLL);
            bool used[2] = {false, false};
            int c = 0;
            for (int f0 = 0; f0 < a.nframes; f0 += per, ++c) {
                const int set = c & 1;
                ScanArgs ac = a;
                ac.nframes = std::min(per, a.nframes - f0);
                ac.frames = a.frames + (size_t)f0 * a.frame_stride;
                ac.counts = a.counts + f0;
                ac.raw = a.raw + (size_t)f0 * a.det_cap;
                ac.queue = p.d_queue.p + (size_t)set * half;
                ac.qcount = p.d_qcount.p + 16 * set;
                if (used[set]) (void)hipStreamWaitEvent(s, p.ev_tail[set], 0);  // the tail of chunk c-2 is done with this set
                (void)hipMemsetAsync(ac.qcount, 0, 16 * sizeof(uint32_t), s);
                launch_tiles<ROT, GUARD>(p, ac, xcd_cap, s, mark, v3);
                (void)hipEventRecord(p.ev_tiles[set], s);
                (void)hipStreamWaitEvent(p.tail_stream, p.ev_tiles[set], 0);
                launch_tail<ROT, GUARD>(p, ac, xcd_cap, p.d_queue2.p + (size_t)set * half2, (uint32_t)half2, p.tail_stream, mark);
                (void)hipEventRecord(p.ev_tail[set], p.tail_stream);
                used[set] = true;
            }
            for (int set = 0; set < 2; ++set)
                if (used[set]) (void)hipStreamWaitEvent(s, p.ev_tail[set], 0);
        }
#ifdef PIGO_DEBUG_BUILD
    } else if (variant == 1) {
        mark("scan_head");
        k_scan_head<ROT, GUARD><<<nb, kThreads, 0, s>>>(a);
        if (a.n_tail_stages > 0) {
            mark("scan_tail");
            k_scan_tail<ROT, GUARD><<<(uint32_t)a.nframes * (uint32_t)a.tail_wgs, kThreads, 0, s>>>(a);
        }
#endif
    } else {
        mark("scan_mono");
        k_scan_mono<ROT, GUARD><<<nb, kThreads, 0, s>>>(a);
    }
    (void)p;
}

pigo_status plan_run_variant(pigo_plan *p, const uint8_t *d_frames, size_t frame_stride, int nframes, pigo_det *d_dets, int32_t *d_counts,
                             hipStream_t s, int variant)
{
    if (!p) return fail(PIGO_ERR_PARAM, "plan is NULL");
    if (nframes < 0 || nframes > p->max_frames) return fail(PIGO_ERR_PARAM, "nframes %d outside [0, %d]", nframes, p->max_frames);
    if (nframes == 0) return PIGO_OK;
    if (!d_frames || !d_dets || !d_counts) return fail(PIGO_ERR_PARAM, "NULL device pointer");
    if (frame_stride < (size_t)p->key.rows * p->key.dim) return fail(PIGO_ERR_PARAM, "frame_stride smaller than rows*dim");
    if (p->key.dim % 4 == 0 && (((uintptr_t)d_frames | (uintptr_t)frame_stride) & 3u))  // the tile / patch copies move aligned dwords
        return fail(PIGO_ERR_PARAM, "d_frames and frame_stride must be multiples of 4 bytes when dim is");
    HIP_TRY(hipSetDevice(p->c->device));
    // (k_scan_one with its own order restore writes the counts itself)
    const bool one_restore = variant == 3 && p->one_ok && p->one.restore && !p->guard && !p->scales.empty() && p->c->ntrees != 0;
    if (!one_restore) HIP_TRY(hipMemsetAsync(d_counts, 0, (size_t)nframes * 4, s));
    p->one_dets = d_dets;
    p->last_nframes = nframes;
    p->n_timed = 0;
    if (p->scales.empty() || p->c->ntrees == 0) return PIGO_OK;  // no window is classified / classifyRegion returns 0.0 (pigo.go:146)

    ScanArgs a = p->args;
    a.frames = d_frames;
    a.frame_stride = frame_stride;
    a.nframes = nframes;
    a.counts = d_counts;
    a.tail_wgs = std::max(8, std::min(256, 2048 / nframes));
    if (variant >= 1 && !(variant == 3 && p->one_ok)) HIP_TRY(hipMemsetAsync(p->d_qcount.p, 0, (size_t)std::max(nframes, 64) * 4, s));

    size_t ev = 0;
    static const bool sync_debug = env_int("PIGO_SYNC_DEBUG", 0) != 0;  // debugging aid: synchronise and report before every kernel
    auto mark = [&](const char *name) {
        if (sync_debug) {
            const hipError_t e = hipDeviceSynchronize();
            fprintf(stderr, "[pigo] before %s: %s\n", name, hipGetErrorString(e));
        }
        if (!p->profiling) return;
        if (ev >= p->events.size()) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return;
            p->events.push_back(e);
            p->ev_names.push_back(name);
        }
        p->ev_names[ev] = name;
        (void)hipEventRecord(p->events[ev], s);
        ++ev;
    };
    if (p->rot) {
        if (p->guard)
            launch_scan<true, true>(*p, a, variant, s, mark);
        else
            launch_scan<true, false>(*p, a, variant, s, mark);
    } else {
        launch_scan<false, false>(*p, a, variant, s, mark);
    }
    if (!one_restore) {
        mark("restore_order");
        dim3 grid((unsigned)((p->det_cap + kThreads - 1) / kThreads), (unsigned)nframes);
        k_restore_order<<<grid, kThreads, 0, s>>>(p->d_raw.p, d_counts, p->det_cap, d_dets);
    }
    mark("end");
    p->n_timed = (int)ev;
    HIP_TRY(hipGetLastError());
    return PIGO_OK;
}

}  // namespace

extern "C" pigo_status pigo_plan_create(pigo_cascade *c, int rows, int cols, int dim, int min_size, int max_size, double shift_factor,
                                        double scale_factor, double angle, int max_frames, int det_cap, pigo_plan **out)
{
    if (!out) return fail(PIGO_ERR_PARAM, "out is NULL");
    *out = nullptr;
    std::unique_ptr<pigo_plan> p;
    PlanKey key{rows, cols, dim, min_size, max_size, shift_factor, scale_factor, angle};
    pigo_status st = plan_build(c, key, max_frames, det_cap, p);
    if (st != PIGO_OK) return st;
    *out = p.release();
    return PIGO_OK;
}

extern "C" void pigo_plan_destroy(pigo_plan *p)
{
    if (!p) return;
    (void)hipSetDevice(p->c->device);
    delete p;
}

extern "C" pigo_status pigo_plan_info(const pigo_plan *p, pigo_plan_info_t *info)
{
    if (!p || !info) return fail(PIGO_ERR_PARAM, "NULL argument");
    info->windows_per_frame = p->windows;
    info->n_scales = (int32_t)p->scales.size();
    info->n_ladder = p->n_ladder;
    info->tiles_per_frame = (int32_t)p->tiles.size();
    info->n_head_trees = p->variant >= 2 ? p->args.nh_lds : p->args.nh;
    info->variant = p->variant;
    info->max_frames = p->max_frames;
    info->det_cap = p->det_cap;
    info->queue_capacity = p->qcap;
    info->workspace_bytes = (int64_t)p->workspace_bytes();
    return PIGO_OK;
}

extern "C" pigo_status pigo_plan_set_variant(pigo_plan *p, int variant)
{
    if (!p) return fail(PIGO_ERR_PARAM, "plan is NULL");
    if (variant < 0 || variant > 3) return fail(PIGO_ERR_PARAM, "variant must be 0, 1, 2 or 3");
    if (variant == 3 && !p->region_ok) return fail(PIGO_ERR_PARAM, "variant 3 not available for this plan (rotated scan, stride not a multiple of 4, ...)");
#ifndef PIGO_DEBUG_BUILD
    if (variant == 1) return fail(PIGO_ERR_PARAM, "variant 1 exists in the debug build only (python -m pigo_amd.build --debug)");
#endif
    if (variant >= 1 && !(p->c->depth == 6 && p->c->ntrees > 0)) return fail(PIGO_ERR_PARAM, "variants 1 and 2 need a depth-6 cascade");
    if (variant == 2 && !p->tile_ok) return fail(PIGO_ERR_PARAM, "variant 2 not available for this cascade");
    p->variant = variant;
    return PIGO_OK;
}

extern "C" pigo_status pigo_plan_run(pigo_plan *p, const uint8_t *d_frames, size_t frame_stride, int nframes, pigo_det *d_dets,
                                     int32_t *d_counts, void *stream)
{
    if (!p) return fail(PIGO_ERR_PARAM, "plan is NULL");
    std::lock_guard<std::mutex> lock(p->mu);  // a plan owns one workspace: launches on it are serialised
    hipStream_t s = (hipStream_t)stream;
    // (a plan of a few frames that runs as ONE launch -- k_scan_one -- has nothing a graph could save)
    if (nframes < 1 || nframes > p->graph_max_frames || p->profiling || !p->cap_stream || (p->variant == 3 && p->one_ok))
        return plan_run_variant(p, d_frames, frame_stride, nframes, d_dets, d_counts, s, p->variant);
    // Small batch: ~10 dependent launches, memsets and stream joins cost more host time than GPU time.  The sequence only
    // depends on the buffers, so it is captured once (on an internal stream: the caller's may be the legacy default stream,
    // which cannot be captured) and replayed on the caller's stream as one graph launch.
    pigo_plan::GraphCache &gc = p->gc;
    if (!(gc.exec && gc.frames == d_frames && gc.stride == frame_stride && gc.nframes == nframes && gc.dets == d_dets && gc.counts == d_counts &&
          gc.variant == p->variant)) {
        HIP_TRY(hipSetDevice(p->c->device));
        if (gc.exec) {
            (void)hipGraphExecDestroy(gc.exec);
            gc.exec = nullptr;
        }
        hipGraph_t graph = nullptr;
        // (the captured sequence stays on ONE stream: a fork onto the plan's side stream inside a capture, next to plan builds and
        // frees on other threads, crashed inside the runtime about once in four runs of the parity suite -- pigo_run_cascade's slots)
        HIP_TRY(hipStreamBeginCapture(p->cap_stream, hipStreamCaptureModeThreadLocal));
        p->no_fork = true;
        const pigo_status st = plan_run_variant(p, d_frames, frame_stride, nframes, d_dets, d_counts, p->cap_stream, p->variant);
        p->no_fork = false;
        const hipError_t e = hipStreamEndCapture(p->cap_stream, &graph);
        if (st != PIGO_OK) {
            if (graph) (void)hipGraphDestroy(graph);
            return st;
        }
        if (e != hipSuccess || !graph) {  // capture not possible here: run the plain way
            (void)hipGetLastError();
            p->graph_max_frames = 0;
            return plan_run_variant(p, d_frames, frame_stride, nframes, d_dets, d_counts, s, p->variant);
        }
        const hipError_t ei = hipGraphInstantiate(&gc.exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (ei != hipSuccess) {
            gc.exec = nullptr;
            (void)hipGetLastError();
            p->graph_max_frames = 0;
            return plan_run_variant(p, d_frames, frame_stride, nframes, d_dets, d_counts, s, p->variant);
        }
        gc.frames = d_frames;
        gc.stride = frame_stride;
        gc.nframes = nframes;
        gc.dets = d_dets;
        gc.counts = d_counts;
        gc.variant = p->variant;
    }
    p->last_nframes = nframes;
    p->n_timed = 0;
    HIP_TRY(hipGraphLaunch(gc.exec, s));
    return PIGO_OK;
}

extern "C" pigo_status pigo_plan_status(pigo_plan *p)
{
    if (!p) return fail(PIGO_ERR_PARAM, "plan is NULL");
    HIP_TRY(hipSetDevice(p->c->device));
    int32_t flags[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpy(flags, p->d_flags.p, 16, hipMemcpyDeviceToHost));
    if (flags[0] || flags[1] || flags[2]) HIP_TRY(hipMemset(p->d_flags.p, 0, 16));
    uint32_t rep[12] = {0};
    if ((flags[0] & 16) && p->one_ok) HIP_TRY(hipMemcpy(rep, p->d_onecnt.p + kOneReport, sizeof rep, hipMemcpyDeviceToHost));
    if (flags[0] && p->one_ok) {  // k_scan_one: a run that overflowed a queue or gave up a hand-off may have left entries and counters behind
        HIP_TRY(hipMemset(p->d_onecnt.p, 0, (size_t)kOneCntWords * 4));
        HIP_TRY(hipMemset(p->d_oneq.p, 0, (size_t)8 * p->one.qcap * sizeof(uint4)));
    }
    p->last_flags[0] = flags[0];
    p->last_flags[1] = flags[1];
    p->last_flags[2] = flags[2];
    if (flags[1]) return fail(PIGO_ERR_PANIC, "the reference would panic: pixel index out of range in classifyRotatedRegion (pigo.go:167-179)");
    if ((flags[0] & 16) && p->one_ok)
        return fail(PIGO_ERR_TIMEOUT,
                    "hand-off timeout inside k_scan_one (queue %u slot %u, workgroup %u: done at claim %u, now %u / %u of %u items; alloc %u / %u, head %u; tags %08x %08x)",
                    rep[0], rep[1], rep[11], rep[2], rep[3], rep[4], rep[8], rep[5], rep[6], rep[7], rep[9], rep[10]);
    if (flags[0])
        return fail(PIGO_ERR_CAPACITY, "survivor queue overflow (%s%s%s%s%s)", (flags[0] & 1) ? "tile LDS queue " : "", (flags[0] & 2) ? "survivor queue " : "",
                    (flags[0] & 4) ? "second-level tail queue " : "", (flags[0] & 8) ? "bucket list " : "", (flags[0] & 16) ? "hand-off timeout inside k_scan_one" : "");
    if (flags[2]) return fail(PIGO_ERR_CAPACITY, "a frame has more than det_cap (%d) detections: its list is truncated (d_counts holds the true count)", p->det_cap);
    return PIGO_OK;
}

extern "C" pigo_status pigo_plan_last_flags(const pigo_plan *p, int32_t *queue_overflow, int32_t *would_panic, int32_t *det_cap_overflow)
{
    if (!p) return fail(PIGO_ERR_PARAM, "plan is NULL");
    if (queue_overflow) *queue_overflow = p->last_flags[0];
    if (would_panic) *would_panic = p->last_flags[1];
    if (det_cap_overflow) *det_cap_overflow = p->last_flags[2];
    return PIGO_OK;
}

extern "C" pigo_status pigo_plan_run_sync(pigo_plan *p, const uint8_t *d_frames, size_t frame_stride, int nframes, pigo_det *d_dets,
                                          int32_t *d_counts, void *stream)
{
    if (!p) return fail(PIGO_ERR_PARAM, "plan is NULL");
    std::lock_guard<std::mutex> lock(p->mu);
    hipStream_t s = (hipStream_t)stream;
    pigo_status st = plan_run_variant(p, d_frames, frame_stride, nframes, d_dets, d_counts, s, p->variant);
    if (st != PIGO_OK) return st;
    HIP_TRY(hipStreamSynchronize(s));
    st = pigo_plan_status(p);
    if (st == PIGO_ERR_CAPACITY && p->last_flags[0] && p->variant >= 1) {  // pathological frame: more survivors than the queue holds
        st = plan_run_variant(p, d_frames, frame_stride, nframes, d_dets, d_counts, s, 0);
        if (st != PIGO_OK) return st;
        HIP_TRY(hipStreamSynchronize(s));
        st = pigo_plan_status(p);
    }
    return st;
}

extern "C" pigo_status pigo_plan_set_profiling(pigo_plan *p, int on)
{
    if (!p) return fail(PIGO_ERR_PARAM, "plan is NULL");
    p->profiling = on != 0;
    return PIGO_OK;
}

extern "C" int pigo_plan_last_timings(pigo_plan *p, const char **names, float *ms, int cap)
{
    if (!p || p->n_timed < 2) return 0;
    (void)hipSetDevice(p->c->device);
    // (launches of one kind are summed: the side chain walks a large batch in chunks, three launches each)
    int n = 0;
    for (int i = 0; i + 1 < p->n_timed; ++i) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, p->events[i], p->events[i + 1]) != hipSuccess) return n;
        int k = 0;
        while (k < n && strcmp(names[k], p->ev_names[i]) != 0) ++k;
        if (k == n) {
            if (n >= cap) continue;
            names[n] = p->ev_names[i];
            ms[n] = 0.f;
            ++n;
        }
        ms[k] += t;
    }
    return n;
}

extern "C" pigo_status pigo_plan_debug_trace(pigo_plan *p, uint64_t *out, int n)
{
    if (!p || !out || n < 16 * 256 || !p->d_stats.p) return fail(PIGO_ERR_PARAM, "bad argument");
    HIP_TRY(hipSetDevice(p->c->device));
    HIP_TRY(hipMemcpy(out, p->d_stats.p + 65536 * 8, 16 * 256 * 8, hipMemcpyDeviceToHost));
    return PIGO_OK;
}

extern "C" pigo_status pigo_plan_debug_stats(pigo_plan *p, uint64_t *out, int n)
{
    if (!p || !out || n < 1) return fail(PIGO_ERR_PARAM, "bad argument");
    memset(out, 0, (size_t)n * 8);
    if (!p->d_stats.p) return PIGO_OK;
    HIP_TRY(hipSetDevice(p->c->device));
    std::vector<unsigned long long> h(65536 * 8);
    HIP_TRY(hipMemcpy(h.data(), p->d_stats.p, h.size() * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(p->d_stats.p, 0, h.size() * 8));
    for (size_t b = 0; b < 65536; ++b)
        for (int k = 0; k < 8 && k < n; ++k) out[k] += h[b * 8 + k];
    if (n >= 16) {  // [8..15]: k_tail_deep phases (wave 0 of every workgroup)
        std::vector<unsigned long long> t(256 * 8);
        HIP_TRY(hipMemcpy(t.data(), p->d_stats.p + 65536 * 8 + 4096, t.size() * 8, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemset(p->d_stats.p + 65536 * 8 + 4096, 0, t.size() * 8));
        for (size_t b = 0; b < 256; ++b)
            for (int k = 0; k < 8; ++k) out[8 + k] += t[b * 8 + k];
    }
    return PIGO_OK;
}

extern "C" pigo_status pigo_plan_last_queue_count(pigo_plan *p, int64_t *n)
{
    if (!p || !n) return fail(PIGO_ERR_PARAM, "NULL argument");
    HIP_TRY(hipSetDevice(p->c->device));
    std::vector<uint32_t> h((size_t)std::max(1, p->last_nframes));
    *n = 0;
    if (p->last_nframes == 0) return PIGO_OK;
    HIP_TRY(hipMemcpy(h.data(), p->d_qcount.p, (size_t)p->last_nframes * 4, hipMemcpyDeviceToHost));
    if (p->variant == 3 && p->one_ok) {  // k_scan_one: the eight queues' `alloc` counters
        std::vector<uint32_t> hc(8 * kOneLine);
        HIP_TRY(hipMemcpy(hc.data(), p->d_onecnt.p, hc.size() * 4, hipMemcpyDeviceToHost));
        for (uint32_t x = 0; x < 8; ++x) *n += hc[x * kOneLine];
        return PIGO_OK;
    }
    if (p->variant >= 2) {
        uint32_t h8[8] = {0};
        HIP_TRY(hipMemcpy(h8, p->d_qcount.p, sizeof h8, hipMemcpyDeviceToHost));
        *n = 0;
        for (uint32_t v : h8) *n += v;  // eight per-XCD queues ([8] counts the second-level tail)
        return PIGO_OK;
    }
    for (uint32_t v : h) *n += v;
    return PIGO_OK;
}

// ---- ClusterDetections ----------------------------------------------------------------------------------------------

extern "C" pigo_status pigo_plan_cluster(pigo_plan *p, const pigo_det *d_dets, const int32_t *d_counts, int nframes, double iou_threshold,
                                         pigo_det *d_sorted, pigo_det *d_clusters, int32_t *d_ccounts, int32_t *d_ties, void *stream)
{
    if (!p) return fail(PIGO_ERR_PARAM, "plan is NULL");
    if (nframes < 0 || nframes > p->max_frames) return fail(PIGO_ERR_PARAM, "nframes outside the plan's range");
    if (nframes == 0) return PIGO_OK;
    if (!d_dets || !d_counts || !d_sorted || !d_clusters || !d_ccounts) return fail(PIGO_ERR_PARAM, "NULL device pointer");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(p->c->device));
    // Lists of a few hundred detections (1080p) are served by one workgroup per frame; plans that can hold long lists (the 4K
    // stress config: thousands of detections, hundreds of clusters per frame) by the seeds / members / compact kernels, which
    // spread a frame's clusters over the chip and have no length limit.  PIGO_CLUSTER_V2=0/1 forces one.
    // (the family and its workspaces are fixed at plan creation: plan_alloc_batch)
    const bool v2 = p->cluster_mode == 1;
    if (!v2 && p->det_cap > 65536) return fail(PIGO_ERR_PARAM, "k_cluster supports det_cap <= 65536 (PIGO_CLUSTER_V2=0 was forced)");
    if (!d_ties) d_ties = p->d_ties.p;
    HIP_TRY(hipMemsetAsync(d_ties, 0, (size_t)nframes * 4, s));
    dim3 grid((unsigned)((p->det_cap + kThreads - 1) / kThreads), (unsigned)nframes);
    if (v2) {  // long lists: partial ranks over kRankSegs segments of the list, then one scatter (the rank array is d_cl_tmpn, free until the sweep)
        uint32_t *d_rank = reinterpret_cast<uint32_t *>(p->d_cl_tmpn.p);
        HIP_TRY(hipMemsetAsync(d_rank, 0, (size_t)nframes * p->det_cap * 4, s));
        k_rank_partial<<<dim3(grid.x, kRankSegs, (unsigned)nframes), kThreads, 0, s>>>(d_dets, d_counts, p->det_cap, d_rank);
        k_scatter_by_rank<<<grid, kThreads, 0, s>>>(d_dets, d_counts, p->det_cap, d_rank, d_sorted, d_ties);
    } else {
        k_sort_by_q<<<grid, kThreads, 0, s>>>(d_dets, d_counts, p->det_cap, d_sorted, d_ties);
    }
    // frames with tied Q values: redo the sort with Go's own (unstable) algorithm so that the tie order -- and with it the
    // seed order and the float32 sum order of ClusterDetections -- is the reference's
    {
        const int lds_keys = std::min(p->det_cap, kGoSortKeys);
        const size_t lds_fixed = sizeof(gosort::PartList) + (size_t)(gosort::kSortThreads / 64) * 2 * gosort::kWaveFifo * 4;
        if (!p->gosort_attr) {
            HIP_TRY(hipFuncSetAttribute((const void *)k_gosort_ties, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds_fixed + (size_t)kGoSortKeys * 10)));
            p->gosort_attr = true;
        }
        k_gosort_ties<<<nframes, gosort::kSortThreads, lds_fixed + (size_t)lds_keys * 10, s>>>(d_dets, d_counts, p->det_cap, d_ties, d_sorted, lds_keys,
                                                                                              p->det_cap > kGoSortKeys ? p->d_gosort_ws.p : nullptr);
    }
    if (v2) {
        k_cluster_seeds<<<nframes, kSeedThreads, 0, s>>>(d_sorted, d_counts, p->det_cap, iou_threshold, p->d_cl_seeds.p, p->d_cl_nseeds.p);
        // one wave per cluster; waves stride over a frame's seeds (the grid cannot depend on the count without a host round trip)
        const unsigned gx = (unsigned)std::max(1, std::min((p->det_cap + 3) / 4, std::max(8, 2048 / nframes)));
        k_cluster_members<<<dim3(gx, (unsigned)nframes), kMemberThreads, 0, s>>>(d_sorted, d_counts, p->det_cap, iou_threshold, p->d_cl_seeds.p, p->d_cl_nseeds.p,
                                                                                  p->d_cl_tmp.p, p->d_cl_tmpn.p);
        k_cluster_compact<<<nframes, 256, 0, s>>>(p->d_cl_nseeds.p, p->det_cap, p->d_cl_tmp.p, p->d_cl_tmpn.p, d_clusters, d_ccounts);
    } else if (p->det_cap <= 256 * 64) {
        k_cluster<256><<<nframes, 256, 0, s>>>(d_sorted, d_counts, p->det_cap, iou_threshold, d_clusters, d_ccounts, p->d_mq.p);
    } else {
        k_cluster<1024><<<nframes, 1024, 0, s>>>(d_sorted, d_counts, p->det_cap, iou_threshold, d_clusters, d_ccounts, p->d_mq.p);
    }
    HIP_TRY(hipGetLastError());
    return PIGO_OK;
}


extern "C" void pigo_sort_by_q(pigo_det *dets, int n)
{
    if (!dets || n <= 1) return;
    gosort::Data x{dets};
    gosort::pdqsort(x, 0, n, gosort::bits_len((unsigned long long)n));
}

extern "C" pigo_status pigo_cluster_detections(pigo_cascade *c, pigo_det *dets, int n, double iou_threshold, pigo_det *out, int cap,
                                               int *n_out)
{
    if (!c) return fail(PIGO_ERR_PARAM, "cascade is NULL");
    if (n_out) *n_out = 0;
    if (n < 0 || (n > 0 && !dets)) return fail(PIGO_ERR_PARAM, "bad detection list");
    if (n == 0) return PIGO_OK;  // clusters := []Detection{}  (pigo.go:280)
    pigo_sort_by_q(dets, n);     // pigo.go:264-266, in place like the reference
    HIP_TRY(hipSetDevice(c->device));
    // a free slot, or a new one: the handle's lock covers the list only (N concurrent callers = N slots, each with its own stream)
    pigo_cascade::ClusterSlot *sl = nullptr;
    {
        std::lock_guard<std::mutex> lock(c->mu);
        for (auto &u : c->cl_slots)
            if (!u->busy) {
                sl = u.get();
                break;
            }
        if (!sl) {
            std::unique_ptr<pigo_cascade::ClusterSlot> u(new (std::nothrow) pigo_cascade::ClusterSlot);
            if (!u) return fail(PIGO_ERR_NOMEM, "out of memory");
            sl = u.get();
            c->cl_slots.push_back(std::move(u));
        }
        sl->busy = true;
    }
    struct Release {
        pigo_cascade *c;
        pigo_cascade::ClusterSlot *sl;
        ~Release()
        {
            std::lock_guard<std::mutex> lock(c->mu);
            sl->busy = false;
        }
    } release{c, sl};
    if (!sl->stream) HIP_TRY(hipStreamCreateWithFlags(&sl->stream, hipStreamNonBlocking));
    hipStream_t s = sl->stream;
    if (n <= kClusterStaged && env_int("PIGO_CLUSTER_V2", -1) < 0) {
        // a short list (what one frame yields): ONE launch that reads the sorted list from pinned host memory into LDS and writes the
        // clusters and their number back to pinned host memory, one synchronisation -- no copy in front of or behind the kernel
        if (!sl->h_cl) HIP_TRY(hipHostMalloc((void **)&sl->h_cl, (size_t)kClusterStaged * sizeof(pigo_det) * 2 + 64, hipHostMallocDefault));
        pigo_det *h_in = sl->h_cl, *h_out = sl->h_cl + kClusterStaged;
        int32_t *h_cnt = reinterpret_cast<int32_t *>(sl->h_cl + 2 * kClusterStaged);  // [0] n, [1] clusters
        memcpy(h_in, dets, (size_t)n * sizeof(pigo_det));
        h_cnt[0] = n;
        h_cnt[1] = -1;
        k_cluster<256, true><<<1, 256, 0, s>>>(h_in, h_cnt, kClusterStaged, iou_threshold, h_out, h_cnt + 1, nullptr);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(s));
        const int32_t ncl = h_cnt[1];
        if (ncl < 0) return fail(PIGO_ERR_HIP, "ClusterDetections: the kernel left no result");
        if (n_out) *n_out = ncl;
        if (ncl > cap) return fail(PIGO_ERR_CAPACITY, "ClusterDetections: %d clusters, capacity %d", ncl, cap);
        if (ncl > 0) {
            if (!out) return fail(PIGO_ERR_PARAM, "out is NULL");
            memcpy(out, h_out, (size_t)ncl * sizeof(pigo_det));
        }
        return PIGO_OK;
    }
    if (sl->d_sorted.n < (size_t)n) HIP_TRY(sl->d_sorted.alloc(n));
    if (sl->d_clusters.n < (size_t)n) HIP_TRY(sl->d_clusters.alloc(n));
    if (sl->d_mq.n < (size_t)n) HIP_TRY(sl->d_mq.alloc(n));
    if (sl->d_small.n < 4) HIP_TRY(sl->d_small.alloc(4));
    const int32_t h_small[2] = {n, 0};
    HIP_TRY(hipMemcpyAsync(sl->d_small.p, h_small, 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(sl->d_sorted.p, dets, (size_t)n * sizeof(pigo_det), hipMemcpyHostToDevice, s));
    const int mode = env_int("PIGO_CLUSTER_V2", -1);
    if (mode >= 0 ? mode != 0 : n > 2048) {  // long list: seeds first, then every cluster on its own wave (no length limit)
        if (sl->d_cl_seeds.n < (size_t)n) HIP_TRY(sl->d_cl_seeds.alloc(n));
        if (sl->d_cl_tmpn.n < (size_t)n) HIP_TRY(sl->d_cl_tmpn.alloc(n));
        if (sl->d_cl_tmp.n < (size_t)n) HIP_TRY(sl->d_cl_tmp.alloc(n));
        k_cluster_seeds<<<1, kSeedThreads, 0, s>>>(sl->d_sorted.p, sl->d_small.p, n, iou_threshold, sl->d_cl_seeds.p, sl->d_small.p + 2);
        k_cluster_members<<<dim3((unsigned)std::max(1, std::min((n + 3) / 4, 2048)), 1u), kMemberThreads, 0, s>>>(sl->d_sorted.p, sl->d_small.p, n, iou_threshold,
                                                                                                                    sl->d_cl_seeds.p, sl->d_small.p + 2, sl->d_cl_tmp.p,
                                                                                                                    sl->d_cl_tmpn.p);
        k_cluster_compact<<<1, 256, 0, s>>>(sl->d_small.p + 2, n, sl->d_cl_tmp.p, sl->d_cl_tmpn.p, sl->d_clusters.p, sl->d_small.p + 1);
    } else if (n <= 256 * 64) {
        k_cluster<256><<<1, 256, 0, s>>>(sl->d_sorted.p, sl->d_small.p, n, iou_threshold, sl->d_clusters.p, sl->d_small.p + 1, sl->d_mq.p);
    } else if (n <= 65536) {
        k_cluster<1024><<<1, 1024, 0, s>>>(sl->d_sorted.p, sl->d_small.p, n, iou_threshold, sl->d_clusters.p, sl->d_small.p + 1, sl->d_mq.p);
    } else {
        (void)hipStreamSynchronize(s);  // (the uploads read the caller's memory)
        return fail(PIGO_ERR_PARAM, "k_cluster supports at most 65536 detections (PIGO_CLUSTER_V2=0 was forced)");
    }
    if (hipGetLastError() != hipSuccess) {
        (void)hipStreamSynchronize(s);
        return fail(PIGO_ERR_HIP, "ClusterDetections: a launch failed");
    }
    int32_t ncl = 0;
    HIP_TRY(hipMemcpyAsync(&ncl, sl->d_small.p + 1, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (n_out) *n_out = ncl;
    if (ncl > cap) return fail(PIGO_ERR_CAPACITY, "ClusterDetections: %d clusters, capacity %d", ncl, cap);
    if (ncl > 0) {
        if (!out) return fail(PIGO_ERR_PARAM, "out is NULL");
        HIP_TRY(hipMemcpyAsync(out, sl->d_clusters.p, (size_t)ncl * sizeof(pigo_det), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    return PIGO_OK;
}

// ---- RunCascade (one frame, host memory) ----------------------------------------------------------------------------------

namespace {

// enqueue one RunCascade on the slot's stream: upload, scan, download of the count, the status flags and the detections
pigo_status slot_enqueue(pigo_cascade::RunSlot &sl)
{
    pigo_plan *p = sl.plan.get();
    // (the upload in pieces, each in flight while the next is copied into the staging buffer, measured flat: 0.292 vs 0.297 ms --
    // every hipMemcpyAsync costs the host what it hides)
    HIP_TRY(hipMemcpyAsync(sl.d_frame.p, sl.h_frame, sl.fbytes, hipMemcpyHostToDevice, sl.stream));
    if (p->variant == 3 && p->one_ok && p->one.restore && !p->guard && !p->scales.empty() && p->c->ntrees != 0) {
        // ONE launch whose last workgroup writes the ordered detections, the count and the flag words straight into the slot's
        // pinned host buffers (device-visible): no copy behind the scan, one stream node less per result
        p->one_host_flags = sl.h_small + 1;
        const pigo_status st = plan_run_variant(p, sl.d_frame.p, sl.fbytes, 1, sl.h_dets, sl.h_small, sl.stream, p->variant);
        p->one_host_flags = nullptr;
        return st;
    }
    pigo_status st = plan_run_variant(p, sl.d_frame.p, sl.fbytes, 1, sl.d_dets.p, sl.d_count.p, sl.stream, p->variant);
    if (st != PIGO_OK) return st;
    HIP_TRY(hipMemcpyAsync(sl.h_small, sl.d_count.p, 4, hipMemcpyDeviceToHost, sl.stream));
    HIP_TRY(hipMemcpyAsync(sl.h_small + 1, p->d_flags.p, 16, hipMemcpyDeviceToHost, sl.stream));
    HIP_TRY(hipMemcpyAsync(sl.h_dets, sl.d_dets.p, (size_t)p->det_cap * sizeof(pigo_det), hipMemcpyDeviceToHost, sl.stream));
    return PIGO_OK;
}

}  // namespace

extern "C" pigo_status pigo_run_cascade(pigo_cascade *c, const uint8_t *pixels, size_t npixels, int rows, int cols, int dim, int min_size,
                                        int max_size, double shift_factor, double scale_factor, double angle, pigo_det *out, int cap,
                                        int *n_out)
{
    if (!c) return fail(PIGO_ERR_PARAM, "cascade is NULL");
    if (n_out) *n_out = 0;
    if (!pixels) return fail(PIGO_ERR_PARAM, "pixels is NULL");
    if (cap < 0 || (cap > 0 && !out)) return fail(PIGO_ERR_PARAM, "bad output buffer");
    if (rows >= 1 && dim >= 1 && npixels < (size_t)rows * (size_t)dim)
        return fail(PIGO_ERR_PARAM, "len(pixels)=%zu < rows*dim=%zu", npixels, (size_t)rows * (size_t)dim);
    const PlanKey key{rows, cols, dim, min_size, max_size, shift_factor, scale_factor, angle};
    const size_t fbytes = (size_t)rows * (size_t)dim;
    int det_cap = std::max(cap, 4096);
    HIP_TRY(hipSetDevice(c->device));
    for (int attempt = 0; attempt < 2; ++attempt) {
        // ---- take a free slot with these parameters, or build one (the list is the only shared state) ----
        pigo_cascade::RunSlot *sl = nullptr;
        {
            std::lock_guard<std::mutex> lock(c->mu);
            for (auto it = c->slots.begin(); it != c->slots.end(); ++it) {
                if (!(*it)->busy && (*it)->key == key && (*it)->plan->det_cap >= det_cap) {
                    c->slots.splice(c->slots.begin(), c->slots, it);
                    sl = c->slots.front().get();
                    sl->busy = true;
                    break;
                }
            }
        }
        if (!sl) {
            // (slots are built concurrently: plan_build stays on its own stream, see there)
            std::unique_ptr<pigo_cascade::RunSlot> ns(new (std::nothrow) pigo_cascade::RunSlot);
            if (!ns) return fail(PIGO_ERR_NOMEM, "out of memory");
            ns->key = key;
            ns->fbytes = fbytes;
            pigo_status st = plan_build(c, key, 1, det_cap, ns->plan);
            if (st != PIGO_OK) return st;
            HIP_TRY(hipStreamCreateWithFlags(&ns->stream, hipStreamNonBlocking));
            HIP_TRY(ns->d_frame.alloc(fbytes));
            HIP_TRY(ns->d_dets.alloc(det_cap));
            HIP_TRY(ns->d_count.alloc(4));
            HIP_TRY(hipHostMalloc((void **)&ns->h_frame, std::max<size_t>(fbytes, 16), hipHostMallocDefault));
            HIP_TRY(hipHostMalloc((void **)&ns->h_dets, (size_t)det_cap * sizeof(pigo_det), hipHostMallocDefault));
            HIP_TRY(hipHostMalloc((void **)&ns->h_small, 32, hipHostMallocDefault));
            ns->busy = true;
            if (env_int("PIGO_GRAPH_FRAMES", 0) >= 1) {  // capture the call once; without a graph the slot enqueues it every time
                // (the captured sequence stays on the slot's one stream: a fork onto the plan's side stream inside a capture,
                // next to plan builds and frees on other threads, crashed once in ~4 runs of the parity suite -- inside the runtime)
                ns->plan->fork_min_frames = 1 << 30;
                hipGraph_t graph = nullptr;
                if (hipStreamBeginCapture(ns->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                    const pigo_status cs = slot_enqueue(*ns);
                    const hipError_t e = hipStreamEndCapture(ns->stream, &graph);
                    if (cs == PIGO_OK && e == hipSuccess && graph && hipGraphInstantiate(&ns->exec, graph, nullptr, nullptr, 0) != hipSuccess) ns->exec = nullptr;
                    if (graph) (void)hipGraphDestroy(graph);
                    if (cs != PIGO_OK) return cs;
                }
                (void)hipGetLastError();
            }
            std::lock_guard<std::mutex> lock(c->mu);
            c->slots.push_front(std::move(ns));
            sl = c->slots.front().get();
            // keep at most 8 idle slots
            size_t idle = 0;
            for (auto it = c->slots.begin(); it != c->slots.end();) {
                if (!(*it)->busy && ++idle > 8)
                    it = c->slots.erase(it);
                else
                    ++it;
            }
        }
        struct Release {
            pigo_cascade *c;
            pigo_cascade::RunSlot *sl;
            ~Release()
            {
                std::lock_guard<std::mutex> lock(c->mu);
                sl->busy = false;
            }
        } release{c, sl};
        pigo_plan *p = sl->plan.get();
        memcpy(sl->h_frame, pixels, fbytes);
        pigo_status st = PIGO_OK;
        if (sl->exec)
            HIP_TRY(hipGraphLaunch(sl->exec, sl->stream));
        else
            st = slot_enqueue(*sl);
        if (st != PIGO_OK) return st;
        HIP_TRY(hipStreamSynchronize(sl->stream));
        int32_t n = sl->h_small[0];
        const int32_t fl_queue = sl->h_small[1], fl_panic = sl->h_small[2], fl_dets = sl->h_small[3];
        if (fl_queue || fl_panic || fl_dets) HIP_TRY(hipMemsetAsync(p->d_flags.p, 0, 16, sl->stream));
        if (fl_queue && p->one_ok) {  // (k_scan_one: an overflowed queue or a hand-off that gave up may have left entries and counters behind)
            HIP_TRY(hipMemsetAsync(p->d_onecnt.p, 0, (size_t)kOneCntWords * 4, sl->stream));
            HIP_TRY(hipMemsetAsync(p->d_oneq.p, 0, (size_t)8 * p->one.qcap * sizeof(uint4), sl->stream));
        }
        if (fl_panic) return fail(PIGO_ERR_PANIC, "the reference would panic: pixel index out of range in classifyRotatedRegion (pigo.go:167-179)");
        if (fl_queue) {  // pathological frame: more survivors than a queue holds -- the monolithic kernel has no queue
            st = plan_run_variant(p, sl->d_frame.p, fbytes, 1, sl->d_dets.p, sl->d_count.p, sl->stream, 0);
            if (st != PIGO_OK) return st;
            HIP_TRY(hipMemcpyAsync(sl->h_small, sl->d_count.p, 4, hipMemcpyDeviceToHost, sl->stream));
            HIP_TRY(hipMemcpyAsync(sl->h_dets, sl->d_dets.p, (size_t)p->det_cap * sizeof(pigo_det), hipMemcpyDeviceToHost, sl->stream));
            HIP_TRY(hipStreamSynchronize(sl->stream));
            HIP_TRY(hipMemsetAsync(p->d_flags.p, 0, 16, sl->stream));
            n = sl->h_small[0];
        }
        if (n > p->det_cap) {  // internal buffer too small: a bigger slot, and rescan
            det_cap = n;
            continue;
        }
        if (n_out) *n_out = n;
        if (n > cap) return fail(PIGO_ERR_CAPACITY, "RunCascade: %d detections, capacity %d", n, cap);
        if (n > 0) memcpy(out, sl->h_dets, (size_t)n * sizeof(pigo_det));
        return PIGO_OK;
    }
    return fail(PIGO_ERR_CAPACITY, "RunCascade: detection count kept growing");
}

// ---- RgbToGrayscale (core/grayscale.go:8-23) ---------------------------------------------------------------------------

namespace {

template <int KIND>
pigo_status launch_gray(const uint8_t *d_pix, size_t frame_stride, int stride, int width, int height, int nframes, uint8_t *d_gray,
                        size_t gray_frame_stride, int gray_dim, hipStream_t stream)
{
    const size_t npx = (size_t)width * (size_t)height;
    const bool lin = stride == 4 * width && gray_dim == width && npx % 4 == 0 && npx / 4 <= 0xffffffffull && frame_stride % 16 == 0 &&
                     gray_frame_stride % 4 == 0 && ((uintptr_t)d_pix % 16) == 0 && ((uintptr_t)d_gray % 4) == 0;
    if (lin) {
        const uint32_t n4 = (uint32_t)(npx / 4);
        const uint32_t per_block = (uint32_t)(kGrayThreads * kGrayQuads);
        dim3 grid((n4 + per_block - 1) / per_block, (unsigned)nframes, 1);
        hipLaunchKernelGGL((k_rgb_to_gray_lin<KIND>), grid, dim3(kGrayThreads), 0, stream, d_pix, frame_stride, d_gray, gray_frame_stride, n4);
    } else {
        dim3 grid((unsigned)((width + kGrayThreads - 1) / kGrayThreads), (unsigned)height, (unsigned)nframes);
        hipLaunchKernelGGL((k_rgb_to_gray_gen<KIND>), grid, dim3(kGrayThreads), 0, stream, d_pix, frame_stride, stride, width, d_gray,
                           gray_frame_stride, gray_dim);
    }
    HIP_TRY(hipGetLastError());
    return PIGO_OK;
}

}  // namespace

extern "C" pigo_status pigo_gray_batch(int device, const uint8_t *d_pix, size_t frame_stride, int stride, int width, int height, int kind,
                                       int nframes, uint8_t *d_gray, size_t gray_frame_stride, int gray_dim, void *stream)
{
    if (kind < PIGO_PIX_NRGBA || kind > PIGO_PIX_CANVAS) return fail(PIGO_ERR_PARAM, "unknown pixel kind %d", kind);
    if (width < 0 || height < 0 || nframes < 0) return fail(PIGO_ERR_PARAM, "negative width/height/nframes");
    if (width >= 65536 * kGrayThreads || height >= 65536 || nframes >= 65536)
        return fail(PIGO_ERR_PARAM, "width/height/nframes too large (%d, %d, %d)", width, height, nframes);
    if (width == 0 || height == 0 || nframes == 0) return PIGO_OK;
    if (!d_pix || !d_gray) return fail(PIGO_ERR_PARAM, "NULL device pointer");
    if ((long long)stride < 4ll * width) return fail(PIGO_ERR_PARAM, "stride=%d < 4*width=%lld", stride, 4ll * width);
    if (gray_dim < width) return fail(PIGO_ERR_PARAM, "gray_dim=%d < width=%d", gray_dim, width);
    const size_t need_src = (size_t)(height - 1) * (size_t)stride + 4u * (size_t)width;
    const size_t need_dst = (size_t)(height - 1) * (size_t)gray_dim + (size_t)width;
    if (nframes > 1 && (frame_stride < need_src || gray_frame_stride < need_dst))
        return fail(PIGO_ERR_PARAM, "frame strides (%zu, %zu) smaller than a frame (%zu, %zu)", frame_stride, gray_frame_stride, need_src,
                    need_dst);
    HIP_TRY(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    switch (kind) {
    case PIGO_PIX_NRGBA: return launch_gray<0>(d_pix, frame_stride, stride, width, height, nframes, d_gray, gray_frame_stride, gray_dim, st);
    case PIGO_PIX_RGBA: return launch_gray<1>(d_pix, frame_stride, stride, width, height, nframes, d_gray, gray_frame_stride, gray_dim, st);
    default: return launch_gray<2>(d_pix, frame_stride, stride, width, height, nframes, d_gray, gray_frame_stride, gray_dim, st);
    }
}

extern "C" pigo_status pigo_rgb_to_grayscale(int device, const uint8_t *pix, size_t npix, int width, int height, int stride, int kind,
                                             uint8_t *gray, size_t cap)
{
    if (kind < PIGO_PIX_NRGBA || kind > PIGO_PIX_CANVAS) return fail(PIGO_ERR_PARAM, "unknown pixel kind %d", kind);
    if (width < 0 || height < 0) return fail(PIGO_ERR_PARAM, "negative width/height");
    const size_t npx = (size_t)width * (size_t)height;
    if (npx == 0) return PIGO_OK;  // make([]uint8, 0)
    if ((long long)stride < 4ll * width) return fail(PIGO_ERR_PARAM, "stride=%d < 4*width=%lld", stride, 4ll * width);
    const size_t need = (size_t)(height - 1) * (size_t)stride + 4u * (size_t)width;
    if (!pix || npix < need)  // src.At(x, y) indexes Pix past its end: the reference panics (grayscale.go:14)
        return fail(PIGO_ERR_PANIC, "len(Pix)=%zu < %zu needed for %dx%d, stride %d", pix ? npix : (size_t)0, need, width, height, stride);
    if (!gray || cap < npx) return fail(PIGO_ERR_CAPACITY, "gray buffer holds %zu bytes, %zu needed", gray ? cap : (size_t)0, npx);
    HIP_TRY(hipSetDevice(device));
    DevBuf<uint8_t> d_src, d_dst;
    if (d_src.alloc((need + 15) & ~(size_t)15) != hipSuccess || d_dst.alloc((npx + 3) & ~(size_t)3) != hipSuccess)
        return fail(PIGO_ERR_NOMEM, "hipMalloc of %zu + %zu bytes failed", need, npx);
    HIP_TRY(hipMemcpy(d_src.p, pix, need, hipMemcpyHostToDevice));
    pigo_status st = pigo_gray_batch(device, d_src.p, need, stride, width, height, kind, 1, d_dst.p, npx, width, nullptr);
    if (st != PIGO_OK) return st;
    HIP_TRY(hipMemcpy(gray, d_dst.p, npx, hipMemcpyDeviceToHost));  // synchronises with the null stream
    return PIGO_OK;
}

// ---- PuplocCascade: UnpackCascade / RunDetector / GetLandmarkPoint (core/puploc.go, core/flploc.go) --------------------

struct pigo_puploc_cascade {
    int device = 0;
    uint32_t stages = 0, trees = 0, depth = 0;
    float scales = 0.0f;
    DevBuf<int8_t> d_codes;
    DevBuf<float> d_preds;
    DevBuf<int32_t> d_flags;
    std::mutex mu;  // serialises the single-request entry points (they share the scratch below)
    DevBuf<uint8_t> d_frame;
    DevBuf<PuplocReq> d_req;
    DevBuf<float> d_rnd, d_pool;
    DevBuf<PuplocOut> d_out;
};

static_assert(sizeof(PuplocReq) == sizeof(pigo_puploc_req) && sizeof(PuplocOut) == sizeof(pigo_puploc), "wire records");

extern "C" pigo_status pigo_puploc_create(const uint8_t *packet, size_t len, int device, pigo_puploc_cascade **out)
{
    if (!out) return fail(PIGO_ERR_PARAM, "out is NULL");
    *out = nullptr;
    if (!packet || len < 16) return fail(PIGO_ERR_PACKET, "UnpackCascade: packet shorter than its 16-byte header (%zu bytes)", len);
    const uint32_t stages = le32(packet), trees = le32(packet + 8), depth = le32(packet + 12);  // puploc.go:51-66
    float scales;
    const uint32_t u = le32(packet + 4);
    memcpy(&scales, &u, 4);
    if (depth < 1 || depth > 14) return fail(PIGO_ERR_PARAM, "UnpackCascade: tree depth %u outside [1, 14]", depth);
    if (stages > 4096 || trees > 4096) return fail(PIGO_ERR_PARAM, "UnpackCascade: %u stages x %u trees is not a plausible cascade", stages, trees);
    const size_t TD = (size_t)1 << depth, ncode = 4 * TD - 4, npred = 2 * TD, ntree = (size_t)stages * trees;
    if (len < 16 + ntree * (ncode + 4 * npred))  // packet[pos : pos+4*depth-4] past the end: the reference panics (puploc.go:75)
        return fail(PIGO_ERR_PACKET, "UnpackCascade: %zu bytes, header needs %zu", len, 16 + ntree * (ncode + 4 * npred));
    std::vector<int8_t> codes(ntree * ncode);
    std::vector<float> preds(ntree * npred);
    size_t pos = 16;
    for (size_t t = 0; t < ntree; ++t) {  // puploc.go:69-93
        memcpy(codes.data() + t * ncode, packet + pos, ncode);
        pos += ncode;
        memcpy(preds.data() + t * npred, packet + pos, 4 * npred);  // little-endian float32, like this host
        pos += 4 * npred;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
        return fail(PIGO_ERR_HIP, "no usable HIP device %d (%d visible): libpigo_hip has no CPU fallback", device, ndev);
    HIP_TRY(hipSetDevice(device));
    std::unique_ptr<pigo_puploc_cascade> c(new pigo_puploc_cascade);
    c->device = device;
    c->stages = stages;
    c->trees = trees;
    c->depth = depth;
    c->scales = scales;
    HIP_TRY(c->d_codes.alloc(codes.size() + 8));  // + slack: the child prefetch of a leaf-level node is never issued, but keep loads in bounds
    HIP_TRY(c->d_preds.alloc(preds.size()));
    HIP_TRY(c->d_flags.alloc(4));
    HIP_TRY(hipMemset(c->d_flags.p, 0, 16));
    if (!codes.empty()) HIP_TRY(hipMemcpy(c->d_codes.p, codes.data(), codes.size(), hipMemcpyHostToDevice));
    if (!preds.empty()) HIP_TRY(hipMemcpy(c->d_preds.p, preds.data(), preds.size() * 4, hipMemcpyHostToDevice));
    *out = c.release();
    return PIGO_OK;
}

extern "C" pigo_status pigo_puploc_info(const pigo_puploc_cascade *c, uint32_t *stages, float *scales, uint32_t *trees, uint32_t *tree_depth)
{
    if (!c) return fail(PIGO_ERR_PARAM, "cascade is NULL");
    if (stages) *stages = c->stages;
    if (scales) *scales = c->scales;
    if (trees) *trees = c->trees;
    if (tree_depth) *tree_depth = c->depth;
    return PIGO_OK;
}

extern "C" void pigo_puploc_destroy(pigo_puploc_cascade *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    delete c;
}

namespace {

pigo_status puploc_launch(pigo_puploc_cascade *c, const uint8_t *d_frames, size_t frame_stride, int nframes, int rows, int cols, int dim,
                          double angle, const PuplocReq *d_reqs, const float *d_rnd, float *d_pool, int n, PuplocOut *d_out, hipStream_t st)
{
    if (rows < 1 || cols < 1) return fail(PIGO_ERR_PANIC, "RunDetector: rows=%d cols=%d (min(nrows-1, ...) indexes before the image)", rows, cols);
    if (dim < cols || rows >= 65536 || dim >= 65536) return fail(PIGO_ERR_PARAM, "RunDetector: need cols <= dim < 65536 and rows < 65536");
    if (nframes < 1 || (nframes > 1 && frame_stride < (size_t)(rows - 1) * dim + cols)) return fail(PIGO_ERR_PARAM, "RunDetector: bad frame batch");
    if (n == 0) return PIGO_OK;
    if (c->stages == 0 || c->trees == 0) return fail(PIGO_ERR_PARAM, "RunDetector: empty cascade");
    PuplocArgs a{};
    a.codes = c->d_codes.p;
    a.preds = c->d_preds.p;
    a.scales = c->scales;
    a.stages = (int)c->stages;
    a.trees = (int)c->trees;
    a.depth = (int)c->depth;
    a.frames = d_frames;
    a.frame_stride = frame_stride;
    a.nrows = rows;
    a.ncols = cols;
    a.dim = dim;
    a.nframes = nframes;
    a.reqs = d_reqs;
    a.rnd = d_rnd;
    a.pool = d_pool;
    a.out = d_out;
    a.flags = c->d_flags.p;
    if (angle > 0.0) {  // puploc.go:252-256
        if (angle > 1.0) angle = 1.0;
        const int k = (int)(32.0 * angle);
        a.qcos = (float)kQCos[k];  // the same 33-entry tables as the face cascade (puploc.go:163-164 == pigo.go:156-157)
        a.qsin = (float)kQSin[k];
        hipLaunchKernelGGL((k_puploc<true>), dim3((unsigned)n), dim3(kPupThreads), 0, st, a);
    } else {
        hipLaunchKernelGGL((k_puploc<false>), dim3((unsigned)n), dim3(kPupThreads), 0, st, a);
    }
    HIP_TRY(hipGetLastError());
    return PIGO_OK;
}

}  // namespace

extern "C" pigo_status pigo_puploc_run_batch(pigo_puploc_cascade *c, const uint8_t *d_frames, size_t frame_stride, int nframes, int rows,
                                             int cols, int dim, double angle, const pigo_puploc_req *d_reqs, const float *d_rnd,
                                             float *d_pool, int n, pigo_puploc *d_out, void *stream)
{
    if (!c) return fail(PIGO_ERR_PARAM, "cascade is NULL");
    if (n < 0) return fail(PIGO_ERR_PARAM, "n < 0");
    if (n > 0 && (!d_frames || !d_reqs || !d_rnd || !d_out)) return fail(PIGO_ERR_PARAM, "NULL device pointer");
    HIP_TRY(hipSetDevice(c->device));
    return puploc_launch(c, d_frames, frame_stride, nframes, rows, cols, dim, angle, reinterpret_cast<const PuplocReq *>(d_reqs), d_rnd, d_pool, n,
                         reinterpret_cast<PuplocOut *>(d_out), (hipStream_t)stream);
}

extern "C" pigo_status pigo_puploc_status(pigo_puploc_cascade *c)
{
    if (!c) return fail(PIGO_ERR_PARAM, "cascade is NULL");
    HIP_TRY(hipSetDevice(c->device));
    int32_t f = 0;
    HIP_TRY(hipMemcpy(&f, c->d_flags.p, 4, hipMemcpyDeviceToHost));
    if (f) {
        HIP_TRY(hipMemset(c->d_flags.p, 0, 4));
        return fail(PIGO_ERR_PANIC, "RunDetector: a request had Perturbs outside [0, 63] or a frame index out of range (the reference panics)");
    }
    return PIGO_OK;
}

extern "C" pigo_status pigo_puploc_run_detector(pigo_puploc_cascade *c, const pigo_puploc *pl, const uint8_t *pixels, size_t npixels, int rows,
                                                int cols, int dim, double angle, int flip_v, const float *rnd, float *pool, pigo_puploc *out)
{
    if (!c || !pl || !out) return fail(PIGO_ERR_PARAM, "NULL argument");
    if (pl->perturbs < 0 || pl->perturbs > kPupPool)  // det.rows[63] = res[0] / det.rows[int(math.Round(-n/2))]: index out of range
        return fail(PIGO_ERR_PANIC, "RunDetector: Perturbs=%d outside [0, 63]", pl->perturbs);
    if (rows < 1 || cols < 1) return fail(PIGO_ERR_PANIC, "RunDetector: rows=%d cols=%d", rows, cols);
    if (dim < cols) return fail(PIGO_ERR_PARAM, "RunDetector: dim=%d < cols=%d", dim, cols);
    const size_t fbytes = (size_t)(rows - 1) * (size_t)dim + (size_t)cols;
    if (!pixels || npixels < fbytes) return fail(PIGO_ERR_PANIC, "RunDetector: len(pixels)=%zu < %zu", pixels ? npixels : (size_t)0, fbytes);
    if (pl->perturbs > 0 && !rnd) return fail(PIGO_ERR_PARAM, "RunDetector: rnd is NULL");
    if (!(std::fabs((double)pl->scale) < 1e6) || std::abs((long long)pl->row) > (1 << 24) || std::abs((long long)pl->col) > (1 << 24))
        return fail(PIGO_ERR_PARAM, "RunDetector: Puploc{%d, %d, %g} outside the supported range", pl->row, pl->col, (double)pl->scale);
    std::lock_guard<std::mutex> lock(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    if (c->d_frame.n < fbytes) HIP_TRY(c->d_frame.alloc(fbytes));
    if (!c->d_req.p) {
        HIP_TRY(c->d_req.alloc(1));
        HIP_TRY(c->d_rnd.alloc(3 * kPupPool));
        HIP_TRY(c->d_pool.alloc(3 * kPupPool));
        HIP_TRY(c->d_out.alloc(1));
    }
    const PuplocReq rq{pl->row, pl->col, pl->scale, pl->perturbs, 0, flip_v ? 1 : 0};
    float hr[3 * kPupPool] = {0};
    // the kernel indexes rnd as [63][3]; Go draws row, col, scale for perturbation 0, then 1, ... -- the same layout
    if (pl->perturbs) memcpy(hr, rnd, sizeof(float) * 3 * (size_t)pl->perturbs);
    HIP_TRY(hipMemcpy(c->d_frame.p, pixels, fbytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_req.p, &rq, sizeof rq, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_rnd.p, hr, sizeof hr, hipMemcpyHostToDevice));
    if (pool) HIP_TRY(hipMemcpy(c->d_pool.p, pool, sizeof(float) * 3 * kPupPool, hipMemcpyHostToDevice));
    pigo_status st = puploc_launch(c, c->d_frame.p, fbytes, 1, rows, cols, dim, angle, c->d_req.p, c->d_rnd.p, pool ? c->d_pool.p : nullptr, 1,
                                   c->d_out.p, nullptr);
    if (st != PIGO_OK) return st;
    PuplocOut o{};
    HIP_TRY(hipMemcpy(&o, c->d_out.p, sizeof o, hipMemcpyDeviceToHost));
    if (pool) HIP_TRY(hipMemcpy(pool, c->d_pool.p, sizeof(float) * 3 * kPupPool, hipMemcpyDeviceToHost));
    out->row = o.row;
    out->col = o.col;
    out->scale = o.scale;
    out->perturbs = 0;  // &Puploc{Row, Col, Scale}: Perturbs stays zero (puploc.go:272-276)
    return PIGO_OK;
}

extern "C" pigo_status pigo_get_landmark_point(pigo_puploc_cascade *c, const pigo_puploc *left_eye, const pigo_puploc *right_eye,
                                               const uint8_t *pixels, size_t npixels, int rows, int cols, int dim, int perturb, int flip_v,
                                               const float *rnd, float *pool, pigo_puploc *out)
{
    if (!left_eye || !right_eye) return fail(PIGO_ERR_PARAM, "NULL eye");
    // flploc.go:37-51, in Go's int (64-bit) and float64
    const long long dx = ((long long)left_eye->row - right_eye->row) * ((long long)left_eye->row - right_eye->row);
    const long long dy = ((long long)left_eye->col - right_eye->col) * ((long long)left_eye->col - right_eye->col);
    const double dist = std::sqrt((double)(dx + dy));
    const double row = (double)((long long)left_eye->row + right_eye->row) / 2.0 + 0.25 * dist;
    const double col = (double)((long long)left_eye->col + right_eye->col) / 2.0 + 0.15 * dist;
    const double scale = 3.0 * dist;
    pigo_puploc flploc;
    flploc.row = (int32_t)(long long)row;
    flploc.col = (int32_t)(long long)col;
    flploc.scale = (float)scale;
    flploc.perturbs = perturb;
    return pigo_puploc_run_detector(c, &flploc, pixels, npixels, rows, cols, dim, 0.0, flip_v, rnd, pool, out);  // flploc.go:53-56
}

// ---- multi-GPU: frames sharded over ranks, one RCCL all-gather of the per-frame lists ---------------------
//
// The reference has no counterpart (RunCascade is a single goroutine, core/pigo.go:212-258); this is north_star's batch
// configuration: one process per GPU, contiguous shards of independent frames, no exchange during the scan, ONE collective at
// the end.  librccl is bound at run time (dlopen) so that single-GPU users of libpigo_hip.so do not need it.

#include <dlfcn.h>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <thread>

namespace {

struct RcclUniqueId {
    char internal[PIGO_COMM_ID_BYTES];  // == ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES 128)
};
typedef void *rccl_comm_t;
typedef int (*fn_ncclGetUniqueId)(RcclUniqueId *);
typedef int (*fn_ncclCommInitRank)(rccl_comm_t *, int, RcclUniqueId, int);
typedef int (*fn_ncclCommDestroy)(rccl_comm_t);
typedef int (*fn_ncclCommAbort)(rccl_comm_t);
typedef int (*fn_ncclAllGather)(const void *, void *, size_t, int, rccl_comm_t, hipStream_t);
typedef const char *(*fn_ncclGetErrorString)(int);
constexpr int kNcclInt32 = 2;  // ncclDataType_t: ncclInt8 0, ncclUint8 1, ncclInt32 2 (rccl.h)

struct Rccl {
    void *h = nullptr;
    fn_ncclGetUniqueId get_id = nullptr;
    fn_ncclCommInitRank init_rank = nullptr;
    fn_ncclCommDestroy destroy = nullptr;
    fn_ncclCommAbort abort = nullptr;
    fn_ncclAllGather all_gather = nullptr;
    fn_ncclGetErrorString err = nullptr;
};

std::mutex g_rccl_mu;
Rccl g_rccl;

pigo_status rccl_load(const Rccl **out)
{
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    if (!g_rccl.h) {
        // a process that already has RCCL (e.g. PyTorch's bundled copy, same soname) shares it; PIGO_RCCL_LIB overrides
        const char *names[] = {getenv("PIGO_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        void *h = nullptr;
        for (const char *n : names)
            if (n && *n && (h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!h) return fail(PIGO_ERR_HIP, "librccl not found (%s)", dlerror());
        Rccl r;
        r.h = h;
        r.get_id = (fn_ncclGetUniqueId)dlsym(h, "ncclGetUniqueId");
        r.init_rank = (fn_ncclCommInitRank)dlsym(h, "ncclCommInitRank");
        r.destroy = (fn_ncclCommDestroy)dlsym(h, "ncclCommDestroy");
        r.abort = (fn_ncclCommAbort)dlsym(h, "ncclCommAbort");
        r.all_gather = (fn_ncclAllGather)dlsym(h, "ncclAllGather");
        r.err = (fn_ncclGetErrorString)dlsym(h, "ncclGetErrorString");
        if (!r.get_id || !r.init_rank || !r.destroy || !r.all_gather) return fail(PIGO_ERR_HIP, "librccl lacks a required symbol");
        g_rccl = r;
    }
    *out = &g_rccl;
    return PIGO_OK;
}

#define RCCL_TRY(r, expr)                                                                                   \
    do {                                                                                                    \
        int e_ = (expr);                                                                                    \
        if (e_ != 0) return fail(PIGO_ERR_HIP, "%s: %s", #expr, (r)->err ? (r)->err(e_) : "rccl error");   \
    } while (0)

}  // namespace

struct pigo_comm {
    int rank = 0, world = 1, device = 0;
    rccl_comm_t comm = nullptr;  // NULL when world == 1: nothing to exchange
    bool aborted = false;        // pigo_comm_abort was called: every further collective on it is refused
    std::mutex mu;               // comm / aborted (the two words only, never held across an RCCL call): pigo_comm_abort may come from another thread
};

extern "C" pigo_status pigo_comm_unique_id(uint8_t id[PIGO_COMM_ID_BYTES])
{
    if (!id) return fail(PIGO_ERR_PARAM, "id is NULL");
    const Rccl *r = nullptr;
    pigo_status st = rccl_load(&r);
    if (st != PIGO_OK) return st;
    RcclUniqueId u;
    RCCL_TRY(r, r->get_id(&u));
    memcpy(id, u.internal, PIGO_COMM_ID_BYTES);
    return PIGO_OK;
}

extern "C" pigo_status pigo_comm_init(const uint8_t id[PIGO_COMM_ID_BYTES], int rank, int world, int device, pigo_comm **out)
{
    if (!out) return fail(PIGO_ERR_PARAM, "out is NULL");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return fail(PIGO_ERR_PARAM, "rank %d outside [0, %d)", rank, world);
    std::unique_ptr<pigo_comm> c(new (std::nothrow) pigo_comm);
    if (!c) return fail(PIGO_ERR_NOMEM, "out of memory");
    c->rank = rank;
    c->world = world;
    c->device = device;
    // world > 1 needs the id; world == 1 WITH an id builds a real one-rank RCCL communicator too (the caller asked for one by
    // calling pigo_comm_unique_id): the collective of pigo_run_batch_sharded then runs through ncclAllGather exactly as on a
    // multi-GPU node, which is how a single-GPU box tests that path.  world == 1 without an id needs no RCCL at all.
    if (world > 1 && !id) return fail(PIGO_ERR_PARAM, "id is NULL");
    if (id) {
        const Rccl *r = nullptr;
        pigo_status st = rccl_load(&r);
        if (st != PIGO_OK) return st;
        HIP_TRY(hipSetDevice(device));
        RcclUniqueId u;
        memcpy(u.internal, id, PIGO_COMM_ID_BYTES);
        // ncclCommInitRank blocks until EVERY rank has called it: a peer that died on the way here (no device, bad plan, crashed
        // process) would hang this rank forever.  The call therefore runs on a helper thread and this one waits for it with a
        // deadline (PIGO_COMM_INIT_TIMEOUT_S, default 300 s; 0 = wait in place, no deadline).  After a timeout the communicator
        // is unusable and the helper thread stays blocked inside RCCL: the host reports the error and ends the process.
        const int timeout_s = std::max(0, env_int("PIGO_COMM_INIT_TIMEOUT_S", 300));
        if (timeout_s == 0) {
            RCCL_TRY(r, r->init_rank(&c->comm, world, u, rank));
        } else {
            struct InitState {
                std::mutex mu;
                std::condition_variable cv;
                bool done = false, abandoned = false;
                int rc = 0;
                rccl_comm_t comm = nullptr;
            };
            std::shared_ptr<InitState> stt = std::make_shared<InitState>();
            const fn_ncclCommInitRank init = r->init_rank;
            const auto abort_fn = r->abort;
            std::thread([stt, init, abort_fn, world, u, rank, device]() {
                (void)hipSetDevice(device);
                rccl_comm_t cm = nullptr;
                const int rc = init(&cm, world, u, rank);
                bool late = false;
                {
                    std::lock_guard<std::mutex> lock(stt->mu);
                    stt->rc = rc;
                    stt->comm = cm;
                    stt->done = true;
                    late = stt->abandoned;
                    stt->cv.notify_all();
                }
                if (late && rc == 0 && cm && abort_fn) (void)abort_fn(cm);  // the waiter has given up: nobody owns this communicator
            }).detach();
            std::unique_lock<std::mutex> lock(stt->mu);
            if (!stt->cv.wait_for(lock, std::chrono::seconds(timeout_s), [&] { return stt->done; })) {
                stt->abandoned = true;
                return fail(PIGO_ERR_HIP, "pigo_comm_init: rank %d of %d waited %d s for its peers in ncclCommInitRank (PIGO_COMM_INIT_TIMEOUT_S); "
                                          "a peer did not join -- this communicator cannot be used", rank, world, timeout_s);
            }
            if (stt->rc != 0) return fail(PIGO_ERR_HIP, "ncclCommInitRank: %s", r->err ? r->err(stt->rc) : "rccl error");
            c->comm = stt->comm;
        }
    }
    *out = c.release();
    return PIGO_OK;
}

// ncclCommAbort: frees the communicator WITHOUT waiting for its outstanding collectives -- what a host calls (from any thread)
// when a peer has failed and pigo_run_batch_sharded's all-gather would otherwise never complete.  The handle stays valid for
// pigo_comm_destroy only.
extern "C" pigo_status pigo_comm_abort(pigo_comm *c)
{
    if (!c) return fail(PIGO_ERR_PARAM, "comm is NULL");
    // The lock only covers the handle's two words.  ncclCommAbort itself runs WITHOUT it: it is what a host calls while another
    // thread sits inside an ncclAllGather enqueue that blocks on the host (lazy connection setup of the first collective, a full
    // proxy queue, a dead peer) -- that thread does not hold the lock either (pigo_run_batch_sharded), and ncclCommAbort is made to
    // run next to a stuck collective, which then returns an error to its caller.
    rccl_comm_t h = nullptr;
    {
        std::lock_guard<std::mutex> lock(c->mu);
        if (!c->comm) return PIGO_OK;  // world 1 without an id: nothing in flight (or aborted already)
        if (!g_rccl.abort) return fail(PIGO_ERR_HIP, "librccl has no ncclCommAbort");
        h = c->comm;
        c->comm = nullptr;
        c->aborted = true;
    }
    (void)hipSetDevice(c->device);
    const int rc = g_rccl.abort(h);
    if (rc != 0) return fail(PIGO_ERR_HIP, "ncclCommAbort: %s", g_rccl.err ? g_rccl.err(rc) : "rccl error");
    return PIGO_OK;
}

extern "C" pigo_status pigo_comm_info(const pigo_comm *c, int *rank, int *world)
{
    if (!c) return fail(PIGO_ERR_PARAM, "comm is NULL");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    return PIGO_OK;
}

extern "C" int pigo_comm_uses_rccl(const pigo_comm *c) { return c && c->comm ? 1 : 0; }

extern "C" void pigo_comm_destroy(pigo_comm *c)
{
    if (!c) return;
    if (c->comm && g_rccl.destroy) {
        (void)hipSetDevice(c->device);
        (void)g_rccl.destroy(c->comm);
    }
    delete c;
}

extern "C" void pigo_shard_bounds(int nframes, int rank, int world, int *lo, int *hi)
{
    if (world < 1) world = 1;
    if (nframes < 0) nframes = 0;
    const int base = nframes / world, rem = nframes % world;
    const int l = rank * base + std::min(rank, rem);
    if (lo) *lo = l;
    if (hi) *hi = l + base + (rank < rem ? 1 : 0);
}

extern "C" size_t pigo_wire_words(int gather_cap) { return gather_cap < 0 ? 0 : 2 + 4 * (size_t)gather_cap; }

extern "C" int pigo_wire_row_flags(const int32_t *wire_row) { return wire_row ? wire_row[1] : 0; }

extern "C" pigo_status pigo_pack_lists(const pigo_det *lists, const int32_t *counts, int nframes, int frames_out, int cap, int gather_cap,
                                       int32_t *wire)
{
    if (nframes < 0 || frames_out < nframes || cap < 0 || gather_cap < 0) return fail(PIGO_ERR_PARAM, "bad list geometry");
    if (frames_out > 0 && !wire) return fail(PIGO_ERR_PARAM, "wire is NULL");
    if (nframes > 0 && (!counts || (!lists && cap > 0))) return fail(PIGO_ERR_PARAM, "NULL list pointer");
    const size_t words = pigo_wire_words(gather_cap);
    for (int f = 0; f < frames_out; ++f) {
        int32_t *row = wire + (size_t)f * words;
        memset(row, 0, words * 4);
        if (f >= nframes) {
            row[1] = PIGO_WIRE_PADDING;
            continue;
        }
        row[0] = counts[f];
        row[1] = (counts[f] > gather_cap ? PIGO_WIRE_TRUNCATED_GATHER : 0) | (counts[f] > cap ? PIGO_WIRE_TRUNCATED_DETCAP : 0);
        const int n = std::max(0, std::min(std::min(counts[f], cap), gather_cap));
        if (n) memcpy(row + 2, lists + (size_t)f * cap, (size_t)n * sizeof(pigo_det));
    }
    return PIGO_OK;
}

extern "C" pigo_status pigo_unpack_list(const int32_t *wire_row, int gather_cap, pigo_det *out, int cap, int *n_out, int *true_count)
{
    if (!wire_row || gather_cap < 0) return fail(PIGO_ERR_PARAM, "bad wire row");
    const int cnt = wire_row[0];
    const int n = std::max(0, std::min(cnt, gather_cap));
    if (true_count) *true_count = cnt;
    if (n_out) *n_out = n;
    if (n > cap) return fail(PIGO_ERR_CAPACITY, "unpack: %d records, capacity %d", n, cap);
    if (n && !out) return fail(PIGO_ERR_PARAM, "out is NULL");
    if (n) memcpy(out, wire_row + 2, (size_t)n * sizeof(pigo_det));
    return PIGO_OK;
}

extern "C" pigo_status pigo_run_batch_sharded(pigo_plan *p, pigo_comm *comm, const uint8_t *d_frames, size_t frame_stride, int nframes_local,
                                              int frames_per_rank, double iou_threshold, int gather_cap, int32_t *d_gathered, void *stream)
{
    if (!p) return fail(PIGO_ERR_PARAM, "plan is NULL");
    if (nframes_local < 0 || nframes_local > p->max_frames || frames_per_rank < nframes_local || frames_per_rank < 1)
        return fail(PIGO_ERR_PARAM, "need 0 <= nframes_local <= min(max_frames, frames_per_rank)");
    if (gather_cap < 1 || gather_cap > p->det_cap) return fail(PIGO_ERR_PARAM, "gather_cap outside [1, det_cap]");
    if (!d_gathered) return fail(PIGO_ERR_PARAM, "d_gathered is NULL");
    const int world = comm ? comm->world : 1, rank = comm ? comm->rank : 0;
    {
        std::unique_lock<std::mutex> cl;
        if (comm) cl = std::unique_lock<std::mutex>(comm->mu);
        if (comm && comm->aborted) return fail(PIGO_ERR_PARAM, "the communicator was aborted (pigo_comm_abort): destroy it and build a new one");
    }
    if (comm && comm->device != p->c->device) return fail(PIGO_ERR_PARAM, "communicator and plan live on different devices");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(p->c->device));
    const size_t words = pigo_wire_words(gather_cap);
    {
        std::lock_guard<std::mutex> lock(p->mu);
        const size_t need = (size_t)p->max_frames * p->det_cap;
        if (p->sh_dets.n < need) HIP_TRY(p->sh_dets.alloc(need));
        if (p->sh_counts.n < (size_t)p->max_frames * 2) HIP_TRY(p->sh_counts.alloc((size_t)p->max_frames * 2));
        if (p->sh_wire.n < (size_t)frames_per_rank * words) HIP_TRY(p->sh_wire.alloc((size_t)frames_per_rank * words));
    }
    const bool clustered = iou_threshold == iou_threshold && iou_threshold >= 0.0;  // NaN / negative: gather the raw RunCascade lists
    if (clustered) {
        std::lock_guard<std::mutex> lock(p->mu);
        const size_t need = (size_t)p->max_frames * p->det_cap;
        if (p->sh_sorted.n < need) HIP_TRY(p->sh_sorted.alloc(need));
        if (p->sh_clusters.n < need) HIP_TRY(p->sh_clusters.alloc(need));
    }
    int32_t *d_counts = p->sh_counts.p, *d_ccounts = p->sh_counts.p + p->max_frames;
    const pigo_det *lists = p->sh_dets.p;
    const int32_t *lcounts = d_counts;
    const size_t row_words = (size_t)frames_per_rank * words;
    // From here on a failure must not leave the peers alone in the collective (they have enqueued their ncclAllGather and
    // would wait for this rank forever): the rank still contributes its rows -- all zero-count padding -- and then reports
    // the error.  A non-OK return of the collective itself is fatal for the communicator (destroy it).
    pigo_status st = PIGO_OK;
    if (nframes_local > 0) {
        st = pigo_plan_run(p, d_frames, frame_stride, nframes_local, p->sh_dets.p, d_counts, stream);
        if (st == PIGO_OK && clustered) {
            st = pigo_plan_cluster(p, p->sh_dets.p, d_counts, nframes_local, iou_threshold, p->sh_sorted.p, p->sh_clusters.p, d_ccounts, nullptr,
                                   stream);
            lists = p->sh_clusters.p;
            lcounts = d_ccounts;
        }
    }
    const std::string first_error = st != PIGO_OK ? g_last_error : std::string();
    if (st == PIGO_OK) {
        // (the rows carry the plan's device flags as the scan in front of this launch left them: a peer sees an overflowed queue or
        // a list cut at det_cap on THIS rank in the rows themselves, without trusting this host to call pigo_plan_status)
        k_pack_lists<<<frames_per_rank, 256, 0, s>>>(lists, lcounts, clustered ? d_counts : nullptr, p->d_flags.p, nframes_local, p->det_cap, gather_cap, 0,
                                                     p->sh_wire.p);
        if (hipGetLastError() != hipSuccess) st = fail(PIGO_ERR_HIP, "k_pack_lists launch failed");
    }
    if (st != PIGO_OK) {  // zero-count padding rows that say why: PIGO_WIRE_RANK_FAILED
        k_pack_lists<<<frames_per_rank, 256, 0, s>>>(p->sh_dets.p, d_counts, nullptr, nullptr, 0, p->det_cap, gather_cap, 1, p->sh_wire.p);
        if (hipGetLastError() != hipSuccess) (void)hipMemsetAsync(p->sh_wire.p, 0, row_words * 4, s);
    }
    // The handle is read under the communicator's lock, the enqueue runs without it: an RCCL enqueue can block on the host, and
    // pigo_comm_abort -- which exists for exactly that state -- must be able to run from another thread (it then makes this call
    // return with RCCL's error).
    rccl_comm_t handle = nullptr;
    if (comm) {
        std::lock_guard<std::mutex> lock(comm->mu);
        if (comm->aborted) return fail(PIGO_ERR_PARAM, "the communicator was aborted (pigo_comm_abort) while this call was preparing its rows");
        handle = comm->comm;
    }
    if (handle) {
        const Rccl *r = nullptr;
        const pigo_status rs = rccl_load(&r);
        if (rs != PIGO_OK) return rs;
        RCCL_TRY(r, r->all_gather(p->sh_wire.p, d_gathered, row_words, kNcclInt32, handle, s));
    } else {
        if (world > 1) return fail(PIGO_ERR_PARAM, "communicator of %d ranks without an RCCL handle", world);
        HIP_TRY(hipMemcpyAsync(d_gathered + (size_t)rank * row_words, p->sh_wire.p, row_words * 4, hipMemcpyDeviceToDevice, s));
    }
    if (st != PIGO_OK) {
        g_last_error = first_error.empty() ? g_last_error : first_error;
        return st;
    }
    return PIGO_OK;
}
