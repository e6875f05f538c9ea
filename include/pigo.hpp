// pigo.hpp -- header-only C++17 mirror of the reference's Go API (package github.com/esimov/pigo/core) over the
// C ABI of libpigo_hip.so (include/pigo_hip.h).  The Go toolchain is not available in this image, so this is
// the host side "above the C ABI" in the compiled-language form the reference itself has; INTEGRATION.md shows
// the equivalent cgo shim.  Names, argument meaning and error behaviour follow core/pigo.go:
//
//     pigo::Pigo pg = pigo::NewPigo().Unpack(cascade_bytes);            // core/pigo.go:46,51
//     pigo::CascadeParams cp{ {pixels, rows, cols, cols}, 20, 1000, 0.1, 1.1 };
//     std::vector<pigo::Detection> dets = pg.RunCascade(cp, 0.0);        // core/pigo.go:212
//     dets = pg.ClusterDetections(dets, 0.2);                            // core/pigo.go:262 (sorts `dets` in place)
//
// Where the Go code would panic (short cascade packet, pixel index out of range) a pigo::Panic is thrown; other
// failures throw std::runtime_error / std::invalid_argument.  There is no CPU fallback.
#ifndef PIGO_HPP
#define PIGO_HPP

#include <cstdint>
#include <array>
#include <memory>
#include <utility>
#include <stdexcept>
#include <string>
#include <functional>
#include <random>
#include <vector>

#include "pigo_hip.h"

namespace pigo {

struct Panic : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// ImageParams, core/pigo.go:29-34
struct ImageParams {
    const std::vector<uint8_t> *Pixels = nullptr;  // row-major gray, stride Dim; owned by the caller, only read
    int Rows = 0, Cols = 0, Dim = 0;
};

// CascadeParams, core/pigo.go:16-22
struct CascadeParams {
    pigo::ImageParams ImageParams;
    int MinSize = 0, MaxSize = 0;
    double ShiftFactor = 0.0, ScaleFactor = 0.0;
};

// Detection, core/pigo.go:195-200
struct Detection {
    int Row = 0, Col = 0, Scale = 0;
    float Q = 0.0f;
};

namespace detail {
inline void check(pigo_status st, const char *what)
{
    if (st == PIGO_OK) return;
    const std::string msg = std::string(what) + ": " + pigo_last_error();
    if (st == PIGO_ERR_PACKET || st == PIGO_ERR_PANIC) throw Panic(msg);
    if (st == PIGO_ERR_PARAM) throw std::invalid_argument(msg);
    throw std::runtime_error(msg);
}
struct Deleter {
    void operator()(pigo_cascade *c) const { pigo_cascade_destroy(c); }
};
}  // namespace detail

class Pigo {
public:
    Pigo() = default;
    explicit Pigo(int device) : device_(device) {}

    // Unpack returns a NEW Pigo and leaves the receiver untouched (core/pigo.go:51,103-109)
    Pigo Unpack(const std::vector<uint8_t> &packet) const
    {
        pigo_cascade *c = nullptr;
        detail::check(pigo_cascade_create(packet.data(), packet.size(), device_, &c), "Unpack");
        Pigo out(device_);
        out.h_.reset(c, detail::Deleter());
        return out;
    }

    // RunCascade, core/pigo.go:212-258: detections with q > 0 in scale-major / row / col order
    std::vector<Detection> RunCascade(const CascadeParams &cp, double angle) const
    {
        need();
        const ImageParams &ip = cp.ImageParams;
        if (!ip.Pixels) throw std::invalid_argument("RunCascade: ImageParams.Pixels is null");
        std::vector<pigo_det> buf(1024);
        for (;;) {
            int n = 0;
            pigo_status st = pigo_run_cascade(h_.get(), ip.Pixels->data(), ip.Pixels->size(), ip.Rows, ip.Cols, ip.Dim, cp.MinSize,
                                              cp.MaxSize, cp.ShiftFactor, cp.ScaleFactor, angle, buf.data(), (int)buf.size(), &n);
            if (st == PIGO_ERR_CAPACITY && n > (int)buf.size()) {
                buf.resize((size_t)n);
                continue;
            }
            detail::check(st, "RunCascade");
            std::vector<Detection> out((size_t)n);
            for (int i = 0; i < n; ++i) out[(size_t)i] = Detection{buf[(size_t)i].row, buf[(size_t)i].col, buf[(size_t)i].scale, buf[(size_t)i].q};
            return out;  // empty vector where Go returns a nil slice (pigo.go:214)
        }
    }

    // ClusterDetections, core/pigo.go:262-308: sorts `detections` in place (like the reference), returns the clusters
    std::vector<Detection> ClusterDetections(std::vector<Detection> &detections, double iouThreshold) const
    {
        need();
        const int n = (int)detections.size();
        std::vector<pigo_det> in((size_t)n), out((size_t)(n > 0 ? n : 1));
        for (int i = 0; i < n; ++i) in[(size_t)i] = pigo_det{detections[(size_t)i].Row, detections[(size_t)i].Col, detections[(size_t)i].Scale, detections[(size_t)i].Q};
        int k = 0;
        detail::check(pigo_cluster_detections(h_.get(), in.data(), n, iouThreshold, out.data(), (int)out.size(), &k), "ClusterDetections");
        for (int i = 0; i < n; ++i) detections[(size_t)i] = Detection{in[(size_t)i].row, in[(size_t)i].col, in[(size_t)i].scale, in[(size_t)i].q};
        std::vector<Detection> clusters((size_t)k);
        for (int i = 0; i < k; ++i) clusters[(size_t)i] = Detection{out[(size_t)i].row, out[(size_t)i].col, out[(size_t)i].scale, out[(size_t)i].q};
        return clusters;
    }

    uint32_t treeDepth() const
    {
        need();
        uint32_t d = 0, n = 0;
        detail::check(pigo_cascade_info(h_.get(), &d, &n), "info");
        return d;
    }
    uint32_t treeNum() const
    {
        need();
        uint32_t d = 0, n = 0;
        detail::check(pigo_cascade_info(h_.get(), &d, &n), "info");
        return n;
    }
    pigo_cascade *handle() const { return h_.get(); }  // for the batch extension (pigo_plan_*)

private:
    void need() const
    {
        if (!h_) throw std::runtime_error("Pigo is not unpacked (call Unpack first)");
    }
    int device_ = 0;
    std::shared_ptr<pigo_cascade> h_;  // immutable after Unpack, shareable between threads like the Go struct
};

// The pixel buffer of a Go image (image.NRGBA / image.RGBA: Pix, Stride, Rect), as RgbToGrayscale's argument
struct Image {
    const std::vector<uint8_t> *Pix = nullptr;  // {R,G,B,A} bytes, row-major, pixel (0,0) first
    int Stride = 0, Width = 0, Height = 0;
    int Kind = PIGO_PIX_NRGBA;                  // PIGO_PIX_NRGBA (GetImage's result) | PIGO_PIX_RGBA | PIGO_PIX_CANVAS
};

// RgbToGrayscale, core/grayscale.go:8-23
inline std::vector<uint8_t> RgbToGrayscale(const Image &src, int device = 0)
{
    std::vector<uint8_t> gray((size_t)src.Width * (size_t)src.Height);
    if (gray.empty()) return gray;
    if (!src.Pix) throw std::invalid_argument("RgbToGrayscale: Image.Pix is null");
    detail::check(pigo_rgb_to_grayscale(device, src.Pix->data(), src.Pix->size(), src.Width, src.Height, src.Stride, src.Kind, gray.data(),
                                        gray.size()),
                  "RgbToGrayscale");
    return gray;
}

// ---- pupil / facial-landmark localisation: core/puploc.go, core/flploc.go -----------------------------------------------

// Puploc, core/puploc.go:14-19
struct Puploc {
    int Row = 0, Col = 0;
    float Scale = 0.0f;
    int Perturbs = 0;
};

// The two inputs of RunDetector that the reference takes from process-global state (see include/pigo_hip.h):
// the rand.Float32() stream and the sync.Pool object.  A default-constructed DetectorState draws from std::mt19937 and
// keeps one pool object, which is what a single-goroutine Go program sees.
struct DetectorState {
    std::function<float()> Float32;   // next value of the uniform [0,1) stream (rand.Float32(), puploc.go:248-250)
    float Pool[3 * 63] = {0};         // rows | cols | scale of the sync.Pool object (puploc.go:228-237)
    bool UsePool = true;              // false: every call sees a brand-new pool object
};

class PuplocCascade {
public:
    explicit PuplocCascade(int device = 0) : device_(device) {}

    // UnpackCascade, core/puploc.go:38-103: returns a NEW cascade
    PuplocCascade UnpackCascade(const std::vector<uint8_t> &packet) const
    {
        pigo_puploc_cascade *h = nullptr;
        detail::check(pigo_puploc_create(packet.data(), packet.size(), device_, &h), "UnpackCascade");
        PuplocCascade out(device_);
        out.h_ = std::shared_ptr<pigo_puploc_cascade>(h, [](pigo_puploc_cascade *c) { pigo_puploc_destroy(c); });
        return out;
    }

    // RunDetector, core/puploc.go:239-277
    Puploc RunDetector(const Puploc &pl, const ImageParams &img, double angle, bool flipV, DetectorState &st) const
    {
        need();
        if (!img.Pixels) throw std::invalid_argument("RunDetector: ImageParams.Pixels is null");
        const std::vector<float> rnd = draw(pl.Perturbs, st);
        const pigo_puploc in{pl.Row, pl.Col, pl.Scale, pl.Perturbs};
        pigo_puploc out{};
        detail::check(pigo_puploc_run_detector(h_.get(), &in, img.Pixels->data(), img.Pixels->size(), img.Rows, img.Cols, img.Dim, angle,
                                               flipV ? 1 : 0, rnd.data(), st.UsePool ? st.Pool : nullptr, &out),
                      "RunDetector");
        return Puploc{out.row, out.col, out.scale, 0};
    }

    // GetLandmarkPoint, core/flploc.go:36-57
    Puploc GetLandmarkPoint(const Puploc &leftEye, const Puploc &rightEye, const ImageParams &img, int perturb, bool flipV,
                            DetectorState &st) const
    {
        need();
        if (!img.Pixels) throw std::invalid_argument("GetLandmarkPoint: ImageParams.Pixels is null");
        const std::vector<float> rnd = draw(perturb, st);
        const pigo_puploc l{leftEye.Row, leftEye.Col, leftEye.Scale, 0}, r{rightEye.Row, rightEye.Col, rightEye.Scale, 0};
        pigo_puploc out{};
        detail::check(pigo_get_landmark_point(h_.get(), &l, &r, img.Pixels->data(), img.Pixels->size(), img.Rows, img.Cols, img.Dim, perturb,
                                              flipV ? 1 : 0, rnd.data(), st.UsePool ? st.Pool : nullptr, &out),
                      "GetLandmarkPoint");
        return Puploc{out.row, out.col, out.scale, 0};
    }

    pigo_puploc_cascade *handle() const { return h_.get(); }  // for pigo_puploc_run_batch

private:
    static std::vector<float> draw(int perturbs, DetectorState &st)
    {
        std::vector<float> rnd(3 * (size_t)(perturbs > 0 ? perturbs : 0) + 3, 0.0f);
        if (!st.Float32) {
            auto gen = std::make_shared<std::mt19937>(std::random_device{}());
            st.Float32 = [gen]() { return (float)((*gen)() >> 8) / 16777216.0f; };
        }
        for (int i = 0; i < 3 * perturbs && i < 3 * 64; ++i) rnd[(size_t)i] = st.Float32();  // row, col, scale of perturbation 0, 1, ...
        return rnd;
    }
    void need() const
    {
        if (!h_) throw std::runtime_error("PuplocCascade is not unpacked (call UnpackCascade first)");
    }
    int device_ = 0;
    std::shared_ptr<pigo_puploc_cascade> h_;
};

// NewPuplocCascade, core/puploc.go:32-34
inline PuplocCascade NewPuplocCascade(int device = 0) { return PuplocCascade(device); }

// NewPigo, core/pigo.go:46
inline Pigo NewPigo(int device = 0) { return Pigo(device); }

// ---- batch / multi-GPU extension (no reference counterpart; pigo_hip.h "batch" and "multi-GPU" sections) ----------------
// Thin owners over the C handles for hosts that keep frames in device memory.  Pointers named d_* are device pointers the
// caller allocated with HIP; `stream` is a hipStream_t (nullptr = default stream).

// One RCCL communicator per process/GPU: rank 0 calls Comm::UniqueId() and ships the bytes to the other ranks.
class Comm {
  public:
    using Id = std::array<uint8_t, PIGO_COMM_ID_BYTES>;
    static Id UniqueId()
    {
        Id id{};
        detail::check(pigo_comm_unique_id(id.data()), "pigo_comm_unique_id");
        return id;
    }
    // world == 1 needs no id (and no RCCL); world == 1 WITH an id builds a real one-rank RCCL communicator
    Comm(const Id *id, int rank, int world, int device)
    {
        pigo_comm *c = nullptr;
        detail::check(pigo_comm_init(id ? id->data() : nullptr, rank, world, device, &c), "pigo_comm_init");
        h_.reset(c, [](pigo_comm *p) { pigo_comm_destroy(p); });
    }
    int Rank() const
    {
        int r = 0, w = 0;
        detail::check(pigo_comm_info(h_.get(), &r, &w), "pigo_comm_info");
        return r;
    }
    int World() const
    {
        int r = 0, w = 0;
        detail::check(pigo_comm_info(h_.get(), &r, &w), "pigo_comm_info");
        return w;
    }
    bool UsesRccl() const { return pigo_comm_uses_rccl(h_.get()) != 0; }
    // ncclCommAbort: a peer failed and the all-gather would never complete; the object can only be destroyed afterwards
    void Abort() { detail::check(pigo_comm_abort(h_.get()), "pigo_comm_abort"); }
    pigo_comm *handle() const { return h_.get(); }

  private:
    std::shared_ptr<pigo_comm> h_;
};

// contiguous shard [lo, hi) of `nframes` frames for `rank`
inline std::pair<int, int> ShardBounds(int nframes, int rank, int world)
{
    int lo = 0, hi = 0;
    pigo_shard_bounds(nframes, rank, world, &lo, &hi);
    return {lo, hi};
}

// one row of the all-gathered wire format -> the frame's (possibly truncated) list, its true length and its PIGO_WIRE_* flags
inline std::vector<Detection> UnpackList(const int32_t *wire_row, int gather_cap, int *true_count = nullptr, int *flags = nullptr)
{
    if (flags) *flags = pigo_wire_row_flags(wire_row);
    std::vector<pigo_det> raw((size_t)(gather_cap > 0 ? gather_cap : 1));
    int n = 0, tc = 0;
    detail::check(pigo_unpack_list(wire_row, gather_cap, raw.data(), (int)raw.size(), &n, &tc), "pigo_unpack_list");
    if (true_count) *true_count = tc;
    std::vector<Detection> out((size_t)n);
    for (int i = 0; i < n; ++i) out[(size_t)i] = Detection{raw[(size_t)i].row, raw[(size_t)i].col, raw[(size_t)i].scale, raw[(size_t)i].q};
    return out;
}

// A scan plan: fixed CascadeParams geometry + angle, workspace for max_frames device-resident frames.
class Plan {
  public:
    Plan(const Pigo &pg, int rows, int cols, int dim, int minSize, int maxSize, double shiftFactor, double scaleFactor, double angle,
         int maxFrames, int detCap)
    {
        pigo_plan *p = nullptr;
        detail::check(pigo_plan_create(pg.handle(), rows, cols, dim, minSize, maxSize, shiftFactor, scaleFactor, angle, maxFrames, detCap, &p),
                      "pigo_plan_create");
        h_.reset(p, [](pigo_plan *q) { pigo_plan_destroy(q); });
    }
    // RunCascade on every frame (asynchronous; call Status() after synchronising the stream)
    void Run(const uint8_t *d_frames, size_t frame_stride, int nframes, pigo_det *d_dets, int32_t *d_counts, void *stream = nullptr) const
    {
        detail::check(pigo_plan_run(h_.get(), d_frames, frame_stride, nframes, d_dets, d_counts, stream), "pigo_plan_run");
    }
    // ClusterDetections on every frame's list
    void Cluster(const pigo_det *d_dets, const int32_t *d_counts, int nframes, double iouThreshold, pigo_det *d_sorted, pigo_det *d_clusters,
                 int32_t *d_ccounts, int32_t *d_ties = nullptr, void *stream = nullptr) const
    {
        detail::check(pigo_plan_cluster(h_.get(), d_dets, d_counts, nframes, iouThreshold, d_sorted, d_clusters, d_ccounts, d_ties, stream),
                      "pigo_plan_cluster");
    }
    // this rank's shard: scan + cluster + one all-gather of the wire rows of all ranks into d_gathered
    void RunBatchSharded(const Comm &comm, const uint8_t *d_frames, size_t frame_stride, int nframes_local, int frames_per_rank,
                         double iouThreshold, int gather_cap, int32_t *d_gathered, void *stream = nullptr) const
    {
        detail::check(pigo_run_batch_sharded(h_.get(), comm.handle(), d_frames, frame_stride, nframes_local, frames_per_rank, iouThreshold,
                                             gather_cap, d_gathered, stream),
                      "pigo_run_batch_sharded");
    }
    void Status() const { detail::check(pigo_plan_status(h_.get()), "pigo_plan_status"); }
    pigo_plan_info_t Info() const
    {
        pigo_plan_info_t info{};
        detail::check(pigo_plan_info(h_.get(), &info), "pigo_plan_info");
        return info;
    }
    pigo_plan *handle() const { return h_.get(); }

  private:
    std::shared_ptr<pigo_plan> h_;
};

}  // namespace pigo
#endif  // PIGO_HPP
