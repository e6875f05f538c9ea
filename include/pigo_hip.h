/*
 * pigo_hip.h -- C ABI of libpigo_hip.so: the MI355X (gfx950) implementation of Pigo's cascade scan.
 *
 * This is the drop-in boundary.  The reference (esimov/pigo, pure Go) has no FFI seam on this path;
 * the seam is the exported Go API of package github.com/esimov/pigo/core, so every entry point below
 * names the Go method it stands in for (file:line under /root/reference).  INTEGRATION.md shows the
 * cgo shim a maintainer adds on the Go side; include/pigo.hpp is the C++ mirror of the same API and
 * pigo_amd/core.py the ctypes one.
 *
 * Conventions
 *   - plain C types only; no torch / HIP types in the signatures (a stream is passed as void*).
 *   - every function returns a pigo_status (0 = ok, < 0 = error) and never aborts the process.
 *     The reference has no error channel on this path: where the Go code would panic (short packet,
 *     pixel index out of range) the C ABI returns PIGO_ERR_PACKET / PIGO_ERR_PANIC and the shim
 *     re-raises it as a Go panic.
 *   - host buffers passed in (packet, pixels, dets) are only read/written during the call and never
 *     retained (cgo pointer rules).
 *   - a handle may be used from any OS thread; calls on one handle are serialised internally.
 *   - the library is GPU-only: there is no CPU fallback.  Without a usable gfx950 device every
 *     entry point that needs one fails with PIGO_ERR_HIP.
 */
#ifndef PIGO_HIP_H
#define PIGO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int pigo_status;
#define PIGO_OK 0
#define PIGO_ERR_PACKET (-1)   /* Unpack: packet shorter than its header says (Go: slice panic, pigo.go:64,81,90) */
#define PIGO_ERR_PARAM (-2)    /* argument outside what the library supports (see each function) */
#define PIGO_ERR_HIP (-3)      /* HIP runtime error / no device; pigo_last_error() has the text */
#define PIGO_ERR_CAPACITY (-4) /* output buffer too small; *n_out holds the required element count */
#define PIGO_ERR_PANIC (-5)    /* the reference would panic: pixel index out of range (rotated scan, quirk Q1) */
#define PIGO_ERR_NOMEM (-6)
#define PIGO_ERR_TIMEOUT (-7)  /* pigo_plan_status: a wave of a one-launch scan (plans of fewer than 8 frames) waited longer than
                                  PIGO_ONE_TIMEOUT_MS (default 2000) for a window another workgroup had claimed -- the device was
                                  taken away from the launch for that long (another process, a debugger).  That run's lists are
                                  incomplete: run the plan again.  pigo_run_cascade does so by itself. */

/* Detection, core/pigo.go:195-200.  16-byte wire/GPU record; the Go shim widens it to Go's
 * {Row, Col, Scale int; Q float32}. */
typedef struct {
    int32_t row, col, scale;
    float q;
} pigo_det;

/* type Pigo (core/pigo.go:37-43): the unpacked cascade, plus its device-resident tables. */
typedef struct pigo_cascade pigo_cascade;

/* A scan plan: CascadeParams (core/pigo.go:16-34) minus the pixels, bound to a cascade, with the
 * per-scale offset tables, the window/tile index space and the device workspace for a batch. */
typedef struct pigo_plan pigo_plan;

/* Thread-local description of the last error returned on this thread ("" if none). */
const char *pigo_last_error(void);

/* Number of visible HIP devices (0 if none / no driver). */
int pigo_device_count(void);

/* ---- (*Pigo).Unpack, core/pigo.go:51-110 ------------------------------------------------------
 * Parses the cascade file, uploads the tables to `device` and returns a new handle.  Like the
 * reference it accepts any tree depth / tree count the header states (depth <= 12 here).
 * PIGO_ERR_PACKET when the reference would panic on a short packet. */
pigo_status pigo_cascade_create(const uint8_t *packet, size_t len, int device, pigo_cascade **out);
pigo_status pigo_cascade_info(const pigo_cascade *c, uint32_t *tree_depth, uint32_t *tree_num);
/* copies of the unpacked tables (treeCodes / treePred / treeThreshold, pigo.go:38-40); any pointer may be NULL */
pigo_status pigo_cascade_tables(const pigo_cascade *c, int8_t *codes, size_t ncodes, float *pred, size_t npred, float *thr,
                                size_t nthr);
void pigo_cascade_destroy(pigo_cascade *c);

/* ---- (*Pigo).RunCascade, core/pigo.go:212-258 ---------------------------------------------------
 * One frame, host memory in, host memory out.  `pixels` = ImageParams.Pixels (row-major gray,
 * stride `dim`), npixels = len(Pixels).  Detections (q > 0 only) are written in the reference's
 * order: scale-major, then row, then col.  angle > 0 takes the classifyRotatedRegion path (angle is
 * clamped to 1.0 like pigo.go:233-235); angle <= 0 the classifyRegion path.
 * Returns PIGO_ERR_CAPACITY (and the needed count in *n_out) if more than `cap` detections exist.
 * Deviations from the reference, all reported as PIGO_ERR_PARAM instead of Go's undefined/panicking
 * behaviour: min_size < 0, dim < cols, npixels < rows*dim, rows/cols/dim >= 65536, non-finite
 * shift/scale factors, more than 2^32-1 windows or more than 2047 non-empty scales. */
pigo_status pigo_run_cascade(pigo_cascade *c, const uint8_t *pixels, size_t npixels, int rows, int cols, int dim, int min_size,
                             int max_size, double shift_factor, double scale_factor, double angle, pigo_det *out, int cap,
                             int *n_out);

/* ---- (*Pigo).ClusterDetections, core/pigo.go:262-308 ---------------------------------------------
 * Sorts `dets` in place by ascending Q exactly like the reference's sort.Slice (Go's pdqsort,
 * restated host-side), then runs the IoU clustering on the GPU.  `out` needs room for up to n
 * clusters.  No limit on n (lists beyond 2048 entries take the seeds / members / compact kernels).
 * Re-entrant like the reference's (a pure function that goroutines call concurrently on one *Pigo,
 * examples/web/main.go:141-144): a call owns a slot of the handle -- pinned staging, a stream, device
 * scratch -- for its duration, the handle's lock covers the slot list only. */
pigo_status pigo_cluster_detections(pigo_cascade *c, pigo_det *dets, int n, double iou_threshold, pigo_det *out, int cap,
                                    int *n_out);
/* The sort step alone (host): sort.Slice(dets, func(i, j) bool { return dets[i].Q < dets[j].Q }), pigo.go:264 */
void pigo_sort_by_q(pigo_det *dets, int n);

/* ---- RgbToGrayscale, core/grayscale.go:8-23 (the step in front of RunCascade) ---------------------
 * `pix` is the Pix buffer of the image the reference would be handed: `height` rows of `stride`
 * bytes, 4 bytes {R,G,B,A} per pixel, pixel (0,0) first.  `kind` says how src.At(x,y).RGBA()
 * (grayscale.go:14) reads it:
 *   PIGO_PIX_NRGBA  *image.NRGBA, what GetImage/DecodeImage return (core/image.go:12-90): c*0x101*A/0xff
 *   PIGO_PIX_RGBA   *image.RGBA  (core/grayscale_test.go:15): c*0x101
 *   PIGO_PIX_CANVAS the wasm front end's own formula on raw canvas RGBA (wasm/canvas/canvas.go:179-191)
 * gray[y*width + x] = uint8((0.299 r + 0.587 g + 0.114 b) / 256) in float64, bit-exact.
 * PIGO_ERR_PANIC when Pix is too short for (width, height, stride) -- src.At would index past it;
 * PIGO_ERR_CAPACITY when `cap` < width*height. */
#define PIGO_PIX_NRGBA 0
#define PIGO_PIX_RGBA 1
#define PIGO_PIX_CANVAS 2
pigo_status pigo_rgb_to_grayscale(int device, const uint8_t *pix, size_t npix, int width, int height, int stride, int kind,
                                  uint8_t *gray, size_t cap);
/* Device-resident batch form (extension): `nframes` frames of `frame_stride` bytes in device memory
 * -> gray frames of `gray_frame_stride` bytes with rows of `gray_dim` >= width bytes (ImageParams.Dim),
 * enqueued on `stream` (hipStream_t, NULL = default).  The output can be fed to pigo_plan_run as is. */
pigo_status pigo_gray_batch(int device, const uint8_t *d_pix, size_t frame_stride, int stride, int width, int height, int kind,
                            int nframes, uint8_t *d_gray, size_t gray_frame_stride, int gray_dim, void *stream);

/* ---- PuplocCascade: pupil / facial-landmark localisation, core/puploc.go + core/flploc.go ---------
 * The step behind ClusterDetections for every face with Scale > 50 (cmd/pigo/main.go:404-564).
 *
 * Two inputs of RunDetector are not among the reference's arguments and are made explicit here:
 *   rnd   the 3*Perturbs float32 values rand.Float32() returns inside RunDetector (puploc.go:248-250),
 *         in draw order: row, col, scale of perturbation 0, then of perturbation 1, ...  The Go shim
 *         draws them from math/rand exactly as the reference would, so results match for any seed.
 *   pool  the sync.Pool object's three 63-entry arrays rows|cols|scale (puploc.go:228-237), 189
 *         floats, read AND written: the reference never clears them and sorts all 63 entries
 *         (puploc.go:267-269), so with Perturbs < 63 its result depends on what the previous user of
 *         the pool object left behind.  NULL = a brand-new object (zeros).  The shim keeps these in
 *         a sync.Pool of its own, which reproduces the reference's behaviour call for call. */
typedef struct pigo_puploc_cascade pigo_puploc_cascade;
/* Puploc, core/puploc.go:14-19 (Go: Row, Col int; Scale float32; Perturbs int) */
typedef struct {
    int32_t row, col;
    float scale;
    int32_t perturbs;
} pigo_puploc;
/* UnpackCascade, core/puploc.go:38-103 (UnpackFlp, core/flploc.go:27-33, is ReadFile + this).  Wire format:
 * {stages u32, scale f32, trees u32, depth u32} then per tree 4*2^depth-4 code bytes + 2*2^depth float32.
 * PIGO_ERR_PACKET where the reference panics on a short packet. */
pigo_status pigo_puploc_create(const uint8_t *packet, size_t len, int device, pigo_puploc_cascade **out);
pigo_status pigo_puploc_info(const pigo_puploc_cascade *c, uint32_t *stages, float *scales, uint32_t *trees, uint32_t *tree_depth);
void pigo_puploc_destroy(pigo_puploc_cascade *c);
/* RunDetector, core/puploc.go:239-277.  PIGO_ERR_PANIC where the reference panics: Perturbs outside
 * [0, 63], rows/cols < 1, pixels shorter than (rows-1)*dim+cols. */
pigo_status pigo_puploc_run_detector(pigo_puploc_cascade *c, const pigo_puploc *pl, const uint8_t *pixels, size_t npixels, int rows,
                                     int cols, int dim, double angle, int flip_v, const float *rnd, float *pool, pigo_puploc *out);
/* GetLandmarkPoint, core/flploc.go:36-57 */
pigo_status pigo_get_landmark_point(pigo_puploc_cascade *c, const pigo_puploc *left_eye, const pigo_puploc *right_eye,
                                    const uint8_t *pixels, size_t npixels, int rows, int cols, int dim, int perturb, int flip_v,
                                    const float *rnd, float *pool, pigo_puploc *out);
/* Device-resident batch form (extension): n independent RunDetector requests against `nframes` gray
 * frames in device memory, one workgroup per request, enqueued on `stream`.  d_rnd: [n][189] floats
 * (request q, perturbation p uses d_rnd[q*189 + 3p .. +2]); d_pool: [n][189] in/out or NULL (fresh).
 * pigo_puploc_status() after synchronising reports a request the reference would have panicked on. */
typedef struct {
    int32_t row, col;
    float scale;
    int32_t perturbs, frame, flip_v;
} pigo_puploc_req;
pigo_status pigo_puploc_run_batch(pigo_puploc_cascade *c, const uint8_t *d_frames, size_t frame_stride, int nframes, int rows, int cols,
                                  int dim, double angle, const pigo_puploc_req *d_reqs, const float *d_rnd, float *d_pool, int n,
                                  pigo_puploc *d_out, void *stream);
pigo_status pigo_puploc_status(pigo_puploc_cascade *c);

/* ---- batch / device-resident extension (BASELINE configs 2-5; no reference counterpart) ----------
 * A plan fixes (rows, cols, dim, MinSize, MaxSize, ShiftFactor, ScaleFactor, angle) and owns the
 * workspace for up to `max_frames` frames with up to `det_cap` raw detections per frame. */
pigo_status pigo_plan_create(pigo_cascade *c, int rows, int cols, int dim, int min_size, int max_size, double shift_factor,
                             double scale_factor, double angle, int max_frames, int det_cap, pigo_plan **out);
void pigo_plan_destroy(pigo_plan *p);

typedef struct {
    int64_t windows_per_frame; /* every (scale,row,col) RunCascade would classify */
    int32_t n_scales;          /* non-empty rungs of the scale ladder */
    int32_t n_ladder;          /* all rungs, including the ones larger than the image */
    int32_t tiles_per_frame;   /* workgroups per frame of the head kernel */
    int32_t n_head_trees;      /* trees evaluated by the dense head kernel */
    int32_t variant;           /* 0 = monolithic scan, 1 = head + survivor-queue tail, 2 = whole cascade per LDS tile, 3 = region kernel */
    int32_t max_frames, det_cap;
    int64_t queue_capacity;    /* survivor-queue entries shared by the batch */
    int64_t workspace_bytes;
} pigo_plan_info_t;
pigo_status pigo_plan_info(const pigo_plan *p, pigo_plan_info_t *info);

/* Selects the scan implementation for this plan:
 *   0 = monolithic lane-per-window kernel (any tree depth; also the overflow fallback),
 *   1 = the first design (dense head kernel + survivor queue + tail kernel, pixels gathered from global memory); it is
 *       compiled into the debug build only (python -m pigo_amd.build --debug) -- PIGO_ERR_PARAM in the release library,
 *   2 = one workgroup takes a tile of windows through the whole cascade out of an LDS copy of the tile's
 *       pixels (default when the cascade has depth 6),
 *   3 = one workgroup per CU owns an LDS-resident region of a frame for a whole group of scales (k_scan_region); the
 *       scales above the groups go through k_scan_big (persistent, next to the region workgroups) and k_tail_deep.  Default
 *       for plans with dim % 4 == 0 and max_frames >= 8, upright and rotated (landscape frames); PIGO_ERR_PARAM for plans
 *       it cannot serve (other strides, rotated scans of portrait frames).
 * All variants produce identical results. */
pigo_status pigo_plan_set_variant(pigo_plan *p, int variant);

/* Asynchronous scan of `nframes` (<= max_frames) device-resident frames.  `d_frames` points to
 * nframes consecutive frames of `frame_stride` bytes each (>= rows*dim) in device memory; when dim is a
 * multiple of 4 the pointer and the stride must be too (frames are copied as aligned dwords).
 * Enqueues on `stream` (a hipStream_t, NULL = default stream):
 *     d_dets   [nframes][det_cap] pigo_det, reference order per frame
 *     d_counts [nframes] int32: detections found (if > det_cap the frame's list is truncated)
 * Nothing is synchronised; call pigo_plan_status() after synchronising the stream.  A plan owns ONE
 * workspace: enqueue all work of a plan on the same stream (or order the streams yourself); use one
 * plan per stream for concurrent batches.
 * Internal streams: a plan for >= 8 frames forks part of its work onto a high-priority side stream of its own (it cannot
 * share the caller's hardware queue) and joins it before this call returns.  A plan for fewer frames is ONE kernel launch on
 * `stream` (k_scan_one: the scan, the hand-over to the deep trees and the order restore inside one persistent grid) -- no side
 * stream, no fork / join, nothing synchronised or probed. */
pigo_status pigo_plan_run(pigo_plan *p, const uint8_t *d_frames, size_t frame_stride, int nframes, pigo_det *d_dets,
                          int32_t *d_counts, void *stream);

/* Asynchronous per-frame ClusterDetections of the lists produced by pigo_plan_run (same stream):
 * sorts each frame's list by ascending Q exactly like the reference's sort.Slice -- a stable order when
 * the frame has no tied Q values (then the two coincide), Go's pdqsort restated on the device otherwise;
 * d_ties[f] (may be NULL) receives the number of tied detections -- then clusters on the GPU.
 *     d_sorted   [nframes][det_cap] the frame's detections, sorted (what the reference leaves in the caller's slice)
 *     d_clusters [nframes][det_cap], d_ccounts [nframes] */
pigo_status pigo_plan_cluster(pigo_plan *p, const pigo_det *d_dets, const int32_t *d_counts, int nframes, double iou_threshold,
                              pigo_det *d_sorted, pigo_det *d_clusters, int32_t *d_ccounts, int32_t *d_ties, void *stream);

/* After the stream has been synchronised: PIGO_OK, PIGO_ERR_PANIC (the reference would have
 * panicked on some frame) or PIGO_ERR_CAPACITY -- either the survivor queue overflowed (results of
 * the last run are incomplete; pigo_plan_run_sync handles this by re-running with the monolithic
 * kernel) or some frame has more than det_cap detections (its list is truncated, and which records
 * were kept is not deterministic; d_counts holds the true count: re-plan with a larger det_cap).
 * pigo_last_error() says which, pigo_plan_last_flags() returns the raw words pigo_plan_status read last: queue_overflow
 * (bit 0 a tile's LDS queue, bit 1 a survivor queue, bit 2 the second-level tail queue: "results incomplete, re-run"),
 * would_panic, det_cap_overflow ("list truncated, re-plan with a larger det_cap"). */
pigo_status pigo_plan_status(pigo_plan *p);
pigo_status pigo_plan_last_flags(const pigo_plan *p, int32_t *queue_overflow, int32_t *would_panic, int32_t *det_cap_overflow);

/* Synchronous convenience wrapper: run + synchronise + overflow fallback. */
pigo_status pigo_plan_run_sync(pigo_plan *p, const uint8_t *d_frames, size_t frame_stride, int nframes, pigo_det *d_dets,
                               int32_t *d_counts, void *stream);

/* Per-kernel timing of the most recent pigo_plan_run when profiling is on (HIP events recorded on
 * the run's stream around each kernel).  names/ms arrays of length `cap`; returns the kernel count. */
pigo_status pigo_plan_set_profiling(pigo_plan *p, int on);
int pigo_plan_last_timings(pigo_plan *p, const char **names, float *ms, int cap);

/* Debug build only (libpigo_hip_debug.so, plans created with PIGO_DEBUG_STATS=1 in the environment; the release
 * library carries no instrumentation in its kernels and returns zeros): accumulated shader-clock totals of
 * k_scan_tile's phases since the last call -- [0] tile/table copies, [1] stage 0, [2] later dense stages,
 * [3] late mode, [4] tiles, [5] windows entering late mode, [6] trees walked in late mode.  Zeros otherwise. */
pigo_status pigo_plan_debug_stats(pigo_plan *p, uint64_t *out, int n);
/* Debug only: raw per-iteration trace of late mode for 16 tiles (n >= 4096 words). */
pigo_status pigo_plan_debug_trace(pigo_plan *p, uint64_t *out, int n);

/* Device statistics of the most recent run (valid after synchronising): survivor-queue entries.  Batches of >= 16
 * frames are pipelined in chunks over two queue sets; the figure then covers the last chunk that used the first set
 * (bench.py reads it after an un-pipelined profiling run). */
pigo_status pigo_plan_last_queue_count(pigo_plan *p, int64_t *n);

/* ---- multi-GPU batch: frames sharded over ranks + ONE RCCL all-gather (BASELINE config 3; no reference counterpart) ----
 * RunCascade is a single goroutine over one image (core/pigo.go:212-258) and frames are independent, so a batch is cut into
 * contiguous shards, one process per GPU, with no exchange during the scan; the per-frame lists are exchanged once at the
 * end.  RCCL has no all-gather-v: every frame travels as one fixed-size row of pigo_wire_words(gather_cap) = 2 + 4*gather_cap
 * int32 -- the TRUE count, a FLAGS word (PIGO_WIRE_* below: what a peer must know about the row without asking the rank that made
 * it), then the first min(count, gather_cap) pigo_det records, zero-padded.  librccl is bound with dlopen at first use
 * (PIGO_RCCL_LIB overrides the name).
 *
 *   rank 0:      pigo_comm_unique_id(id)            -- ncclGetUniqueId; the host program ships the 128 bytes to every rank
 *   every rank:  pigo_comm_init(id, rank, world, device, &comm)   -- ncclCommInitRank (collective)
 *                pigo_shard_bounds(nframes, rank, world, &lo, &hi)
 *                pigo_run_batch_sharded(plan, comm, d_frames + lo*stride, stride, hi - lo, frames_per_rank, iou, gather_cap,
 *                                       d_gathered, stream)
 * world == 1 needs no RCCL at all: with id == NULL the "gather" is one device-to-device copy.  world == 1 WITH an id (from
 * pigo_comm_unique_id) builds a real one-rank RCCL communicator and pigo_run_batch_sharded goes through ncclAllGather exactly
 * as on a multi-GPU node -- the way to exercise the collective on a single-GPU box (pigo_comm_uses_rccl tells which).
 * Errors: a rank whose scan fails inside pigo_run_batch_sharded still contributes zero-count padding rows to the collective
 * before it returns the error, so its peers do not hang.  Only SYNCHRONOUS refusals take that path (bad arguments, a launch
 * error): survivor-queue overflow and lists truncated at det_cap are raised on the device and reported by pigo_plan_status()
 * after the stream has been synchronised -- every rank should check it, and every PEER sees both in the flags word of that
 * rank's rows (PIGO_WIRE_QUEUE_OVERFLOW, PIGO_WIRE_TRUNCATED_DETCAP), packed on the device behind the scan.  pigo_comm_init waits for its peers with a deadline (PIGO_COMM_INIT_TIMEOUT_S, default 300 s,
 * 0 = none) and returns PIGO_ERR_HIP when a peer does not join; a failed pigo_comm_init or collective is fatal for the
 * communicator on every rank: pigo_comm_abort (if a collective may be outstanding), pigo_comm_destroy, start over. */
typedef struct pigo_comm pigo_comm;
#define PIGO_COMM_ID_BYTES 128
pigo_status pigo_comm_unique_id(uint8_t id[PIGO_COMM_ID_BYTES]);
pigo_status pigo_comm_init(const uint8_t id[PIGO_COMM_ID_BYTES], int rank, int world, int device, pigo_comm **out);
pigo_status pigo_comm_info(const pigo_comm *c, int *rank, int *world);
int pigo_comm_uses_rccl(const pigo_comm *c); /* 1: the all-gather is ncclAllGather; 0: world == 1 without an id, a plain copy */
/* ncclCommAbort: give up the communicator WITHOUT waiting for outstanding collectives -- for a host whose peer rank has failed
 * (its pigo_run_batch_sharded all-gather would never complete).  May be called from another thread than the one blocked on the
 * stream.  Afterwards the handle only accepts pigo_comm_destroy. */
pigo_status pigo_comm_abort(pigo_comm *c);
void pigo_comm_destroy(pigo_comm *c);
/* contiguous shard [lo, hi) of `nframes` frames for `rank`; earlier ranks take the remainder */
void pigo_shard_bounds(int nframes, int rank, int world, int *lo, int *hi);
size_t pigo_wire_words(int gather_cap);
/* flags word of a wire row (row[1]) */
#define PIGO_WIRE_TRUNCATED_GATHER 1 /* count > gather_cap: the row holds the first gather_cap records of the list */
#define PIGO_WIRE_TRUNCATED_DETCAP 2 /* the producing rank's list (or the raw list its clusters were made from) was cut at det_cap: incomplete */
#define PIGO_WIRE_QUEUE_OVERFLOW 4   /* the producing rank's scan overflowed a survivor queue: its lists may miss detections, rescan */
#define PIGO_WIRE_WOULD_PANIC 8      /* a frame of the producing rank's batch makes the reference panic (pigo.go:167-179) */
#define PIGO_WIRE_RANK_FAILED 16     /* the producing rank's scan was refused (bad arguments, launch error): all its rows are padding */
#define PIGO_WIRE_PADDING 32         /* no frame behind this row (short shard) -- not "a frame without detections" */
int pigo_wire_row_flags(const int32_t *wire_row);
/* Scan (+ per-frame ClusterDetections when iou_threshold >= 0; a negative or NaN threshold gathers the raw RunCascade lists)
 * this rank's `nframes_local` device-resident frames and all-gather the wire rows of all ranks:
 *     d_gathered [world * frames_per_rank][pigo_wire_words(gather_cap)] int32, device memory; rank r's frames are rows
 *     [r*frames_per_rank, r*frames_per_rank + its nframes_local), the rest of its rows are zero-count padding.
 * Everything is enqueued on `stream`; the collective is the last operation.  Every rank of the communicator must call
 * with the same frames_per_rank and gather_cap.  pigo_plan_status() applies as for pigo_plan_run. */
pigo_status pigo_run_batch_sharded(pigo_plan *p, pigo_comm *comm, const uint8_t *d_frames, size_t frame_stride, int nframes_local,
                                   int frames_per_rank, double iou_threshold, int gather_cap, int32_t *d_gathered, void *stream);
/* The wire format on the host (what pigo_run_batch_sharded packs on the device): rows [nframes, frames_out) are padding. */
pigo_status pigo_pack_lists(const pigo_det *lists, const int32_t *counts, int nframes, int frames_out, int cap, int gather_cap,
                            int32_t *wire);
pigo_status pigo_unpack_list(const int32_t *wire_row, int gather_cap, pigo_det *out, int cap, int *n_out, int *true_count);

#ifdef __cplusplus
}
#endif
#endif /* PIGO_HIP_H */
