/*
 * tinyorb.h -- C ABI of the MI355X-native ORB feature-extraction front-end.
 *
 * This is the drop-in boundary for the `tinyslam::orb` module of ccaven/tinyslam
 * (src/lib.rs:1, src/orb.rs).  Each entry point names the reference item it replaces;
 * INTEGRATION.md shows the Rust `extern "C"` shim a maintainer would add on the reference
 * side.  Plain pointers and sizes only: no C++, HIP or torch types cross this boundary.
 *
 * Semantics follow the reference's literal behaviour:
 * keypoints are FAST-12 corners on an R16Float luminance pyramid (vertically flipped frame),
 * orientation is the 16-pixel ring centroid in milliradians (negative angles stored as 0),
 * descriptors are 256-bit rotated BRIEF on the reference's (x-only, UV-offset) blur.
 *
 * Threading: calls on one OrbProgram must be serialised by the caller; distinct programs are
 * independent (one program = one device + its streams), like one wgpu Device/Queue per
 * OrbProgram in the reference (orb.rs:47-51), and may be created, used and destroyed from
 * different host threads at the same time (tests/test_gpu_round4.py::test_programs_on_different_threads).
 */
#ifndef TINYORB_H
#define TINYORB_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TINYORB_ABI_VERSION 5

/* status codes (the reference panics instead: orb.rs:553 unwrap, label look-ups) */
#define ORB_OK 0
#define ORB_EINVAL 1    /* bad argument / configuration (reference: wgpu validation panic) */
#define ORB_EHIP 2      /* HIP runtime failure; text in orb_last_error() */
#define ORB_ECAPACITY 3 /* more corners detected than max_features; first max_features kept */
#define ORB_ESTATE 4    /* call out of order (e.g. read before extract) */

#define ORB_MAX_HIERARCHY_DEPTH 10 /* orb.rs:67 MAX_HIERARCHY_DEPTH */

/* wgpu::Extent3d as used by OrbConfig.image_size (orb.rs:41, 120, 227) */
typedef struct OrbExtent3d {
    uint32_t width;
    uint32_t height;
    uint32_t depth_or_array_layers; /* must be 1 */
} OrbExtent3d;

/* orb.rs:40-45 `pub struct OrbConfig` -- same field order */
typedef struct OrbConfig {
    OrbExtent3d image_size;
    uint32_t max_features;
    uint32_t hierarchy_depth; /* 1..=10; the reference needs >= 2 */
    float initial_threshold;  /* intensity units, [0,1] */
} OrbConfig;

/* orb.rs:10-17 `#[repr(C)] CornerData` == WGSL `Feature` (fast.wgsl:1-6); 16 bytes.
 * x,y are in the octave's own pixel grid of the vertically flipped image (Q2);
 * angle is milliradians 0..3141 (Q7). */
typedef struct CornerData {
    uint32_t x;
    uint32_t y;
    uint32_t angle;
    uint32_t octave;
} CornerData;

/* orb.rs:19-23 `#[repr(C)] CornerDescriptor`; 32 bytes = 8 little-endian u32 words
 * (brief.wgsl:15); bit i of word k is BRIEF test 32k+i (brief.wgsl:47,63). */
typedef struct CornerDescriptor {
    uint8_t bits[32];
} CornerDescriptor;

/* Build-side options with no counterpart in the reference. Zero-initialise for defaults. */
typedef struct OrbOptions {
    int32_t device;       /* HIP device ordinal */
    uint32_t max_batch;   /* frames per batched call; 0 -> 1 */
    uint32_t flags;       /* ORB_FLAG_* */
    uint32_t fast_arc;    /* 0 -> 12 (the reference's FAST-12, fast.wgsl:56-60; 9 with ORB_FLAG_INTENDED);
                           * 9..16: corner = run of >= fast_arc */
    /* Two things the reference's WGSL leaves to the adapter it runs on, as switches (the defaults, 0 and 0, are the
     * canonical decisions CRD-6 / CRD-5 of SURVEY.md 8a; until a dump from the reference itself pins them --
     * rust/dump_config0, tools/pin_oracle.py -- a caller who knows the adapter can follow it).  For the reference's
     * detector only (RGBA or Y8 input): refused together with ORB_FLAG_INTENDED, ORB_FLAG_NMS or a fast_arc other
     * than 12, which have no reference behaviour to follow. */
    uint32_t oob_policy;          /* ORB_OOB_*: what a textureLoad OUTSIDE the addressed level returns (fast.wgsl:78,86,103
                                   * at octaves >= 1, whose guard uses the level-0 size; brief.wgsl:59-60) */
    uint32_t sampler_weight_bits; /* 0: bilinear weights are the exact binary32 fractions; n = 1..23: a sampler that holds
                                   * them in n fractional bits, rounded to nearest, halves up (8 is common): the blur's
                                   * lerps (gaussian_blur_x.wgsl:53-58) and the blit of an odd-sized level (blit.wgsl:35) */
    uint32_t fp_contract;         /* CRD-13: the arithmetic WGSL leaves to the adapter's shader compiler, a mask of
                                   * ORB_FP_*.  0 = every binary32 product and sum of the shaders rounded on its own, dot() reduced from its
                                   * first component (the default).  ORB_FP_CONTRACT_LUMINANCE / _BLUR / _ROTATION: that stage's
                                   * product-and-sum pairs as fused multiply-adds -- dot() (grayscale.wgsl:36), `result += sample *
                                   * weight` (gaussian_blur_x.wgsl:58), matrix * vector (brief.wgsl:53-54).  ORB_FP_LAST_TERM_FIRST: dot()
                                   * and matrix * vector reduced from the last component / column down (Mesa's lowering) instead of the
                                   * first up.  Carried by the fused AND the per-stage kernels at full speed (the luminance and rotation
                                   * forms are template instances, the blur's is a set of scalar constants).  RGBA input, the reference's
                                   * detector.  ABI 4 knew the values 0 and 1 (= every stage contracted, per-stage kernels only). */
    uint32_t angle_bins;          /* ORB_FLAG_INTENDED only (IM-6b): 0 = a descriptor is rotated by its keypoint's milliradian
                                   * code (6284 rotated patterns: a 6.4 MB table that misses the 4 MB L2 of an XCD); N = 8..6284 = by the
                                   * centre of its angle bin, bin = code * N / 6284, centre code = (bin * 6284 + 3142) / N -- 1024 bins are a
                                   * 1 MB table (OpenCV's ORB uses 30).  The keypoint's reported angle stays the milliradian code.
                                   * (Took the last reserved word.) */
} OrbOptions;

#define ORB_OOB_ZERO 0u  /* 0.0 -- Vulkan robust image access; naga: image_load = Unchecked on such devices */
#define ORB_OOB_CLAMP 1u /* every coordinate clamped into [0, size - 1] */
#define ORB_OOB_UMIN 2u  /* naga's `Restrict` policy as its SPIR-V writer emits it: min(coordinate AS UNSIGNED, size - 1) --
                          * a negative coordinate lands on the LAST column / row of the level */

#define ORB_FP_CONTRACT_LUMINANCE 1u
#define ORB_FP_CONTRACT_BLUR 2u
#define ORB_FP_CONTRACT_ROTATION 4u
#define ORB_FP_CONTRACT_ALL 7u
#define ORB_FP_LAST_TERM_FIRST 8u

#define ORB_FLAG_STAGED 1u        /* force the one-kernel-per-stage pipeline (cross-check of the fused path) */
#define ORB_FLAG_DOUBLE_OUTPUT 2u /* two sets of output slabs: batch k+1 computes while batch k is collated */
#define ORB_FLAG_NMS 4u           /* opt-in, NOT in the reference: 3x3 non-maximum suppression per
                                   * octave on the arc score sum(|v - c| - threshold); the counter is then the number
                                   * of survivors. */
#define ORB_FLAG_INTENDED 8u      /* opt-in, NOT in the reference: the algorithm the reference's
                                   * README describes, with the shaders' accidents repaired -- BT.601 luminance (0.299),
                                   * no vertical mirror, a true separable 7-tap Gaussian (X then Y), the octave's own
                                   * border guard, angle codes over the full circle (0..6283 mrad), BRIEF rotated by
                                   * +theta, fast_arc 0 -> 9, optional ORB_FLAG_NMS, and when more than max_features
                                   * keypoints remain the max_features best by score are kept (ties: smaller octave, y,
                                   * x).  Definitions IM-1..IM-8 in DESIGN.md section 8; has its own fused kernels (widths that are a
                                   * multiple of 4), ORB_FLAG_STAGED selects the per-stage cross-check. */

#define ORB_FLAG_INPUT_Y8 16u     /* opt-in, NOT in the reference's code (its roadmap: README.md:42 "Use Y channel of YUV
                                   * stream directly"): frames are ONE byte per pixel (W*H bytes,
                                   * tightly packed) and the grey image is that sample, gray(x,y) = f16(Y(x, H-1-y)/255)
                                   * -- the vertical mirror of the reference's full-screen pass is kept, everything
                                   * downstream is the literal path, unchanged.  Not combined with ORB_FLAG_INTENDED, ORB_FLAG_NMS
                                   * or a fast_arc other than 12 (those have no Y8 definition to check against). */

#define ORB_FLAG_SINGLE_BLOCKING_WAIT 32u /* how orb_extract_corners waits for the device.  Default: the host thread polls a completion word
                                   * in pinned memory (the shortest latency; a spinning thread on a shared box now and then loses its CPU
                                   * for a scheduler slice).  With this flag (or TINYORB_SINGLE_WAIT=block in the environment) the spin is
                                   * bounded to 50 us, after which the thread sleeps until the completion interrupt -- what the reference's
                                   * device.poll(Wait) does (orb.rs:547). */

typedef struct OrbProgram OrbProgram; /* opaque; replaces orb.rs:47-51 `OrbProgram` */

/* ---- lifetime: replaces the struct literal + OrbProgram::init (orb.rs:107-219) ---- */
int orb_program_create(const OrbConfig *config, const OrbOptions *options, OrbProgram **out);
void orb_program_destroy(OrbProgram *p);
/* last error text of this program (or of the failed create when p == NULL) */
const char *orb_last_error(const OrbProgram *p);
uint32_t orb_abi_version(void);
/* "fused" (one kernel per pyramid level + BRIEF) or "staged" (one kernel per reference stage;
 * taken with ORB_FLAG_STAGED or for shapes the fused kernels do not cover: more than 2^26 pixels; the reference's
 * algorithm (RGBA or Y8 input): width > 4096 only -- any other width and any halving run fused; the arc/NMS extensions
 * and the intended mode also need a width that is a multiple of 4, arc/NMS a level 0 that halves exactly.  The band
 * height of the fused kernels is chosen per level from 64 / 32 / 16 / 8 rows). */
const char *orb_pipeline(const OrbProgram *p);
/* Empty unless the program runs on the staged kernels WITHOUT having asked for them: then the reason, e.g.
 * "staged pipeline (...): width 5120 exceeds 4096 (...)".  The same line goes to stderr once per process (silence it
 * with TINYORB_QUIET=1): the staged kernels run at about 1/7 of the fused rate and nobody should find that out late. */
const char *orb_pipeline_note(const OrbProgram *p);

/* ---- single-frame API, one call per reference method ---- */
/* orb.rs:567-583 write_input_image: tightly packed RGBA8, rows of 4*width bytes (ORB_FLAG_INPUT_Y8: rows of width bytes). */
int orb_write_input_image(OrbProgram *p, const uint8_t *bytes, size_t len);
/* The same upload without the wait (orb.rs:567-583 returns after queue.write_texture, before the copy has happened): `bytes`
 * lies in PINNED host memory (orb_host_alloc, or registered with the HIP runtime by the caller) and goes up on the program's
 * copy stream; the call returns at once, orb_upload_sync() waits until the array may be reused.  Images are extracted in the
 * order they were written, and ONE may be written ahead: a camera loop uploads frame k + 1 under the kernels of frame k --
 *     write_pinned(f0);  loop { write_pinned(f[k+1]); extract_corners(&n) [frame k]; read_corners; read_descriptors; }
 * A third write while two images wait returns ORB_ESTATE.  With nothing written since, extract_corners works on the last
 * image again. */
int orb_write_input_image_pinned(OrbProgram *p, const uint8_t *bytes_pinned, size_t len);
/* orb.rs:585-589 set_threshold */
int orb_set_threshold(OrbProgram *p, float threshold);
/* orb.rs:469-557 extract_corners: runs the whole pipeline, blocks until the results are in
 * host staging, returns the RAW detection counter (may exceed max_features, Q9/Q19; then the
 * status is ORB_ECAPACITY and the first max_features records are valid). */
int orb_extract_corners(OrbProgram *p, uint32_t *corner_count);
/* orb.rs:559-561 read_corners: copies min(n, max_features) records from staging. */
int orb_read_corners(OrbProgram *p, CornerData *dst, size_t n);
/* orb.rs:563-565 read_descriptors */
int orb_read_descriptors(OrbProgram *p, CornerDescriptor *dst, size_t n);

/* ---- batched mode (BASELINE.json configs[3], [4]): independent frames, one call ---- */
/* frames_dev: n_frames contiguous RGBA8 frames in DEVICE memory.  stream: a hipStream_t (or
 * NULL for the program's own stream).  Asynchronous: results stay device-resident in the
 * program's output slabs until the next batched call. */
int orb_extract_batch_device(OrbProgram *p, const uint8_t *frames_dev, uint32_t n_frames, void *stream);
/* Same with host frames: they are pinned in place for the call and uploaded in chunks on a copy stream while the
 * kernels of the chunks already on the device run.  Blocks until the uploads are done; results as above. */
int orb_extract_batch_host(OrbProgram *p, const uint8_t *frames_host, uint32_t n_frames);
/* The same for frames that already lie in PINNED host memory (orb_host_alloc, or registered with the HIP runtime by the
 * caller), without pinning and without blocking: returns once the chunked uploads and the kernels are enqueued.
 * orb_upload_sync() waits until the last upload has left the host array (which may then be reused); the kernels may
 * still be running (orb_batch_sync). */
int orb_extract_batch_pinned(OrbProgram *p, const uint8_t *frames_pinned, uint32_t n_frames);
int orb_upload_sync(OrbProgram *p);
/* Wait for the last batched call to finish. */
int orb_batch_sync(OrbProgram *p);
/* Raw per-frame counters of the last batch (synchronises). */
int orb_batch_counts(OrbProgram *p, uint32_t *totals, uint32_t n_frames);
/* Copy up to n records of one frame of the last batch to the host (synchronises). */
int orb_batch_read(OrbProgram *p, uint32_t frame, CornerData *corners, CornerDescriptor *descriptors, size_t n);
/* Selects which output set (0 or 1; 1 needs ORB_FLAG_DOUBLE_OUTPUT) the next batched call writes and the
 * batch read/buffer calls refer to. */
int orb_batch_select_output(OrbProgram *p, uint32_t set);
/* Device pointers of the output slabs, for a device-side collate (RCCL gather):
 * counts[max_batch] u32, corners[max_batch][max_features], descriptors[max_batch][max_features]. */
int orb_batch_device_buffers(OrbProgram *p, void **counts, void **corners, void **descriptors);

/* ---- bulk read-back (orb.rs:537-565: the reference copies counter + corners + descriptors to host staging after
 * every frame; the batched mode returns a whole batch in one go) ----
 * Packs the STORED records (min(counter, max_features) per frame) of the first n_frames frames of the last batch
 * back to back, in frame order, into caller-allocated buffers:
 *   counts[n_frames]       raw per-frame counters (may be NULL)
 *   offsets[n_frames + 1]  exclusive prefix of the stored counts; offsets[n_frames] = total records (may be NULL)
 *   corners / descriptors  [capacity] records; records past `capacity` are dropped (compare offsets[n_frames]).
 * orb_batch_read_all: HOST buffers.  With pinned memory (orb_host_alloc, or registered with the HIP runtime by the
 * caller) the device writes them directly over PCIe, asynchronously on `stream` (NULL: the stream of the batch; then
 * orb_batch_sync() waits for it; another stream is ordered behind the batch and waited for with orb_stream_sync()).
 * Pageable buffers are pinned for the duration of the call, which then blocks.
 * orb_batch_compact_device: the same into DEVICE buffers (payload of a device-side collate). */
int orb_batch_read_all(OrbProgram *p, uint32_t n_frames, uint32_t *counts, uint64_t *offsets, CornerData *corners,
                       CornerDescriptor *descriptors, size_t capacity, void *stream);
int orb_batch_compact_device(OrbProgram *p, uint32_t n_frames, uint32_t *counts_dev, uint64_t *offsets_dev,
                             CornerData *corners_dev, CornerDescriptor *descriptors_dev, size_t capacity, void *stream);
/* The same read-back in two steps, for a host that streams batches: the packed records go to program-owned device
 * memory first and cross PCIe as two exact-size DMA copies (about 56 GB/s on an MI355X box; the device writing pinned
 * host memory itself, as orb_batch_read_all does, reaches about 31 GB/s).
 * orb_batch_pack:  enqueues the packing of the first n_frames frames of the last batch (of the selected output set, see
 *                  orb_batch_select_output) behind that batch; asynchronous.
 * orb_batch_fetch: waits ON THE HOST until the pack of output set `set` (0 or 1) is done, fills counts / offsets (either
 *                  may be NULL) and enqueues the copies of min(total, capacity) records on `stream` (NULL: the program's
 *                  stream); with pinned destinations it returns while they are in flight (orb_stream_sync waits).
 * With ORB_FLAG_DOUBLE_OUTPUT: pack batch k (set k % 2), then fetch batch k - 1 on another stream -- its copies overlap
 * batch k's kernels.  A set may be packed again once its fetch has completed. */
int orb_batch_pack(OrbProgram *p, uint32_t n_frames, void *stream);
int orb_batch_fetch(OrbProgram *p, uint32_t set, uint32_t *counts, uint64_t *offsets, CornerData *corners,
                    CornerDescriptor *descriptors, size_t capacity, void *stream);
/* ---- transport records (multi-GPU collate; not in the reference, which has one device).  On the wire between GPUs a
 * keypoint is ORB_TRANSPORT_RECORD_BYTES = 40 bytes instead of 16 + 32: ten u32 words {x | y << 16, angle | octave << 16,
 * descriptor[8]} (angle code < 6284, octave < 8; frames of 65536 or more texels in either direction are refused with
 * ORB_EINVAL), records of a batch back to back in frame order.  Lossless.
 * orb_batch_pack_transport:  packs the stored records of the first n_frames frames of the batch in output set `set` into
 *     dst_dev[capacity_records] and writes offsets_dev[n_frames + 1] (exclusive prefix of min(counter, max_features);
 *     device memory, may be NULL).  Enqueued on `stream` (NULL: the program's stream) WITHOUT implicit ordering: the caller
 *     orders `stream` behind the batch's kernels (an event, or the same stream).
 * orb_unpack_transport:      n_segments runs of records in src_dev (run i: count[i] records from record src_first[i]) are
 *     expanded into corners_dev / descriptors_dev from record dst_first[i] on; the three arrays are HOST memory, read
 *     during the call.  Enqueued on `stream` (NULL: the program's stream), no implicit ordering. */
#define ORB_TRANSPORT_RECORD_BYTES 40
int orb_batch_pack_transport(OrbProgram *p, uint32_t set, uint32_t n_frames, void *dst_dev, size_t capacity_records,
                             uint64_t *offsets_dev, void *stream);
int orb_unpack_transport(OrbProgram *p, const void *src_dev, uint32_t n_segments, const uint64_t *src_first,
                         const uint64_t *count, const uint64_t *dst_first, CornerData *corners_dev,
                         CornerDescriptor *descriptors_dev, void *stream);
/* Pinned, device-visible host memory for callers without a HIP binding of their own. */
int orb_host_alloc(size_t nbytes, void **out);
void orb_host_free(void *ptr);
/* Waits for `stream` (a hipStream_t; NULL: the program's own stream) on the program's device. */
int orb_stream_sync(OrbProgram *p, void *stream);
/* The program's own stream (a hipStream_t), for callers that order their own work behind it. */
void *orb_program_stream(OrbProgram *p);

/* ---- one node, several GPUs (BASELINE.json configs[4]).  NOT in the reference, which drives one wgpu
 * device (orb.rs:47-51): an OrbNode owns one OrbProgram per listed device of this process and shards a batch of
 * independent frames over them in contiguous ranges (rank r gets frames [F*r/n, F*(r+1)/n) -- no data-path
 * collective).  The only exchange is the collate to the first device: every other device sends its packed records by
 * grouped ncclSend/ncclRecv of their EXACT sizes (RCCL over xGMI, one link per peer; librccl is loaded on first use, a
 * single-GPU program never needs it); the per-frame counters reach the host through pinned memory, no all-gather.
 *
 * A job runs through three stages -- extract, collate_begin, collate_end -- and up to two jobs may be outstanding, so
 * that a host which streams batches overlaps the collate of batch k with the kernels of batch k+1:
 *     orb_node_extract_batch(k);  orb_node_collate_begin() [job k-1];  orb_node_collate_end() [job k-1];  ...
 * One blocking call per job stays available: orb_node_extract_batch + orb_node_collate (the reference's call shape,
 * orb.rs:469-557, is one blocking call per frame). ---- */
typedef struct OrbNode OrbNode;
/* options->device is ignored (devices[] decides); options->max_batch is the largest shard of one device;
 * ORB_FLAG_DOUBLE_OUTPUT is implied.  ORB_FLAG_INPUT_Y8 nodes take one byte per pixel, as their programs do.
 * TINYORB_NODE_LOOPBACK in the environment (tests): 1 = devices may repeat and the exchange uses device copies, not RCCL;
 * 2 = every rank on ONE device and the exchange through RCCL all the same: a one-rank communicator over that device, each
 * rank's records sent to the communicator's own rank (ncclSend + ncclRecv in one group; with n_devices == 1 rank 0's own
 * records take that way too) -- how a one-GPU box executes the RCCL code. */
int orb_node_create(const int *devices, int n_devices, const OrbConfig *config, const OrbOptions *options, OrbNode **out);
void orb_node_destroy(OrbNode *node);
const char *orb_node_last_error(const OrbNode *node);
int orb_node_device_count(const OrbNode *node);
/* The program of rank `rank` (borrowed; e.g. for orb_synth_frames_device or orb_set_threshold on every rank). */
OrbProgram *orb_node_program(OrbNode *node, int rank);
/* Frame range [*lo, *hi) of rank `rank` for a job of n_frames frames. */
int orb_node_shard(const OrbNode *node, uint32_t n_frames, int rank, uint32_t *lo, uint32_t *hi);
/* Stage 1.  frames_dev[r]: rank r's shard, resident on ITS device (frames lo_r.. of the job, contiguous).  Asynchronous:
 * every device runs its shard on its own stream and packs the results behind it on a second one.  ORB_ESTATE when two
 * jobs are already outstanding. */
int orb_node_extract_batch(OrbNode *node, const uint8_t *const *frames_dev, uint32_t n_frames);
/* The same from one host array of n_frames frames: every shard is uploaded in chunks while its first chunks already
 * compute (orb_extract_batch_pinned; the array is pinned in place for the call).  Returns when the uploads are done. */
int orb_node_extract_batch_host(OrbNode *node, const uint8_t *frames_host, uint32_t n_frames);
/* Stage 2, for the oldest job that has not begun it: waits on the host for that job's per-frame counters (its kernels
 * and pack; later jobs keep the devices busy meanwhile) and enqueues the exchange -- exact-size transport records to
 * the first device, expanded there behind its own records.  Does not wait for the exchange. */
int orb_node_collate_begin(OrbNode *node);
/* Stage 3, for the oldest job (begins its exchange if nobody has): blocks until the collated result is on the first
 * device.  counts[n_frames] and offsets[n_frames + 1] are HOST arrays (as in orb_batch_read_all; either may be NULL);
 * *corners_dev / *descriptors_dev receive the addresses of the packed records (frame order), which stay valid while
 * the next TWO jobs are extracted (three result buffers). */
int orb_node_collate_end(OrbNode *node, uint32_t *counts, uint64_t *offsets, void **corners_dev, void **descriptors_dev);
/* orb_node_collate_begin + orb_node_collate_end for the oldest job: the blocking form. */
int orb_node_collate(OrbNode *node, uint32_t *counts, uint64_t *offsets, void **corners_dev, void **descriptors_dev);
/* Jobs extracted and not yet ended (0..2). */
int orb_node_pending(const OrbNode *node);
/* How the records of ranks >= 1 reach the first device: "rccl" (ncclSend/ncclRecv between devices), "rccl-self"
 * (TINYORB_NODE_LOOPBACK=2), "copies" (TINYORB_NODE_LOOPBACK=1) or "none" (one device, nothing to exchange); and the
 * number of ncclSend + ncclRecv pairs enqueued so far.  A failure in the middle of an enqueue sequence (HIP or RCCL) takes
 * the node out of service: every later extract / collate returns ORB_ESTATE -- destroy it and create a new one. */
const char *orb_node_exchange_backend(const OrbNode *node);
uint64_t orb_node_rccl_pairs(const OrbNode *node);
/* Where a job's results end up.  ORB_NODE_RESULTS_ROOT (default): collated on the first device, as described above.
 * ORB_NODE_RESULTS_SHARDED: nothing is exchanged -- every rank packs the stored records of its shard back to back, in frame
 * order, into buffers on ITS OWN device (for a consumer that runs where the frames were extracted, e.g. a matcher per GPU:
 * with eight GPUs the collate into one device runs at the rate of its links, the kernels do not).  The three stages and their
 * overlap stay; orb_node_collate_end fills counts[] and offsets[] as before (offsets[] keep counting across ranks: rank r's
 * records are offsets[first frame of r] .. and lie at the start of its own buffer) and returns NULL record pointers;
 * orb_node_shard_result hands out rank r's buffers of the job ended last (valid while the next two jobs are extracted).
 * May only be changed while no job is outstanding. */
#define ORB_NODE_RESULTS_ROOT 0
#define ORB_NODE_RESULTS_SHARDED 1
int orb_node_set_results(OrbNode *node, int where);
int orb_node_shard_result(OrbNode *node, int rank, uint32_t *n_frames, uint64_t *n_records, void **corners_dev,
                          void **descriptors_dev);
/* Copies the records of the job ended last to the host (sharded results: rank by rank, i.e. still in frame order); capacity in
 * records. */
int orb_node_read_collated(OrbNode *node, CornerData *corners, CornerDescriptor *descriptors, size_t capacity);

/* ---- descriptor matching (the SLAM stage that consumes this path's output; NOT in the
 * reference, definition is the build's own) ----
 * Brute-force Hamming matching between consecutive frames of the last batch: for every stored keypoint i of frame
 * f (query) the stored keypoint j of frame f+1 with the smallest popcount(desc_f[i] ^ desc_f+1[j]); ties go to the
 * smallest j.  `second` is the smallest distance over all other j (for a ratio test).  Without candidates:
 * index = ORB_MATCH_NONE, distance = second = 0xffff; with one candidate second = 0xffff.
 * Runs on the matrix cores (descriptors as +-1 in fp4, block-scaled MFMA: csrc/orb_kernels_match.h; the first call
 * allocates 128 bytes per record of a batch for them) when max_features <= 16383, else on the vector unit. */
typedef struct {
    uint32_t index;
    uint16_t distance;
    uint16_t second;
} OrbMatch;
#define ORB_MATCH_NONE 0xffffffffu
/* Matches frame f against f+1 for f in [0, n_frames - 1) of the last batch, asynchronously on `stream` (NULL: the
 * program's stream; it is ordered after the batch that produced the descriptors when that ran on the same stream).  A program has ONE
 * result buffer (and one buffer of expanded descriptors), whichever output set the batch went to: a call overwrites the matches of the
 * call before -- read them first -- and a call on another stream than the last one is ordered behind it. */
int orb_match_consecutive(OrbProgram *p, uint32_t n_frames, void *stream);
/* Copy up to n matches of the queries of `frame` to the host (synchronises). */
int orb_match_read(OrbProgram *p, uint32_t frame, OrbMatch *dst, size_t n);

/* Keypoint coordinates are in the octave's own pixel grid (fast.wgsl:143-150).  Centre of that pixel in level-0
 * pixel units, for consumers that work across octaves: a level-m texel covers 2^m level-0
 * pixels (exact halving; for odd sizes the blit's own mapping, blit.wgsl:17-36, differs by less than a pixel). */
void orb_corner_level0_xy(const CornerData *c, float *x0, float *y0);

/* ---- inspection (parity tests) ---- */
#define ORB_PLANE_GRAY 0
#define ORB_PLANE_BLUR 1
/* Copies one pyramid level (binary16 bit patterns, row-major) of a frame of the last call.
 * The fused pipeline does not materialise every plane; it returns ORB_ESTATE for those. */
int orb_debug_read_plane(OrbProgram *p, uint32_t frame, int kind, uint32_t level, uint16_t *dst, size_t n_texels);
/* Level geometry: width/height of mip `level` (max(1, W>>level), wgpu mip chain orb.rs:224-233). */
int orb_level_size(const OrbProgram *p, uint32_t level, uint32_t *width, uint32_t *height);
/* Device scalar helpers, so the tests can pin CRD-3 / CRD-9 on the GPU itself (host arrays). */
int orb_debug_f32_to_f16(OrbProgram *p, const float *src, uint16_t *dst, size_t n);
int orb_debug_angle_code(OrbProgram *p, const float *cy, const float *cx, uint32_t *dst, size_t n);
/* The program's table of the BRIEF pattern rotated by every angle code (brief.wgsl:50-57 evaluated once per code instead
 * of once per keypoint): codes x 256 tests x {a, b} int16 BYTE offsets 2 * (ry * pitch + rx) into a window of binary16 texels,
 * stored [code][lane 0..63][test lane + 64 e, e = 0..3][a, b].  Writes min(n_entries, codes * 512) values; *codes and *pitch
 * (either may be NULL) say what the table was built for. */
int orb_debug_rot_table(OrbProgram *p, int16_t *dst, size_t n_entries, uint32_t *codes, uint32_t *pitch);

/* ---- measurement ---- */
#define ORB_KERNEL_COUNT 22
/* When enabled every kernel launch is bracketed by hipEvents on its stream. */
int orb_profile_enable(OrbProgram *p, int enable);
int orb_profile_reset(OrbProgram *p);
/* Accumulated device time and number of launches of kernel `id` (synchronises). */
int orb_profile_get(OrbProgram *p, int id, double *total_ms, uint64_t *launches);
const char *orb_kernel_name(int id);

/* ---- synthetic frames, generated on the device ---- */
#define ORB_SYN_GRADIENT 1u
#define ORB_SYN_BLOBS 2u
#define ORB_SYN_WEDGES 4u
#define ORB_SYN_NOISE 8u
#define ORB_SYN_Y8 16u /* one byte per pixel: (77 R + 150 G + 29 B + 128) >> 8 of the recipe (ORB_FLAG_INPUT_Y8 programs) */
/* Frame i gets seed seed0+i.  frames_dev == NULL allocates/uses the program's own input slab
 * (max_batch frames) and returns its address in *out_dev. */
int orb_synth_frames_device(OrbProgram *p, uint8_t *frames_dev, uint32_t n_frames, uint32_t seed0, uint32_t flags,
                            uint8_t **out_dev);
/* Diagnostic: cycle sums of the phases of k_front (2 x 16 values: level 0, levels >= 1); filled only by a
 * diagnostic build (TINYORB_BUILD_STAMPS=1) when the program was created with TINYORB_STAMPS=1. */
int orb_debug_stamps(OrbProgram *p, unsigned long long *dst, size_t n);
/* Device-to-host copy helper for tests that have no other HIP binding. */
int orb_copy_to_host(OrbProgram *p, void *dst_host, const void *src_dev, size_t nbytes);

#ifdef __cplusplus
}
#endif
#endif /* TINYORB_H */
