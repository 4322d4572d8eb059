// orb_api.hip -- libtinyorb: the C ABI of include/tinyorb.h over the HIP kernels.
//
// Host-side counterpart of src/orb.rs in the reference: OrbProgram::init (orb.rs:107-219) becomes
// orb_program_create (device allocations instead of wgpu textures/buffers/pipelines), and
// extract_corners (orb.rs:469-557) becomes a short sequence of kernel launches on one HIP stream
// instead of 3*D render passes + D+1 compute dispatches + 3 staging copies.
#include <hip/hip_runtime.h>
#include <chrono>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <algorithm>
#include <vector>

#include "../../include/tinyorb.h"
#include "orb_front_launch.h"
#include "orb_kernels_fused.h"
#include "orb_kernels_brief.h"
#include "orb_kernels_intended.h"
#include "orb_kernels_staged.h"
#include "orb_kernels_collate.h"
#include "orb_kernels_match.h"

using namespace orb;

namespace {

enum KernelId { KID_GRAY = 0, KID_MIP, KID_BLUR, KID_FAST, KID_BRIEF, KID_FUSED_L0, KID_FUSED_LN, KID_SYNTH, KID_BRIEF_ROWS, KID_PREFIX, KID_FRONT_I, KID_SELECT_I, KID_BRIEF_I, KID_MATCH, KID_COMPACT, KID_BRIEF_T, KID_BRIEF_NF, KID_PACK_T, KID_UNPACK_T, KID_BRIEF_ONE, KID_FRONT_I_LN, KID_DESC_EXPAND };
const char* const kKernelNames[ORB_KERNEL_COUNT] = {"k_grayscale", "k_mip",      "k_blur_rows", "k_fast",       "k_brief",
                                                    "k_front_l0",  "k_front_ln", "k_synth",     "k_brief_rows", "k_slot_prefix",
                                                    "k_front_i_l0", "k_select_i", "k_brief_i",   "k_match",      "k_compact",
                                                    "k_brief_t",   "k_brief_nf",  "k_compact_transport", "k_unpack_transport",
                                                    "k_brief_one", "k_front_i_ln", "k_desc_expand"};

thread_local std::string g_create_error;

// k_front's instances by arithmetic form (orb_front_launch.h: front_form())
hipError_t (*const kFrontLaunch[8])(const FrontLaunch&) = {front_launch_fp0, front_launch_fp1, front_launch_fp2, front_launch_fp3,
                                                           front_launch_fp4, front_launch_fp5, front_launch_fp6, front_launch_fp7};
hipError_t (*const kFrontPairLaunch[8])(const FrontPairLaunch&) = {front_pair_launch_fp0, front_pair_launch_fp1, front_pair_launch_fp2, front_pair_launch_fp3,
                                                                   front_pair_launch_fp4, front_pair_launch_fp5, front_pair_launch_fp6, front_pair_launch_fp7};
hipError_t (*const kFrontSetMaxLds[8])(int) = {front_set_max_lds_fp0, front_set_max_lds_fp1, front_set_max_lds_fp2, front_set_max_lds_fp3,
                                               front_set_max_lds_fp4, front_set_max_lds_fp5, front_set_max_lds_fp6, front_set_max_lds_fp7};

struct ProfSpan {
    int kid;
    hipEvent_t start, stop;
};

}  // namespace

struct OrbProgram {
    OrbConfig cfg{};
    OrbOptions opt{};
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;  // chunked uploads of orb_extract_batch_host
    hipEvent_t order_event = nullptr;   // orders work on a caller's stream behind the last batch
    std::vector<hipEvent_t> upload_events;  // one per chunk of a chunked upload (reused)
    hipEvent_t upload_done = nullptr;       // behind the last host-to-device copy of the last host batch
    // orb_batch_pack / orb_batch_fetch: packed records of an output set on the device, its counters and offsets in pinned
    // host memory (written by the packing kernel), and the event that says the pack is done
    CornerData* d_pack_c[2] = {nullptr, nullptr};
    CornerDescriptor* d_pack_d[2] = {nullptr, nullptr};
    uint32_t* h_pack_counts[2] = {nullptr, nullptr};
    uint64_t* h_pack_offsets[2] = {nullptr, nullptr};
    hipEvent_t pack_event[2] = {nullptr, nullptr};
    uint32_t pack_n[2] = {0u, 0u};
    uint32_t cur_set = 0;
    Pyramid pyr{};
    size_t frame_bytes = 0;
    uint32_t max_batch = 1;
    float threshold = 0.f;
    uint32_t arc = 12;  // FAST arc length (opt-in extension, 9..16)
    uint32_t oob = kOobZero;  // OrbOptions::oob_policy
    float wq = 0.0f;          // OrbOptions::sampler_weight_bits as 2^bits (0: exact weights)
    // opt-in NMS (staged pipeline): provisional detections + score planes
    uint32_t cap_prov = 0;
    uint32_t* d_prov_counts = nullptr;
    CornerData* d_prov = nullptr;
    float* d_prov_scores = nullptr;
    float* d_score_planes = nullptr;
    ScoreLayout score_layout{};
    // "intended" mode: survivors of the NMS with their scores, input of the top-K cut
    bool intended = false;
    bool input_y8 = false;  // ORB_FLAG_INPUT_Y8: frames are one byte per pixel
    // fused "intended" pipeline (orb_kernels_intended.h): tile slots, their segments (record + score), the cut
    bool fused_i = false;
    bool igauss_fused = false;  // fused_i: k_front_i blurs its own tile (phase G); false: k_gauss over the stored grey plane (TINYORB_I_GAUSS_KERNEL=1)
    bool igrey_stored = false;  // fused_i: k_front_i<true> stores the level-0 grey plane (k_gauss or k_mip reads it)
    bool fused_x = false;  // the reference's algorithm with the opt-in arc / NMS on the tile kernels
    uint32_t* d_xband_counts = nullptr;  // band-slot counters written by the blur-only k_front launches (unused)
    RowsGeom xrows{};
    uint32_t xband_slots = 0;  // 16-row bands per frame over all levels (grid of the blur-only launches)
    IBriefGeom itiles{};
    CornerData* d_iseg = nullptr;
    float* d_iseg_scores = nullptr;
    uint32_t* d_iseg_counts = nullptr;
    uint32_t* d_iseg_before = nullptr;
    unsigned long long* d_thr_key = nullptr;
    uint32_t ibrief_lds = 0;
    MatchRecord* d_matches = nullptr;  // [max_batch][max_features], allocated by the first orb_match_consecutive
    uint8_t* d_desc8 = nullptr;        // [max_batch][max_features][128 or 256]: the descriptors as +-1 in fp4 (k_match_fp4) or +-127 in int8 (k_match_mfma)
    int match_valu = -1;               // which matcher (0 fp4, 1 vector unit: TINYORB_MATCH_VALU=1, 2 int8: TINYORB_MATCH_I8=1); read once
    // d_matches and d_desc8 are ONE buffer each per program, shared by both output sets and by whatever stream the caller passes: a match
    // on another stream is ordered behind the one before (its expand kernel would overwrite rows the earlier match still reads)
    hipEvent_t match_done = nullptr;
    hipStream_t match_stream = nullptr;
    uint32_t* d_prov2_counts = nullptr;
    CornerData* d_prov2 = nullptr;
    float* d_prov2_scores = nullptr;

    uint8_t* d_input = nullptr;  // max_batch frames (single-frame API, host batches, synth)
    // single-frame API: up to two images may be written ahead of extract_corners (orb_write_input_image_pinned uploads frame
    // k + 1 on the copy stream while frame k is extracted); slab 0 is d_input's first frame, slab 1 d_input_alt
    uint8_t* d_input_alt = nullptr;
    uint32_t in_last = 0;                 // slab of the image the last extract_corners worked on (and the next one will, if nothing is pending)
    uint32_t in_pending = 0;              // images written and not yet extracted (0..2)
    uint32_t in_queue[2] = {0u, 0u};      // their slabs, oldest first
    hipEvent_t in_uploaded[2] = {nullptr, nullptr};  // behind the asynchronous upload into the slab
    bool in_async[2] = {false, false};    // the slab's image came through the copy stream: extract_corners orders its kernels behind in_uploaded
    // environment switches (experiments and cross-checks), read once when the program is created
    struct {
        bool single_memcpy = false, single_split = false, single_serial = false, single_sync = false, single_block = false;
        bool no_swizzle = false, quiet = false;
        int phase_mask = -1, brief_i_mask = -1, lds_pad = 0;
    } env;
    // device-visible addresses of the pinned single-frame staging arrays (resolved once)
    uint32_t* dv_count = nullptr;
    CornerData* dv_corners = nullptr;
    CornerDescriptor* dv_desc = nullptr;
    uint16_t* d_gray = nullptr;  // max_batch x pyr.stride
    uint16_t* d_blur = nullptr;
    uint16_t* d_blur_rowc = nullptr;  // fused path: [max_batch][row_stride] blur row constants (columns < qa)
    uint32_t blur_qa[kMaxLevels] = {0};  // fused path: per level, columns [0, qa) of the blur plane live in d_blur_rowc
    uint32_t* d_counts = nullptr;  // currently selected output set (orb_batch_select_output)
    CornerData* d_corners = nullptr;
    CornerDescriptor* d_desc = nullptr;
    uint32_t* out_counts[2] = {nullptr, nullptr};
    CornerData* out_corners[2] = {nullptr, nullptr};
    CornerDescriptor* out_desc[2] = {nullptr, nullptr};
    CornerData* d_seg = nullptr;     // fused path: [max_batch][n_slots][seg_classes][seg_cap] band segments
    uint32_t* d_seg_counts = nullptr;  // [max_batch][n_slots][seg_classes]
    uint32_t* d_seg_before = nullptr;  // [max_batch][seg_classes][n_slots] exclusive prefix of the stored counts
    BandGeom bands{};
    RowsGeom rows{};
    BriefTGeom brieft{};       // thread-per-keypoint BRIEF of the fused literal pipelines (plain and arc/NMS)
    bool use_brief_t = false;
    uint32_t band_rows_lvl[kMaxLevels] = {0};  // band height of the plain fused path per level: kFrontRows or kFrontRowsWide (chosen at create)
    uint32_t ln_threads[kMaxLevels] = {0};     // levels >= 1: threads of a band's workgroup, kFrontThreadsLN or kFrontThreadsLNBig (chosen at create)
    uint32_t tile_w_lvl[kMaxLevels] = {0};     // 0: full-width bands; else the level runs on column tiles of this width (k_front<..., TILED>)
    uint32_t seg_classes = 1;  // lists per band slot of the plain fused path: 2 with k_brief_t (angle code 0 / the rest)
    uint32_t* d_pattern = nullptr;
    float* d_cos = nullptr;
    float* d_sin = nullptr;
    uint4* d_rot = nullptr;  // the pattern rotated by every angle code (k_rot_table), for k_brief_nf or k_brief_i
    unsigned long long* d_stamps = nullptr;  // TINYORB_STAMPS=1: phase cycle sums of k_front (2 x 16 slots)

    // host staging of the single-frame API (orb.rs:216-218 staging buffers)
    uint32_t* h_count = nullptr;      // pinned: [0] the raw counter of the last single-frame extract, [kSingleDoneWord] its completion sequence number (own cache line)
    uint32_t* d_single_done = nullptr;  // workgroups of k_brief_one that have finished (the last one publishes the sequence number and clears it)
    hipEvent_t single_done_ev = nullptr;  // blocking-sync event behind k_brief_one (TINYORB_SINGLE_WAIT=block / ORB_FLAG_SINGLE_BLOCKING_WAIT)
    uint32_t single_seq = 0;
    CornerData* h_corners = nullptr;
    CornerDescriptor* h_desc = nullptr;
    bool single_valid = false;

    uint32_t last_batch = 0;  // frames of the last batched call
    hipStream_t last_stream = nullptr;
    bool planes_valid = false;
    bool fused = false;
    uint32_t max_lds = 0;
    uint32_t n_cus = 256;

    bool profiling = false;
    std::vector<ProfSpan> pending;
    std::vector<hipEvent_t> event_pool;
    double prof_ms[ORB_KERNEL_COUNT] = {0};
    uint64_t prof_n[ORB_KERNEL_COUNT] = {0};

    std::string pipeline_note;  // why the staged kernels were chosen when nobody asked for them (else empty)
    std::string err;
};

namespace {

int fail(OrbProgram* p, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (p)
        p->err = buf;
    else
        g_create_error = buf;
    return code;
}

#define HIP_TRY(p, expr)                                                                                  \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) return fail((p), ORB_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// Bracket one kernel launch with events when profiling is on.
struct LaunchScope {
    OrbProgram* p;
    hipStream_t s;
    int kid;
    hipEvent_t start = nullptr, stop = nullptr;
    LaunchScope(OrbProgram* p_, hipStream_t s_, int kid_) : p(p_), s(s_), kid(kid_) {
        if (!p->profiling) return;
        auto get = [&]() -> hipEvent_t {
            if (!p->event_pool.empty()) {
                hipEvent_t e = p->event_pool.back();
                p->event_pool.pop_back();
                return e;
            }
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) return nullptr;
            return e;
        };
        start = get();
        stop = get();
        if (start && stop) (void)hipEventRecord(start, s);
    }
    ~LaunchScope() {
        if (start && stop) {
            (void)hipEventRecord(stop, s);
            p->pending.push_back({kid, start, stop});
        }
    }
};

int drain_profile(OrbProgram* p) {
    for (auto& sp : p->pending) {
        HIP_TRY(p, hipEventSynchronize(sp.stop));
        float ms = 0.f;
        HIP_TRY(p, hipEventElapsedTime(&ms, sp.start, sp.stop));
        p->prof_ms[sp.kid] += ms;
        p->prof_n[sp.kid] += 1;
        p->event_pool.push_back(sp.start);
        p->event_pool.push_back(sp.stop);
    }
    p->pending.clear();
    return ORB_OK;
}

void layout_pyramid(uint32_t W, uint32_t H, uint32_t depth, Pyramid* pyr) {
    memset(pyr, 0, sizeof *pyr);
    pyr->depth = depth;
    uint32_t off = 0;
    for (uint32_t m = 0; m < depth; m++) {
        uint32_t w = W >> m, h = H >> m;  // wgpu mip chain: max(1, dim >> m)
        pyr->w[m] = w ? w : 1u;
        pyr->h[m] = h ? h : 1u;
        pyr->off[m] = off;
        off += pyr->w[m] * pyr->h[m];
        off = (off + 7u) & ~7u;  // keep every level 16-byte aligned
    }
    pyr->stride = (off + 63u) & ~63u;
    uint32_t rows = 0;
    for (uint32_t m = 0; m < depth; m++) {
        pyr->row_off[m] = rows;
        rows += pyr->h[m];
    }
    pyr->row_stride = (rows + 7u) & ~7u;
}

// The staged pipeline for `n` frames starting at device pointer `frames` (orb.rs:469-534).
int run_staged(OrbProgram* p, const uint8_t* frames, uint32_t n, hipStream_t s) {
    const Pyramid& pyr = p->pyr;
    const uint32_t W = pyr.w[0], H = pyr.h[0], D = pyr.depth, cap = p->cfg.max_features;
    HIP_TRY(p, hipMemsetAsync(p->d_counts, 0, sizeof(uint32_t) * n, s));  // orb.rs:475 clear_buffer(counter)
    {
        LaunchScope ls(p, s, KID_GRAY);
        dim3 grid((W + 1023u) / 1024u, H, n);
        if (p->input_y8)
            hipLaunchKernelGGL(k_grayscale_y8, grid, dim3(256), 0, s, frames, p->frame_bytes, p->d_gray, pyr);
        else if (p->intended)
            hipLaunchKernelGGL(k_grayscale<true>, grid, dim3(256), 0, s, frames, p->frame_bytes, p->d_gray, pyr);
        else
            hipLaunchKernelGGL(k_grayscale<false>, grid, dim3(256), 0, s, frames, p->frame_bytes, p->d_gray, pyr, p->opt.fp_contract);
    }
    for (uint32_t m = 1; m < D; m++) {  // orb.rs:413-429
        LaunchScope ls(p, s, KID_MIP);
        dim3 grid((pyr.w[m] + 63u) / 64u, (pyr.h[m] + 4u * kMipRows - 1u) / (4u * kMipRows), n);
        hipLaunchKernelGGL(k_mip, grid, dim3(64, 4), 0, s, p->d_gray, pyr, m, (float)pyr.w[m - 1] / (float)pyr.w[m], (float)pyr.h[m - 1] / (float)pyr.h[m], p->wq);
    }
    for (uint32_t m = 0; m < D; m++) {  // orb.rs:432-466 (both passes)
        LaunchScope ls(p, s, KID_BLUR);
        if (p->intended) {
            hipLaunchKernelGGL(k_gauss, dim3((pyr.w[m] + kGaussTW - 1u) / kGaussTW, (pyr.h[m] + kGaussTH - 1u) / kGaussTH, n), dim3(256), 0, s,
                               p->d_gray, p->d_blur, pyr, m);
            continue;
        }
        dim3 grid(pyr.h[m], 1, n);
        size_t lds = (size_t)pyr.w[m] * 2u * sizeof(uint16_t);
        hipLaunchKernelGGL(k_blur_rows, grid, dim3(256), lds, s, p->d_gray, p->d_blur, pyr, m, p->wq, p->opt.fp_contract);
    }
    const bool nms = (p->opt.flags & ORB_FLAG_NMS) != 0u;
    const uint32_t im = p->intended ? 1u : 0u;
    const bool prov = nms || p->intended;  // the detector writes the provisional list
    if (prov) HIP_TRY(p, hipMemsetAsync(p->d_prov_counts, 0, sizeof(uint32_t) * n, s));
    if (nms && p->intended) HIP_TRY(p, hipMemsetAsync(p->d_prov2_counts, 0, sizeof(uint32_t) * n, s));
    if (nms) {
        HIP_TRY(p, hipMemsetAsync(p->d_score_planes, 0, sizeof(float) * (size_t)p->score_layout.stride * n, s));
    }
    uint32_t width = W, height = H;  // orb.rs:501-519
    for (uint32_t oct = 0; oct < D; oct++) {
        const uint32_t gw = ((width + 7u) / 8u) * 8u, gh = ((height + 7u) / 8u) * 8u;
        if (gw && gh) {
            LaunchScope ls(p, s, KID_FAST);
            dim3 grid((gw + 15u) / 16u, (gh + 15u) / 16u, n);
            if (prov)
                hipLaunchKernelGGL(k_fast, grid, dim3(16, 16), 0, s, p->d_gray, pyr, oct, gw, gh, p->threshold, p->arc, im,
                                   p->d_prov_counts, p->d_prov, p->cap_prov, p->d_prov_scores,
                                   nms ? p->d_score_planes : (float*)nullptr, p->score_layout);
            else
                hipLaunchKernelGGL(k_fast, grid, dim3(16, 16), 0, s, p->d_gray, pyr, oct, gw, gh, p->threshold, p->arc, im,
                                   p->d_counts, p->d_corners, cap, (float*)nullptr, (float*)nullptr, p->score_layout, p->oob);
        }
        width /= 2u;
        height /= 2u;
    }
    const uint32_t nms_blocks = std::min<uint32_t>((p->cap_prov + 255u) / 256u, 64u);
    if (nms && !p->intended) {
        LaunchScope ls(p, s, KID_FAST);
        hipLaunchKernelGGL(k_nms, dim3(nms_blocks, 1, n), dim3(256), 0, s, p->d_prov_counts, p->d_prov,
                           p->d_prov_scores, p->cap_prov, p->d_score_planes, p->score_layout, p->d_counts, p->d_corners,
                           cap, (float*)nullptr);
    }
    if (p->intended) {  // [NMS ->] top-K cut -> final list (IM-7, IM-8)
        LaunchScope ls(p, s, KID_FAST);
        if (nms)
            hipLaunchKernelGGL(k_nms, dim3(nms_blocks, 1, n), dim3(256), 0, s, p->d_prov_counts,
                               p->d_prov, p->d_prov_scores, p->cap_prov, p->d_score_planes, p->score_layout,
                               p->d_prov2_counts, p->d_prov2, p->cap_prov, p->d_prov2_scores);
        hipLaunchKernelGGL(k_topk, dim3(n), dim3(1024), 0, s, nms ? p->d_prov2_counts : p->d_prov_counts,
                           nms ? p->d_prov2 : p->d_prov, nms ? p->d_prov2_scores : p->d_prov_scores, p->cap_prov,
                           p->d_counts, p->d_corners, cap);
    }
    {  // orb.rs:523-534
        LaunchScope ls(p, s, KID_BRIEF);
        BriefTables tab{p->d_pattern, p->d_cos, p->d_sin, p->d_rot};
        uint32_t bx = (cap + 127u) / 128u;  // ~32 keypoints per wave at a full frame
        if (bx > 64u) bx = 64u;
        if (bx < 1u) bx = 1u;
        hipLaunchKernelGGL(k_brief, dim3(bx, 1, n), dim3(256), 0, s, p->d_blur, pyr, p->d_counts, p->d_corners, cap,
                           p->d_desc, tab, im, p->oob, p->opt.fp_contract, p->opt.angle_bins);
    }
    HIP_TRY(p, hipGetLastError());
    p->planes_valid = true;
    return ORB_OK;
}

// Can the fused per-level kernels handle this configuration?  (Otherwise: staged pipeline.)
bool fused_eligible(const OrbProgram* p) {
    if (p->opt.flags & (ORB_FLAG_STAGED | ORB_FLAG_NMS | ORB_FLAG_INTENDED)) return false;
    if (p->arc != 12u) return false;  // the fused FAST phase is specialised for the reference's 12-run
    const Pyramid& pyr = p->pyr;
    // index arithmetic: v_mul_i32_i24 takes 24-bit operands (rows, widths < 2^14 here) and returns 32 bits; RGBA byte
    // offsets inside a frame are 4 * W * H < 2^32
    if ((uint64_t)pyr.w[0] * pyr.h[0] > (1ull << 26) || pyr.h[0] > 16384u) return false;
    if (pyr.w[0] > (uint32_t)kFrontMaxWidthTiled || pyr.w[0] < 8u) return false;
    // rows of any width: k_front<..., UA> loads RGBA texel by texel (4-byte aligned) and Y8 byte by byte
    return true;  // a level 1 that is not an exact half is built by k_mip from the stored level-0 plane (FrontGeom::store_grey)
}

uint32_t front_bands(const Pyramid& pyr, uint32_t lvl, uint32_t band_rows) {
    const uint32_t gh = (((pyr.h[0] >> lvl) + 7u) / 8u) * 8u;  // orb.rs:513, 518
    const uint32_t rows = pyr.h[lvl] > gh ? pyr.h[lvl] : gh;
    return (rows + band_rows - 1) / band_rows;
}

// tile_w = 0: full-width bands; else column tiles of that width (rounded up to 8; FrontGeom::tiled)
FrontGeom front_geometry(const Pyramid& pyr, uint32_t lvl, uint32_t gw, uint32_t gh, uint32_t n_frames,
                         uint32_t band_rows = kFrontRows, uint32_t tile_w = 0, uint32_t threads = 0) {
    FrontGeom g{};
    g.lvl = lvl;
    g.rows = band_rows;
    g.gw = gw;
    g.gh = gh;
    const uint32_t w = pyr.w[lvl], h = pyr.h[lvl];
    const uint32_t rows = h > gh ? h : gh;
    g.n_bands = (rows + band_rows - 1) / band_rows;
    g.n_frames = n_frames;
    const uint32_t cols = (w > gw ? w : gw) + 4u;
    g.ls = kLdsPad + ((cols + 7u) & ~7u);
    g.ts = (w + 7u) & ~7u;
    g.write_mip = (lvl + 1 < pyr.depth && w == 2u * pyr.w[lvl + 1] && h == 2u * pyr.h[lvl + 1]) ? 1u : 0u;
    g.xcd_swizzle = (n_frames % 8u == 0u) ? 1u : 0u;
    {   // constant stretches of the literal blur (tap 1 is monotone in x): see k_front phase C
        uint32_t P = 0;
        while (P < w && blur_tap(P, w, kBlurOffHost).i1 == 0) P++;
        uint32_t Q = 0;
        if (P > 0) while (Q < w && (uint32_t)blur_tap(Q, w, kBlurOffHost).i1 < P) Q++;
        g.blur_p = P;
        g.blur_q = Q;
        g.n_var = w - Q;
    }
    if (tile_w) {
        const uint32_t dom = ((w > gw ? w : gw) + 7u) & ~7u;  // columns of the level / of its dispatch domain
        g.tiled = 1u;
        g.tw = std::min((tile_w + 7u) & ~7u, dom);
        g.n_ct = (dom + g.tw - 1u) / g.tw;
        g.xb = 3u;
        while ((1u << g.xb) < g.tw) g.xb++;
        g.ls = kLdsPad + ((g.tw + 4u + 7u) & ~7u);
        g.ts = std::min(g.tw, (w + 7u) & ~7u);
        const BlurTap t = blur_tap(w - 1u, w, kBlurOffHost);
        g.far_i0 = (uint32_t)t.i0;
        g.far_i1 = (uint32_t)t.i1;
        // queues B (3 ts entries) and C (ts), or the blur column table (24 bytes per column that varies), whichever is larger
        g.tmp_halfs = std::max<uint32_t>(2u * kFrontTmpRows * g.ts, (uint32_t)(sizeof(BlurCol) / 2u) * g.n_var);
        g.tmp_halfs = (g.tmp_halfs + 7u) & ~7u;
    }
    {   // pre-test items of a band (tile): rows x ceil(dispatch columns / 16) at level 0 and on 1024 threads, / 8 otherwise (orb_front_body.inc, B1)
        const uint32_t cols_t = g.tiled ? std::min(g.tw, gw) : gw, iw = (lvl == 0 || threads == (uint32_t)kFrontThreadsLNBig) ? 16u : 8u;
        g.ovf_words = (band_rows * ((cols_t + iw - 1u) / iw) + 31u) / 32u;
    }
    g.n_classes = 1u;
    g.phase_mask = 15u;  // run_fused_range applies the program's experiment switches (TINYORB_PHASE_MASK, TINYORB_NO_SWIZZLE)
    return g;
}

// Can tile 0 of a tiled level do the band's blur from its own columns?  The column table's grey texels lie in [0, ~0.12 w].
bool front_tile0_holds_blur(const FrontGeom& g, uint32_t w) {
    if (!g.tiled || g.n_ct == 1u) return true;
    uint32_t hi = 0;
    for (uint32_t x = g.blur_q; x < w; x++) {
        const BlurTap t2 = blur_tap(x, w, kBlurOffHost);
        for (int j : {t2.i0, t2.i1})
            if ((uint32_t)j >= g.blur_p) hi = std::max<uint32_t>(hi, (uint32_t)blur_tap((uint32_t)j, w, kBlurOffHost).i1);
    }
    return hi + 1u <= g.tw;
}

// Geometry of k_brief_t over the band (or tile) slots described by `rg`; false when the frame is too large for its LDS
// staging (then k_brief_rows does the work).
bool brieft_geometry(const OrbProgram* p, const RowsGeom& rg, uint32_t n_classes, BriefTGeom* out) {
    BriefTGeom g{};
    g.n_slots = rg.n_slots;
    g.seg_cap = rg.seg_cap;
    g.n_classes = n_classes;
    uint32_t rows = 0;
    for (uint32_t m = 0; m < p->pyr.depth; m++) {
        g.flat_end[m] = rg.flat_end[m];
        g.qa[m] = rg.qa[m];
        // 18 zero rows in front of a level and 26 behind it: the patch reaches 18 rows past the keypoint, and at
        // octaves >= 1 a keypoint may sit up to 7 rows below the level's last row (8-rounded dispatch, Q8)
        g.row_base[m] = rows + (uint32_t)kBriefHalo;
        rows += p->pyr.h[m] + 2u * (uint32_t)kBriefHalo + 8u;
    }
    g.rows_padded = rows;
    g.oob = p->oob;
    *out = g;
    if (getenv("TINYORB_BRIEF_ROWS")) return false;  // A/B and cross-check: the wave-per-keypoint kernel
    return rg.n_slots >= 1u && rg.n_slots * n_classes <= kBriefTMaxSlots && rows <= kBriefTMaxRows;
}

// k_slot_prefix + BRIEF over the band (or tile) slots `rows_geom` of frames [f0, f0 + n).
int launch_brief(OrbProgram* p, hipStream_t s, uint32_t n, const RowsGeom& rows_geom, const uint16_t* d_blur,
                 const uint16_t* d_blur_rowc, const uint32_t* seg_counts, uint32_t* seg_before, const CornerData* seg,
                 uint32_t* d_counts, CornerData* d_corners, CornerDescriptor* d_desc) {
    const uint32_t cap = p->cfg.max_features;
    const BriefTables tab{p->d_pattern, p->d_cos, p->d_sin, p->d_rot};
    const bool use_t = p->use_brief_t;
    {
        LaunchScope ls(p, s, KID_PREFIX);
        hipLaunchKernelGGL(k_slot_prefix, dim3(n), dim3(64), 0, s, seg_counts, seg_before, d_counts, rows_geom.n_slots,
                           rows_geom.seg_cap, use_t ? p->brieft.n_classes : 1u);
    }
    if (use_t) {
        const BriefTGeom& tg = p->brieft;
        const dim3 grid(n, (cap + (uint32_t)kBriefNfChunk - 1u) / (uint32_t)kBriefNfChunk);  // k_brief_nf's
        {
            LaunchScope ls(p, s, KID_BRIEF_T);
#define BRIEF_T_LAUNCH(ROT_)                                                                                                         \
    hipLaunchKernelGGL((k_brief_t<kBriefTWaves, ROT_>), dim3(n, (cap + kBriefTThreads - 1u) / kBriefTThreads), dim3(kBriefTThreads), \
                       brieft_lds_bytes(tg), s, d_blur_rowc, p->pyr, tg, seg_counts, seg_before, seg, d_corners, cap, d_desc, tab)
            switch (rot_form(p->opt.fp_contract)) {  // the rotation's form (OrbOptions::fp_contract): a template parameter of the straight-line tests
                case 1: BRIEF_T_LAUNCH(1); break;
                case 2: BRIEF_T_LAUNCH(2); break;
                default: BRIEF_T_LAUNCH(0); break;
            }
#undef BRIEF_T_LAUNCH
        }
        LaunchScope ls(p, s, KID_BRIEF_NF);
        if (p->oob != kOobZero)
            hipLaunchKernelGGL(k_brief_nf<true>, grid, dim3(256), 0, s, d_blur, d_blur_rowc, p->pyr, tg, seg_counts, seg_before, d_corners,
                               cap, d_desc, tab);
        else
            hipLaunchKernelGGL(k_brief_nf<false>, grid, dim3(256), 0, s, d_blur, d_blur_rowc, p->pyr, tg, seg_counts, seg_before, d_corners,
                               cap, d_desc, tab);
    } else {
        LaunchScope ls(p, s, KID_BRIEF_ROWS);
        RowsGeom rg = rows_geom;
        rg.oob = p->oob;
        rg.fp = p->opt.fp_contract;
        rg.split = 1u;  // small batches: several workgroups per band slot so that the chip still sees ~2000 of them
        while (rg.split < 16u && rg.n_slots * n * rg.split < 2048u) rg.split *= 2u;
        hipLaunchKernelGGL(k_brief_rows, dim3(rg.n_slots * rg.split, n), dim3(256), 0, s, d_blur, d_blur_rowc, p->pyr, rg,
                           seg_counts, seg_before, seg, d_corners, cap, d_desc, tab);
    }
    return ORB_OK;
}

// The fused pipeline for frames [f0, f0 + n) of the batch on stream s: one k_front launch per level + BRIEF
// (with_brief = false: the caller launches its own BRIEF kernel -- the single-frame path's k_brief_one).
// first_level > 0 (single-frame call): the levels below it have been launched already (k_front_pair).
int run_fused_range(OrbProgram* p, const uint8_t* frames_all, uint32_t f0, uint32_t n, hipStream_t s, bool with_brief = true,
                    uint32_t first_level = 0) {
    const Pyramid& pyr = p->pyr;
    const uint32_t D = pyr.depth, cap = p->cfg.max_features;
    const uint8_t* frames = frames_all + (size_t)f0 * p->frame_bytes;
    uint16_t* const d_gray = p->d_gray + (size_t)f0 * pyr.stride;
    uint16_t* const d_blur = p->d_blur + (size_t)f0 * pyr.stride;
    uint16_t* const d_blur_rowc = p->d_blur_rowc + (size_t)f0 * pyr.row_stride;
    const size_t lists = (size_t)p->bands.n_slots * p->seg_classes;  // lists per frame
    uint32_t* const d_seg_counts = p->d_seg_counts + (size_t)f0 * lists;
    uint32_t* const d_seg_before = p->d_seg_before + (size_t)f0 * lists;
    CornerData* const d_seg = p->d_seg + (size_t)f0 * lists * p->bands.seg_cap;
    uint32_t* const d_counts = p->d_counts + f0;
    CornerData* const d_corners = p->d_corners + (size_t)f0 * cap;
    CornerDescriptor* const d_desc = p->d_desc + (size_t)f0 * cap;
    // orb.rs:475 clear_buffer(counter): every band slot's count is rewritten by its k_front block and
    // counts[] by k_slot_prefix, so nothing needs clearing here.
    uint32_t width = pyr.w[0], height = pyr.h[0];  // orb.rs:501-519
    for (uint32_t lvl = 0; lvl < D; lvl++) {
        const uint32_t gw = ((width + 7u) / 8u) * 8u, gh = ((height + 7u) / 8u) * 8u;
        width /= 2u;
        height /= 2u;
        if (lvl < first_level) continue;  // launched by the caller
        if (lvl > 0 && !(pyr.w[lvl - 1] == 2u * pyr.w[lvl] && pyr.h[lvl - 1] == 2u * pyr.h[lvl])) {
            hipStream_t sm = s;
            LaunchScope ls(p, sm, KID_MIP);  // inexact reduction (odd source size): generic bilinear blit
            dim3 grid((pyr.w[lvl] + 63u) / 64u, (pyr.h[lvl] + 4u * kMipRows - 1u) / (4u * kMipRows), n);
            hipLaunchKernelGGL(k_mip, grid, dim3(64, 4), 0, sm, d_gray, pyr, lvl, (float)pyr.w[lvl - 1] / (float)pyr.w[lvl], (float)pyr.h[lvl - 1] / (float)pyr.h[lvl], p->wq);
        }
        FrontGeom g = front_geometry(pyr, lvl, gw ? gw : 8u, gh, n, p->band_rows_lvl[lvl], p->tile_w_lvl[lvl], p->ln_threads[lvl]);
        hipStream_t s_lvl = s;
        if (gw == 0) g.gh = 0;  // no FAST dispatch at this octave (orb.rs:511-515 with width 0)
        g.slot_base = p->bands.slot_base[lvl];
        g.n_slots = p->bands.n_slots;
        g.seg_cap = p->bands.seg_cap;
        g.n_classes = p->seg_classes;
        g.stamps = p->d_stamps;
        if (p->env.phase_mask >= 0) g.phase_mask = (uint32_t)p->env.phase_mask;
        if (p->env.no_swizzle) g.xcd_swizzle = 0u;
        g.oob = p->oob;
        g.wq = p->wq;
        g.fp = p->opt.fp_contract;
        g.store_grey = (lvl == 0 && D > 1 && !(pyr.w[0] == 2u * pyr.w[1] && pyr.h[0] == 2u * pyr.h[1])) ? 1u : 0u;
        if (g.n_bands * (g.tiled ? g.n_ct : 1u) != p->bands.slot_base[lvl + 1] - p->bands.slot_base[lvl])
            return fail(p, ORB_EINVAL, "internal: band count mismatch at level %u", lvl);
        if (!g.tiled && sizeof(BlurCol) * (size_t)g.n_var > 8u * (size_t)g.ts)  // the column table borrows the queues' storage
            return fail(p, ORB_EINVAL, "internal: blur column table of level %u does not fit", lvl);
        uint32_t lds = front_lds_bytes(g);
        lds += (uint32_t)p->env.lds_pad;  // TINYORB_LDS_PAD: occupancy experiments only
        if (lds > p->max_lds) return fail(p, ORB_EINVAL, "level %u needs %u bytes of LDS", lvl, lds);
        const dim3 grid(g.n_bands * (g.tiled ? g.n_ct : 1u) * n);
        // The kernel instances live in orb_front_inst.hip, one translation unit per arithmetic form (orb_front_launch.h); the form a
        // launch takes carries the luminance bits only where a luminance is computed (level 0 from RGBA).
        FrontLaunch L{frames, p->frame_bytes, d_gray, d_blur, d_blur_rowc, pyr, g, p->threshold, d_seg_counts, d_seg, s_lvl, grid.x, lds,
                      p->band_rows_lvl[lvl], p->ln_threads[lvl], p->input_y8,
                      // rows not aligned to a quad, or the level-0 plane is needed (level 1 not an exact half): the general variant
                      lvl == 0 && ((pyr.w[0] & 3u) || g.store_grey), p->oob != kOobZero, false};
        {
            LaunchScope ls(p, s_lvl, lvl == 0 ? KID_FUSED_L0 : KID_FUSED_LN);
            const hipError_t e = kFrontLaunch[front_form(p->opt.fp_contract, lvl == 0 && !p->input_y8)](L);
            if (e != hipSuccess) return fail(p, ORB_EHIP, "k_front (level %u) failed to launch: %s", lvl, hipGetErrorString(e));
        }
    }
    // orb.rs:523-534, plus the compaction of the band segments into the final lists
    if (with_brief) launch_brief(p, s, n, p->rows, d_blur, d_blur_rowc, d_seg_counts, d_seg_before, d_seg, d_counts, d_corners, d_desc);
    HIP_TRY(p, hipGetLastError());
    return ORB_OK;
}

int run_fused(OrbProgram* p, const uint8_t* frames, uint32_t n, hipStream_t s) {
    // One launch per kernel for the whole batch.  Cutting the batch was measured twice and lost both times (NOTEBOOK.md): round 1, whole
    // pipelines of 2 / 4 / 8 chunks on two streams: + 5 %, + 4 %, + 6 % time; round 4 (TINYORB_BATCH_SPLIT, profiles/r04_split_ab.txt), the
    // BRIEF kernels of sub-range i on a side stream under the front kernels of sub-range i + 1: 45 us MORE per cut -- k_front's two
    // workgroups per CU hold all of the CU's LDS and wave slots, k_brief_t is as bound by the vector units as k_front is, and every
    // cross-queue dependency costs a signal round trip.  The experiment's code left the tree in round 5.
    if (int rc = run_fused_range(p, frames, 0, n, s)) return rc;
    p->planes_valid = true;  // except the level-0 grey plane, which the fused path keeps in LDS only
    return ORB_OK;
}

// Tile width of a level for the fused "intended" pipeline.
uint32_t itile_width(uint32_t w) {
    const uint32_t w8 = (w + 7u) & ~7u;
    return w8 < (uint32_t)kITileW ? w8 : (uint32_t)kITileW;
}

bool fused_i_eligible(const OrbProgram* p) {
    if (!p->intended || (p->opt.flags & ORB_FLAG_STAGED)) return false;
    const Pyramid& pyr = p->pyr;
    if ((uint64_t)pyr.w[0] * pyr.h[0] > (1ull << 26)) return false;  // 32-bit byte offsets; W, H <= 16384 (create)
    if ((pyr.w[0] & 3u) != 0u || pyr.w[0] < 8u) return false;       // level 0 is read as RGBA quads
    return true;
}

// The fused "intended" pipeline: k_front_i per level (+ k_mip where a level is not an exact half), k_gauss per
// level, k_select_i, k_brief_i.
int run_fused_i(OrbProgram* p, const uint8_t* frames, uint32_t n, hipStream_t s) {
    const Pyramid& pyr = p->pyr;
    const uint32_t D = pyr.depth, cap = p->cfg.max_features;
    const IBriefGeom& bg = p->itiles;
    for (uint32_t lvl = 0; lvl < D; lvl++) {
        if (lvl > 0 && !(pyr.w[lvl - 1] == 2u * pyr.w[lvl] && pyr.h[lvl - 1] == 2u * pyr.h[lvl])) {
            LaunchScope ls(p, s, KID_MIP);
            dim3 grid((pyr.w[lvl] + 63u) / 64u, (pyr.h[lvl] + 4u * kMipRows - 1u) / (4u * kMipRows), n);
            hipLaunchKernelGGL(k_mip, grid, dim3(64, 4), 0, s, p->d_gray, pyr, lvl, (float)pyr.w[lvl - 1] / (float)pyr.w[lvl], (float)pyr.h[lvl - 1] / (float)pyr.h[lvl]);
        }
        IGeom g{};
        g.lvl = lvl;
        g.tw = bg.tw[lvl];
        g.n_ct = bg.n_ct[lvl];
        g.n_bands = (pyr.h[lvl] + kFrontRows - 1) / kFrontRows;
        g.ls = kIPad + g.tw + 16u;
        g.write_mip = (lvl + 1 < D && pyr.w[lvl] == 2u * pyr.w[lvl + 1] && pyr.h[lvl] == 2u * pyr.h[lvl + 1]) ? 1u : 0u;
        g.xcd_swizzle = (n % 8u == 0u) ? 1u : 0u;
        g.slot_base = bg.slot_base[lvl];
        g.n_slots = bg.n_slots;
        g.seg_cap = bg.seg_cap;
        g.arc = p->arc;
        g.nms = (p->opt.flags & ORB_FLAG_NMS) ? 1u : 0u;
        g.phase_mask = p->env.phase_mask >= 0 ? (uint32_t)p->env.phase_mask : 63u;
        g.literal = 0u;
        g.dw = pyr.w[lvl];
        g.dh = pyr.h[lvl];
        g.gx1 = (uint32_t)((int)pyr.w[lvl] - 16);
        g.gy1 = (uint32_t)((int)pyr.h[lvl] - 16);
        g.blur = p->igauss_fused ? 1u : 0u;
        g.store_grey = p->igrey_stored ? 1u : 0u;
        if (g.n_bands * g.n_ct != bg.slot_base[lvl + 1] - bg.slot_base[lvl])
            return fail(p, ORB_EINVAL, "internal: tile count mismatch at level %u", lvl);
        if (g.blur && !ifront_gauss_fits(g.tw)) return fail(p, ORB_EINVAL, "internal: tile width %u too wide for the fused Gaussian", g.tw);
        const dim3 grid(g.n_bands * g.n_ct * n);
        LaunchScope ls(p, s, lvl == 0 ? KID_FRONT_I : KID_FRONT_I_LN);  // level 0 (RGBA in) and the levels above are different launches: one id each
        if (lvl == 0)
            hipLaunchKernelGGL(k_front_i<true>, grid, dim3(kIThreads), ifront_lds_bytes(g), s, frames, p->frame_bytes,
                               p->d_gray, p->d_blur, pyr, g, p->threshold, p->d_iseg_counts, p->d_iseg, p->d_iseg_scores);
        else
            hipLaunchKernelGGL(k_front_i<false>, grid, dim3(kIThreads), ifront_lds_bytes(g), s, frames, p->frame_bytes,
                               p->d_gray, p->d_blur, pyr, g, p->threshold, p->d_iseg_counts, p->d_iseg, p->d_iseg_scores);
    }
    for (uint32_t m = 0; m < D && !p->igauss_fused; m++) {
        LaunchScope ls(p, s, KID_BLUR);
        hipLaunchKernelGGL(k_gauss, dim3((pyr.w[m] + kGaussTW - 1u) / kGaussTW, (pyr.h[m] + kGaussTH - 1u) / kGaussTH, n), dim3(256), 0, s, p->d_gray,
                           p->d_blur, pyr, m);
    }
    {
        LaunchScope ls(p, s, KID_SELECT_I);
        hipLaunchKernelGGL(k_select_i, dim3(n), dim3(1024), 0, s, p->d_iseg_counts, p->d_iseg, p->d_iseg_scores, bg.n_slots,
                           bg.seg_cap, cap, p->d_counts, p->d_thr_key, p->d_iseg_before);
    }
    {
        LaunchScope ls(p, s, KID_BRIEF_I);
        IBriefGeom bgl = bg;
        bgl.xcd_swizzle = (n % 8u == 0u) ? 1u : 0u;
        bgl.phase_mask = p->env.brief_i_mask >= 0 ? (uint32_t)p->env.brief_i_mask : 3u;
        bgl.angle_bins = p->opt.angle_bins;
        hipLaunchKernelGGL(k_brief_i, dim3(bg.group_base[D] * n), dim3(kIBriefThreads), p->ibrief_lds, s, p->d_blur, pyr, bgl,
                           p->d_iseg_counts, p->d_iseg_before, p->d_thr_key, p->d_iseg, p->d_iseg_scores, p->d_corners, cap,
                           p->d_desc, BriefTables{p->d_pattern, p->d_cos, p->d_sin, p->d_rot});
    }
    HIP_TRY(p, hipGetLastError());
    p->planes_valid = true;
    return ORB_OK;
}

// Dispatch grid of octave `lvl` in the reference (orb.rs:501-519: halved and 8-rounded per octave).
void reference_grid(const Pyramid& pyr, uint32_t lvl, uint32_t* gw, uint32_t* gh) {
    uint32_t width = pyr.w[0], height = pyr.h[0];
    for (uint32_t m = 0; m < lvl; m++) {
        width /= 2u;
        height /= 2u;
    }
    *gw = ((width + 7u) / 8u) * 8u;
    *gh = ((height + 7u) / 8u) * 8u;
}

bool fused_x_eligible(const OrbProgram* p) {
    if (p->intended || p->input_y8 || (p->opt.flags & ORB_FLAG_STAGED)) return false;  // the tile kernel reads RGBA
    if (p->arc == 12u && !(p->opt.flags & ORB_FLAG_NMS)) return false;  // that is the plain fused pipeline
    const Pyramid& pyr = p->pyr;
    if ((uint64_t)pyr.w[0] * pyr.h[0] > (1ull << 26) || pyr.h[0] > 16384u) return false;
    if ((pyr.w[0] & 3u) != 0u || pyr.w[0] > (uint32_t)kFrontMaxWidth || pyr.w[0] < 8u) return false;
    if (pyr.depth > 1 && !(pyr.w[0] == 2u * pyr.w[1] && pyr.h[0] == 2u * pyr.h[1])) return false;
    return true;
}

// The reference's algorithm with the opt-in arc length / NMS, fused: k_front_i per level
// in its literal setting (grey plane, detector, NMS, mip), k_front<false> per level with only its blur phase (row
// constants + stored tail from the grey plane), k_slot_prefix, k_brief_rows over the tile slots.
int run_fused_x(OrbProgram* p, const uint8_t* frames, uint32_t n, hipStream_t s) {
    const Pyramid& pyr = p->pyr;
    const uint32_t D = pyr.depth;
    const IBriefGeom& bg = p->itiles;
    uint32_t band_base = 0;
    for (uint32_t lvl = 0; lvl < D; lvl++) {
        if (lvl > 0 && !(pyr.w[lvl - 1] == 2u * pyr.w[lvl] && pyr.h[lvl - 1] == 2u * pyr.h[lvl])) {
            LaunchScope ls(p, s, KID_MIP);
            dim3 grid((pyr.w[lvl] + 63u) / 64u, (pyr.h[lvl] + 4u * kMipRows - 1u) / (4u * kMipRows), n);
            hipLaunchKernelGGL(k_mip, grid, dim3(64, 4), 0, s, p->d_gray, pyr, lvl, (float)pyr.w[lvl - 1] / (float)pyr.w[lvl], (float)pyr.h[lvl - 1] / (float)pyr.h[lvl]);
        }
        uint32_t gw, gh;
        reference_grid(pyr, lvl, &gw, &gh);
        IGeom g{};
        g.lvl = lvl;
        g.tw = bg.tw[lvl];
        g.n_ct = bg.n_ct[lvl];
        g.n_bands = bg.n_bands[lvl];
        g.ls = kIPad + g.tw + 16u;
        g.write_mip = (lvl + 1 < D && pyr.w[lvl] == 2u * pyr.w[lvl + 1] && pyr.h[lvl] == 2u * pyr.h[lvl + 1]) ? 1u : 0u;
        g.xcd_swizzle = (n % 8u == 0u) ? 1u : 0u;
        g.slot_base = bg.slot_base[lvl];
        g.n_slots = bg.n_slots;
        g.seg_cap = bg.seg_cap;
        g.arc = p->arc;
        g.nms = (p->opt.flags & ORB_FLAG_NMS) ? 1u : 0u;
        g.phase_mask = 63u;
        g.literal = 1u;
        g.store_grey = 1u;  // the literal blur below reads the grey plane
        g.blur = 0u;
        g.dw = std::max(pyr.w[lvl], gw);
        g.dh = std::max(pyr.h[lvl], gh);
        g.gx1 = std::min<uint32_t>(gw, pyr.w[0] > 16u ? pyr.w[0] - 16u : 0u);  // fast.wgsl:77 with level-0 dimensions (Q8)
        g.gy1 = std::min<uint32_t>(gh, pyr.h[0] > 16u ? pyr.h[0] - 16u : 0u);
        if (g.n_bands * g.n_ct != bg.slot_base[lvl + 1] - bg.slot_base[lvl])
            return fail(p, ORB_EINVAL, "internal: tile count mismatch at level %u", lvl);
        if (gw && gh) {
            LaunchScope ls(p, s, lvl == 0 ? KID_FRONT_I : KID_FRONT_I_LN);  // level 0 (RGBA in) and the levels above are different launches: one id each
            const dim3 grid(g.n_bands * g.n_ct * n);
            if (lvl == 0)
                hipLaunchKernelGGL(k_front_i<true>, grid, dim3(kIThreads), ifront_lds_bytes(g), s, frames, p->frame_bytes,
                                   p->d_gray, p->d_blur, pyr, g, p->threshold, p->d_iseg_counts, p->d_iseg, p->d_iseg_scores);
            else
                hipLaunchKernelGGL(k_front_i<false>, grid, dim3(kIThreads), ifront_lds_bytes(g), s, frames, p->frame_bytes,
                                   p->d_gray, p->d_blur, pyr, g, p->threshold, p->d_iseg_counts, p->d_iseg, p->d_iseg_scores);
        }
        // else: no FAST dispatch at this octave (orb.rs:511-515 with width 0).  Its tile slots were zeroed at create and
        // no kernel of this program ever writes them, so there is nothing to clear (clearing the whole counter array
        // here would erase the counts the earlier levels have just produced).
        {   // literal blur of this level from its grey plane: k_front's staging + phase C, nothing else
            FrontGeom fg = front_geometry(pyr, lvl, gw ? gw : 8u, gh, n);
            fg.phase_mask = 8u;
            fg.write_mip = 0u;
            fg.slot_base = band_base;
            fg.n_slots = p->xband_slots;
            fg.seg_cap = 1u;
            fg.n_classes = 1u;
            fg.stamps = nullptr;
            band_base += fg.n_bands;
            const uint32_t lds = front_lds_bytes(fg);
            if (lds > p->max_lds) return fail(p, ORB_EINVAL, "level %u needs %u bytes of LDS", lvl, lds);
            LaunchScope ls(p, s, KID_FUSED_LN);
            const FrontLaunch L{frames, p->frame_bytes, p->d_gray, p->d_blur, p->d_blur_rowc, pyr, fg, p->threshold, p->d_xband_counts, p->d_iseg, s,
                                fg.n_bands * n, lds, (uint32_t)kFrontRows, (uint32_t)kFrontThreadsLN, false, false, false, true};
            const hipError_t e = kFrontLaunch[0](L);  // k_front<false> (16 rows, 512 threads), from the unit that instantiates it
            if (e != hipSuccess) return fail(p, ORB_EHIP, "k_front (blur of level %u) failed to launch: %s", lvl, hipGetErrorString(e));
        }
    }
    launch_brief(p, s, n, p->xrows, p->d_blur, p->d_blur_rowc, p->d_iseg_counts, p->d_iseg_before, p->d_iseg, p->d_counts,
                 p->d_corners, p->d_desc);
    HIP_TRY(p, hipGetLastError());
    p->planes_valid = true;
    return ORB_OK;
}

int launch_compact(OrbProgram* p, uint32_t n, uint32_t* counts, uint64_t* offsets, CornerData* corners, CornerDescriptor* desc,
                   size_t capacity, void* stream, hipStream_t* used);  // below, with the bulk read-back

int run_pipeline(OrbProgram* p, const uint8_t* frames, uint32_t n, hipStream_t s) {
    if (p->fused_i) return run_fused_i(p, frames, n, s);
    if (p->fused_x) return run_fused_x(p, frames, n, s);
    return p->fused ? run_fused(p, frames, n, s) : run_staged(p, frames, n, s);
}

int ensure_input(OrbProgram* p) {
    if (!p->d_input) HIP_TRY(p, hipMalloc(&p->d_input, p->frame_bytes * p->max_batch));
    return ORB_OK;
}

}  // namespace

extern "C" {

uint32_t orb_abi_version(void) { return TINYORB_ABI_VERSION; }

const char* orb_last_error(const OrbProgram* p) { return p ? p->err.c_str() : g_create_error.c_str(); }

const char* orb_pipeline(const OrbProgram* p) { return p ? ((p->fused || p->fused_i || p->fused_x) ? "fused" : "staged") : ""; }

const char* orb_pipeline_note(const OrbProgram* p) { return p ? p->pipeline_note.c_str() : ""; }

const char* orb_kernel_name(int id) { return (id >= 0 && id < ORB_KERNEL_COUNT) ? kKernelNames[id] : ""; }

int orb_program_create(const OrbConfig* config, const OrbOptions* options, OrbProgram** out) {
    if (!out) return fail(nullptr, ORB_EINVAL, "out is NULL");
    *out = nullptr;
    if (!config) return fail(nullptr, ORB_EINVAL, "config is NULL");
    const uint32_t W = config->image_size.width, H = config->image_size.height;
    if (W == 0 || H == 0 || config->image_size.depth_or_array_layers != 1)
        return fail(nullptr, ORB_EINVAL, "image_size must be WxHx1 with W,H > 0");
    if ((uint64_t)W * H > (1ull << 28)) return fail(nullptr, ORB_EINVAL, "image too large");
    if (config->hierarchy_depth < 1 || config->hierarchy_depth > ORB_MAX_HIERARCHY_DEPTH)
        return fail(nullptr, ORB_EINVAL, "hierarchy_depth must be 1..=10 (orb.rs:66-67)");
    if (config->max_features == 0 || config->max_features > (1u << 24))
        return fail(nullptr, ORB_EINVAL, "max_features must be 1..=2^24");
    if (!(config->initial_threshold >= 0.f)) return fail(nullptr, ORB_EINVAL, "initial_threshold must be >= 0");
    if (options && options->fast_arc != 0 && (options->fast_arc < 9 || options->fast_arc > 16))
        return fail(nullptr, ORB_EINVAL, "fast_arc must be 0 (= 12) or 9..16");
    if (options && (options->flags & ORB_FLAG_INPUT_Y8) &&
        ((options->flags & (ORB_FLAG_INTENDED | ORB_FLAG_NMS)) || (options->fast_arc != 0 && options->fast_arc != 12)))
        return fail(nullptr, ORB_EINVAL, "ORB_FLAG_INPUT_Y8 is defined for the reference's detector only (no "
                                         "ORB_FLAG_INTENDED, ORB_FLAG_NMS or fast_arc other than 12)");
    if (options && (options->oob_policy > ORB_OOB_UMIN || options->sampler_weight_bits > 23u))
        return fail(nullptr, ORB_EINVAL, "oob_policy must be ORB_OOB_ZERO / _CLAMP / _UMIN and sampler_weight_bits 0..23");
    if (options && (options->oob_policy != ORB_OOB_ZERO || options->sampler_weight_bits != 0u) &&
        ((options->flags & (ORB_FLAG_INTENDED | ORB_FLAG_NMS)) || (options->fast_arc != 0 && options->fast_arc != 12)))
        return fail(nullptr, ORB_EINVAL, "oob_policy / sampler_weight_bits follow the reference's adapter: they are defined for the "
                                         "reference's detector only (no ORB_FLAG_INTENDED, ORB_FLAG_NMS or fast_arc other than 12)");
    if (options && options->fp_contract > (ORB_FP_CONTRACT_ALL | ORB_FP_LAST_TERM_FIRST))
        return fail(nullptr, ORB_EINVAL, "fp_contract is a mask of ORB_FP_CONTRACT_LUMINANCE / _BLUR / _ROTATION and ORB_FP_LAST_TERM_FIRST");
    if (options && options->fp_contract &&
        ((options->flags & (ORB_FLAG_INTENDED | ORB_FLAG_NMS | ORB_FLAG_INPUT_Y8)) || (options->fast_arc != 0 && options->fast_arc != 12)))
        return fail(nullptr, ORB_EINVAL, "fp_contract follows the reference's shader compiler: it is defined for the reference's detector on RGBA "
                                         "input only (no ORB_FLAG_INTENDED, ORB_FLAG_NMS, ORB_FLAG_INPUT_Y8 or fast_arc other than 12)");
    if (options && options->angle_bins && (!(options->flags & ORB_FLAG_INTENDED) || options->angle_bins < 8u || options->angle_bins > 6284u))
        return fail(nullptr, ORB_EINVAL, "angle_bins is an option of ORB_FLAG_INTENDED (IM-6b): 0, or 8..6284 bins of the full circle");
    if (options && (options->flags & ORB_FLAG_INTENDED) && (W > 16384u || H > 16384u))
        return fail(nullptr, ORB_EINVAL, "ORB_FLAG_INTENDED needs W, H <= 16384 (14-bit coordinates in the top-K key)");

    OrbProgram* p = new (std::nothrow) OrbProgram();
    if (!p) return fail(nullptr, ORB_EINVAL, "out of host memory");
    p->cfg = *config;
    if (options) p->opt = *options;
    p->device = p->opt.device;
    p->max_batch = p->opt.max_batch ? p->opt.max_batch : 1u;
    p->threshold = config->initial_threshold;  // orb.rs:178
    p->intended = (p->opt.flags & ORB_FLAG_INTENDED) != 0u;
    p->input_y8 = (p->opt.flags & ORB_FLAG_INPUT_Y8) != 0u;
    p->arc = p->opt.fast_arc ? p->opt.fast_arc : (p->intended ? 9u : 12u);
    p->oob = p->opt.oob_policy;
    p->wq = p->opt.sampler_weight_bits ? (float)(1u << p->opt.sampler_weight_bits) : 0.0f;
    p->frame_bytes = (size_t)W * H * (p->input_y8 ? 1u : 4u);
    layout_pyramid(W, H, config->hierarchy_depth, &p->pyr);
    {   // experiment / cross-check switches: the environment is read here and nowhere on a per-call path
        auto on = [](const char* name) { const char* e = getenv(name); return e && *e && atoi(e) != 0; };
        auto num = [](const char* name, int dflt) { const char* e = getenv(name); return e && *e ? atoi(e) : dflt; };
        p->env.single_memcpy = on("TINYORB_SINGLE_MEMCPY");
        p->env.single_split = on("TINYORB_SINGLE_SPLIT");
        p->env.single_serial = on("TINYORB_SINGLE_SERIAL");
        p->env.single_sync = on("TINYORB_SINGLE_SYNC");
        {   // how orb_extract_corners waits: "poll" (default) spins on the completion word; "block" spins for kSingleSpinUs, then sleeps
            // in hipEventSynchronize on a blocking-sync event (an interrupt instead of a spin -- the reference blocks in device.poll(Wait),
            // orb.rs:547); ORB_FLAG_SINGLE_BLOCKING_WAIT asks for the same from code
            const char* w = getenv("TINYORB_SINGLE_WAIT");
            p->env.single_block = (w && !strcmp(w, "block")) || (p->opt.flags & ORB_FLAG_SINGLE_BLOCKING_WAIT) != 0u;
        }
        p->env.no_swizzle = on("TINYORB_NO_SWIZZLE");
        p->env.quiet = getenv("TINYORB_QUIET") != nullptr;
        p->env.phase_mask = num("TINYORB_PHASE_MASK", -1);
        p->env.brief_i_mask = num("TINYORB_BRIEF_I_MASK", -1);
        p->env.lds_pad = num("TINYORB_LDS_PAD", 0);
    }

    auto bail = [&](int code) {
        g_create_error = p->err;
        orb_program_destroy(p);
        return code;
    };
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0) {
        fail(p, ORB_EHIP, "no HIP device available (%s)", e == hipSuccess ? "count is 0" : hipGetErrorString(e));
        return bail(ORB_EHIP);
    }
    if (p->device < 0 || p->device >= n_dev) {
        fail(p, ORB_EINVAL, "device %d out of range (0..%d)", p->device, n_dev - 1);
        return bail(ORB_EINVAL);
    }
#define CREATE_TRY(expr)                                                            \
    do {                                                                            \
        hipError_t e2_ = (expr);                                                    \
        if (e2_ != hipSuccess) {                                                    \
            fail(p, ORB_EHIP, "%s failed: %s", #expr, hipGetErrorString(e2_));      \
            return bail(ORB_EHIP);                                                  \
        }                                                                           \
    } while (0)
    CREATE_TRY(hipSetDevice(p->device));
    CREATE_TRY(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
    {
        int lds_max = 0;
        CREATE_TRY(hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, p->device));
        p->max_lds = (uint32_t)lds_max;
        int cus = 0;
        CREATE_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, p->device));
        p->n_cus = cus > 0 ? (uint32_t)cus : 256u;
        p->fused = fused_eligible(p);
        if (p->fused) {
            // Band height per level, from kFrontBandHeights (64, 32, 16, 8 rows: x of a queue entry in 9, 10, 11, 12 bits).
            // Two workgroups per CU matter more than the smaller halo share of a taller band (0.62 against 0.45 ms at 720p
            // with one), and a band should hold about 20 k pixels (the 1024 threads then take 2.5 pre-test items each): a
            // level takes the tallest band of which two fit the CU's LDS -- level 0: 64 rows up to about 330 wide, 32 up to
            // 700, 16 up to 1390, 8 up to 1980 --, else the tallest that fits at all (16 rows up to 2048, 8 rows up to 4096).
            uint32_t need = 0;
            {
                uint32_t width = W, height = H;
                const char* force = getenv("TINYORB_BAND_ROWS");  // experiments: one band height for every level that can have it
                for (uint32_t lvl = 0; lvl < p->pyr.depth; lvl++) {
                    const uint32_t gw = ((width + 7u) / 8u) * 8u, gh = ((height + 7u) / 8u) * 8u, w = p->pyr.w[lvl];
                    width /= 2u;
                    height /= 2u;
                    // the tallest band of which two fit a CU; else the tallest that fits at all
                    uint32_t rows = 0, rows_lds = 0, fits = 0, fits_lds = 0, want = 0, want_lds = 0;
                    const uint32_t forced = force ? (uint32_t)atoi(force) : 0u;
                    auto pick = [&](uint32_t threads) {
                        rows = rows_lds = fits = fits_lds = want = want_lds = 0u;
                        for (int cand : kFrontBandHeights) {
                            if (std::max(w, gw) > (1u << front_x_bits(cand)) || std::max(w, gw) > (uint32_t)kFrontMaxWidthWide) continue;
                            // levels >= 1: beyond about 32 pixels per thread a band only gets longer (at 512 threads: 640 wide, 32 rows
                            // 4 % slower than 16; 480 wide, 32 rows 16 % faster than 16)
                            if (lvl > 0 && (uint32_t)cand * gw > 32u * threads && cand > kFrontRowsWide) continue;
                            const uint32_t lds = front_lds_bytes(front_geometry(p->pyr, lvl, gw, gh, 1, (uint32_t)cand, 0u, threads));
                            if (lds > p->max_lds) continue;
                            if (forced == (uint32_t)cand) want = (uint32_t)cand, want_lds = lds;
                            if (!fits) fits = (uint32_t)cand, fits_lds = lds;
                            if (!rows && 2u * lds <= p->max_lds) rows = (uint32_t)cand, rows_lds = lds;
                        }
                    };
                    // Levels >= 1 take level 0's shape -- 1024 threads, sixteen-pixel pre-test items -- where a band of which two fit
                    // a CU holds about as many pixels as level 0's at 1280 columns (18 k and more: 640 columns x 32 rows, 320 x 64,
                    // 1280 x 16; measured at 720p, 640x480, 2560x1440: 5 ... 21 % off k_front<false>), and 512 threads otherwise
                    // (480 columns: the tallest band that fits twice is 32 x 480 = 15 k pixels, 13 % slower on 1024 threads; 160
                    // columns: 27 % slower).  Not for programs of one frame (flat bands), forced heights, tiles or an out-of-level
                    // policy (those instances exist for 512 threads only).
                    p->ln_threads[lvl] = (uint32_t)kFrontThreadsLN;
                    if (lvl > 0 && p->max_batch > 1u && !forced && !getenv("TINYORB_TILE_W") && p->oob == kOobZero && !getenv("TINYORB_LN_512")) {
                        pick((uint32_t)kFrontThreadsLNBig);
                        if (rows != 0u && rows * gw >= 18432u) p->ln_threads[lvl] = (uint32_t)kFrontThreadsLNBig;
                    }
                    if (p->ln_threads[lvl] != (uint32_t)kFrontThreadsLNBig) pick(lvl == 0 ? (uint32_t)kFrontThreadsL0 : (uint32_t)kFrontThreadsLN);
                    const bool two_per_cu = rows != 0u;
                    if (!rows) rows = fits, rows_lds = fits_lds;
                    // A program for one frame at a time (the reference's call shape) is after latency, not throughput: its 45 + 23
                    // bands cannot fill 256 CUs anyway, so it takes the flattest bands -- twice the workgroups, each with half
                    // the work on its critical path (k_front<true> 16.3 -> 13.6 us at 720p).
                    if (p->max_batch == 1u && two_per_cu && !want && !getenv("TINYORB_NO_LATENCY_BANDS")) {
                        const uint32_t flat = (uint32_t)kFrontRowsWide;
                        const uint32_t lds = front_lds_bytes(front_geometry(p->pyr, lvl, gw, gh, 1, flat));
                        if (std::max(w, gw) <= (1u << front_x_bits((int)flat)) && lds <= p->max_lds) rows = flat, rows_lds = lds;
                    }
                    if (want) rows = want, rows_lds = want_lds;
                    // Too wide for two full-width bands per CU (one workgroup per CU costs a third of the rate: 0.62 against
                    // 0.45 ms at 720p), or for any: column tiles of about kFrontTileW columns -- 16 rows x 1280 columns is the
                    // shape the kernel is tuned on --, equal ones, wide enough for tile 0 to hold the columns the band's blur
                    // reads (front_tile0_holds_blur); 8 rows where 16-row tiles of that width do not fit a CU twice.
                    const char* force_w = getenv("TINYORB_TILE_W");
                    if ((!two_per_cu && !want) || force_w) {
                        const uint32_t dom = (std::max(w, gw) + 7u) & ~7u;
                        uint32_t want_w = force_w ? std::max(64u, (uint32_t)atoi(force_w)) : (uint32_t)kFrontTileW, tw = dom, t_rows = 0, t_lds = 0;
                        for (; tw >= dom || t_rows == 0u; want_w += 64u) {
                            const uint32_t n_ct = std::max(1u, (dom + want_w - 1u) / want_w);
                            tw = std::min(dom, (((dom + n_ct - 1u) / n_ct) + 7u) & ~7u);
                            t_rows = 0;
                            for (uint32_t cand : {16u, 8u}) {
                                const FrontGeom tg = front_geometry(p->pyr, lvl, gw, gh, 1, cand, tw);
                                const uint32_t lds = front_lds_bytes(tg);
                                const uint32_t items = lvl == 0 ? (tg.tw + 12u) / 4u : (tg.tw + 20u) / 8u;  // staged items of a row <= threads
                                if (!front_tile0_holds_blur(tg, w) || cand > (1u << (15u - tg.xb)) || lds > p->max_lds ||
                                    items > (lvl == 0 ? (uint32_t)kFrontThreadsL0 : (uint32_t)kFrontThreadsLN))
                                    continue;
                                if (!t_rows || (2u * lds <= p->max_lds && 2u * t_lds > p->max_lds)) t_rows = cand, t_lds = lds;
                                if (2u * t_lds <= p->max_lds) break;
                            }
                            if (t_rows || want_w >= dom) break;
                        }
                        if (t_rows) {
                            rows = t_rows, rows_lds = t_lds;
                            p->tile_w_lvl[lvl] = tw;
                        }
                    }
                    p->band_rows_lvl[lvl] = rows;
                    need = rows == 0 ? p->max_lds + 1u : std::max(need, rows_lds);
                }
            }
            if (need > p->max_lds) {
                p->fused = false;
            } else {
                BandGeom& bg = p->bands;
                uint32_t slots = 0;
                uint64_t band_px = 0;  // pixels of the largest band (tile): no band can hold more corners
                for (uint32_t lvl = 0; lvl < p->pyr.depth; lvl++) {
                    bg.slot_base[lvl] = slots;
                    const uint64_t lvl_cols = ((uint64_t)(W >> lvl) + 7u) / 8u * 8u;
                    uint32_t n_ct = 1;
                    if (p->tile_w_lvl[lvl]) n_ct = (uint32_t)((lvl_cols + p->tile_w_lvl[lvl] - 1u) / p->tile_w_lvl[lvl]);
                    slots += front_bands(p->pyr, lvl, p->band_rows_lvl[lvl]) * n_ct;
                    band_px = std::max<uint64_t>(band_px, (uint64_t)p->band_rows_lvl[lvl] * (p->tile_w_lvl[lvl] ? p->tile_w_lvl[lvl] : lvl_cols));
                }
                bg.slot_base[p->pyr.depth] = slots;
                bg.n_slots = slots;
                bg.seg_cap = (uint32_t)(band_px < config->max_features ? band_px : config->max_features);
                RowsGeom& rg = p->rows;
                rg.n_slots = bg.n_slots;
                rg.seg_cap = bg.seg_cap;
                for (uint32_t lvl = 0; lvl <= p->pyr.depth; lvl++) rg.slot_base[lvl] = bg.slot_base[lvl];
                for (uint32_t lvl = 0; lvl < p->pyr.depth; lvl++) {
                    // columns [0, qa) of the level's blur are kept as one constant per row (k_front, phase C)
                    const uint32_t q = front_geometry(p->pyr, lvl, 8, 8, 1).blur_q, qa = q & ~7u;
                    p->blur_qa[lvl] = qa;
                    rg.qa[lvl] = qa;
                    // a keypoint is "flat" when every sample column lies below Q, not only below qa: the stored columns [qa, Q) hold
                    // the row constant too
                    rg.flat_end[lvl] = q > (uint32_t)kBriefHalo ? q - kBriefHalo : 0u;
                }
                // The attribute belongs to the function on this device, not to the program: always raise it to the
                // device's limit, so that a later, smaller program never lowers it under a live, larger one.
                p->use_brief_t = brieft_geometry(p, rg, 2u, &p->brieft);
                p->seg_classes = p->use_brief_t ? 2u : 1u;
                for (auto set_max_lds : kFrontSetMaxLds) CREATE_TRY(set_max_lds((int)p->max_lds));  // every instance of k_front / k_front_pair
            }
        }
    }
    p->fused_i = fused_i_eligible(p);
    if (p->fused_i) {
        IBriefGeom& bg = p->itiles;
        uint32_t slots = 0;
        for (uint32_t lvl = 0; lvl < p->pyr.depth; lvl++) {
            bg.tw[lvl] = itile_width(p->pyr.w[lvl]);
            bg.n_ct[lvl] = (p->pyr.w[lvl] + bg.tw[lvl] - 1u) / bg.tw[lvl];
            bg.slot_base[lvl] = slots;
            slots += ((p->pyr.h[lvl] + kFrontRows - 1) / kFrontRows) * bg.n_ct[lvl];
        }
        bg.slot_base[p->pyr.depth] = slots;
        bg.n_slots = slots;
        uint32_t groups = 0, max_rows = 0;
        for (uint32_t lvl = 0; lvl < p->pyr.depth; lvl++) {
            bg.n_bands[lvl] = (p->pyr.h[lvl] + kFrontRows - 1) / kFrontRows;
            bg.group_base[lvl] = groups;
            groups += ((bg.n_bands[lvl] + kIBriefStack - 1) / kIBriefStack) * bg.n_ct[lvl];
            const uint32_t rows = std::min<uint32_t>(bg.n_bands[lvl], kIBriefStack) * kFrontRows + 2u * kBriefHalo;
            if (rows > max_rows) max_rows = rows;
        }
        bg.group_base[p->pyr.depth] = groups;
        // The Gaussian (IM-3) inside k_front_i, straight from the tile; the level-0 grey plane is then only stored when a level
        // that is not an exact half needs it (k_mip reads planes).
        p->igauss_fused = !getenv("TINYORB_I_GAUSS_KERNEL");
        for (uint32_t lvl = 0; lvl < p->pyr.depth; lvl++) p->igauss_fused = p->igauss_fused && ifront_gauss_fits(bg.tw[lvl]);
        p->igrey_stored = !p->igauss_fused || (p->pyr.depth > 1 && !(p->pyr.w[0] == 2u * p->pyr.w[1] && p->pyr.h[0] == 2u * p->pyr.h[1]));
        // a tile's segment holds every keypoint the tile can have (one per pixel), whatever max_features is: the
        // top-K cut (IM-8) must see all candidates, not the ones that happened to be appended first
        bg.seg_cap = (uint32_t)kFrontRows * bg.tw[0];
        bg.pitch = (uint32_t)kITileW + 2u * kIBriefApronX;
        p->ibrief_lds = max_rows * bg.pitch * (uint32_t)sizeof(uint16_t);
        // as for k_front: raise to the device limit (less the kernel's static LDS), never to this program's own need
        hipFuncAttributes fa{};
        hipError_t ea = hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_brief_i));
        if (ea == hipSuccess && p->ibrief_lds + (uint32_t)fa.sharedSizeBytes > p->max_lds) {
            fail(p, ORB_EINVAL, "k_brief_i needs %u bytes of LDS", p->ibrief_lds + (uint32_t)fa.sharedSizeBytes);
            return bail(ORB_EINVAL);
        }
        if (ea == hipSuccess)
            ea = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_brief_i), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)(p->max_lds - (uint32_t)fa.sharedSizeBytes));
        if (ea != hipSuccess) {
            fail(p, ORB_EHIP, "hipFuncSetAttribute(k_brief_i): %s", hipGetErrorString(ea));
            return bail(ORB_EHIP);
        }
    }
    p->fused_x = fused_x_eligible(p);
    if (p->fused_x) {
        // tiles over the reference's dispatch grid of every octave; blur-only k_front launches over 16-row bands
        IBriefGeom& bg = p->itiles;
        RowsGeom& rg = p->xrows;
        uint32_t slots = 0, bands = 0, need = 0;
        for (uint32_t lvl = 0; lvl < p->pyr.depth; lvl++) {
            uint32_t gw, gh;
            reference_grid(p->pyr, lvl, &gw, &gh);
            const uint32_t dw = std::max(p->pyr.w[lvl], gw), dh = std::max(p->pyr.h[lvl], gh);
            bg.tw[lvl] = itile_width(dw);
            bg.n_ct[lvl] = (dw + bg.tw[lvl] - 1u) / bg.tw[lvl];
            bg.n_bands[lvl] = (dh + kFrontRows - 1) / kFrontRows;
            bg.slot_base[lvl] = slots;
            rg.slot_base[lvl] = slots;
            slots += bg.n_bands[lvl] * bg.n_ct[lvl];
            const FrontGeom fg = front_geometry(p->pyr, lvl, gw ? gw : 8u, gh, 1);
            bands += fg.n_bands;
            need = std::max(need, front_lds_bytes(fg));
            const uint32_t qa = fg.blur_q & ~7u;
            p->blur_qa[lvl] = qa;
            rg.qa[lvl] = qa;
            rg.flat_end[lvl] = fg.blur_q > (uint32_t)kBriefHalo ? fg.blur_q - kBriefHalo : 0u;
        }
        bg.slot_base[p->pyr.depth] = slots;
        rg.slot_base[p->pyr.depth] = slots;
        bg.n_slots = slots;
        bg.seg_cap = (uint32_t)kFrontRows * bg.tw[0];
        rg.n_slots = slots;
        rg.seg_cap = bg.seg_cap;
        p->xband_slots = bands;
        if (need > p->max_lds) {
            p->fused_x = false;
        } else {
            p->use_brief_t = brieft_geometry(p, rg, 1u, &p->brieft);  // the tile kernels keep one list per tile
            hipError_t ea = kFrontSetMaxLds[0]((int)p->max_lds);  // (the unit that holds k_front<false>)
            if (ea != hipSuccess) {
                fail(p, ORB_EHIP, "hipFuncSetAttribute(k_front): %s", hipGetErrorString(ea));
                return bail(ORB_EHIP);
            }
        }
    }
    if (!(p->fused || p->fused_i || p->fused_x) && !(p->opt.flags & ORB_FLAG_STAGED)) {
        // A silent fall to the per-stage kernels is a 7x performance cliff: say why, once, and keep it readable.
        const Pyramid& py = p->pyr;
        char why[256];
        const bool plain = !p->intended && !(p->opt.flags & ORB_FLAG_NMS) && p->arc == 12u;  // the reference's own algorithm
        if ((py.w[0] & 3u) != 0u && !plain)
            snprintf(why, sizeof why, "width %u is not a multiple of 4 (the tile kernels read RGBA quads)", py.w[0]);
        else if (py.w[0] < 8u)
            snprintf(why, sizeof why, "width %u is below 8", py.w[0]);
        else if (!p->intended && py.w[0] > (uint32_t)kFrontMaxWidthTiled)
            snprintf(why, sizeof why, "width %u exceeds %d", py.w[0], kFrontMaxWidthTiled);
        else if ((uint64_t)py.w[0] * py.h[0] > (1ull << 26) || py.h[0] > 16384u)
            snprintf(why, sizeof why, "%ux%u exceeds the fused kernels' 2^26-pixel / 16384-row index range", py.w[0], py.h[0]);
        else if (!p->intended && !plain && py.depth > 1 && !(py.w[0] == 2u * py.w[1] && py.h[0] == 2u * py.h[1]))
            snprintf(why, sizeof why, "level 0 (%ux%u) does not halve exactly and hierarchy_depth > 1", py.w[0], py.h[0]);
        else if (p->input_y8 && ((p->opt.flags & ORB_FLAG_NMS) || p->arc != 12u))
            snprintf(why, sizeof why, "the tile kernels of the arc/NMS extensions read RGBA");
        else
            snprintf(why, sizeof why, "the band does not fit in %u bytes of LDS", p->max_lds);
        p->pipeline_note = std::string("staged pipeline (one kernel per reference stage, about 1/7 of the fused rate): ") + why;
        static bool warned = false;
        if (!warned && !p->env.quiet) {
            warned = true;
            fprintf(stderr, "libtinyorb: %s\n", p->pipeline_note.c_str());
        }
    }
    const size_t B = p->max_batch, cap = config->max_features;
    CREATE_TRY(hipMalloc(&p->d_gray, B * p->pyr.stride * sizeof(uint16_t)));
    CREATE_TRY(hipMalloc(&p->d_blur, B * p->pyr.stride * sizeof(uint16_t)));
    for (int set = 0; set < ((p->opt.flags & ORB_FLAG_DOUBLE_OUTPUT) ? 2 : 1); set++) {
        CREATE_TRY(hipMalloc(&p->out_counts[set], B * sizeof(uint32_t)));
        CREATE_TRY(hipMalloc(&p->out_corners[set], B * cap * sizeof(CornerData)));
        CREATE_TRY(hipMalloc(&p->out_desc[set], B * cap * sizeof(CornerDescriptor)));
        CREATE_TRY(hipMemset(p->out_counts[set], 0, B * sizeof(uint32_t)));
        CREATE_TRY(hipMemset(p->out_corners[set], 0, B * cap * sizeof(CornerData)));
        CREATE_TRY(hipMemset(p->out_desc[set], 0, B * cap * sizeof(CornerDescriptor)));
    }
    p->d_counts = p->out_counts[0];
    p->d_corners = p->out_corners[0];
    p->d_desc = p->out_desc[0];
    if (p->fused) {
        CREATE_TRY(hipMalloc(&p->d_blur_rowc, B * p->pyr.row_stride * sizeof(uint16_t)));
        CREATE_TRY(hipMemset(p->d_blur_rowc, 0, B * p->pyr.row_stride * sizeof(uint16_t)));
        const size_t lists = (size_t)p->bands.n_slots * p->seg_classes;
        CREATE_TRY(hipMalloc(&p->d_seg, B * lists * (size_t)p->bands.seg_cap * sizeof(CornerData)));
        CREATE_TRY(hipMalloc(&p->d_seg_counts, B * lists * sizeof(uint32_t)));
        CREATE_TRY(hipMemset(p->d_seg_counts, 0, B * lists * sizeof(uint32_t)));
        CREATE_TRY(hipMalloc(&p->d_seg_before, B * lists * sizeof(uint32_t)));
    }
    if (p->fused_x) {
        CREATE_TRY(hipMalloc(&p->d_blur_rowc, B * p->pyr.row_stride * sizeof(uint16_t)));
        CREATE_TRY(hipMemset(p->d_blur_rowc, 0, B * p->pyr.row_stride * sizeof(uint16_t)));
        CREATE_TRY(hipMalloc(&p->d_xband_counts, B * (size_t)p->xband_slots * sizeof(uint32_t)));
    }
    if (p->fused_i || p->fused_x) {
        const size_t n_seg = B * (size_t)p->itiles.n_slots;
        CREATE_TRY(hipMalloc(&p->d_iseg, n_seg * p->itiles.seg_cap * sizeof(CornerData)));
        CREATE_TRY(hipMalloc(&p->d_iseg_scores, n_seg * p->itiles.seg_cap * sizeof(float)));
        CREATE_TRY(hipMalloc(&p->d_iseg_counts, n_seg * sizeof(uint32_t)));
        CREATE_TRY(hipMemset(p->d_iseg_counts, 0, n_seg * sizeof(uint32_t)));
        CREATE_TRY(hipMalloc(&p->d_iseg_before, n_seg * sizeof(uint32_t)));
        CREATE_TRY(hipMalloc(&p->d_thr_key, B * sizeof(unsigned long long)));
    } else if (p->opt.flags & (ORB_FLAG_NMS | ORB_FLAG_INTENDED)) {
        // score planes: one float per dispatch-grid pixel of every octave + a 1-px border; provisional list
        uint32_t off = 0, width = W, height = H;
        for (uint32_t m = 0; m < p->pyr.depth; m++) {
            const uint32_t gw = ((width + 7u) / 8u) * 8u, gh = ((height + 7u) / 8u) * 8u;
            p->score_layout.off[m] = off;
            p->score_layout.pitch[m] = gw + 2u;
            off += (gw + 2u) * (gh + 2u);
            width /= 2u;
            height /= 2u;
        }
        p->score_layout.stride = (off + 63u) & ~63u;
        // The provisional list must hold every detection (the NMS and the top-K cut have to see all of them): one
        // slot per texel of the pyramid is the worst case and what is allocated (20 bytes per texel and frame).
        p->cap_prov = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(p->pyr.stride, cap), 1u << 24);
        if (p->opt.flags & ORB_FLAG_NMS)
        CREATE_TRY(hipMalloc(&p->d_score_planes, B * (size_t)p->score_layout.stride * sizeof(float)));
        CREATE_TRY(hipMalloc(&p->d_prov_counts, B * sizeof(uint32_t)));
        CREATE_TRY(hipMalloc(&p->d_prov, B * (size_t)p->cap_prov * sizeof(CornerData)));
        CREATE_TRY(hipMalloc(&p->d_prov_scores, B * (size_t)p->cap_prov * sizeof(float)));
        if (p->intended && (p->opt.flags & ORB_FLAG_NMS)) {
            CREATE_TRY(hipMalloc(&p->d_prov2_counts, B * sizeof(uint32_t)));
            CREATE_TRY(hipMalloc(&p->d_prov2, B * (size_t)p->cap_prov * sizeof(CornerData)));
            CREATE_TRY(hipMalloc(&p->d_prov2_scores, B * (size_t)p->cap_prov * sizeof(float)));
        }
    }
    CREATE_TRY(hipMalloc(&p->d_pattern, 256 * sizeof(uint32_t)));
    CREATE_TRY(hipMalloc(&p->d_cos, ORB_ANGLE_STEPS_FULL * sizeof(float)));
    CREATE_TRY(hipMalloc(&p->d_sin, ORB_ANGLE_STEPS_FULL * sizeof(float)));
    CREATE_TRY(hipMemcpy(p->d_pattern, ORB_BRIEF_PATTERN, 1024, hipMemcpyHostToDevice));
    CREATE_TRY(hipMemcpy(p->d_cos, ORB_COS_BITS, ORB_ANGLE_STEPS_FULL * 4, hipMemcpyHostToDevice));
    CREATE_TRY(hipMemcpy(p->d_sin, ORB_SIN_BITS, ORB_ANGLE_STEPS_FULL * 4, hipMemcpyHostToDevice));
    {   // the pattern rotated by every angle code, for the consumer this program has: k_brief_i's window (intended, full circle) or
        // k_brief_nf's patch (the reference's codes 0..3141)
        // (IM-6b, OrbOptions::angle_bins: one entry per angle bin instead of one per milliradian code)
        const uint32_t n_codes = p->fused_i ? (p->opt.angle_bins ? p->opt.angle_bins : (uint32_t)ORB_ANGLE_STEPS_FULL) : (uint32_t)ORB_ANGLE_STEPS;
        const int pitch = p->fused_i ? (int)p->itiles.pitch : kNfPatchCols;
        CREATE_TRY(hipMalloc(&p->d_rot, (size_t)n_codes * 64u * sizeof(uint4)));
        hipLaunchKernelGGL(k_rot_table, dim3(n_codes), dim3(64), 0, p->stream, p->d_pattern, p->d_cos, p->d_sin, pitch, p->fused_i ? 1 : 0, p->d_rot,
                           p->opt.fp_contract, p->fused_i ? p->opt.angle_bins : 0u);
        CREATE_TRY(hipGetLastError());
        CREATE_TRY(hipStreamSynchronize(p->stream));
    }
    if (p->fused && getenv("TINYORB_STAMPS")) {  // diagnostic builds only (tools/stamps.py)
        CREATE_TRY(hipMalloc(&p->d_stamps, 32 * sizeof(unsigned long long)));
        CREATE_TRY(hipMemset(p->d_stamps, 0, 32 * sizeof(unsigned long long)));
    }
    CREATE_TRY(hipHostMalloc(&p->h_count, kSingleCountWords * sizeof(uint32_t), hipHostMallocDefault));
    memset(p->h_count, 0, kSingleCountWords * sizeof(uint32_t));
    CREATE_TRY(hipMalloc(&p->d_single_done, sizeof(uint32_t)));
    CREATE_TRY(hipMemset(p->d_single_done, 0, sizeof(uint32_t)));
    CREATE_TRY(hipHostMalloc(&p->h_corners, cap * sizeof(CornerData), hipHostMallocDefault));
    CREATE_TRY(hipHostMalloc(&p->h_desc, cap * sizeof(CornerDescriptor), hipHostMallocDefault));
    {   // device-visible addresses of the staging arrays, once (null: the runtime cannot map them -> the three copies of orb.rs:537-547)
        void *dc = nullptr, *dk = nullptr, *dd = nullptr;
        if (hipHostGetDevicePointer(&dc, p->h_count, 0) == hipSuccess && hipHostGetDevicePointer(&dk, p->h_corners, 0) == hipSuccess &&
            hipHostGetDevicePointer(&dd, p->h_desc, 0) == hipSuccess) {
            p->dv_count = static_cast<uint32_t*>(dc);
            p->dv_corners = static_cast<CornerData*>(dk);
            p->dv_desc = static_cast<CornerDescriptor*>(dd);
        } else {
            (void)hipGetLastError();
        }
    }
#undef CREATE_TRY
    *out = p;
    return ORB_OK;
}

void orb_program_destroy(OrbProgram* p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    for (auto& sp : p->pending) {
        (void)hipEventDestroy(sp.start);
        (void)hipEventDestroy(sp.stop);
    }
    for (auto e : p->event_pool) (void)hipEventDestroy(e);
    (void)hipFree(p->d_input);
    (void)hipFree(p->d_input_alt);
    for (hipEvent_t ev : p->in_uploaded)
        if (ev) (void)hipEventDestroy(ev);
    (void)hipFree(p->d_gray);
    (void)hipFree(p->d_blur);
    (void)hipFree(p->d_blur_rowc);
    for (int set = 0; set < 2; set++) {
        (void)hipFree(p->out_counts[set]);
        (void)hipFree(p->out_corners[set]);
        (void)hipFree(p->out_desc[set]);
    }
    (void)hipFree(p->d_seg);
    (void)hipFree(p->d_seg_counts);
    (void)hipFree(p->d_seg_before);
    (void)hipFree(p->d_score_planes);
    (void)hipFree(p->d_prov_counts);
    (void)hipFree(p->d_prov);
    (void)hipFree(p->d_prov_scores);
    (void)hipFree(p->d_matches);
    (void)hipFree(p->d_xband_counts);
    (void)hipFree(p->d_iseg);
    (void)hipFree(p->d_iseg_scores);
    (void)hipFree(p->d_iseg_counts);
    (void)hipFree(p->d_iseg_before);
    (void)hipFree(p->d_thr_key);
    (void)hipFree(p->d_prov2_counts);
    (void)hipFree(p->d_prov2);
    (void)hipFree(p->d_prov2_scores);
    (void)hipFree(p->d_pattern);
    (void)hipFree(p->d_cos);
    (void)hipFree(p->d_sin);
    if (p->d_rot) (void)hipFree(p->d_rot);
    (void)hipFree(p->d_stamps);
    (void)hipFree(p->d_desc8);
    if (p->match_done) (void)hipEventDestroy(p->match_done);
    if (p->single_done_ev) (void)hipEventDestroy(p->single_done_ev);
    if (p->h_count) (void)hipHostFree(p->h_count);
    if (p->d_single_done) (void)hipFree(p->d_single_done);
    if (p->h_corners) (void)hipHostFree(p->h_corners);
    if (p->h_desc) (void)hipHostFree(p->h_desc);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    if (p->copy_stream) (void)hipStreamDestroy(p->copy_stream);
    if (p->order_event) (void)hipEventDestroy(p->order_event);
    for (hipEvent_t ev : p->upload_events) (void)hipEventDestroy(ev);
    if (p->upload_done) (void)hipEventDestroy(p->upload_done);
    for (int set = 0; set < 2; set++) {
        (void)hipFree(p->d_pack_c[set]);
        (void)hipFree(p->d_pack_d[set]);
        if (p->h_pack_counts[set]) (void)hipHostFree(p->h_pack_counts[set]);
        if (p->h_pack_offsets[set]) (void)hipHostFree(p->h_pack_offsets[set]);
        if (p->pack_event[set]) (void)hipEventDestroy(p->pack_event[set]);
    }
    delete p;
}

// Slab an image written now goes to, and its place in the queue of images waiting for extract_corners (at most two: the
// one being extracted next and the one uploaded under it).  With nothing pending the slab of the last extract is reused.
// ahead = false (the blocking write, the reference's): the last write wins -- an image that is still waiting is overwritten,
// the queue never grows past one.  ahead = true (orb_write_input_image_pinned): the image is queued behind a waiting one.
// Nothing is published here: the queue, the pending count and the slab's "came through the copy stream" mark change only once the image
// is really on its way (publish_input_slab), so that a failed allocation, event creation or copy leaves the program as it was.
static int pick_input_slab(OrbProgram* p, bool ahead, uint32_t* slab_out, uint8_t** dst, bool* appended) {
    if (ahead && p->in_pending >= 2u)
        return fail(p, ORB_ESTATE, "two images are already waiting for extract_corners");
    if (int rc = ensure_input(p)) return rc;
    uint32_t slab;
    if (p->in_pending == 0u || ahead) {
        slab = p->in_pending == 0u ? p->in_last : (p->in_queue[0] ^ 1u);
        *appended = true;
    } else {
        slab = p->in_queue[p->in_pending - 1u];  // overwrite the newest waiting image
        if (p->in_async[slab]) HIP_TRY(p, hipEventSynchronize(p->in_uploaded[slab]));  // its upload must not land on top of this one
        *appended = false;
    }
    if (slab == 1u && !p->d_input_alt) HIP_TRY(p, hipMalloc(&p->d_input_alt, p->frame_bytes));
    *slab_out = slab;
    *dst = slab ? p->d_input_alt : p->d_input;
    return ORB_OK;
}
static void publish_input_slab(OrbProgram* p, uint32_t slab, bool appended, bool async) {
    if (appended) p->in_queue[p->in_pending++] = slab;
    p->in_async[slab] = async;
}
// A copy into a slab that was already queued (the blocking write overwrites the newest waiting image) failed half-way: that image
// is gone, the queue must not name it any more.
static void drop_newest_input(OrbProgram* p, bool appended) {
    if (!appended && p->in_pending) p->in_pending--;
}

int orb_write_input_image(OrbProgram* p, const uint8_t* bytes, size_t len) {
    if (!p) return ORB_EINVAL;
    if (!bytes || len != p->frame_bytes)
        return fail(p, ORB_EINVAL, "write_input_image: expected %zu bytes (4*W*H), got %zu", p->frame_bytes, len);
    HIP_TRY(p, hipSetDevice(p->device));
    uint32_t slab = 0;
    uint8_t* dst = nullptr;
    bool appended = false;
    if (int rc = pick_input_slab(p, false, &slab, &dst, &appended)) return rc;
    hipError_t e = hipMemcpyAsync(dst, bytes, len, hipMemcpyHostToDevice, p->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(p->stream);  // the caller's slice may be reused right away
    if (e != hipSuccess) {
        drop_newest_input(p, appended);
        return fail(p, ORB_EHIP, "write_input_image: the upload failed: %s", hipGetErrorString(e));
    }
    publish_input_slab(p, slab, appended, false);
    return ORB_OK;
}

int orb_write_input_image_pinned(OrbProgram* p, const uint8_t* bytes_pinned, size_t len) {
    if (!p) return ORB_EINVAL;
    if (!bytes_pinned || len != p->frame_bytes)
        return fail(p, ORB_EINVAL, "write_input_image_pinned: expected %zu bytes, got %zu", p->frame_bytes, len);
    HIP_TRY(p, hipSetDevice(p->device));
    uint32_t slab = 0;
    uint8_t* dst = nullptr;
    bool appended = false;
    if (int rc = pick_input_slab(p, true, &slab, &dst, &appended)) return rc;
    if (!p->copy_stream) HIP_TRY(p, hipStreamCreateWithFlags(&p->copy_stream, hipStreamNonBlocking));
    if (!p->in_uploaded[slab]) HIP_TRY(p, hipEventCreateWithFlags(&p->in_uploaded[slab], hipEventDisableTiming));
    if (!p->upload_done) HIP_TRY(p, hipEventCreateWithFlags(&p->upload_done, hipEventDisableTiming));
    // The slab is free: the extract that last read it has returned (extract_corners blocks).  The copy runs on the copy
    // stream, beside the kernels of the image extracted meanwhile; extract_corners orders its kernels behind it.
    HIP_TRY(p, hipMemcpyAsync(dst, bytes_pinned, len, hipMemcpyHostToDevice, p->copy_stream));
    HIP_TRY(p, hipEventRecord(p->in_uploaded[slab], p->copy_stream));
    HIP_TRY(p, hipEventRecord(p->upload_done, p->copy_stream));  // orb_upload_sync(): the host array may be reused
    publish_input_slab(p, slab, appended, true);  // only now: every call above succeeded, the image is on its way
    return ORB_OK;
}

int orb_set_threshold(OrbProgram* p, float threshold) {
    if (!p) return ORB_EINVAL;
    if (!(threshold >= 0.f)) return fail(p, ORB_EINVAL, "threshold must be >= 0");
    p->threshold = threshold;
    return ORB_OK;
}

int orb_extract_corners(OrbProgram* p, uint32_t* corner_count) {
    if (!p) return ORB_EINVAL;
    if (!corner_count) return fail(p, ORB_EINVAL, "corner_count is NULL");
    HIP_TRY(p, hipSetDevice(p->device));
    if (int rc = ensure_input(p)) return rc;
    hipStream_t s = p->stream;
    const size_t cap = p->cfg.max_features;
    // the oldest image written and not yet extracted; with none pending, the last one again (as the reference would)
    uint32_t slab = p->in_last;
    if (p->in_pending) {
        slab = p->in_queue[0];
        p->in_queue[0] = p->in_queue[1];
        p->in_pending--;
        p->in_last = slab;
    }
    const uint8_t* const d_in = slab ? p->d_input_alt : p->d_input;
    if (p->in_async[slab]) {  // uploaded on the copy stream (orb_write_input_image_pinned)
        HIP_TRY(p, hipStreamWaitEvent(s, p->in_uploaded[slab], 0));
        p->in_async[slab] = false;
    }
    // orb.rs:537-547: counter, corners and descriptors go to host staging, then block.  The reference copies the three
    // whole buffers; here a kernel writes the counter and the STORED records into the (pinned, device-visible) staging
    // memory: no size has to reach the host first, and 48 bytes per keypoint cross PCIe instead of 48 * max_features.
    // What lies behind the stored records in the staging arrays is stale, as it is in the reference's buffers.
    void *dc = p->dv_count, *dk = p->dv_corners, *dd = p->dv_desc;
    const bool direct = !p->env.single_memcpy && dc != nullptr;
    if (direct && p->fused && p->use_brief_t && p->oob == kOobZero && !p->env.single_split) {  // k_brief_one is built for the default policy
        // Three launches per frame: one k_front per level, then k_brief_one -- slot prefix, both BRIEF kernels and the
        // write to host staging in one (a dependent launch costs 6-10 us whatever it does, and this call is the
        // reference's only shape).
        // Levels 0 and 1 in ONE launch when level 1 can build its rows from the frame (k_front_pair): RGBA or Y8 input, level 1 an
        // exact half of a level 0 whose width is a multiple of 8, both on full-width bands of 8 rows (the latency shape).
        const Pyramid& py = p->pyr;
        const bool pair_ok = py.depth >= 2u && py.w[0] == 2u * py.w[1] && py.h[0] == 2u * py.h[1] && (py.w[0] & 7u) == 0u &&
                             p->tile_w_lvl[0] == 0u && p->tile_w_lvl[1] == 0u && p->band_rows_lvl[0] == 8u && p->band_rows_lvl[1] == 8u &&
                             !p->d_stamps && !p->env.single_serial &&
                             // a band stages a row with one thread per four texels (level 1 from the frame: per four of ITS texels):
                             // both rows must fit the 1024 threads, else no thread would stage anything (orb_front_body.inc)
                             (py.w[0] >> 2) <= (uint32_t)kFrontThreadsL0 &&
                             ((((std::max(py.w[1], ((py.w[0] / 2u + 7u) / 8u) * 8u) + 4u + 7u) & ~7u)) >> 2) <= (uint32_t)kFrontThreadsL0;
        if (pair_ok) {
            FrontGeom g[2];
            uint32_t width = py.w[0], height = py.h[0], lds = 0;
            for (uint32_t lvl = 0; lvl < 2u; lvl++) {
                const uint32_t gw = ((width + 7u) / 8u) * 8u, gh = ((height + 7u) / 8u) * 8u;
                width /= 2u;
                height /= 2u;
                g[lvl] = front_geometry(py, lvl, gw ? gw : 8u, gh, 1, 8u, 0u);
                if (gw == 0) g[lvl].gh = 0;
                g[lvl].slot_base = p->bands.slot_base[lvl];
                g[lvl].n_slots = p->bands.n_slots;
                g[lvl].seg_cap = p->bands.seg_cap;
                g[lvl].n_classes = p->seg_classes;
                g[lvl].xcd_swizzle = 0u;
                g[lvl].wq = p->wq;
                g[lvl].fp = p->opt.fp_contract;
                if (p->env.phase_mask >= 0) g[lvl].phase_mask = (uint32_t)p->env.phase_mask;
                lds = std::max(lds, front_lds_bytes(g[lvl]));
            }
            // a level-1 band stages a row with one thread per four texels, a level-0 band with one per quad (orb_front_body.inc)
            if (lds > p->max_lds || ((g[1].ls - (uint32_t)kLdsPad) >> 2) > (uint32_t)kFrontThreadsL0 || (py.w[0] >> 2) > (uint32_t)kFrontThreadsL0)
                return fail(p, ORB_EINVAL, "internal: k_front_pair does not fit this frame (%u bytes of LDS, %u columns)", lds, py.w[0]);
            {
                LaunchScope ls(p, s, KID_FUSED_L0);
                const FrontPairLaunch PL{d_in, p->frame_bytes, p->d_gray, p->d_blur, p->d_blur_rowc, py, g[0], g[1], p->threshold, p->d_seg_counts, p->d_seg, s, lds,
                                         p->input_y8};
                const hipError_t e = kFrontPairLaunch[front_form(p->opt.fp_contract, !p->input_y8)](PL);
                if (e != hipSuccess) return fail(p, ORB_EHIP, "k_front_pair failed to launch: %s", hipGetErrorString(e));
            }
            if (py.depth > 2u)
                if (int rc = run_fused_range(p, d_in, 0, 1, s, false, 2u)) return rc;
        } else if (int rc = run_fused_range(p, d_in, 0, 1, s, false)) {
            return rc;
        }
        const uint32_t seq = ++p->single_seq ? p->single_seq : ++p->single_seq;  // never 0
        {
            LaunchScope ls(p, s, KID_BRIEF_ONE);
#define BRIEF_ONE_LAUNCH(ROT_)                                                                                                                                  \
    hipLaunchKernelGGL(k_brief_one<ROT_>, dim3((unsigned)((cap + kBriefOneChunk - 1u) / kBriefOneChunk)), dim3(kBriefOneThreads), brieft_lds_bytes(p->brieft), s, \
                       p->d_blur, p->d_blur_rowc, p->pyr, p->brieft, p->d_seg_counts, p->d_seg_before, p->d_seg, p->d_counts, p->d_corners, (uint32_t)cap,      \
                       p->d_desc, BriefTables{p->d_pattern, p->d_cos, p->d_sin, p->d_rot}, static_cast<uint32_t*>(dc), static_cast<CornerData*>(dk),            \
                       static_cast<CornerDescriptor*>(dd), p->d_single_done, seq)
            switch (rot_form(p->opt.fp_contract)) {  // the rotation's form (OrbOptions::fp_contract)
                case 1: BRIEF_ONE_LAUNCH(1); break;
                case 2: BRIEF_ONE_LAUNCH(2); break;
                default: BRIEF_ONE_LAUNCH(0); break;
            }
#undef BRIEF_ONE_LAUNCH
        }
        HIP_TRY(p, hipGetLastError());
        if (p->env.single_block) {
            if (!p->single_done_ev) HIP_TRY(p, hipEventCreateWithFlags(&p->single_done_ev, hipEventBlockingSync | hipEventDisableTiming));
            HIP_TRY(p, hipEventRecord(p->single_done_ev, s));
        }
        p->planes_valid = true;
        // Block (orb.rs:549): the last workgroup of k_brief_one to finish publishes the call's sequence number in pinned memory
        // behind everything the launch wrote there, and the host polls that word -- hipStreamSynchronize costs 5 us more
        // than the poll (tools/ubench/launch_floor.hip: 11.6 against 6.5 us for an empty launch).  The poll gives up after
        // 20 ms (a faulted kernel never publishes) and a stream synchronisation reports what happened.
        // A host thread that spins on a shared box now and then loses its CPU for a scheduler slice (two to four iterations of 200
        // take 20 ms, NOTEBOOK.md): with TINYORB_SINGLE_WAIT=block / ORB_FLAG_SINGLE_BLOCKING_WAIT the spin is bounded (kSingleSpinUs: the call
        // normally completes inside it) and the thread then SLEEPS in hipEventSynchronize on a blocking-sync event recorded behind the launch
        // -- woken by the completion interrupt, as the reference's device.poll(Wait) is (orb.rs:547).
        bool seen = false;
        constexpr int kSingleSpinUs = 50;
        if (!p->profiling && !p->env.single_sync) {
            const volatile uint32_t* const done = p->h_count + kSingleDoneWord;
            const auto t0 = std::chrono::steady_clock::now();
            const auto limit = p->env.single_block ? std::chrono::microseconds(kSingleSpinUs) : std::chrono::microseconds(20000);
            const uint32_t check = p->env.single_block ? 63u : 1023u;
            for (uint32_t spins = 0; !(seen = __atomic_load_n(done, __ATOMIC_ACQUIRE) == seq); spins++)
                if ((spins & check) == check && std::chrono::steady_clock::now() - t0 > limit) break;
            if (!seen && p->env.single_block) {
                HIP_TRY(p, hipEventSynchronize(p->single_done_ev));  // sleeps; the kernel is over when it returns, its stores are in host memory
                seen = __atomic_load_n(done, __ATOMIC_ACQUIRE) == seq;
            }
        }
        if (!seen) {
            HIP_TRY(p, hipStreamSynchronize(s));
            // not published within 20 ms (or polling is off): whatever happened, the launch is over now -- a launch that
            // was cut short must not leave its workgroup count behind for the next call to add to
            if (!p->profiling && !p->env.single_sync) HIP_TRY(p, hipMemsetAsync(p->d_single_done, 0, sizeof(uint32_t), s));
        }
#ifdef TINYORB_STAMPS
        if (getenv("TINYORB_PRINT_STAMPS")) {  // k_brief_one's joints, workgroups 0 and 24, in units of 10 ns
            fprintf(stderr, "k_brief_one stamps (load+stage, scan, flat body, other body, fence) x 10 ns:");
            for (int i = 1; i <= 10; i++) fprintf(stderr, " %u%s", p->h_count[i], i == 5 ? " |" : "");
            fprintf(stderr, "\n");
        }
#endif
        p->single_valid = true;
        p->last_batch = 1;
        p->last_stream = s;
        *corner_count = *p->h_count;  // raw counter, orb.rs:550-556
        if (*corner_count > cap) return fail(p, ORB_ECAPACITY, "%u corners detected, max_features is %zu", *corner_count, cap);
        return ORB_OK;
    }
    if (int rc = run_pipeline(p, d_in, 1, s)) return rc;
    if (direct) {
        p->last_batch = 1;
        p->last_stream = s;
        if (int rc = launch_compact(p, 1, static_cast<uint32_t*>(dc), nullptr, static_cast<CornerData*>(dk),
                                    static_cast<CornerDescriptor*>(dd), cap, s, nullptr))
            return rc;
    } else {
        (void)hipGetLastError();
        HIP_TRY(p, hipMemcpyAsync(p->h_count, p->d_counts, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(p, hipMemcpyAsync(p->h_corners, p->d_corners, cap * sizeof(CornerData), hipMemcpyDeviceToHost, s));
        HIP_TRY(p, hipMemcpyAsync(p->h_desc, p->d_desc, cap * sizeof(CornerDescriptor), hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(p, hipStreamSynchronize(s));
    p->single_valid = true;
    p->last_batch = 1;
    p->last_stream = s;
    *corner_count = *p->h_count;  // raw counter, orb.rs:550-556
    if (*corner_count > cap) return fail(p, ORB_ECAPACITY, "%u corners detected, max_features is %zu", *corner_count, cap);
    return ORB_OK;
}

int orb_read_corners(OrbProgram* p, CornerData* dst, size_t n) {
    if (!p) return ORB_EINVAL;
    if (!dst && n) return fail(p, ORB_EINVAL, "dst is NULL");
    if (!p->single_valid) return fail(p, ORB_ESTATE, "read_corners before extract_corners");
    const size_t cap = p->cfg.max_features;
    memcpy(dst, p->h_corners, (n < cap ? n : cap) * sizeof(CornerData));
    return ORB_OK;
}

int orb_read_descriptors(OrbProgram* p, CornerDescriptor* dst, size_t n) {
    if (!p) return ORB_EINVAL;
    if (!dst && n) return fail(p, ORB_EINVAL, "dst is NULL");
    if (!p->single_valid) return fail(p, ORB_ESTATE, "read_descriptors before extract_corners");
    const size_t cap = p->cfg.max_features;
    memcpy(dst, p->h_desc, (n < cap ? n : cap) * sizeof(CornerDescriptor));
    return ORB_OK;
}

int orb_extract_batch_device(OrbProgram* p, const uint8_t* frames_dev, uint32_t n_frames, void* stream) {
    if (!p) return ORB_EINVAL;
    if (!frames_dev) return fail(p, ORB_EINVAL, "frames_dev is NULL");
    if (n_frames == 0 || n_frames > p->max_batch)
        return fail(p, ORB_EINVAL, "n_frames %u outside 1..=max_batch (%u)", n_frames, p->max_batch);
    HIP_TRY(p, hipSetDevice(p->device));
    hipStream_t s = stream ? (hipStream_t)stream : p->stream;
    if (int rc = run_pipeline(p, frames_dev, n_frames, s)) return rc;
    p->last_batch = n_frames;
    p->last_stream = s;
    p->single_valid = false;
    return ORB_OK;
}

// Chunked upload of pinned host frames on the copy stream, the kernels of chunk c running while chunk c+1 is still
// crossing PCIe (only the fused path can work on a sub-range of the batch).  Asynchronous.
static int upload_chunked_and_run(OrbProgram* p, const uint8_t* frames_pinned, uint32_t n_frames, hipStream_t s) {
    const uint32_t chunk = 16;
    const size_t total = p->frame_bytes * n_frames;
    if (!p->upload_done) HIP_TRY(p, hipEventCreateWithFlags(&p->upload_done, hipEventDisableTiming));
    if (!p->fused || n_frames <= chunk) {
        HIP_TRY(p, hipMemcpyAsync(p->d_input, frames_pinned, total, hipMemcpyHostToDevice, s));
        HIP_TRY(p, hipEventRecord(p->upload_done, s));
        return run_pipeline(p, p->d_input, n_frames, s);
    }
    if (!p->copy_stream) HIP_TRY(p, hipStreamCreateWithFlags(&p->copy_stream, hipStreamNonBlocking));
    const size_t n_chunks = (n_frames + chunk - 1) / chunk;
    while (p->upload_events.size() < n_chunks) {
        hipEvent_t ev = nullptr;
        HIP_TRY(p, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        p->upload_events.push_back(ev);
    }
    // the input slab is reused: the first upload must not overtake the kernels of the previous batch that still read it
    if (!p->order_event) HIP_TRY(p, hipEventCreateWithFlags(&p->order_event, hipEventDisableTiming));
    HIP_TRY(p, hipEventRecord(p->order_event, s));
    HIP_TRY(p, hipStreamWaitEvent(p->copy_stream, p->order_event, 0));
    for (uint32_t c = 0, f0 = 0; f0 < n_frames; c++, f0 += chunk) {
        const uint32_t m = n_frames - f0 < chunk ? n_frames - f0 : chunk;
        HIP_TRY(p, hipMemcpyAsync(p->d_input + (size_t)f0 * p->frame_bytes, frames_pinned + (size_t)f0 * p->frame_bytes,
                                  (size_t)m * p->frame_bytes, hipMemcpyHostToDevice, p->copy_stream));
        HIP_TRY(p, hipEventRecord(p->upload_events[c], p->copy_stream));
        HIP_TRY(p, hipStreamWaitEvent(s, p->upload_events[c], 0));
        if (int rc = run_fused_range(p, p->d_input, f0, m, s)) return rc;
    }
    HIP_TRY(p, hipEventRecord(p->upload_done, p->copy_stream));
    p->planes_valid = true;
    return ORB_OK;
}

int orb_extract_batch_pinned(OrbProgram* p, const uint8_t* frames_pinned, uint32_t n_frames) {
    if (!p) return ORB_EINVAL;
    if (!frames_pinned) return fail(p, ORB_EINVAL, "frames_pinned is NULL");
    if (n_frames == 0 || n_frames > p->max_batch)
        return fail(p, ORB_EINVAL, "n_frames %u outside 1..=max_batch (%u)", n_frames, p->max_batch);
    HIP_TRY(p, hipSetDevice(p->device));
    if (int rc = ensure_input(p)) return rc;
    if (int rc = upload_chunked_and_run(p, frames_pinned, n_frames, p->stream)) return rc;
    p->last_batch = n_frames;
    p->last_stream = p->stream;
    p->single_valid = false;
    return ORB_OK;
}

int orb_upload_sync(OrbProgram* p) {
    if (!p) return ORB_EINVAL;
    HIP_TRY(p, hipSetDevice(p->device));
    if (p->upload_done) HIP_TRY(p, hipEventSynchronize(p->upload_done));  // recorded behind the last upload
    return ORB_OK;
}

int orb_extract_batch_host(OrbProgram* p, const uint8_t* frames_host, uint32_t n_frames) {
    if (!p) return ORB_EINVAL;
    if (!frames_host) return fail(p, ORB_EINVAL, "frames_host is NULL");
    if (n_frames == 0 || n_frames > p->max_batch)
        return fail(p, ORB_EINVAL, "n_frames %u outside 1..=max_batch (%u)", n_frames, p->max_batch);
    HIP_TRY(p, hipSetDevice(p->device));
    if (int rc = ensure_input(p)) return rc;
    hipStream_t s = p->stream;
    // Ingest (README.md:42 of the reference): the caller's frames are pinned in place for the
    // duration of the call and uploaded in chunks on a copy stream; the kernels of chunk c run while chunk c+1 is
    // still crossing PCIe.
    const size_t total = p->frame_bytes * n_frames;
    const bool pinned = hipHostRegister(const_cast<uint8_t*>(frames_host), total, hipHostRegisterDefault) == hipSuccess;
    if (!pinned) (void)hipGetLastError();
    int rc = ORB_OK;
    if (!pinned) {  // pageable source: the copy is staged by the runtime
        hipError_t e = hipMemcpyAsync(p->d_input, frames_host, total, hipMemcpyHostToDevice, s);
        if (e != hipSuccess) rc = fail(p, ORB_EHIP, "upload failed: %s", hipGetErrorString(e));
        if (!rc) rc = run_pipeline(p, p->d_input, n_frames, s);
    } else {
        rc = upload_chunked_and_run(p, frames_host, n_frames, s);
        // the host pages must stay pinned until the last upload has finished
        if (p->upload_done && hipEventSynchronize(p->upload_done) != hipSuccess && !rc) rc = fail(p, ORB_EHIP, "upload sync failed");
        if (rc) (void)hipStreamSynchronize(s), (void)(p->copy_stream && hipStreamSynchronize(p->copy_stream));
        (void)hipHostUnregister(const_cast<uint8_t*>(frames_host));
    }
    if (rc) return rc;
    p->last_batch = n_frames;
    p->last_stream = s;
    p->single_valid = false;
    return ORB_OK;
}

int orb_batch_sync(OrbProgram* p) {
    if (!p) return ORB_EINVAL;
    HIP_TRY(p, hipSetDevice(p->device));
    HIP_TRY(p, hipStreamSynchronize(p->last_stream ? p->last_stream : p->stream));
    return ORB_OK;
}

int orb_batch_counts(OrbProgram* p, uint32_t* totals, uint32_t n_frames) {
    if (!p) return ORB_EINVAL;
    if (!totals || n_frames > p->last_batch) return fail(p, ORB_EINVAL, "batch_counts: bad arguments");
    if (int rc = orb_batch_sync(p)) return rc;
    HIP_TRY(p, hipMemcpy(totals, p->d_counts, sizeof(uint32_t) * n_frames, hipMemcpyDeviceToHost));
    return ORB_OK;
}

int orb_batch_read(OrbProgram* p, uint32_t frame, CornerData* corners, CornerDescriptor* descriptors, size_t n) {
    if (!p) return ORB_EINVAL;
    if (frame >= p->last_batch) return fail(p, ORB_EINVAL, "batch_read: frame %u not in the last batch", frame);
    if (int rc = orb_batch_sync(p)) return rc;
    const size_t cap = p->cfg.max_features;
    if (n > cap) n = cap;
    if (corners)
        HIP_TRY(p, hipMemcpy(corners, p->d_corners + (size_t)frame * cap, n * sizeof(CornerData), hipMemcpyDeviceToHost));
    if (descriptors)
        HIP_TRY(p, hipMemcpy(descriptors, p->d_desc + (size_t)frame * cap, n * sizeof(CornerDescriptor),
                             hipMemcpyDeviceToHost));
    return ORB_OK;
}

}  // extern "C"

namespace {

// Launches k_compact for the first n frames of the last batch on `stream` (NULL: the batch's own stream); a foreign
// stream is first ordered behind the batch with an event.
int launch_compact(OrbProgram* p, uint32_t n, uint32_t* counts, uint64_t* offsets, CornerData* corners,
                   CornerDescriptor* desc, size_t capacity, void* stream, hipStream_t* used) {
    hipStream_t batch = p->last_stream ? p->last_stream : p->stream;
    hipStream_t s = stream ? (hipStream_t)stream : batch;
    if (s != batch) {
        if (!p->order_event) HIP_TRY(p, hipEventCreateWithFlags(&p->order_event, hipEventDisableTiming));
        HIP_TRY(p, hipEventRecord(p->order_event, batch));
        HIP_TRY(p, hipStreamWaitEvent(s, p->order_event, 0));
    }
    const uint32_t cap = p->cfg.max_features;
    {
        LaunchScope ls(p, s, KID_COMPACT);
        hipLaunchKernelGGL(k_compact, dim3(n, (cap + kCompactChunk - 1u) / kCompactChunk), dim3(256), 0, s, p->d_counts,
                           p->d_corners, p->d_desc, cap, n, counts, reinterpret_cast<unsigned long long*>(offsets), corners, desc,
                           (unsigned long long)capacity);
    }
    HIP_TRY(p, hipGetLastError());
    if (used) *used = s;
    return ORB_OK;
}

}  // namespace

extern "C" {

int orb_batch_compact_device(OrbProgram* p, uint32_t n_frames, uint32_t* counts_dev, uint64_t* offsets_dev,
                             CornerData* corners_dev, CornerDescriptor* descriptors_dev, size_t capacity, void* stream) {
    if (!p) return ORB_EINVAL;
    if (n_frames == 0 || n_frames > p->last_batch) return fail(p, ORB_EINVAL, "compact: n_frames %u not in the last batch (%u)", n_frames, p->last_batch);
    if (!corners_dev || !descriptors_dev) return fail(p, ORB_EINVAL, "compact: destination is NULL");
    HIP_TRY(p, hipSetDevice(p->device));
    return launch_compact(p, n_frames, counts_dev, offsets_dev, corners_dev, descriptors_dev, capacity, stream, nullptr);
}

int orb_batch_read_all(OrbProgram* p, uint32_t n_frames, uint32_t* counts, uint64_t* offsets, CornerData* corners,
                       CornerDescriptor* descriptors, size_t capacity, void* stream) {
    if (!p) return ORB_EINVAL;
    if (n_frames == 0 || n_frames > p->last_batch) return fail(p, ORB_EINVAL, "read_all: n_frames %u not in the last batch (%u)", n_frames, p->last_batch);
    if (!corners || !descriptors) return fail(p, ORB_EINVAL, "read_all: destination is NULL");
    HIP_TRY(p, hipSetDevice(p->device));
    // Device-visible addresses of the host buffers.  Pinned memory (orb_host_alloc / hipHostRegister by the caller)
    // resolves directly; anything else is registered for the duration of the call, which then has to block.
    struct Buf { void* host; size_t bytes; void* dev; bool registered; };
    Buf bufs[4] = {{counts, sizeof(uint32_t) * n_frames, nullptr, false},
                   {offsets, sizeof(uint64_t) * ((size_t)n_frames + 1u), nullptr, false},
                   {corners, sizeof(CornerData) * capacity, nullptr, false},
                   {descriptors, sizeof(CornerDescriptor) * capacity, nullptr, false}};
    bool blocking = false;
    int rc = ORB_OK;
    for (Buf& b : bufs) {
        if (!b.host || b.bytes == 0) continue;
        if (hipHostGetDevicePointer(&b.dev, b.host, 0) == hipSuccess && b.dev) continue;
        (void)hipGetLastError();
        hipError_t e = hipHostRegister(b.host, b.bytes, hipHostRegisterMapped);
        if (e == hipSuccess) {
            b.registered = true;
            e = hipHostGetDevicePointer(&b.dev, b.host, 0);
        }
        if (e != hipSuccess) {
            rc = fail(p, ORB_EHIP, "read_all: host buffer cannot be made device-visible: %s", hipGetErrorString(e));
            break;
        }
        blocking = true;
    }
    hipStream_t used = nullptr;
    if (!rc)
        rc = launch_compact(p, n_frames, (uint32_t*)bufs[0].dev, (uint64_t*)bufs[1].dev, (CornerData*)bufs[2].dev,
                            (CornerDescriptor*)bufs[3].dev, capacity, stream, &used);
    if (blocking || rc) {
        if (used && hipStreamSynchronize(used) != hipSuccess && !rc) rc = fail(p, ORB_EHIP, "read_all: stream sync failed");
        for (Buf& b : bufs)
            if (b.registered) (void)hipHostUnregister(b.host);
    }
    return rc;
}

int orb_batch_pack(OrbProgram* p, uint32_t n_frames, void* stream) {
    if (!p) return ORB_EINVAL;
    if (n_frames == 0 || n_frames > p->last_batch) return fail(p, ORB_EINVAL, "pack: n_frames %u not in the last batch (%u)", n_frames, p->last_batch);
    HIP_TRY(p, hipSetDevice(p->device));
    const uint32_t set = p->cur_set;
    const size_t B = p->max_batch, cap = p->cfg.max_features;
    if (!p->d_pack_c[set]) {
        HIP_TRY(p, hipMalloc(&p->d_pack_c[set], B * cap * sizeof(CornerData)));
        HIP_TRY(p, hipMalloc(&p->d_pack_d[set], B * cap * sizeof(CornerDescriptor)));
        HIP_TRY(p, hipHostMalloc(&p->h_pack_counts[set], B * sizeof(uint32_t), hipHostMallocMapped));
        HIP_TRY(p, hipHostMalloc(&p->h_pack_offsets[set], (B + 1u) * sizeof(uint64_t), hipHostMallocMapped));
        HIP_TRY(p, hipEventCreateWithFlags(&p->pack_event[set], hipEventDisableTiming));
    }
    void *dc = nullptr, *dof = nullptr;
    HIP_TRY(p, hipHostGetDevicePointer(&dc, p->h_pack_counts[set], 0));
    HIP_TRY(p, hipHostGetDevicePointer(&dof, p->h_pack_offsets[set], 0));
    hipStream_t used = nullptr;
    if (int rc = launch_compact(p, n_frames, (uint32_t*)dc, (uint64_t*)dof, p->d_pack_c[set], p->d_pack_d[set], B * cap, stream, &used))
        return rc;
    HIP_TRY(p, hipEventRecord(p->pack_event[set], used));
    p->pack_n[set] = n_frames;
    return ORB_OK;
}

int orb_batch_fetch(OrbProgram* p, uint32_t set, uint32_t* counts, uint64_t* offsets, CornerData* corners,
                    CornerDescriptor* descriptors, size_t capacity, void* stream) {
    if (!p) return ORB_EINVAL;
    if (set > 1u || !p->pack_event[set] || p->pack_n[set] == 0u) return fail(p, ORB_ESTATE, "fetch: output set %u has not been packed", set);
    if (!corners || !descriptors) return fail(p, ORB_EINVAL, "fetch: destination is NULL");
    HIP_TRY(p, hipSetDevice(p->device));
    HIP_TRY(p, hipEventSynchronize(p->pack_event[set]));  // the sizes are on the host now
    const uint32_t n = p->pack_n[set];
    if (counts) memcpy(counts, p->h_pack_counts[set], n * sizeof(uint32_t));
    if (offsets) memcpy(offsets, p->h_pack_offsets[set], ((size_t)n + 1u) * sizeof(uint64_t));
    const uint64_t total = p->h_pack_offsets[set][n];
    const size_t m = total < capacity ? (size_t)total : capacity;
    hipStream_t s = stream ? (hipStream_t)stream : p->stream;
    if (m) {
        HIP_TRY(p, hipMemcpyAsync(corners, p->d_pack_c[set], m * sizeof(CornerData), hipMemcpyDeviceToHost, s));
        HIP_TRY(p, hipMemcpyAsync(descriptors, p->d_pack_d[set], m * sizeof(CornerDescriptor), hipMemcpyDeviceToHost, s));
    }
    return ORB_OK;
}

int orb_batch_pack_transport(OrbProgram* p, uint32_t set, uint32_t n_frames, void* dst_dev, size_t capacity_records,
                             uint64_t* offsets_dev, void* stream) {
    if (!p) return ORB_EINVAL;
    if (set > 1u || !p->out_counts[set]) return fail(p, ORB_EINVAL, "pack_transport: output set %u does not exist", set);
    if (n_frames == 0 || n_frames > p->max_batch) return fail(p, ORB_EINVAL, "pack_transport: n_frames %u outside 1..=max_batch (%u)", n_frames, p->max_batch);
    if (!dst_dev) return fail(p, ORB_EINVAL, "pack_transport: destination is NULL");
    if (p->pyr.w[0] >= 65536u || p->pyr.h[0] >= 65536u) return fail(p, ORB_EINVAL, "pack_transport: coordinates do not fit 16 bits");
    HIP_TRY(p, hipSetDevice(p->device));
    hipStream_t s = stream ? (hipStream_t)stream : p->stream;
    const uint32_t cap = p->cfg.max_features;
    {
        LaunchScope ls(p, s, KID_PACK_T);
        hipLaunchKernelGGL(k_compact_transport, dim3(n_frames, (cap + kCompactChunk - 1u) / kCompactChunk), dim3(256), 0, s,
                           p->out_counts[set], p->out_corners[set], p->out_desc[set], cap, n_frames,
                           reinterpret_cast<unsigned long long*>(offsets_dev), static_cast<TransportRecord*>(dst_dev),
                           (unsigned long long)capacity_records);
    }
    HIP_TRY(p, hipGetLastError());
    return ORB_OK;
}

int orb_unpack_transport(OrbProgram* p, const void* src_dev, uint32_t n_segments, const uint64_t* src_first, const uint64_t* count,
                         const uint64_t* dst_first, CornerData* corners_dev, CornerDescriptor* descriptors_dev, void* stream) {
    if (!p) return ORB_EINVAL;
    if (!src_dev || !corners_dev || !descriptors_dev || (n_segments && (!src_first || !count || !dst_first)))
        return fail(p, ORB_EINVAL, "unpack_transport: NULL argument");
    HIP_TRY(p, hipSetDevice(p->device));
    hipStream_t s = stream ? (hipStream_t)stream : p->stream;
    for (uint32_t s0 = 0; s0 < n_segments; s0 += (uint32_t)kUnpackSegments) {
        const uint32_t m = std::min<uint32_t>(n_segments - s0, (uint32_t)kUnpackSegments);
        UnpackGeom g{};
        unsigned long long most = 0;
        for (uint32_t i = 0; i < m; i++) {
            g.src_first[i] = src_first[s0 + i], g.count[i] = count[s0 + i], g.dst_first[i] = dst_first[s0 + i];
            most = std::max<unsigned long long>(most, g.count[i]);
        }
        if (most == 0) continue;
        const unsigned long long chunks = std::min<unsigned long long>((most + 255u) / 256u, 4096u);
        LaunchScope ls(p, s, KID_UNPACK_T);
        hipLaunchKernelGGL(k_unpack_transport, dim3((uint32_t)chunks, m), dim3(256), 0, s,
                           static_cast<const TransportRecord*>(src_dev), g, corners_dev, descriptors_dev);
    }
    HIP_TRY(p, hipGetLastError());
    return ORB_OK;
}

int orb_host_alloc(size_t nbytes, void** out) {
    if (!out || nbytes == 0) return ORB_EINVAL;
    *out = nullptr;
    hipError_t e = hipHostMalloc(out, nbytes, hipHostMallocMapped | hipHostMallocPortable);
    if (e != hipSuccess) return fail(nullptr, ORB_EHIP, "hipHostMalloc(%zu) failed: %s", nbytes, hipGetErrorString(e));
    return ORB_OK;
}

void orb_host_free(void* ptr) {
    if (ptr) (void)hipHostFree(ptr);
}

int orb_stream_sync(OrbProgram* p, void* stream) {
    if (!p) return ORB_EINVAL;
    HIP_TRY(p, hipSetDevice(p->device));
    HIP_TRY(p, hipStreamSynchronize(stream ? (hipStream_t)stream : p->stream));
    return ORB_OK;
}

void* orb_program_stream(OrbProgram* p) { return p ? (void*)p->stream : nullptr; }

}  // extern "C"

extern "C" {

void orb_corner_level0_xy(const CornerData* c, float* x0, float* y0) {
    if (!c || !x0 || !y0) return;
    const float s = (float)(1u << (c->octave & 31u));
    *x0 = ((float)c->x + 0.5f) * s - 0.5f;
    *y0 = ((float)c->y + 0.5f) * s - 0.5f;
}

int orb_match_consecutive(OrbProgram* p, uint32_t n_frames, void* stream) {
    if (!p) return ORB_EINVAL;
    if (n_frames < 2u || n_frames > p->last_batch)
        return fail(p, ORB_EINVAL, "match_consecutive: need 2..%u frames of the last batch", p->last_batch);
    if (p->cfg.max_features > (1u << 23)) return fail(p, ORB_EINVAL, "match_consecutive: max_features must be <= 2^23");
    HIP_TRY(p, hipSetDevice(p->device));
    const size_t cap = p->cfg.max_features;
    if (!p->d_matches) HIP_TRY(p, hipMalloc(&p->d_matches, (size_t)p->max_batch * cap * sizeof(MatchRecord)));
    hipStream_t s = stream ? (hipStream_t)stream : (p->last_stream ? p->last_stream : p->stream);
    if (p->match_valu < 0) {  // which matcher: 0 the block-scaled fp4 form on the matrix cores (default), 1 the vector unit, 2 the int8 form
        const char* e = getenv("TINYORB_MATCH_VALU");
        const char* e8 = getenv("TINYORB_MATCH_I8");
        p->match_valu = (e && atoi(e) != 0) ? 1 : ((e8 && atoi(e8) != 0) ? 2 : 0);
    }
    // The matrix-core matchers (orb_kernels_match.h) need the descriptors as signed elements: 128 (fp4) or 256 (int8) bytes per
    // record of the batch.  A capacity beyond their key's index range, or no memory for the bytes, leaves the vector-unit kernel.
    const bool i8 = p->match_valu == 2;
    bool mfma = p->match_valu != 1 && cap <= (size_t)(i8 ? kMatchMaxCap : kMatch4MaxCap);
    if (mfma && !p->d_desc8 && hipMalloc(&p->d_desc8, (size_t)p->max_batch * cap * (i8 ? 256u : 128u)) != hipSuccess) {
        (void)hipGetLastError();
        p->d_desc8 = nullptr;
        mfma = false;
    }
    const dim3 grid_e((unsigned)((cap + 31u) / 32u), n_frames), grid_m(n_frames - 1u, (unsigned)((cap + kMatchQueriesPerWg - 1u) / kMatchQueriesPerWg));
    // one result buffer and one expanded-descriptor buffer per program: a match on another stream than the last one waits for it
    if (!p->match_done) HIP_TRY(p, hipEventCreateWithFlags(&p->match_done, hipEventDisableTiming));
    if (p->match_stream && p->match_stream != s) HIP_TRY(p, hipStreamWaitEvent(s, p->match_done, 0));
    if (mfma && !i8) {
        {
            LaunchScope ls(p, s, KID_DESC_EXPAND);
            hipLaunchKernelGGL(k_desc_expand4, grid_e, dim3(256), 0, s, p->d_counts, p->d_desc, (uint32_t)cap, p->d_desc8);
        }
        LaunchScope ls(p, s, KID_MATCH);
        hipLaunchKernelGGL(k_match_fp4, grid_m, dim3(64 * kMatchWaves), 0, s, p->d_counts, p->d_desc8, (uint32_t)cap, p->d_matches);
    } else if (mfma) {
        {
            LaunchScope ls(p, s, KID_DESC_EXPAND);
            hipLaunchKernelGGL(k_desc_expand, grid_e, dim3(256), 0, s, p->d_counts, p->d_desc, (uint32_t)cap, p->d_desc8);
        }
        LaunchScope ls(p, s, KID_MATCH);
        hipLaunchKernelGGL(k_match_mfma, grid_m, dim3(64 * kMatchWaves), 0, s, p->d_counts, p->d_desc8, (uint32_t)cap, p->d_matches);
    } else {
        LaunchScope ls(p, s, KID_MATCH);
        hipLaunchKernelGGL(k_match, dim3(n_frames - 1u, (unsigned)((cap + 64u * kMatchQ - 1u) / (64u * kMatchQ))), dim3(64), 0, s, p->d_counts,
                           p->d_desc, (uint32_t)cap, p->d_matches);
    }
    HIP_TRY(p, hipGetLastError());
    HIP_TRY(p, hipEventRecord(p->match_done, s));
    p->match_stream = s;
    p->last_stream = s;
    return ORB_OK;
}

int orb_match_read(OrbProgram* p, uint32_t frame, OrbMatch* dst, size_t n) {
    if (!p || !dst) return ORB_EINVAL;
    if (!p->d_matches || frame + 1u >= p->last_batch) return fail(p, ORB_EINVAL, "match_read: no matches for frame %u", frame);
    if (int rc = orb_batch_sync(p)) return rc;
    const size_t cap = p->cfg.max_features;
    if (n > cap) n = cap;
    static_assert(sizeof(OrbMatch) == sizeof(MatchRecord), "OrbMatch layout");
    HIP_TRY(p, hipMemcpy(dst, p->d_matches + (size_t)frame * cap, n * sizeof(OrbMatch), hipMemcpyDeviceToHost));
    return ORB_OK;
}

int orb_batch_select_output(OrbProgram* p, uint32_t set) {
    if (!p) return ORB_EINVAL;
    if (set > 1u || !p->out_counts[set])
        return fail(p, ORB_EINVAL, "output set %u does not exist (create with ORB_FLAG_DOUBLE_OUTPUT)", set);
    p->d_counts = p->out_counts[set];
    p->d_corners = p->out_corners[set];
    p->d_desc = p->out_desc[set];
    p->cur_set = set;
    return ORB_OK;
}

int orb_batch_device_buffers(OrbProgram* p, void** counts, void** corners, void** descriptors) {
    if (!p) return ORB_EINVAL;
    if (counts) *counts = p->d_counts;
    if (corners) *corners = p->d_corners;
    if (descriptors) *descriptors = p->d_desc;
    return ORB_OK;
}

int orb_level_size(const OrbProgram* p, uint32_t level, uint32_t* width, uint32_t* height) {
    if (!p || level >= p->pyr.depth) return ORB_EINVAL;
    if (width) *width = p->pyr.w[level];
    if (height) *height = p->pyr.h[level];
    return ORB_OK;
}

int orb_debug_read_plane(OrbProgram* p, uint32_t frame, int kind, uint32_t level, uint16_t* dst, size_t n_texels) {
    if (!p) return ORB_EINVAL;
    if (!dst || level >= p->pyr.depth || frame >= p->last_batch || (kind != ORB_PLANE_GRAY && kind != ORB_PLANE_BLUR))
        return fail(p, ORB_EINVAL, "debug_read_plane: bad arguments");
    if (!p->planes_valid || ((p->fused || (p->fused_i && !p->igrey_stored)) && kind == ORB_PLANE_GRAY && level == 0))
        return fail(p, ORB_ESTATE, "plane not materialised by the last call");
    const size_t texels = (size_t)p->pyr.w[level] * p->pyr.h[level];
    if (n_texels != texels) return fail(p, ORB_EINVAL, "debug_read_plane: expected %zu texels", texels);
    if (int rc = orb_batch_sync(p)) return rc;
    const uint16_t* base = (kind == ORB_PLANE_GRAY ? p->d_gray : p->d_blur) + (size_t)frame * p->pyr.stride + p->pyr.off[level];
    HIP_TRY(p, hipMemcpy(dst, base, texels * sizeof(uint16_t), hipMemcpyDeviceToHost));
    if ((p->fused || p->fused_x) && kind == ORB_PLANE_BLUR && p->blur_qa[level] > 0) {
        // the fused path keeps columns [0, qa) of a blur level as one constant per row (k_front, phase C)
        const uint32_t w = p->pyr.w[level], h = p->pyr.h[level], qa = p->blur_qa[level];
        std::vector<uint16_t> rowc(h);
        HIP_TRY(p, hipMemcpy(rowc.data(), p->d_blur_rowc + (size_t)frame * p->pyr.row_stride + p->pyr.row_off[level],
                             h * sizeof(uint16_t), hipMemcpyDeviceToHost));
        for (uint32_t y = 0; y < h; y++)
            for (uint32_t x = 0; x < qa && x < w; x++) dst[(size_t)y * w + x] = rowc[y];
    }
    return ORB_OK;
}

int orb_debug_f32_to_f16(OrbProgram* p, const float* src, uint16_t* dst, size_t n) {
    if (!p || !src || !dst) return ORB_EINVAL;
    if (n == 0) return ORB_OK;
    HIP_TRY(p, hipSetDevice(p->device));
    float* d_src = nullptr;
    uint16_t* d_dst = nullptr;
    HIP_TRY(p, hipMalloc(&d_src, n * sizeof(float)));
    hipError_t e = hipMalloc(&d_dst, n * sizeof(uint16_t));
    if (e == hipSuccess) e = hipMemcpy(d_src, src, n * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_probe_f16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, p->stream, d_src, d_dst, n);
        e = hipStreamSynchronize(p->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(dst, d_dst, n * sizeof(uint16_t), hipMemcpyDeviceToHost);
    (void)hipFree(d_src);
    (void)hipFree(d_dst);
    if (e != hipSuccess) return fail(p, ORB_EHIP, "debug_f32_to_f16: %s", hipGetErrorString(e));
    return ORB_OK;
}

int orb_debug_angle_code(OrbProgram* p, const float* cy, const float* cx, uint32_t* dst, size_t n) {
    if (!p || !cy || !cx || !dst) return ORB_EINVAL;
    if (n == 0) return ORB_OK;
    HIP_TRY(p, hipSetDevice(p->device));
    float *d_cy = nullptr, *d_cx = nullptr;
    uint32_t* d_dst = nullptr;
    HIP_TRY(p, hipMalloc(&d_cy, n * sizeof(float)));
    hipError_t e = hipMalloc(&d_cx, n * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&d_dst, n * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemcpy(d_cy, cy, n * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_cx, cx, n * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_probe_angle, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, p->stream, d_cy, d_cx, d_dst, n);
        e = hipStreamSynchronize(p->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(dst, d_dst, n * sizeof(uint32_t), hipMemcpyDeviceToHost);
    (void)hipFree(d_cy);
    (void)hipFree(d_cx);
    (void)hipFree(d_dst);
    if (e != hipSuccess) return fail(p, ORB_EHIP, "debug_angle_code: %s", hipGetErrorString(e));
    return ORB_OK;
}

int orb_debug_rot_table(OrbProgram* p, int16_t* dst, size_t n_entries, uint32_t* codes, uint32_t* pitch) {
    if (!p) return ORB_EINVAL;
    if (!p->d_rot) return fail(p, ORB_ESTATE, "this program has no rotated-pattern table");
    const uint32_t n_codes = p->fused_i ? (p->opt.angle_bins ? p->opt.angle_bins : (uint32_t)ORB_ANGLE_STEPS_FULL) : (uint32_t)ORB_ANGLE_STEPS;
    if (codes) *codes = n_codes;
    if (pitch) *pitch = p->fused_i ? p->itiles.pitch : (uint32_t)kNfPatchCols;
    const size_t n = std::min(n_entries, (size_t)n_codes * 512u);
    if (n && !dst) return fail(p, ORB_EINVAL, "dst is NULL");
    HIP_TRY(p, hipSetDevice(p->device));
    if (n) HIP_TRY(p, hipMemcpy(dst, p->d_rot, n * sizeof(int16_t), hipMemcpyDeviceToHost));
    return ORB_OK;
}

int orb_profile_enable(OrbProgram* p, int enable) {
    if (!p) return ORB_EINVAL;
    p->profiling = enable != 0;
    return ORB_OK;
}

int orb_profile_reset(OrbProgram* p) {
    if (!p) return ORB_EINVAL;
    HIP_TRY(p, hipSetDevice(p->device));
    if (int rc = drain_profile(p)) return rc;
    for (int i = 0; i < ORB_KERNEL_COUNT; i++) {
        p->prof_ms[i] = 0;
        p->prof_n[i] = 0;
    }
    return ORB_OK;
}

int orb_profile_get(OrbProgram* p, int id, double* total_ms, uint64_t* launches) {
    if (!p || id < 0 || id >= ORB_KERNEL_COUNT) return ORB_EINVAL;
    HIP_TRY(p, hipSetDevice(p->device));
    if (int rc = drain_profile(p)) return rc;
    if (total_ms) *total_ms = p->prof_ms[id];
    if (launches) *launches = p->prof_n[id];
    return ORB_OK;
}

int orb_synth_frames_device(OrbProgram* p, uint8_t* frames_dev, uint32_t n_frames, uint32_t seed0, uint32_t flags,
                            uint8_t** out_dev) {
    if (!p) return ORB_EINVAL;
    if (n_frames == 0) return fail(p, ORB_EINVAL, "n_frames is 0");
    HIP_TRY(p, hipSetDevice(p->device));
    if (!frames_dev) {
        if (n_frames > p->max_batch) return fail(p, ORB_EINVAL, "n_frames %u > max_batch %u", n_frames, p->max_batch);
        if (int rc = ensure_input(p)) return rc;
        frames_dev = p->d_input;
        // the program's own slab is what extract_corners works on next: an upload still in flight on the copy stream
        // (orb_write_input_image_pinned) must not land on top of the synthesised frames -- wait for every slab's before the state is reset
        for (uint32_t sl = 0; sl < 2u; sl++)
            if (p->in_async[sl] && p->in_uploaded[sl]) HIP_TRY(p, hipEventSynchronize(p->in_uploaded[sl]));
        p->in_pending = 0u, p->in_last = 0u, p->in_async[0] = p->in_async[1] = false;
    }
    const uint32_t W = p->pyr.w[0], H = p->pyr.h[0];
    if (p->input_y8) flags |= ORB_SYN_Y8;  // a Y8 program's frames are one byte per pixel
    if ((flags & ORB_SYN_Y8) && !p->input_y8) return fail(p, ORB_EINVAL, "ORB_SYN_Y8 needs a program created with ORB_FLAG_INPUT_Y8");
    {
        LaunchScope ls(p, p->stream, KID_SYNTH);
        hipLaunchKernelGGL(k_synth, dim3((W + 255u) / 256u, H, n_frames), dim3(256), 0, p->stream, frames_dev,
                           p->frame_bytes, W, H, seed0, flags);
    }
    HIP_TRY(p, hipGetLastError());
    HIP_TRY(p, hipStreamSynchronize(p->stream));
    if (out_dev) *out_dev = frames_dev;
    return ORB_OK;
}

int orb_debug_stamps(OrbProgram* p, unsigned long long* dst, size_t n) {
    if (!p || !dst) return ORB_EINVAL;
    if (!p->d_stamps) return fail(p, ORB_ESTATE, "stamps are collected only with TINYORB_STAMPS=1");
    if (n > 32) n = 32;
    HIP_TRY(p, hipSetDevice(p->device));
    HIP_TRY(p, hipDeviceSynchronize());
    HIP_TRY(p, hipMemcpy(dst, p->d_stamps, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return ORB_OK;
}

int orb_copy_to_host(OrbProgram* p, void* dst_host, const void* src_dev, size_t nbytes) {
    if (!p || !dst_host || !src_dev) return ORB_EINVAL;
    HIP_TRY(p, hipSetDevice(p->device));
    HIP_TRY(p, hipMemcpy(dst_host, src_dev, nbytes, hipMemcpyDeviceToHost));
    return ORB_OK;
}

}  // extern "C"
