// orb_node.hip -- one process, several GPUs of one node: sharding of a batch of independent frames and the collate of
// the results on the first device (include/tinyorb.h, "one node, several GPUs"; BASELINE.json configs[4]).
//
// The reference has no counterpart (one wgpu device per OrbProgram, orb.rs:47-51).  Built only on the public C ABI of
// libtinyorb + HIP + RCCL.  Frames are independent, so there is NO data-path collective: rank r extracts the contiguous
// range [F*r/n, F*(r+1)/n) on its own device and stream, all devices at once.  The collate is the only exchange:
//   1. every rank packs its stored records back to back (k_compact) -- rank 0 straight into the collate buffer;
//   2. ncclAllGather of the per-frame counters (4 B per frame);
//   3. the counters go to the host once, which gives every rank's exact payload size;
//   4. one group of ncclSend/ncclRecv: rank r -> rank 0, exactly S_r * 16 B of keypoints and S_r * 32 B of descriptors,
//      received at rank r's offset of the frame-ordered collate buffer.  xGMI is point to point: every peer uses its own
//      link to rank 0, nothing is relayed, nothing is padded.
// librccl is opened with dlopen on first use, so a single-GPU user of libtinyorb never loads it.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/tinyorb.h"

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

thread_local std::string g_node_create_error;

}  // namespace

struct OrbNode {
    int n = 0;
    std::vector<int> devices;
    std::vector<OrbProgram*> progs;
    std::vector<hipStream_t> streams;
    OrbConfig cfg{};
    uint32_t max_batch = 1;
    size_t frame_bytes = 0;
    Rccl rccl;
    std::vector<ncclComm_t> comms;
    std::vector<uint32_t*> d_sendcounts;          // [max_batch] per device, zero padded
    std::vector<uint32_t*> d_allcounts;           // [n * max_batch] per device
    std::vector<CornerData*> d_pack_c;            // rank 0 only: the collated keypoints [n * max_batch * cap] (else null)
    std::vector<CornerDescriptor*> d_pack_d;
    bool loopback = false;                        // TINYORB_NODE_LOOPBACK=1: the exchange runs as device copies, not RCCL (see orb_node_create)
    std::vector<void*> d_wire;                    // 40-byte transport records (tinyorb.h): rank r >= 1 its own [max_batch * cap]
                                                  // to send, rank 0 the received ones [(n - 1) * max_batch * cap] (null when n == 1)
    std::vector<uint8_t*> d_frames;               // per device, only for orb_node_extract_batch_host
    uint32_t* h_allcounts = nullptr;              // pinned [n * max_batch]
    std::vector<uint32_t> shard_n;                // frames of each rank in the last job
    uint32_t last_frames = 0;
    bool extracted = false, collated = false;
    uint64_t total_records = 0;
    std::string err;
};

namespace {

int nfail(OrbNode* node, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (node)
        node->err = buf;
    else
        g_node_create_error = buf;
    return code;
}

#define NODE_HIP(node, expr)                                                                                  \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess) return nfail((node), ORB_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)
#define NODE_NCCL(node, expr)                                                                                  \
    do {                                                                                                       \
        ncclResult_t r_ = (expr);                                                                              \
        if (r_ != ncclSuccess)                                                                                 \
            return nfail((node), ORB_EHIP, "%s failed: %s", #expr,                                            \
                         (node)->rccl.GetErrorString ? (node)->rccl.GetErrorString(r_) : "rccl error");      \
    } while (0)
#define NODE_ORB(node, prog, expr)                                                             \
    do {                                                                                       \
        int rc_ = (expr);                                                                      \
        if (rc_ != ORB_OK && rc_ != ORB_ECAPACITY) return nfail((node), rc_, "%s: %s", #expr, orb_last_error(prog)); \
    } while (0)

int load_rccl(OrbNode* node) {
    Rccl& R = node->rccl;
    if (R.handle) return ORB_OK;
    const char* cands[] = {getenv("TINYORB_RCCL_PATH"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1",
                           "/opt/rocm/lib/librccl.so"};
    std::string tried;
    for (const char* c : cands) {
        if (!c || !*c) continue;
        R.handle = dlopen(c, RTLD_NOW | RTLD_LOCAL);
        if (R.handle) break;
        tried += std::string(c) + " ";
    }
    if (!R.handle) return nfail(node, ORB_EHIP, "librccl not found (tried: %s); set TINYORB_RCCL_PATH", tried.c_str());
    auto sym = [&](const char* name) { return dlsym(R.handle, name); };
    R.CommInitAll = reinterpret_cast<decltype(R.CommInitAll)>(sym("ncclCommInitAll"));
    R.CommDestroy = reinterpret_cast<decltype(R.CommDestroy)>(sym("ncclCommDestroy"));
    R.AllGather = reinterpret_cast<decltype(R.AllGather)>(sym("ncclAllGather"));
    R.Send = reinterpret_cast<decltype(R.Send)>(sym("ncclSend"));
    R.Recv = reinterpret_cast<decltype(R.Recv)>(sym("ncclRecv"));
    R.GroupStart = reinterpret_cast<decltype(R.GroupStart)>(sym("ncclGroupStart"));
    R.GroupEnd = reinterpret_cast<decltype(R.GroupEnd)>(sym("ncclGroupEnd"));
    R.GetErrorString = reinterpret_cast<decltype(R.GetErrorString)>(sym("ncclGetErrorString"));
    if (!R.CommInitAll || !R.CommDestroy || !R.AllGather || !R.Send || !R.Recv || !R.GroupStart || !R.GroupEnd) {
        dlclose(R.handle);
        R = Rccl{};
        return nfail(node, ORB_EHIP, "librccl lacks a required entry point");
    }
    return ORB_OK;
}

int ensure_comms(OrbNode* node) {
    if (node->loopback || !node->comms.empty()) return ORB_OK;
    if (int rc = load_rccl(node)) return rc;
    node->comms.assign(node->n, nullptr);
    ncclResult_t r = node->rccl.CommInitAll(node->comms.data(), node->n, node->devices.data());
    if (r != ncclSuccess) {
        node->comms.clear();
        return nfail(node, ORB_EHIP, "ncclCommInitAll over %d devices failed: %s", node->n,
                     node->rccl.GetErrorString ? node->rccl.GetErrorString(r) : "rccl error");
    }
    return ORB_OK;
}

void shard_range(uint32_t n_frames, int n, int rank, uint32_t* lo, uint32_t* hi) {
    *lo = (uint32_t)(((uint64_t)n_frames * (uint64_t)rank) / (uint64_t)n);
    *hi = (uint32_t)(((uint64_t)n_frames * (uint64_t)(rank + 1)) / (uint64_t)n);
}

int check_job(OrbNode* node, uint32_t n_frames) {
    if (n_frames == 0) return nfail(node, ORB_EINVAL, "n_frames is 0");
    for (int r = 0; r < node->n; r++) {
        uint32_t lo, hi;
        shard_range(n_frames, node->n, r, &lo, &hi);
        if (hi - lo > node->max_batch)
            return nfail(node, ORB_EINVAL, "rank %d would get %u frames, max_batch is %u", r, hi - lo, node->max_batch);
    }
    return ORB_OK;
}

}  // namespace

extern "C" {

const char* orb_node_last_error(const OrbNode* node) { return node ? node->err.c_str() : g_node_create_error.c_str(); }

int orb_node_device_count(const OrbNode* node) { return node ? node->n : 0; }

OrbProgram* orb_node_program(OrbNode* node, int rank) {
    return (node && rank >= 0 && rank < node->n) ? node->progs[rank] : nullptr;
}

int orb_node_shard(const OrbNode* node, uint32_t n_frames, int rank, uint32_t* lo, uint32_t* hi) {
    if (!node || !lo || !hi || rank < 0 || rank >= node->n) return ORB_EINVAL;
    shard_range(n_frames, node->n, rank, lo, hi);
    return ORB_OK;
}

void orb_node_destroy(OrbNode* node) {
    if (!node) return;
    for (int r = 0; r < node->n; r++) {
        if (r < (int)node->devices.size()) (void)hipSetDevice(node->devices[r]);
        if (r < (int)node->streams.size() && node->streams[r]) (void)hipStreamSynchronize(node->streams[r]);
        if (r < (int)node->comms.size() && node->comms[r]) (void)node->rccl.CommDestroy(node->comms[r]);
        if (r < (int)node->d_sendcounts.size()) (void)hipFree(node->d_sendcounts[r]);
        if (r < (int)node->d_allcounts.size()) (void)hipFree(node->d_allcounts[r]);
        if (r < (int)node->d_pack_c.size()) (void)hipFree(node->d_pack_c[r]);
        if (r < (int)node->d_pack_d.size()) (void)hipFree(node->d_pack_d[r]);
        if (r < (int)node->d_wire.size()) (void)hipFree(node->d_wire[r]);
        if (r < (int)node->d_frames.size()) (void)hipFree(node->d_frames[r]);
        if (r < (int)node->progs.size()) orb_program_destroy(node->progs[r]);
    }
    if (node->h_allcounts) (void)hipHostFree(node->h_allcounts);
    // the RCCL handle stays open: unloading a library with live background threads is not safe
    delete node;
}

int orb_node_create(const int* devices, int n_devices, const OrbConfig* config, const OrbOptions* options, OrbNode** out) {
    if (!out) return nfail(nullptr, ORB_EINVAL, "out is NULL");
    *out = nullptr;
    if (!devices || n_devices <= 0 || n_devices > 64) return nfail(nullptr, ORB_EINVAL, "need 1..64 devices");
    if (!config) return nfail(nullptr, ORB_EINVAL, "config is NULL");
    // TINYORB_NODE_LOOPBACK=1 (test facility): the counters and records of the ranks move by device copies instead of RCCL,
    // and a device may be listed more than once -- the whole n > 1 data path (shards, transport records, offsets, expansion
    // on rank 0) can then run on one GPU, which RCCL refuses.
    const char* lb = getenv("TINYORB_NODE_LOOPBACK");
    const bool loopback = lb && atoi(lb) != 0;
    for (int a = 0; a < n_devices && !loopback; a++)
        for (int b = a + 1; b < n_devices; b++)
            if (devices[a] == devices[b]) return nfail(nullptr, ORB_EINVAL, "device %d listed twice", devices[a]);
    OrbNode* node = new (std::nothrow) OrbNode();
    if (!node) return nfail(nullptr, ORB_EINVAL, "out of host memory");
    node->loopback = loopback;
    node->n = n_devices;
    node->devices.assign(devices, devices + n_devices);
    node->cfg = *config;
    OrbOptions opt{};
    if (options) opt = *options;
    node->max_batch = opt.max_batch ? opt.max_batch : 1u;
    node->frame_bytes = (size_t)config->image_size.width * config->image_size.height * 4u;
    node->shard_n.assign(n_devices, 0u);
    auto bail = [&](int code) {
        g_node_create_error = node->err;
        orb_node_destroy(node);
        return code;
    };
    const size_t cap = config->max_features, B = node->max_batch;
    for (int r = 0; r < n_devices; r++) {
        OrbOptions o = opt;
        o.device = devices[r];
        OrbProgram* p = nullptr;
        int rc = orb_program_create(config, &o, &p);
        if (rc != ORB_OK) {
            nfail(node, rc, "device %d: %s", devices[r], orb_last_error(nullptr));
            return bail(rc);
        }
        node->progs.push_back(p);
        node->streams.push_back((hipStream_t)orb_program_stream(p));
        uint32_t *sc = nullptr, *ac = nullptr;
        CornerData* pc = nullptr;
        CornerDescriptor* pd = nullptr;
        void* wire = nullptr;
        const size_t pack = (size_t)n_devices * B * cap;                                // rank 0: every frame's records
        const size_t wire_records = (r == 0 ? (size_t)(n_devices - 1) : 1u) * B * cap;  // rank 0 receives, the others send
        hipError_t e = hipSetDevice(devices[r]);
        if (e == hipSuccess) e = hipMalloc(&sc, B * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc(&ac, (size_t)n_devices * B * sizeof(uint32_t));
        if (e == hipSuccess && r == 0) e = hipMalloc(&pc, pack * sizeof(CornerData));
        if (e == hipSuccess && r == 0) e = hipMalloc(&pd, pack * sizeof(CornerDescriptor));
        if (e == hipSuccess && wire_records) e = hipMalloc(&wire, wire_records * (size_t)ORB_TRANSPORT_RECORD_BYTES);
        node->d_sendcounts.push_back(sc);
        node->d_allcounts.push_back(ac);
        node->d_pack_c.push_back(pc);
        node->d_pack_d.push_back(pd);
        node->d_wire.push_back(wire);
        node->d_frames.push_back(nullptr);
        if (e != hipSuccess) {
            nfail(node, ORB_EHIP, "device %d: allocation failed: %s", devices[r], hipGetErrorString(e));
            return bail(ORB_EHIP);
        }
    }
    if (hipHostMalloc(&node->h_allcounts, (size_t)n_devices * B * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) {
        nfail(node, ORB_EHIP, "pinned host allocation failed");
        return bail(ORB_EHIP);
    }
    *out = node;
    return ORB_OK;
}

int orb_node_extract_batch(OrbNode* node, const uint8_t* const* frames_dev, uint32_t n_frames) {
    if (!node) return ORB_EINVAL;
    if (!frames_dev) return nfail(node, ORB_EINVAL, "frames_dev is NULL");
    if (int rc = check_job(node, n_frames)) return rc;
    node->extracted = node->collated = false;
    for (int r = 0; r < node->n; r++) {  // asynchronous per device: all shards run at once
        uint32_t lo, hi;
        shard_range(n_frames, node->n, r, &lo, &hi);
        node->shard_n[r] = hi - lo;
        if (hi == lo) continue;
        if (!frames_dev[r]) return nfail(node, ORB_EINVAL, "frames_dev[%d] is NULL", r);
        NODE_ORB(node, node->progs[r], orb_extract_batch_device(node->progs[r], frames_dev[r], hi - lo, nullptr));
    }
    node->last_frames = n_frames;
    node->extracted = true;
    return ORB_OK;
}

int orb_node_extract_batch_host(OrbNode* node, const uint8_t* frames_host, uint32_t n_frames) {
    if (!node) return ORB_EINVAL;
    if (!frames_host) return nfail(node, ORB_EINVAL, "frames_host is NULL");
    if (int rc = check_job(node, n_frames)) return rc;
    std::vector<const uint8_t*> ptrs(node->n, nullptr);
    for (int r = 0; r < node->n; r++) {
        uint32_t lo, hi;
        shard_range(n_frames, node->n, r, &lo, &hi);
        if (hi == lo) continue;
        NODE_HIP(node, hipSetDevice(node->devices[r]));
        if (!node->d_frames[r]) NODE_HIP(node, hipMalloc(&node->d_frames[r], node->frame_bytes * node->max_batch));
        // pageable source: the copy is staged by the runtime and the source may be reused on return
        NODE_HIP(node, hipMemcpyAsync(node->d_frames[r], frames_host + (size_t)lo * node->frame_bytes,
                                      (size_t)(hi - lo) * node->frame_bytes, hipMemcpyHostToDevice, node->streams[r]));
        ptrs[r] = node->d_frames[r];
    }
    return orb_node_extract_batch(node, ptrs.data(), n_frames);
}

int orb_node_collate(OrbNode* node, uint32_t* counts, uint64_t* offsets, void** corners_dev, void** descriptors_dev) {
    if (!node) return ORB_EINVAL;
    if (!node->extracted) return nfail(node, ORB_ESTATE, "collate before extract_batch");
    if (int rc = ensure_comms(node)) return rc;
    const int n = node->n;
    const uint32_t B = node->max_batch;
    const size_t cap = node->cfg.max_features;
    const Rccl& R = node->rccl;
    // 1. pack every shard on its own device (rank 0 directly at the head of the collate buffer) and stage the counters
    for (int r = 0; r < n; r++) {
        NODE_HIP(node, hipSetDevice(node->devices[r]));
        NODE_HIP(node, hipMemsetAsync(node->d_sendcounts[r], 0, B * sizeof(uint32_t), node->streams[r]));
        if (node->shard_n[r] == 0) continue;
        void* d_counts = nullptr;
        NODE_ORB(node, node->progs[r], orb_batch_device_buffers(node->progs[r], &d_counts, nullptr, nullptr));
        NODE_HIP(node, hipMemcpyAsync(node->d_sendcounts[r], d_counts, node->shard_n[r] * sizeof(uint32_t),
                                      hipMemcpyDeviceToDevice, node->streams[r]));
        if (r == 0)  // rank 0's own records go straight to the head of the collated arrays
            NODE_ORB(node, node->progs[0],
                     orb_batch_compact_device(node->progs[0], node->shard_n[0], nullptr, nullptr, node->d_pack_c[0],
                                              node->d_pack_d[0], (size_t)B * cap, nullptr));
        else         // the others pack 40-byte transport records (set 0: a node never switches output sets)
            NODE_ORB(node, node->progs[r],
                     orb_batch_pack_transport(node->progs[r], 0, node->shard_n[r], node->d_wire[r], (size_t)B * cap, nullptr,
                                              node->streams[r]));
    }
    // 2. counters of every frame to every rank
    if (node->loopback) {
        for (int r = 0; r < n; r++) {
            NODE_HIP(node, hipSetDevice(node->devices[r]));
            NODE_HIP(node, hipStreamSynchronize(node->streams[r]));  // every rank's counters and records are in place
        }
        for (int r = 0; r < n; r++) {
            NODE_HIP(node, hipSetDevice(node->devices[r]));
            for (int q = 0; q < n; q++)
                NODE_HIP(node, hipMemcpyAsync(node->d_allcounts[r] + (size_t)q * B, node->d_sendcounts[q], B * sizeof(uint32_t),
                                              hipMemcpyDefault, node->streams[r]));
        }
    } else {
        NODE_NCCL(node, R.GroupStart());
        for (int r = 0; r < n; r++)
            NODE_NCCL(node, R.AllGather(node->d_sendcounts[r], node->d_allcounts[r], B, ncclUint32, node->comms[r], node->streams[r]));
        NODE_NCCL(node, R.GroupEnd());
    }
    // 3. one copy to the host: exact payload sizes
    NODE_HIP(node, hipSetDevice(node->devices[0]));
    NODE_HIP(node, hipMemcpyAsync(node->h_allcounts, node->d_allcounts[0], (size_t)n * B * sizeof(uint32_t),
                                  hipMemcpyDeviceToHost, node->streams[0]));
    NODE_HIP(node, hipStreamSynchronize(node->streams[0]));
    std::vector<uint64_t> rank_records(n, 0), rank_offset(n + 1, 0);
    for (int r = 0; r < n; r++) {
        for (uint32_t f = 0; f < node->shard_n[r]; f++) {
            const uint32_t raw = node->h_allcounts[(size_t)r * B + f];
            rank_records[r] += raw < cap ? raw : cap;
        }
        rank_offset[r + 1] = rank_offset[r] + rank_records[r];
    }
    // 4. transport records of ranks 1.. to rank 0, exact sizes, each peer on its own link; rank 0 then expands them
    //    behind its own records (orb_unpack_transport), in rank = frame order
    std::vector<uint64_t> wire_first(n, 0), wire_count(n, 0), dst_first(n, 0);
    if (!node->loopback) NODE_NCCL(node, R.GroupStart());
    uint64_t at = 0;
    for (int r = 1; r < n; r++) {
        wire_first[r - 1] = at, wire_count[r - 1] = rank_records[r], dst_first[r - 1] = rank_offset[r];
        if (rank_records[r] == 0) continue;
        const size_t bytes = (size_t)rank_records[r] * ORB_TRANSPORT_RECORD_BYTES;
        uint8_t* const landing = static_cast<uint8_t*>(node->d_wire[0]) + (size_t)at * ORB_TRANSPORT_RECORD_BYTES;
        if (node->loopback) {  // the records were complete before the counters were exchanged (stream synchronised above)
            NODE_HIP(node, hipSetDevice(node->devices[0]));
            NODE_HIP(node, hipMemcpyAsync(landing, node->d_wire[r], bytes, hipMemcpyDefault, node->streams[0]));
        } else {
            NODE_NCCL(node, R.Send(node->d_wire[r], bytes, ncclUint8, 0, node->comms[r], node->streams[r]));
            NODE_NCCL(node, R.Recv(landing, bytes, ncclUint8, r, node->comms[0], node->streams[0]));
        }
        at += rank_records[r];
    }
    if (!node->loopback) NODE_NCCL(node, R.GroupEnd());
    if (n > 1 && at > 0)
        NODE_ORB(node, node->progs[0],
                 orb_unpack_transport(node->progs[0], node->d_wire[0], (uint32_t)(n - 1), wire_first.data(), wire_count.data(),
                                      dst_first.data(), node->d_pack_c[0], node->d_pack_d[0], node->streams[0]));
    for (int r = 0; r < n; r++) {
        NODE_HIP(node, hipSetDevice(node->devices[r]));
        NODE_HIP(node, hipStreamSynchronize(node->streams[r]));
    }
    // frame-ordered counters and offsets for the caller
    uint64_t off = 0;
    uint32_t f_out = 0;
    for (int r = 0; r < n; r++)
        for (uint32_t f = 0; f < node->shard_n[r]; f++, f_out++) {
            const uint32_t raw = node->h_allcounts[(size_t)r * B + f];
            if (counts) counts[f_out] = raw;
            if (offsets) offsets[f_out] = off;
            off += raw < cap ? raw : cap;
        }
    if (offsets) offsets[f_out] = off;
    node->total_records = off;
    node->collated = true;
    if (corners_dev) *corners_dev = node->d_pack_c[0];
    if (descriptors_dev) *descriptors_dev = node->d_pack_d[0];
    return ORB_OK;
}

int orb_node_read_collated(OrbNode* node, CornerData* corners, CornerDescriptor* descriptors, size_t capacity) {
    if (!node) return ORB_EINVAL;
    if (!node->collated) return nfail(node, ORB_ESTATE, "read_collated before collate");
    const size_t m = node->total_records < capacity ? (size_t)node->total_records : capacity;
    NODE_HIP(node, hipSetDevice(node->devices[0]));
    if (corners && m) NODE_HIP(node, hipMemcpy(corners, node->d_pack_c[0], m * sizeof(CornerData), hipMemcpyDeviceToHost));
    if (descriptors && m)
        NODE_HIP(node, hipMemcpy(descriptors, node->d_pack_d[0], m * sizeof(CornerDescriptor), hipMemcpyDeviceToHost));
    return ORB_OK;
}

}  // extern "C"
