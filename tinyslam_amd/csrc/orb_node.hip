// orb_node.hip -- one process, several GPUs of one node: sharding of a batch of independent frames and the collate of
// the results on the first device (include/tinyorb.h, "one node, several GPUs"; BASELINE.json configs[4]).
//
// The reference has no counterpart (one wgpu device per OrbProgram, orb.rs:47-51).  Built only on the public C ABI of
// libtinyorb + HIP + RCCL.  Frames are independent, so there is NO data-path collective: rank r extracts the contiguous
// range [F*r/n, F*(r+1)/n) on its own device and stream, all devices at once.  The collate is the only exchange, and it
// is a PIPELINE of three stages per job, so that batch k+1 computes while batch k is collated:
//   extract  kernels of the shard on the device's compute stream (output set k % 2); behind them, on the device's PACK
//            stream, the stored records are packed back to back -- rank 0 with k_compact straight into the collated
//            arrays, the others as 40-byte transport records into a wire buffer -- and the per-frame counters and
//            offsets land in pinned host memory (written by the packing kernel / one small copy): no host
//            synchronisation, and no all-gather -- one process sees every rank's counters;
//   begin    the host waits for the pack events only (the next job's kernels are already queued), reads the exact
//            payload sizes and enqueues, on every device's EXCHANGE stream, one group of ncclSend/ncclRecv: rank r ->
//            rank 0, exactly S_r * 40 bytes.  xGMI is point to point: every peer uses its own link to rank 0, nothing
//            is relayed, nothing is padded.  Rank 0 then expands the records behind its own (k_unpack_transport);
//   end      waits for that exchange and hands out the frame-ordered result.
// Buffers: two output sets and two wire buffers per device (slot k % 2), three collated result buffers on rank 0
// (k % 3: a result stays valid while the next two jobs are extracted), all ordered by events.
// librccl is opened with dlopen on first use, so a single-GPU user of libtinyorb never loads it.
//
// Three ways to move the records (NodeXchg): RCCL between distinct devices (the product), device copies between ranks
// that share a device (TINYORB_NODE_LOOPBACK=1, tests), and RCCL on ONE device (TINYORB_NODE_LOOPBACK=2): a one-rank
// communicator over the device, every rank's transport records sent to the communicator's own rank by ncclSend + ncclRecv
// inside the same group the multi-device path uses -- with n == 1 rank 0's own records take that way too --, so that a
// one-GPU box executes the RCCL code (dlopen, symbols, communicator, grouped point-to-point, stream ordering).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <new>
#include <string>
#include <vector>

#include "../../include/tinyorb.h"

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

thread_local std::string g_node_create_error;

enum NodeXchg { XCHG_RCCL = 0, XCHG_COPIES = 1, XCHG_RCCL_SELF = 2 };

constexpr int kSlots = 2;      // output sets / wire buffers / pinned counter sets
constexpr int kCollSlots = 3;  // collated result buffers on rank 0

struct NodeJob {
    uint32_t n_frames = 0;
    int slot = 0, coll = 0;
    bool exchanging = false;
    std::vector<uint32_t> shard_n;
};

}  // namespace

struct OrbNode {
    int n = 0;
    std::vector<int> devices;
    std::vector<OrbProgram*> progs;
    std::vector<hipStream_t> streams;       // compute: the program's own stream
    std::vector<hipStream_t> pack_streams;  // packing of a finished shard
    std::vector<hipStream_t> xchg_streams;  // Send/Recv (or loopback copies) and the expansion on rank 0
    OrbConfig cfg{};
    uint32_t max_batch = 1;
    size_t frame_bytes = 0;
    Rccl rccl;
    std::vector<ncclComm_t> comms;
    int xchg = XCHG_RCCL;   // how the records move (NodeXchg; TINYORB_NODE_LOOPBACK, see orb_node_create)
    int first_sender = 1;   // ranks [first_sender, n) pack transport records and send them; 0 only with XCHG_RCCL_SELF and n == 1
    uint64_t rccl_pairs = 0;  // ncclSend + ncclRecv pairs enqueued so far
    bool failed = false;      // an enqueue failed half-way: the pipeline state is unknown, every further job is refused
    // per slot (job k uses slot k % kSlots) and rank
    std::vector<void*> d_send[kSlots];         // 40-byte transport records (tinyorb.h) of a sending rank [max_batch * cap]; null for the others
    void* d_recv[kSlots] = {nullptr, nullptr}; // rank 0: the received ones [(n - first_sender) * max_batch * cap] (null without senders)
    std::vector<uint32_t*> h_counts[kSlots];   // pinned [max_batch]: raw per-frame counters of the rank's shard
    std::vector<uint64_t*> h_offsets[kSlots];  // pinned [max_batch + 1]: exclusive prefix of the stored counts
    std::vector<hipEvent_t> ev_kernels[kSlots], ev_pack[kSlots], ev_xchg[kSlots];
    std::vector<char> pack_set[kSlots], xchg_set[kSlots];  // has the event been recorded (is there anything to wait for)?
    CornerData* d_coll_c[kCollSlots] = {nullptr, nullptr, nullptr};  // rank 0: collated keypoints [n * max_batch * cap]
    CornerDescriptor* d_coll_d[kCollSlots] = {nullptr, nullptr, nullptr};
    std::vector<uint8_t*> d_frames;            // per device, only for orb_node_extract_batch_host
    std::deque<NodeJob> jobs;                  // outstanding jobs, oldest first (at most kSlots)
    uint64_t job_seq = 0;
    bool collated = false;                     // a job has been ended: read_collated has something to read
    uint64_t total_records = 0;
    int last_coll = 0;
    // Sharded results (orb_node_set_results): nothing is exchanged, every rank packs its own records into buffers on ITS device
    bool sharded = false;
    std::vector<CornerData*> d_shard_c[kCollSlots];      // per rank [max_batch * cap]
    std::vector<CornerDescriptor*> d_shard_d[kCollSlots];
    std::vector<uint64_t> last_shard_records;            // of the job ended last, per rank
    std::vector<uint32_t> last_shard_frames;
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
    R.Send = reinterpret_cast<decltype(R.Send)>(sym("ncclSend"));
    R.Recv = reinterpret_cast<decltype(R.Recv)>(sym("ncclRecv"));
    R.GroupStart = reinterpret_cast<decltype(R.GroupStart)>(sym("ncclGroupStart"));
    R.GroupEnd = reinterpret_cast<decltype(R.GroupEnd)>(sym("ncclGroupEnd"));
    R.GetErrorString = reinterpret_cast<decltype(R.GetErrorString)>(sym("ncclGetErrorString"));
    if (!R.CommInitAll || !R.CommDestroy || !R.Send || !R.Recv || !R.GroupStart || !R.GroupEnd) {
        dlclose(R.handle);
        R = Rccl{};
        return nfail(node, ORB_EHIP, "librccl lacks a required entry point");
    }
    return ORB_OK;
}

int ensure_comms(OrbNode* node) {
    if (node->xchg == XCHG_COPIES || !node->comms.empty()) return ORB_OK;
    if (node->xchg == XCHG_RCCL && node->n == 1) return ORB_OK;  // nothing to exchange: a single-GPU node never loads librccl
    if (int rc = load_rccl(node)) return rc;
    // XCHG_RCCL_SELF: every rank lives on devices[0]; ONE communicator of one rank over it, all transfers are sends to self
    const int n_comm = node->xchg == XCHG_RCCL_SELF ? 1 : node->n;
    node->comms.assign(n_comm, nullptr);
    ncclResult_t r = node->rccl.CommInitAll(node->comms.data(), n_comm, node->devices.data());
    if (r != ncclSuccess) {
        node->comms.clear();
        return nfail(node, ORB_EHIP, "ncclCommInitAll over %d device(s) failed: %s", n_comm,
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
    if ((int)node->jobs.size() >= kSlots)
        return nfail(node, ORB_ESTATE, "%d jobs are outstanding: orb_node_collate_end the oldest one first", kSlots);
    for (int r = 0; r < node->n; r++) {
        uint32_t lo, hi;
        shard_range(n_frames, node->n, r, &lo, &hi);
        if (hi - lo > node->max_batch)
            return nfail(node, ORB_EINVAL, "rank %d would get %u frames, max_batch is %u", r, hi - lo, node->max_batch);
    }
    return ORB_OK;
}

// Device-visible address of pinned host memory on the current device.
template <typename T>
T* dev_ptr(T* host) {
    void* d = nullptr;
    if (hipHostGetDevicePointer(&d, host, 0) != hipSuccess) return nullptr;
    return static_cast<T*>(d);
}

// After a failure in the middle of an enqueue sequence the events, output sets and wire buffers no longer match what
// the job queue says.  Drain every stream, drop the outstanding jobs and refuse further ones: the caller cannot retry
// into a half-enqueued job (destroy the node and create a new one).
int poison(OrbNode* node, int rc) {
    node->failed = true;
    for (int r = 0; r < node->n; r++) {
        if (hipSetDevice(node->devices[r]) != hipSuccess) continue;
        (void)hipStreamSynchronize(node->streams[r]);
        (void)hipStreamSynchronize(node->pack_streams[r]);
        (void)hipStreamSynchronize(node->xchg_streams[r]);
    }
    (void)hipGetLastError();
    node->jobs.clear();
    return rc;
}

int check_alive(OrbNode* node) {
    if (!node->failed) return ORB_OK;
    const std::string why = node->err;
    return nfail(node, ORB_ESTATE, "node unusable after an earlier failure (%s)", why.c_str());
}

// The exchange of one job: exact-size transfers rank r -> rank 0, then the expansion behind rank 0's own records.
// An RCCL error inside the group still closes the group (a communicator left in an open group hangs the next call).
int enqueue_exchange(OrbNode* node, NodeJob& job) {
    const int n = node->n, slot = job.slot, s0 = node->first_sender;
    const uint32_t B = node->max_batch;
    const size_t cap = node->cfg.max_features;
    const Rccl& R = node->rccl;
    std::vector<uint64_t> rank_records(n, 0), rank_offset(n + 1, 0);
    for (int r = 0; r < n; r++) {
        rank_records[r] = job.shard_n[r] ? node->h_offsets[slot][r][job.shard_n[r]] : 0u;
        if (rank_records[r] > (uint64_t)B * cap) return nfail(node, ORB_EHIP, "internal: rank %d reports %llu records", r, (unsigned long long)rank_records[r]);
        rank_offset[r + 1] = rank_offset[r] + rank_records[r];
    }
    // run i of the landing area = sender s0 + i
    const int n_send = n - s0;
    std::vector<uint64_t> wire_first(n_send > 0 ? n_send : 1, 0), wire_count(n_send > 0 ? n_send : 1, 0), dst_first(n_send > 0 ? n_send : 1, 0);
    uint64_t at = 0;
    for (int r = s0; r < n; r++) {
        wire_first[r - s0] = at, wire_count[r - s0] = rank_records[r], dst_first[r - s0] = rank_offset[r];
        at += rank_records[r];
    }
    if (n_send > 0 && at > 0) {
        const bool self = node->xchg == XCHG_RCCL_SELF;
        // every sender's exchange stream waits for its pack; the landing area is reused in stream order.  On one device with
        // one communicator (self) everything goes through rank 0's exchange stream.
        for (int r = s0; r < n; r++) {
            if (rank_records[r] == 0) continue;
            NODE_HIP(node, hipSetDevice(node->devices[r]));
            NODE_HIP(node, hipStreamWaitEvent(node->xchg_streams[self ? 0 : r], node->ev_pack[slot][r], 0));
        }
        if (node->xchg == XCHG_COPIES) {
            for (int r = s0; r < n; r++) {
                if (rank_records[r] == 0) continue;
                const size_t bytes = (size_t)rank_records[r] * ORB_TRANSPORT_RECORD_BYTES;
                uint8_t* const landing = static_cast<uint8_t*>(node->d_recv[slot]) + (size_t)wire_first[r - s0] * ORB_TRANSPORT_RECORD_BYTES;
                NODE_HIP(node, hipSetDevice(node->devices[r]));
                NODE_HIP(node, hipMemcpyAsync(landing, node->d_send[slot][r], bytes, hipMemcpyDefault, node->xchg_streams[r]));
                NODE_HIP(node, hipEventRecord(node->ev_xchg[slot][r], node->xchg_streams[r]));
                node->xchg_set[slot][r] = 1;
                NODE_HIP(node, hipSetDevice(node->devices[0]));
                NODE_HIP(node, hipStreamWaitEvent(node->xchg_streams[0], node->ev_xchg[slot][r], 0));
            }
        } else {
            ncclResult_t bad = ncclSuccess;
            const char* what = "";
            if (self) NODE_HIP(node, hipSetDevice(node->devices[0]));
            ncclResult_t g = R.GroupStart();
            if (g != ncclSuccess) return nfail(node, ORB_EHIP, "ncclGroupStart failed: %s", R.GetErrorString ? R.GetErrorString(g) : "rccl error");
            for (int r = s0; r < n && bad == ncclSuccess; r++) {
                if (rank_records[r] == 0) continue;
                const size_t bytes = (size_t)rank_records[r] * ORB_TRANSPORT_RECORD_BYTES;
                uint8_t* const landing = static_cast<uint8_t*>(node->d_recv[slot]) + (size_t)wire_first[r - s0] * ORB_TRANSPORT_RECORD_BYTES;
                // self: the one-rank communicator's peer 0 is itself; sends and receives of a group match in order
                bad = R.Send(node->d_send[slot][r], bytes, ncclUint8, 0, node->comms[self ? 0 : r], node->xchg_streams[self ? 0 : r]);
                what = "ncclSend";
                if (bad != ncclSuccess) break;
                bad = R.Recv(landing, bytes, ncclUint8, self ? 0 : r, node->comms[0], node->xchg_streams[0]);
                what = "ncclRecv";
                if (bad == ncclSuccess) node->rccl_pairs++;
            }
            g = R.GroupEnd();  // always: the group must not stay open
            if (bad != ncclSuccess) return nfail(node, ORB_EHIP, "%s failed: %s", what, R.GetErrorString ? R.GetErrorString(bad) : "rccl error");
            if (g != ncclSuccess) return nfail(node, ORB_EHIP, "ncclGroupEnd failed: %s", R.GetErrorString ? R.GetErrorString(g) : "rccl error");
            for (int r = s0; r < n; r++) {
                if (rank_records[r] == 0) continue;
                NODE_HIP(node, hipSetDevice(node->devices[r]));
                NODE_HIP(node, hipEventRecord(node->ev_xchg[slot][r], node->xchg_streams[self ? 0 : r]));  // the send buffer is free again
                node->xchg_set[slot][r] = 1;
            }
        }
        NODE_HIP(node, hipSetDevice(node->devices[0]));
        NODE_ORB(node, node->progs[0],
                 orb_unpack_transport(node->progs[0], node->d_recv[slot], (uint32_t)n_send, wire_first.data(), wire_count.data(),
                                      dst_first.data(), node->d_coll_c[job.coll], node->d_coll_d[job.coll], node->xchg_streams[0]));
    }
    NODE_HIP(node, hipSetDevice(node->devices[0]));
    NODE_HIP(node, hipEventRecord(node->ev_xchg[slot][0], node->xchg_streams[0]));
    node->xchg_set[slot][0] = 1;
    job.exchanging = true;
    return ORB_OK;
}

int begin_oldest(OrbNode* node) {
    NodeJob* job = nullptr;
    for (NodeJob& j : node->jobs)
        if (!j.exchanging) {
            job = &j;
            break;
        }
    if (!job) return nfail(node, ORB_ESTATE, "collate_begin: no extracted job is waiting for its exchange");
    if (node->sharded) {  // nothing to exchange: the job is complete when its packs are
        job->exchanging = true;
        return ORB_OK;
    }
    if (int rc = ensure_comms(node)) return rc;  // nothing enqueued yet: the job stays, the caller may retry
    // the only host wait of the pipeline: the counters of THIS job (its kernels + pack; the next job is already queued)
    for (int r = 0; r < node->n; r++) {
        if (job->shard_n[r] == 0) continue;
        NODE_HIP(node, hipSetDevice(node->devices[r]));
        NODE_HIP(node, hipEventSynchronize(node->ev_pack[job->slot][r]));
    }
    if (int rc = enqueue_exchange(node, *job)) return poison(node, rc);
    return ORB_OK;
}

}  // namespace

extern "C" {

const char* orb_node_last_error(const OrbNode* node) { return node ? node->err.c_str() : g_node_create_error.c_str(); }

int orb_node_device_count(const OrbNode* node) { return node ? node->n : 0; }

int orb_node_pending(const OrbNode* node) { return node ? (int)node->jobs.size() : 0; }

const char* orb_node_exchange_backend(const OrbNode* node) {
    if (!node) return "";
    if (node->n - node->first_sender <= 0) return "none";
    return node->xchg == XCHG_COPIES ? "copies" : node->xchg == XCHG_RCCL_SELF ? "rccl-self" : "rccl";
}

uint64_t orb_node_rccl_pairs(const OrbNode* node) { return node ? node->rccl_pairs : 0u; }

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
        if (r < (int)node->pack_streams.size() && node->pack_streams[r]) (void)hipStreamSynchronize(node->pack_streams[r]);
        if (r < (int)node->xchg_streams.size() && node->xchg_streams[r]) (void)hipStreamSynchronize(node->xchg_streams[r]);
    }
    for (int r = 0; r < node->n; r++) {
        if (r < (int)node->devices.size()) (void)hipSetDevice(node->devices[r]);
        if (r < (int)node->comms.size() && node->comms[r]) (void)node->rccl.CommDestroy(node->comms[r]);
        for (int s = 0; s < kSlots; s++) {
            if (r < (int)node->d_send[s].size()) (void)hipFree(node->d_send[s][r]);
            if (r == 0) (void)hipFree(node->d_recv[s]);
            if (r < (int)node->h_counts[s].size() && node->h_counts[s][r]) (void)hipHostFree(node->h_counts[s][r]);
            if (r < (int)node->h_offsets[s].size() && node->h_offsets[s][r]) (void)hipHostFree(node->h_offsets[s][r]);
            if (r < (int)node->ev_kernels[s].size() && node->ev_kernels[s][r]) (void)hipEventDestroy(node->ev_kernels[s][r]);
            if (r < (int)node->ev_pack[s].size() && node->ev_pack[s][r]) (void)hipEventDestroy(node->ev_pack[s][r]);
            if (r < (int)node->ev_xchg[s].size() && node->ev_xchg[s][r]) (void)hipEventDestroy(node->ev_xchg[s][r]);
        }
        if (r < (int)node->d_frames.size()) (void)hipFree(node->d_frames[r]);
        if (r < (int)node->pack_streams.size() && node->pack_streams[r]) (void)hipStreamDestroy(node->pack_streams[r]);
        if (r < (int)node->xchg_streams.size() && node->xchg_streams[r]) (void)hipStreamDestroy(node->xchg_streams[r]);
        if (r < (int)node->progs.size()) orb_program_destroy(node->progs[r]);
    }
    for (int c = 0; c < kCollSlots; c++)
        for (size_t r = 0; r < node->d_shard_c[c].size(); r++) {
            if (r < node->devices.size()) (void)hipSetDevice(node->devices[r]);
            (void)hipFree(node->d_shard_c[c][r]);
            if (r < node->d_shard_d[c].size()) (void)hipFree(node->d_shard_d[c][r]);
        }
    if (!node->devices.empty()) (void)hipSetDevice(node->devices[0]);
    for (int c = 0; c < kCollSlots; c++) {
        (void)hipFree(node->d_coll_c[c]);
        (void)hipFree(node->d_coll_d[c]);
    }
    // the RCCL handle stays open: unloading a library with live background threads is not safe
    delete node;
}

int orb_node_create(const int* devices, int n_devices, const OrbConfig* config, const OrbOptions* options, OrbNode** out) {
    if (!out) return nfail(nullptr, ORB_EINVAL, "out is NULL");
    *out = nullptr;
    if (!devices || n_devices <= 0 || n_devices > 64) return nfail(nullptr, ORB_EINVAL, "need 1..64 devices");
    if (!config) return nfail(nullptr, ORB_EINVAL, "config is NULL");
    // TINYORB_NODE_LOOPBACK (test facility; the head of this file): 1 = the records of the ranks move by device copies
    // instead of RCCL and a device may be listed more than once -- the whole n > 1 data path (shards, transport records,
    // offsets, expansion on rank 0, the pipeline's events) can then run on one GPU, which RCCL refuses; 2 = the same ranks
    // on ONE device, but the records move through RCCL: a one-rank communicator, ncclSend/ncclRecv to itself.
    const char* lb = getenv("TINYORB_NODE_LOOPBACK");
    const int xchg = lb ? atoi(lb) : 0;
    if (xchg < 0 || xchg > 2) return nfail(nullptr, ORB_EINVAL, "TINYORB_NODE_LOOPBACK must be 0, 1 or 2");
    for (int a = 0; a < n_devices; a++)
        for (int b = a + 1; b < n_devices; b++) {
            if (xchg == XCHG_RCCL && devices[a] == devices[b]) return nfail(nullptr, ORB_EINVAL, "device %d listed twice", devices[a]);
            if (xchg == XCHG_RCCL_SELF && devices[a] != devices[b])
                return nfail(nullptr, ORB_EINVAL, "TINYORB_NODE_LOOPBACK=2 runs every rank on one device (got %d and %d)", devices[a], devices[b]);
        }
    OrbNode* node = new (std::nothrow) OrbNode();
    if (!node) return nfail(nullptr, ORB_EINVAL, "out of host memory");
    node->xchg = xchg;
    node->first_sender = (xchg == XCHG_RCCL_SELF && n_devices == 1) ? 0 : 1;
    node->n = n_devices;
    node->devices.assign(devices, devices + n_devices);
    node->cfg = *config;
    OrbOptions opt{};
    if (options) opt = *options;
    opt.flags |= ORB_FLAG_DOUBLE_OUTPUT;  // batch k+1 computes while batch k is packed
    node->max_batch = opt.max_batch ? opt.max_batch : 1u;
    // ORB_FLAG_INPUT_Y8 programs take one byte per pixel: the shard offsets into a host array follow the flag
    node->frame_bytes = (size_t)config->image_size.width * config->image_size.height * ((opt.flags & ORB_FLAG_INPUT_Y8) ? 1u : 4u);
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
        node->d_frames.push_back(nullptr);
        hipStream_t ps = nullptr, xs = nullptr;
        hipError_t e = hipSetDevice(devices[r]);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&ps, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&xs, hipStreamNonBlocking);
        node->pack_streams.push_back(ps);
        node->xchg_streams.push_back(xs);
        const size_t send_records = r >= node->first_sender ? B * cap : 0u;
        const size_t recv_records = r == 0 ? (size_t)(n_devices - node->first_sender) * B * cap : 0u;  // rank 0 receives
        for (int s = 0; s < kSlots; s++) {
            void* wire = nullptr;
            uint32_t* hc = nullptr;
            uint64_t* ho = nullptr;
            hipEvent_t ek = nullptr, ep = nullptr, ex = nullptr;
            if (e == hipSuccess && send_records) e = hipMalloc(&wire, send_records * (size_t)ORB_TRANSPORT_RECORD_BYTES);
            if (e == hipSuccess && recv_records) e = hipMalloc(&node->d_recv[s], recv_records * (size_t)ORB_TRANSPORT_RECORD_BYTES);
            if (e == hipSuccess) e = hipHostMalloc(&hc, B * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocPortable);
            if (e == hipSuccess) e = hipHostMalloc(&ho, (B + 1u) * sizeof(uint64_t), hipHostMallocMapped | hipHostMallocPortable);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&ek, hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&ep, hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&ex, hipEventDisableTiming);
            if (hc) memset(hc, 0, B * sizeof(uint32_t));
            if (ho) memset(ho, 0, (B + 1u) * sizeof(uint64_t));
            node->d_send[s].push_back(wire);
            node->h_counts[s].push_back(hc);
            node->h_offsets[s].push_back(ho);
            node->ev_kernels[s].push_back(ek);
            node->ev_pack[s].push_back(ep);
            node->ev_xchg[s].push_back(ex);
            node->pack_set[s].push_back(0);
            node->xchg_set[s].push_back(0);
        }
        if (r == 0) {
            const size_t pack = (size_t)n_devices * B * cap;  // every frame's records
            for (int c = 0; c < kCollSlots && e == hipSuccess; c++) {
                e = hipMalloc(&node->d_coll_c[c], pack * sizeof(CornerData));
                if (e == hipSuccess) e = hipMalloc(&node->d_coll_d[c], pack * sizeof(CornerDescriptor));
            }
        }
        if (e != hipSuccess) {
            nfail(node, ORB_EHIP, "device %d: allocation failed: %s", devices[r], hipGetErrorString(e));
            return bail(ORB_EHIP);
        }
    }
    *out = node;
    return ORB_OK;
}

// Stage 1 of a job: the kernels of every shard, and behind them the packing of its results (see the head of this file).
static int enqueue_job(OrbNode* node, const uint8_t* const* frames_dev, const uint8_t* frames_pinned, uint32_t n_frames) {
    NodeJob job;
    job.n_frames = n_frames;
    job.slot = (int)(node->job_seq % (uint64_t)kSlots);
    job.coll = (int)(node->job_seq % (uint64_t)kCollSlots);
    job.shard_n.assign(node->n, 0u);
    const int slot = job.slot;
    const size_t cap = node->cfg.max_features, B = node->max_batch;
    for (int r = 0; r < node->n; r++) {  // asynchronous per device: all shards run at once
        uint32_t lo, hi;
        shard_range(n_frames, node->n, r, &lo, &hi);
        const uint32_t m = hi - lo;
        job.shard_n[r] = m;
        if (m == 0) continue;
        OrbProgram* const prog = node->progs[r];
        NODE_HIP(node, hipSetDevice(node->devices[r]));
        NODE_ORB(node, prog, orb_batch_select_output(prog, (uint32_t)slot));
        // output set `slot` was last read by the pack of the job two back
        if (node->pack_set[slot][r]) NODE_HIP(node, hipStreamWaitEvent(node->streams[r], node->ev_pack[slot][r], 0));
        if (frames_dev)
            NODE_ORB(node, prog, orb_extract_batch_device(prog, frames_dev[r], m, nullptr));
        else
            NODE_ORB(node, prog, orb_extract_batch_pinned(prog, frames_pinned + (size_t)lo * node->frame_bytes, m));
        NODE_HIP(node, hipEventRecord(node->ev_kernels[slot][r], node->streams[r]));
        hipStream_t ps = node->pack_streams[r];
        NODE_HIP(node, hipStreamWaitEvent(ps, node->ev_kernels[slot][r], 0));
        uint32_t* const dc = dev_ptr(node->h_counts[slot][r]);
        uint64_t* const dof = dev_ptr(node->h_offsets[slot][r]);
        if (!dc || !dof) return nfail(node, ORB_EHIP, "pinned counters are not visible to device %d", node->devices[r]);
        if (node->sharded) {  // results stay where they were computed: every rank packs into its own buffers, nothing travels
            NODE_ORB(node, prog, orb_batch_compact_device(prog, m, dc, dof, node->d_shard_c[job.coll][r], node->d_shard_d[job.coll][r],
                                                          B * cap, ps));
        } else if (r < node->first_sender) {  // rank 0's own records go straight to the head of the collated arrays
            NODE_ORB(node, prog, orb_batch_compact_device(prog, m, dc, dof, node->d_coll_c[job.coll], node->d_coll_d[job.coll],
                                                          B * cap, ps));
        } else {       // the others pack 40-byte transport records; the wire buffer was last read by the sends two jobs back
            if (node->xchg_set[slot][r]) NODE_HIP(node, hipStreamWaitEvent(ps, node->ev_xchg[slot][r], 0));
            void* d_counts = nullptr;
            NODE_ORB(node, prog, orb_batch_device_buffers(prog, &d_counts, nullptr, nullptr));
            NODE_ORB(node, prog, orb_batch_pack_transport(prog, (uint32_t)slot, m, node->d_send[slot][r], B * cap, dof, ps));
            NODE_HIP(node, hipMemcpyAsync(node->h_counts[slot][r], d_counts, m * sizeof(uint32_t), hipMemcpyDeviceToHost, ps));
        }
        NODE_HIP(node, hipEventRecord(node->ev_pack[slot][r], ps));
        node->pack_set[slot][r] = 1;
    }
    node->jobs.push_back(std::move(job));
    node->job_seq++;
    return ORB_OK;
}

// Everything that can be refused is refused before the first enqueue; a failure after that leaves devices with work the
// job queue does not know about, so it takes the node out of service (poison).
static int submit_job(OrbNode* node, const uint8_t* const* frames_dev, const uint8_t* frames_pinned, uint32_t n_frames) {
    if (int rc = check_alive(node)) return rc;
    if (int rc = check_job(node, n_frames)) return rc;
    for (int r = 0; r < node->n && frames_dev; r++) {
        uint32_t lo, hi;
        shard_range(n_frames, node->n, r, &lo, &hi);
        if (hi > lo && !frames_dev[r]) return nfail(node, ORB_EINVAL, "frames_dev[%d] is NULL", r);
    }
    if (int rc = enqueue_job(node, frames_dev, frames_pinned, n_frames)) return poison(node, rc);
    return ORB_OK;
}

int orb_node_extract_batch(OrbNode* node, const uint8_t* const* frames_dev, uint32_t n_frames) {
    if (!node) return ORB_EINVAL;
    if (!frames_dev) return nfail(node, ORB_EINVAL, "frames_dev is NULL");
    return submit_job(node, frames_dev, nullptr, n_frames);
}

int orb_node_extract_batch_host(OrbNode* node, const uint8_t* frames_host, uint32_t n_frames) {
    if (!node) return ORB_EINVAL;
    if (!frames_host) return nfail(node, ORB_EINVAL, "frames_host is NULL");
    if (int rc = check_alive(node)) return rc;
    if (int rc = check_job(node, n_frames)) return rc;
    // Every shard goes up in 16-frame chunks from the caller's array, pinned in place for the duration of the uploads,
    // on its device's copy stream while the kernels of the chunks already there run (orb_extract_batch_pinned): all
    // links at once.  Returns when the uploads are done (the array may be reused); the kernels may still be running.
    const size_t total = node->frame_bytes * n_frames;
    const bool pinned = hipHostRegister(const_cast<uint8_t*>(frames_host), total, hipHostRegisterPortable) == hipSuccess;
    if (!pinned) (void)hipGetLastError();  // already pinned by the caller (orb_host_alloc), or not pinnable: the runtime stages the copies
    int rc = submit_job(node, nullptr, frames_host, n_frames);
    const bool queued = rc == ORB_OK;
    for (int r = 0; r < node->n; r++) {
        const int rs = orb_upload_sync(node->progs[r]);
        if (rs != ORB_OK && rc == ORB_OK) rc = nfail(node, rs, "upload to device %d: %s", node->devices[r], orb_last_error(node->progs[r]));
    }
    if (pinned) (void)hipHostUnregister(const_cast<uint8_t*>(frames_host));
    // an upload that failed behind a job that IS queued: the job's kernels read frames that never arrived -- the node goes out of
    // service like after any other failure in the middle of an enqueue sequence (a retry must not collate that job)
    if (queued && rc != ORB_OK) return poison(node, rc);
    return rc;
}

int orb_node_collate_begin(OrbNode* node) {
    if (!node) return ORB_EINVAL;
    if (int rc = check_alive(node)) return rc;
    return begin_oldest(node);
}

int orb_node_collate_end(OrbNode* node, uint32_t* counts, uint64_t* offsets, void** corners_dev, void** descriptors_dev) {
    if (!node) return ORB_EINVAL;
    if (int rc = check_alive(node)) return rc;
    if (node->jobs.empty()) return nfail(node, ORB_ESTATE, "collate before extract_batch");
    if (!node->jobs.front().exchanging)
        if (int rc = begin_oldest(node)) return rc;
    NodeJob& job = node->jobs.front();
    const int slot = job.slot;
    const size_t cap = node->cfg.max_features;
    // a wait that fails here leaves a job whose exchange is in an unknown state: out of service, as in begin_oldest / submit_job
    hipError_t we = hipSuccess;
    if (node->sharded) {
        for (int r = 0; r < node->n && we == hipSuccess; r++) {
            if (job.shard_n[r] == 0) continue;
            we = hipSetDevice(node->devices[r]);
            if (we == hipSuccess) we = hipEventSynchronize(node->ev_pack[slot][r]);  // rank r's records are packed on its device
        }
    } else {
        we = hipSetDevice(node->devices[0]);
        if (we == hipSuccess && job.shard_n[0]) we = hipEventSynchronize(node->ev_pack[slot][0]);  // rank 0's own records are in place
        if (we == hipSuccess) we = hipEventSynchronize(node->ev_xchg[slot][0]);                    // and everybody else's
    }
    if (we != hipSuccess) return poison(node, nfail(node, ORB_EHIP, "collate_end: waiting for the job failed: %s", hipGetErrorString(we)));
    // frame-ordered counters and offsets for the caller
    uint64_t off = 0;
    uint32_t f_out = 0;
    node->last_shard_records.assign(node->n, 0u);
    node->last_shard_frames.assign(node->n, 0u);
    for (int r = 0; r < node->n; r++) {
        const uint64_t off_r = off;
        for (uint32_t f = 0; f < job.shard_n[r]; f++, f_out++) {
            const uint32_t raw = node->h_counts[slot][r][f];
            if (counts) counts[f_out] = raw;
            if (offsets) offsets[f_out] = off;
            off += raw < cap ? raw : cap;
        }
        node->last_shard_records[r] = off - off_r;
        node->last_shard_frames[r] = job.shard_n[r];
    }
    if (offsets) offsets[f_out] = off;
    node->total_records = off;
    node->last_coll = job.coll;
    node->collated = true;
    // sharded: there is no collated array -- the records of rank r lie on device r (orb_node_shard_result)
    if (corners_dev) *corners_dev = node->sharded ? nullptr : (void*)node->d_coll_c[job.coll];
    if (descriptors_dev) *descriptors_dev = node->sharded ? nullptr : (void*)node->d_coll_d[job.coll];
    node->jobs.pop_front();
    return ORB_OK;
}

int orb_node_collate(OrbNode* node, uint32_t* counts, uint64_t* offsets, void** corners_dev, void** descriptors_dev) {
    return orb_node_collate_end(node, counts, offsets, corners_dev, descriptors_dev);  // begins the exchange if nobody has
}

int orb_node_set_results(OrbNode* node, int where) {
    if (!node) return ORB_EINVAL;
    if (int rc = check_alive(node)) return rc;
    if (where != ORB_NODE_RESULTS_ROOT && where != ORB_NODE_RESULTS_SHARDED) return nfail(node, ORB_EINVAL, "results: ORB_NODE_RESULTS_ROOT or _SHARDED");
    if (!node->jobs.empty()) return nfail(node, ORB_ESTATE, "set_results with %zu jobs outstanding", node->jobs.size());
    if (where == ORB_NODE_RESULTS_SHARDED && node->d_shard_c[0].empty()) {
        // Allocate into local vectors and commit them only when every hipMalloc succeeded: a failed call must not leave the node with
        // half of its buffers (a retry would take "not empty" for "allocated" and pack through null pointers).
        const size_t pack = (size_t)node->max_batch * node->cfg.max_features;
        std::vector<CornerData*> sc[kCollSlots];
        std::vector<CornerDescriptor*> sd[kCollSlots];
        hipError_t bad = hipSuccess;
        for (int c = 0; c < kCollSlots && bad == hipSuccess; c++) {
            sc[c].assign(node->n, nullptr);
            sd[c].assign(node->n, nullptr);
            for (int r = 0; r < node->n && bad == hipSuccess; r++) {
                bad = hipSetDevice(node->devices[r]);
                if (bad == hipSuccess) bad = hipMalloc(&sc[c][r], pack * sizeof(CornerData));
                if (bad == hipSuccess) bad = hipMalloc(&sd[c][r], pack * sizeof(CornerDescriptor));
            }
        }
        if (bad != hipSuccess) {
            for (int c = 0; c < kCollSlots; c++)
                for (size_t r = 0; r < sc[c].size(); r++) {
                    (void)hipSetDevice(node->devices[r]);
                    if (sc[c][r]) (void)hipFree(sc[c][r]);
                    if (r < sd[c].size() && sd[c][r]) (void)hipFree(sd[c][r]);
                }
            return nfail(node, ORB_EHIP, "set_results: allocating the shard buffers failed: %s", hipGetErrorString(bad));
        }
        for (int c = 0; c < kCollSlots; c++) {
            node->d_shard_c[c] = std::move(sc[c]);
            node->d_shard_d[c] = std::move(sd[c]);
        }
    }
    node->sharded = where == ORB_NODE_RESULTS_SHARDED;
    node->collated = false;
    return ORB_OK;
}

int orb_node_shard_result(OrbNode* node, int rank, uint32_t* n_frames, uint64_t* n_records, void** corners_dev, void** descriptors_dev) {
    if (!node || rank < 0 || rank >= node->n) return ORB_EINVAL;
    if (!node->sharded) return nfail(node, ORB_ESTATE, "shard_result: the node collates on the first device (orb_node_set_results)");
    if (!node->collated) return nfail(node, ORB_ESTATE, "shard_result before collate_end");
    if (n_frames) *n_frames = node->last_shard_frames[rank];
    if (n_records) *n_records = node->last_shard_records[rank];
    if (corners_dev) *corners_dev = node->d_shard_c[node->last_coll][rank];
    if (descriptors_dev) *descriptors_dev = node->d_shard_d[node->last_coll][rank];
    return ORB_OK;
}

int orb_node_read_collated(OrbNode* node, CornerData* corners, CornerDescriptor* descriptors, size_t capacity) {
    if (!node) return ORB_EINVAL;
    if (!node->collated) return nfail(node, ORB_ESTATE, "read_collated before collate");
    if (node->sharded) {  // convenience for tests and small jobs: rank by rank, in frame order
        size_t at = 0;
        for (int r = 0; r < node->n && at < capacity; r++) {
            const size_t m = std::min<size_t>((size_t)node->last_shard_records[r], capacity - at);
            if (m == 0) continue;
            NODE_HIP(node, hipSetDevice(node->devices[r]));
            if (corners) NODE_HIP(node, hipMemcpy(corners + at, node->d_shard_c[node->last_coll][r], m * sizeof(CornerData), hipMemcpyDeviceToHost));
            if (descriptors)
                NODE_HIP(node, hipMemcpy(descriptors + at, node->d_shard_d[node->last_coll][r], m * sizeof(CornerDescriptor), hipMemcpyDeviceToHost));
            at += m;
        }
        return ORB_OK;
    }
    const size_t m = node->total_records < capacity ? (size_t)node->total_records : capacity;
    NODE_HIP(node, hipSetDevice(node->devices[0]));
    if (corners && m) NODE_HIP(node, hipMemcpy(corners, node->d_coll_c[node->last_coll], m * sizeof(CornerData), hipMemcpyDeviceToHost));
    if (descriptors && m)
        NODE_HIP(node, hipMemcpy(descriptors, node->d_coll_d[node->last_coll], m * sizeof(CornerDescriptor), hipMemcpyDeviceToHost));
    return ORB_OK;
}

}  // extern "C"
