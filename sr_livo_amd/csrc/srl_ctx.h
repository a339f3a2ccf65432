// srl_ctx.h -- internal: the context object behind the opaque srl_ctx handle of include/srlivo_hip.h
#pragma once
#include "../../include/srlivo_hip.h"
#include "srl_device.h"
#include "host/tr1_relation.h"

#include <hip/hip_runtime.h>
#include "srl_rccl.h"

#include <string>
#include <vector>

// scratch hash table of the per-frame kernels that is never cleared between frames (srl_frame_scratch.h)
struct SrlEpochTable {
    unsigned long long *keyw = nullptr;       // (epoch16 << 48) | key48
    unsigned long long *minw = nullptr;       // optional companion: {~frame counter, smallest index}
    unsigned cap = 0;                         // slots allocated (power of two)
    unsigned epoch16 = 0;
    unsigned counter32 = 0;
};

// The word behind a rank's inbox rows that any rank sets when it gives a session up (srl_peer_set_deadline_ms): a rank whose pass times
// out looks at its own inbox's word and ends the session at once instead of waiting for its own deadline.
#define SRL_PEER_POISON_WORD SRL_PEER_INBOX_GRANULES
// ... and the word behind it: an identity of the DEVICE the inbox lives on (hash of its PCI bus id, never 0), written at srl_peer_export.
// srl_peer_attach compares the peers' words with its own: ranks that share a device (a test arrangement -- production is one process per
// GPU) do not arm launches, whose waiting workgroups would hold the compute units the other rank's kernel needs.
#define SRL_PEER_DEVICE_WORD (SRL_PEER_INBOX_GRANULES + 1)
#define SRL_PEER_INBOX_ALLOC_GRANULES (SRL_PEER_INBOX_GRANULES + 8)

struct srl_ctx {
    int device = 0;
    bool counted = false;              // this context is in the per-device census (srl_capi.cpp: g_live_ctx)
    hipStream_t stream = nullptr;
    std::string err;

    // map
    SrlMapSlot *d_table = nullptr;
    unsigned table_cap = 0;            // slots (power of two)
    unsigned char *d_slabs = nullptr;
    unsigned slab_cap = 0;             // slabs allocated
    int num_voxels = 0;
    long long num_points = 0;

    // sweep (this rank's shard)
    double *d_raw = nullptr;           // SoA x|y|z, stride sweep_cap
    int sweep_cap = 0;
    int n = 0;                         // shard size
    int shard_begin = 0;
    int total_n = 0;
    bool sweep_loaded = false;         // a sweep (possibly empty) has been uploaded / selected
    int search_select_mode = 0;        // srl_debug_set_search_select_mode: selection path of srl_search_neighbors (tests)
    int select_mode = 0;               // srl_debug_set_select_mode: selection path of srl_build_residuals (tests; 0 = automatic)
    int ablate = 0;                    // srl_debug_set_ablate (profiling tools only; never set by the product)
    void (*overlap_fn)(void *) = nullptr;   // srl_build_residuals_overlap: host work to run while the kernels are in flight
    void *overlap_user = nullptr;

    // the NEXT sweep (srl_sweep_prefetch / srl_sweep_swap): uploaded on its own stream while the current one is solved
    double *d_raw_next = nullptr;      // SoA, stride next_cap
    double *d_stage_next = nullptr;    // AoS staging of the prefetch: the DMA's target (no kernel runs on the copy stream)
    double *d_stage_cur = nullptr;     // ... of the sweep that is current now (swapped with d_stage_next by srl_sweep_swap): its points are
                                       // transposed into d_raw by the first pass that touches them (SrlAssocArgs::aos)
    int soa_valid_n = 0;               // leading keypoints of the current sweep whose SoA planes in d_raw are filled
    int next_cap = 0, stage_next_cap = 0, stage_cur_cap = 0;   // capacities (points) of d_raw_next / d_stage_next / d_stage_cur
    int next_n = -1, next_begin = 0, next_total = 0;   // next_n < 0: nothing prefetched
    hipStream_t copy_stream = nullptr;
    hipStream_t prefix_stream = nullptr;  // the PREFIX of a prefix-first prefetch: a stream (and with it a DMA engine) of its own, beside the tail
    int num_cu = 256;                 // compute units of the device (launch-shape policy)
    hipEvent_t next_ready = nullptr;      // the prefetched sweep has landed completely (the slot's FULL event: up_ev[next_slot][1])
    // PREFIX-FIRST upload.  With a finite max_num_residuals a solve visits only the first few thousand keypoints of
    // a sweep (srl_build_residuals: the prefix pass), yet the whole sweep had to cross PCIe before its first pass could fire.  The prefetch
    // therefore goes out as TWO DMAs -- the prefix the running solve's passes visit (prefix_hint) and the rest -- each with its own event and
    // on a stream of its own (chained on one stream the two copies and their markers took ~63 us per 1.5 MB sweep: every hop between a DMA
    // engine and the stream's queue costs; side by side the prefix lands ~11 us after the call and the tail in the ~36 us one copy takes);
    // two event pairs alternating between consecutive prefetches.  srl_sweep_swap keeps a waiting launch once the PREFIX has landed; a
    // pass that needs more (the whole-shard repeat, other options, anything launched normally) first orders the stream behind the FULL event.
    hipEvent_t up_ev[2][2] = {};          // [slot][0 = prefix landed, 1 = everything landed]
    bool up_one_dma[2] = {true, true};    // the slot's upload went out as ONE DMA: only its full event was recorded
    int next_slot = 0;                    // slot of the prefetch in flight / last issued
    int next_prefix_n = 0;                // points of it covered by the prefix event (== next_n: one DMA)
    int prefix_hint = 0;                  // keypoints the passes of the running solve visit when that is a prefix of the shard (0: whole sweeps)
    // neighbourhood bounds (SrlAssocArgs::bound_in): 4 floats per keypoint, written by every pass; entries [0, bound_n) were written on the
    // CURRENT sweep and map with the options of bound_sig -- reset by whatever changes one of them
    float *d_bound = nullptr;
    int bound_n = 0;
    int bound_sig[4] = {0, 0, 0, 0};      // K, voxel neighbourhood, occupancy threshold, (float) voxel size bits
    int bound_mode = 1;                   // srl_debug_set_bound_culling: 0 = never use them
    bool tail_pending = false;            // the CURRENT sweep's tail may still be in flight: stream not yet ordered behind up_ev[cur_slot][1]
    int cur_slot = 0, cur_prefix_n = 0;   // ... its slot, and how many of its points are known to have landed

    // frame-resident pipeline (srl_frame_*)
    double *d_frame_raw = nullptr;     // AoS n x 3
    double *d_frame_world = nullptr;   // AoS n x 3
    int frame_cap = 0;
    int frame_n = -1;
    int frame_world_n = -1;            // points of d_frame_world as the last srl_frame_commit left them (-1: none, or a newer frame uploaded since)
    // undistorted sweep (srl_frame_undistort): inputs, imu_point, corrected raw_point
    double *d_corr_in = nullptr, *d_corr_rel = nullptr, *d_corr_imu = nullptr, *d_corr_raw = nullptr;
    int *d_corr_seg = nullptr;
    int corr_cap = 0;
    int corr_n = -1;

    // srl_debug_frame_timing: per-stage wall time of the frame pipeline (the stream is synchronised at every stage boundary while on)
    bool frame_timing = false;
    double frame_stage_us[16] = {};
    long long frame_stage_last_ns = 0;

    // page-locked sources: event behind the last DMA that read a caller's buffer (srl_sweep_wait)
    hipEvent_t upload_ev = nullptr;
    bool upload_pending = false;

    // pinned staging ring of srl_sweep_upload (pageable sources): CPU copy of one chunk overlaps the DMA of the previous
    static constexpr int RING_SLOTS = 4;
    static constexpr int RING_SLOT_BYTES = 256 * 1024;
    char *h_ring = nullptr;
    hipEvent_t ring_ev[RING_SLOTS] = {};
    bool ring_busy[RING_SLOTS] = {};
    unsigned ring_next = 0;

    // pinned host scratch (grow-only) for the small D2H / H2D hops of the frame pipeline: pageable copies are staged
    // synchronously by the runtime and cost more than the kernels around them
    char *h_scratch = nullptr;
    size_t h_scratch_bytes = 0;
    // frame pipeline <-> host exchange (page-locked, coherent, device-mapped): the selection kernels write the voxel list straight into
    // it and the gather reads the ordered index list out of it -- no copy commands, the host waits on a tagged word, not on the stream
    char *h_frame_x = nullptr;
    size_t h_frame_x_bytes = 0;
    unsigned *d_frame_sync = nullptr;            // [0]: blocks of the emitting kernel that are done (reset by the last one)
    unsigned frame_tag = 0;                      // tag of the last exchange (never 0)
    // keypoint order on the device (srl_frame_kernels.hip): the growth schedule of std::tr1::unordered_map, recorded from a real container
    SrlTr1Sched *d_tr1_sched = nullptr;
    int tr1_steps = -1;                          // steps of the uploaded schedule (-1: not uploaded)
    unsigned tr1_first[SRL_TR1_MAX_STEPS] = {0}, tr1_nb[SRL_TR1_MAX_STEPS + 1] = {0};
    int frame_order_mode = 0;                    // srl_debug_set_frame_order_mode: 0 = device order where it applies, 1 = the host replay (tests)
    int frame_order_used = 0;                    // what the last selection did: 1 = device order, 2 = host replay, 3 = device order overflowed -> host replay
    // addPointsToMap on the device: counters of the last insert, folded into num_voxels / num_points lazily (srl_map_settle)
    int *h_insert_cnt = nullptr;                 // pinned: [0] segments, [1] new voxels, [2] points added
    hipEvent_t ev_insert = nullptr, ev_world = nullptr;
    bool insert_pending = false;
    hipEvent_t ev_frame_read = nullptr;          // main stream: the last kernel reading d_frame_raw (the next upload, on the copy stream, waits for it)
    SrlEpochTable sel_table, ins_table;          // keypoint selection (with first-index words) / frame insertion

    // work buffers
    double *d_rec = nullptr;
    unsigned char *d_status = nullptr;
    double *d_partials = nullptr;
    SrlBlockInfo *d_binfo = nullptr;
    int work_cap = 0;                  // keypoints capacity of rec/status
    int block_cap = 0;
    SrlDevOut *d_out = nullptr;
    SrlDevOut *h_out = nullptr;        // pinned
    long long *d_count = nullptr;
    long long *h_count = nullptr;      // pinned
    unsigned long long *d_rec_granules = nullptr;   // fused ordered cut: 16 tagged granules per keypoint (the record), grown on demand
    size_t rec_granule_cap = 0;
    unsigned long long *d_granules = nullptr;   // published rows of the fused final reduction: 512 workgroups x 64 granules
    int force_kpw = 0, force_wpb = 0;  // srl_debug_set_launch_shape (0 = automatic)
    bool fuse_reduce = true;           // srl_debug_set_fused_reduce(0): always run the separate reduce kernel (A/B, tests)
    SrlMailbox *h_mail = nullptr;      // host-mapped fine-grained mailbox the reduce kernel publishes into
    unsigned long long seq = 0;

    // ARMED launches (srl_capi.cpp: arm_next / pose_box_write / srl_ctx_disarm): the kernel of the NEXT pass is enqueued while the
    // current one runs and waits, resident, for its pose
    int arm_mode = 1;                           // srl_set_armed_launch: 0 off, 1 on (armed where it is likely to fire: arm_wanted in srl_capi.cpp), 2 always
    int passes_in_solve = 0;                    // srl_build_residuals calls since the last srl_solve_end / sweep change
    int expected_passes = 0;                    // passes the previous solve took (0: unknown): no launch is armed behind the pass expected to be the
                                                // last one unless a prefetched sweep is waiting (the armed launch then becomes ITS first pass)
    int pose_box_kind = -1;                     // -1: not chosen yet (1 where possible); 0: pinned host memory, workgroup 0 relays into device memory; 1: fine-grained device memory the host writes through the PCIe BAR
    bool pose_box_dev_visible = false;
    unsigned long long *h_pose_box = nullptr;   // where the HOST writes the pose granules (kind 1: the CPU-visible device pointer)
    unsigned long long *pose_box_pinned = nullptr;   // kind 0 allocation (hipHostMalloc)
    unsigned long long *pose_box_dev = nullptr;      // kind 1 allocation (hipExtMallocWithFlags, fine-grained)
    unsigned long long *d_pose_relay = nullptr;
    bool armed = false;
    SrlAssocArgs armed_sig;                     // the armed launch's arguments with the pose zeroed: a pass must equal them to fire it
    int armed_nb = 0, armed_kpw = 0;
    int armed_nblocks = 0;                      // grid of the armed launch (a pass over fewer keypoints may fire it: the surplus workgroups find empty tiles)
    const double *armed_raw = nullptr, *armed_alt = nullptr;   // x planes of the sweep buffer the launch was armed on / of the context's other buffer
    const double *armed_alt_aos = nullptr;      // the staging buffer the launch reads when it is fired for the other buffer
    int armed_raw_cap = 0, armed_alt_cap = 0;   // ... and their strides (capacity in points)
    int armed_ring = -1;                        // light profiling: ring slot of the armed launch's event pair (-1: none)
    long long armed_at_ns = 0;                  // steady clock at arm time
    double arm_host_linger_us = 150.0;          // an armed launch older than this is cancelled, never fired (the kernel's own bound is longer)
    unsigned arm_linger_ticks = 30000u;         // 300 us of the 100 MHz clock: after that a waiting launch leaves by itself (a device-wide
                                                // synchronisation from outside this library waits that long at most)
    unsigned long long arm_stats[4] = {0, 0, 0, 0};   // armed, fired, cancelled, expired
    bool ring_void[512] = {};
    bool ring_marker[512] = {};                 // light profiling with a period: an entry that only serves as the NEXT launch's start (not timed itself)
    int prof_period = 1;                        // srl_set_profiling_period: the light profiling times every prof_period-th association launch
    unsigned long long prof_count = 0;          // association launches seen by the light profiling (own and armed)
    long long ring_last_count = -2;             // prof_count of the launch behind the newest ring entry
    bool armed_measured = false, cur_measured = true;     // is the armed launch / the launch of the pass now running one of the timed ones?
    // srl_timing_mark: "the sums start here" without a read-back -- launches enqueued before the mark carry an older generation and are
    // left out when their events are read (their passes' bytes likewise)
    unsigned timing_gen = 0, armed_gen = 0, cur_gen = 0;
    unsigned ring_gen[512] = {};
    int ring_prev[512] = {};                    // light profiling: ring slot whose END event is this launch's start (armed launches), -1 = own start event
    long long *h_arm_stamps = nullptr;          // srl_debug_pass_stamps: 64 rows x 16 slots the armed kernels file (host-mapped)
    long long arm_host_stamps[64][4] = {};      // per pass (row seq & 63), steady-clock ns: call entry, pose written / launch returned, result seen, fired?                   // light profiling: event pairs of cancelled armed launches (not counted)

    // taps
    bool taps = false;
    int tap_cap = 0, tap_K = 0;
    int *d_tap_ids = nullptr;
    int *d_tap_ncand = nullptr;
    double *d_tap_normal = nullptr, *d_tap_a2d = nullptr, *d_tap_offset = nullptr;
    bool taps_valid = false;
    int last_K = 0;
    long long last_visited_local = -1;

    // comm
    int nranks = 1, rank = 0;
    ncclComm_t comm = nullptr;
    ncclComm_t parked_comm = nullptr;  // srl_comm_suspend: the communicator set aside while the context runs unsharded
    int parked_nranks = 0, parked_rank = 0;
    bool force_coll = false;           // env SRL_FORCE_COLLECTIVES=1: run the RCCL calls even with one rank (test hook)
    srl_allreduce_fn cb_ar = nullptr;
    srl_allgather_i64_fn cb_ag = nullptr;
    void *cb_user = nullptr;
    long long *d_gather = nullptr;     // nranks (capacity d_gather_cap: grow-only, see ensure_gather)
    size_t d_gather_cap = 0;
    // direct peer exchange (srl_peer_export / srl_peer_attach): the sharded sum without an RCCL call on the data path
    unsigned long long *d_inbox = nullptr;     // fine-grained device memory the peers store into (SRL_PEER_INBOX_GRANULES)
    SrlPeerTable *d_peer = nullptr;            // the table the kernels read (device copy)
    void *peer_mapped[SRL_MAX_PEERS] = {};     // HIP IPC mappings of the other ranks' inboxes (closed at detach)
    bool peer_on = false;
    int peer_seen = 0;                 // inboxes mapped at srl_peer_attach (srl_comm_info: equal to nranks when every rank's handle opened)
    bool peer_shares_device = false;   // a peer rank's inbox lives on THIS device (SRL_PEER_DEVICE_WORD): launches are not armed (arm_mode 1)
    bool comm_shares_device = false;   // ... the same for the RCCL transport (device identities all-gathered at srl_comm_init_rank)
    bool peer_failed = false;          // a row of this session never arrived: no further exchange until srl_peer_export + srl_peer_attach
    unsigned long long *peer_inbox[SRL_MAX_PEERS] = {};     // every rank's inbox as mapped into this process (host copy of SrlPeerTable::inbox)
    int peer_deadline_ms = 10000;      // srl_peer_set_deadline_ms: how long a pass keeps re-polling for a late rank's row before the session is given up
    unsigned long long peer_retries = 0;       // passes repeated because a row had not arrived within one kernel's bounded spin (srl_comm_info)
    unsigned long long peer_seq = 0;           // exchange counter: advances in lock-step on every rank
    SrlMailbox *d_mail = nullptr;              // device-side mailbox: where a FUSED pass leaves its rank's result for the RCCL all-reduce
    bool dbg_gather = false;           // srl_debug_set_gather_counts: d_gather is preloaded (no all-gather)

    // timing
    // scratch pool: device blocks handed out to the map-insert / frame pipeline calls and kept for the next call
    // (hipMalloc / hipFree per call cost more than the kernels they serve; hipFree also synchronises the device)
    struct PoolBlock { void *p; size_t bytes; };
    std::vector<PoolBlock> pool_free;

    int last_nblocks = 0;
    int profiling = 0;                 // 0 off, 1 full (4 events + sync per call), 2 light (assoc kernel only, read lazily)
    static constexpr int PROF_RING = 512;       // (= sizeof ring_void)
    hipEvent_t ring[PROF_RING][2] = {};
    unsigned ring_head = 0, ring_tail = 0;   // [tail, head) recorded and not yet read

    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    srl_timing timing = {};
    int last_nb = 1;
};

#define HIPCHK(ctx, call)                                                                      \
    do {                                                                                       \
        hipError_t e__ = (call);                                                               \
        if (e__ != hipSuccess) {                                                               \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);                   \
            return SRL_ERR_HIP;                                                                \
        }                                                                                      \
    } while (0)

// RCCL calls go through the run-time table: NCCLCHK(ctx, AllReduce(...)) = srl_rccl()->AllReduce(...)
#define NCCLCHK(ctx, call)                                                                     \
    do {                                                                                       \
        const SrlRccl *rc__ = srl_rccl();                                                      \
        if (!rc__) { (ctx)->err = srl_rccl_error(); return SRL_ERR_COMM; }                     \
        ncclResult_t r__ = rc__->call;                                                         \
        if (r__ != ncclSuccess) {                                                              \
            (ctx)->err = std::string(#call) + ": " + rc__->GetErrorString(r__);                \
            return SRL_ERR_COMM;                                                               \
        }                                                                                      \
    } while (0)

// cancel an armed launch, if any (every entry point that touches the stream or waits for it calls this first: an armed launch
// that nobody fires would hold the stream until its bound)
extern "C" int srl_ctx_disarm(srl_ctx *ctx);
#define SRL_DISARM(ctx) do { if ((ctx)->armed) { int rcd__ = srl_ctx_disarm(ctx); if (rcd__) return rcd__; } } while (0)

// frame pipeline stage stamps (only while srl_debug_frame_timing is on): close stage `slot` here
#include <chrono>
inline void srl_stage_begin(srl_ctx *ctx) {
    if (!ctx->frame_timing) return;
    hipStreamSynchronize(ctx->stream);
    ctx->frame_stage_last_ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
inline void srl_stage_end(srl_ctx *ctx, int slot) {
    if (!ctx->frame_timing) return;
    hipStreamSynchronize(ctx->stream);
    const long long now = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
    ctx->frame_stage_us[slot] += (double)(now - ctx->frame_stage_last_ns) * 1e-3;
    ctx->frame_stage_last_ns = now;
}

// fold the counters of a deferred insert into the map's totals (every reader of num_voxels / num_points calls this first)
inline int srl_map_settle(srl_ctx *ctx) {
    if (!ctx->insert_pending) return SRL_OK;
    HIPCHK(ctx, hipEventSynchronize(ctx->ev_insert));
    ctx->num_voxels += ctx->h_insert_cnt[1];
    ctx->num_points += ctx->h_insert_cnt[2];
    ctx->insert_pending = false;
    return SRL_OK;
}

inline int ensure_copy_stream(srl_ctx *ctx) {
    if (!ctx->copy_stream) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    return SRL_OK;
}
// the frame's raw points have just been read for the last time by what is enqueued so far
inline int srl_mark_frame_read(srl_ctx *ctx) {
    if (!ctx->ev_frame_read) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_frame_read, hipEventDisableTiming));
    HIPCHK(ctx, hipEventRecord(ctx->ev_frame_read, ctx->stream));
    return SRL_OK;
}

inline int ensure_frame_exchange(srl_ctx *ctx, size_t bytes) {
    if (!ctx->d_frame_sync) {
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_frame_sync, 256));
        HIPCHK(ctx, hipMemsetAsync(ctx->d_frame_sync, 0, 256, ctx->stream));
    }
    if (bytes <= ctx->h_frame_x_bytes) return SRL_OK;
    if (ctx->h_frame_x) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));      // a gather may still be reading its index list out of the old block
        HIPCHK(ctx, hipHostFree(ctx->h_frame_x)); ctx->h_frame_x = nullptr; ctx->h_frame_x_bytes = 0;
    }
    const size_t cap = ((bytes + bytes / 2 + 4095) / 4096) * 4096;
    HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_frame_x, cap, hipHostMallocCoherent | hipHostMallocMapped));
    ctx->h_frame_x_bytes = cap;
    return SRL_OK;
}

inline int ensure_host_scratch(srl_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->h_scratch_bytes) return SRL_OK;
    if (ctx->h_scratch) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));      // a DMA out of the old block may still be queued (srl_frame_select_keypoints' index list)
        HIPCHK(ctx, hipHostFree(ctx->h_scratch)); ctx->h_scratch = nullptr; ctx->h_scratch_bytes = 0;
    }
    const size_t cap = ((bytes + bytes / 2 + 4095) / 4096) * 4096;
    HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_scratch, cap, hipHostMallocDefault));
    ctx->h_scratch_bytes = cap;
    return SRL_OK;
}

template <class T>
int ensure(srl_ctx *ctx, T *&p, size_t count) {
    if (p) { HIPCHK(ctx, hipFree(p)); p = nullptr; }
    if (count == 0) count = 1;
    HIPCHK(ctx, hipMalloc((void **)&p, count * sizeof(T)));
    return SRL_OK;
}

// The per-rank counts of the ordered cut.  Grow-only and never re-allocated for a size that fits: hipFree waits for the WHOLE
// device, and a second rank of the same process may already sit in a pass that waits for THIS rank's row (a second peer
// session on live contexts: its attach used to free + re-allocate here and stalled until the peer's bounded spin gave up).
inline int ensure_gather(srl_ctx *ctx, size_t nranks) {
    if (ctx->d_gather && nranks <= ctx->d_gather_cap) return SRL_OK;
    const size_t cap = nranks < 64 ? 64 : nranks;
    int rc = ensure(ctx, ctx->d_gather, cap);
    if (rc) { ctx->d_gather_cap = 0; return rc; }
    ctx->d_gather_cap = cap;
    return SRL_OK;
}

// RAII scratch block from the context's pool (best fit, grow-only; everything is released by srl_ctx_destroy)
struct DevBuf {
    srl_ctx *ctx = nullptr;
    void *p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { if (p && ctx) ctx->pool_free.push_back({p, bytes}); else if (p) hipFree(p); }
    hipError_t alloc(srl_ctx *c, size_t want) {
        ctx = c;
        if (want < 256) want = 256;
        int best = -1;
        for (int i = 0; i < (int)c->pool_free.size(); i++)
            if (c->pool_free[i].bytes >= want && (best < 0 || c->pool_free[i].bytes < c->pool_free[best].bytes)) best = i;
        if (best >= 0 && c->pool_free[best].bytes <= 4 * want + (1u << 20)) {
            p = c->pool_free[best].p; bytes = c->pool_free[best].bytes;
            c->pool_free.erase(c->pool_free.begin() + best);
            return hipSuccess;
        }
        const size_t cap = ((want + want / 4 + 255) / 256) * 256;
        hipError_t e = hipMalloc(&p, cap);
        bytes = (e == hipSuccess) ? cap : 0;
        return e;
    }
    template <class T> T *as() { return reinterpret_cast<T *>(p); }
};
