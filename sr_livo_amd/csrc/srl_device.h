// srl_device.h -- internal device-side data layout and kernel launch interface (gfx950 only).
//
// Data layout in HBM:
//   * map slabs : one 256-byte, 256-byte-aligned record per voxel, in voxel CREATION order
//                 (slab index == voxel index, so point id = slab*20 + slot matches the reference's
//                 per-voxel push_back order, lioOptimization.cpp:413-443).  20 x (x,y,z) FP32 packed
//                 AoS (240 B) + count + key.  One wave streams a slab as 60 consecutive dwords.
//   * hash table: open addressing, power-of-two capacity >= 2 x voxels (load <= 0.5), 16-byte slots
//                 {packed int16x3 key, slab, count}; replaces tsl::robin_map<voxel, voxelBlock>
//                 (cloudMap.h:171) for find() only -- iteration order is never observed on the path.
//   * sweep     : raw points SoA  x[N] | y[N] | z[N]  FP64 (coalesced per-thread loads in phase 0).
//   * records   : per keypoint 8 doubles {J[6], distance, weight} + status byte (ordered cut-off path
//                 and parity taps).
//   * partials  : per workgroup 32 doubles (21 upper-tri HtH, 6 Hth, loss) + counts.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>


#define SRL_CAP 20
#define SRL_SLAB_BYTES 256
#define SRL_MAX_SLABS 16777215u      // slab * 256 + slot offset must fit 32 bits (kernels address slabs with 32-bit byte offsets)
#ifndef SRL_KPB
#define SRL_KPB 64            // keypoints per workgroup
#endif
#ifndef SRL_ASSOC_WAVES_PER_SIMD
#define SRL_ASSOC_WAVES_PER_SIMD 4
#endif
#define SRL_BLOCK 256         // threads per workgroup (4 waves)
#define SRL_SURV_CAP 64       // per-wave survivor scratch entries (general path; more survivors -> extraction)
#define SRL_WAVE_SCRATCH 2048  // bytes of LDS scratch per wave (fast path: 64 x 16 B records + 66 keys + 33 ranked d2 = 1816 B; heap replay 1024 B)
#define SRL_MAXK 32
#define SRL_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull
#define SRL_TABLE_FACTOR 4u   // hash slots per voxel capacity (load <= 0.25: a 2-slot probe almost always resolves)
#define SRL_PART_STRIDE 32
#define SRL_ROW_GRANULES 128  // published row of a workgroup (fused final reduction): 64 granules = 32 doubles, 8 = acceptance mask of <= 256 keypoints
#define SRL_FUSED_MAX_BLOCKS 2048    // workgroups of a fused pass (published rows): 256k keypoints in 256-keypoint workgroups = 1 024
#define SRL_FUSED_GROUP 256          // grids beyond 2 groups reduce in two levels: the last workgroup of every group of 256 sums its group's rows into a
#define SRL_FUSED_MAX_GROUPS (SRL_FUSED_MAX_BLOCKS / SRL_FUSED_GROUP)   // "super row" (stored behind the rows); the grid's last workgroup adds the super rows
#define SRL_FUSED_CUT_MAX_KPB 64   // fused ordered cut: the finisher re-reads one workgroup's records, one granule per thread

struct SrlMapSlot {
    unsigned long long key;
    unsigned slab;
    unsigned count;
};
static_assert(sizeof(SrlMapSlot) == 16, "slot must be 16 bytes");

struct SrlSlab {
    float xyz[SRL_CAP][3];
    unsigned count;
    unsigned pad;
    unsigned long long key;
};
static_assert(sizeof(SrlSlab) == SRL_SLAB_BYTES, "slab must be 256 bytes");

struct SrlBlockInfo {          // per-workgroup integer results
    int accepted;              // residuals accepted in this block
    unsigned sum_pk;           // candidates visited
    int nan_first;             // 1 + block-local index of the first keypoint whose planarity is NaN (optimize.cpp:348-350), 0 = none
    int num_fallback;
    int planes;                // keypoints with >= min_number_neighbors neighbours (the ones that reach the break test, optimize.cpp:107)
    int pad[3];
};
static_assert(sizeof(SrlBlockInfo) == 32, "block info is 32 bytes");

struct SrlDevOut {             // result of the reduce kernel (device, then copied to host)
    double HtH[36];
    double Hth[6];
    double loss;
    double d_num_res;          // counts carried as doubles so one all-reduce covers everything
    double d_total_accepted;
    double d_sum_pk;
    double d_nan;
    double d_fallback;
    double d_visited;          // keypoints visited by the sequential loop in this shard (summed by the all-reduce)
    double d_timeout;          // fused finisher: 1 when a workgroup's row never arrived (summed too: every rank repeats the pass together)
    long long last_visited;    // local index of last visited keypoint (n-1 if no cut)
    long long pad;
};

struct SrlMailbox {            // host-mapped (fine-grained) memory the reduce kernel publishes into (single rank)
    SrlDevOut out;             // plain form: the record ...
    unsigned long long seq;    // ... = launch sequence number once `out` is complete
    unsigned long long expired;   // sequence number of the last ARMED launch that gave up waiting for its pose (its own word: an armed
                                  // launch expires on its own schedule and must not touch the result a slower host has not read yet)
    unsigned long long pad[10];   // (g starts on a 64-byte line)
    // tagged form (the fused finisher of a single-context pass): word w of the record as granules g[2w], g[2w + 1] = {low 32 bits of the
    // sequence number, 32-bit half} -- "the data is the flag": no drain and no second PCIe write behind the data
    unsigned long long g[2 * (sizeof(SrlDevOut) / 8)];
};
static_assert(sizeof(SrlDevOut) == 52 * 8 && offsetof(SrlDevOut, pad) == 51 * 8, "SrlDevOut is 52 words, the marker word last");
static_assert(offsetof(SrlMailbox, g) % 64 == 0, "the tagged record starts on a cache line");

// ---- armed launches: the pose box
#define SRL_POSE_BOX_CTRL 42       // granule index of the control word {epoch, code | SRL_ARM_ALT}; granules 2d, 2d + 1 = halves of pose double d (Rn R t)
#define SRL_POSE_BOX_N 43          // granule {epoch, n}: keypoints of the pass (an armed launch can be fired for ANOTHER sweep: srl_sweep_swap)
#define SRL_POSE_BOX_TLAST 44      // granules 44..49: halves of t_last[3] (optimize.cpp:25: the previous frame's translation -- per sweep, like the pose)
#define SRL_POSE_BOX_USED 50       // granules a launch waits for
#define SRL_POSE_BOX_WRITTEN 56    // granules the host writes (whole 64-byte lines: seven)
#define SRL_POSE_BOX_GRANULES 64   // allocated
#define SRL_ARM_GO 1u
#define SRL_ARM_CANCEL 2u
#define SRL_ARM_EXPIRED 3u
#define SRL_ARM_CODE_MASK 3u
#define SRL_ARM_ALT 4u             // flag beside SRL_ARM_GO: the pass runs on the launch's ALTERNATE sweep buffer (alt_x / alt_y / alt_z)

#define SRL_REDUCED_DOUBLES 50 // leading doubles of SrlDevOut that are summed over the shards (HtH .. d_timeout)

// ---- direct peer exchange of the sharded sum (srl_peer_attach): no RCCL call on the data path ---------------------------
// Every rank owns an INBOX in fine-grained device memory that its peers can store into (same process: the raw pointer; other
// processes: a HIP IPC mapping).  An exchange = every rank stores its row, as tagged 8-byte granules {epoch, 32-bit half}
// ("the data is the flag", like the rows of the fused reduction), into the inbox of EVERY rank including itself, then polls
// its own inbox for the rows of all ranks and adds them in rank order -- the same bits on every rank, one xGMI store hop.
// Slots alternate with the exchange counter: a rank can only be one exchange ahead of a peer (it needs the peer's row of
// exchange e + 1, which the peer sends after finishing e), so two slots never collide.
#define SRL_MAX_PEERS 8
#define SRL_PEER_ROW 64        // doubles per row (>= SRL_REDUCED_DOUBLES; a row = 2 x 64 granules: low halves, then high halves)
#define SRL_PEER_INBOX_GRANULES (2 * SRL_MAX_PEERS * 2 * SRL_PEER_ROW)   // [slot][source rank][half][lane]
#define SRL_PEER_TIMEOUT_MARK 0x9EE9ll   // SrlDevOut::pad when a PEER's row never arrived (0x7117: a workgroup's row of the own launch)
struct SrlPeerTable {          // device memory, written once at srl_peer_attach
    unsigned long long *inbox[SRL_MAX_PEERS];   // inbox of rank r as mapped into this process
    int nranks, rank;
};

struct SrlAssocArgs {
    // sweep
    const double *raw_x, *raw_y, *raw_z;
    // A prefetched sweep arrives AoS (n x 3) by DMA alone -- no kernel on the copy stream: while the association kernels hold every
    // compute unit a transpose kernel could not start before the solve it is meant to overlap had ended.  The first pass over such a
    // sweep reads its points from `aos` and writes the SoA planes raw_x / raw_y / raw_z as it goes (every later pass reads the planes);
    // null: the planes are valid.
    const double *aos;
    int n;
    // map
    const SrlMapSlot *table;
    unsigned table_mask;
    const unsigned char *slabs;
    int write_rec;                     // per-keypoint records + status are needed (ordered cut-off can trigger, or taps); else skipped
    unsigned inf_off;                  // byte offset of the slab whose 20 points are (+inf, +inf, +inf): lanes without a candidate load from it
    // pose (computed on the host exactly like the reference: optimize.cpp:35 and :95)
    double Rn[9];       // end_quat.normalized().toRotationMatrix()
    double R[9];        // end_quat.toRotationMatrix()
    double t[3];
    double t_last[3];
    double R_il[9];
    double t_il[3];
    // options
    double size_voxel;
    double max_dist;
    double lambda_w, lambda_n;
    double power_planarity;
    double nbr_scale;   // max_dist * min_number_neighbors (optimize.cpp:88)
    int K;              // max_number_neighbors
    int min_nb;         // min_number_neighbors
    int thr_cap;        // threshold_voxel_capacity
    int select_mode;
    int ablate;             // debug only (env SRL_ABLATE): bit0 skip phase 2, bit1 stop after compaction, bit2 stop after probe, bit3 skip probe
    // fused final reduction (single rank, no ordered cut possible, no taps): the last workgroup to finish sums the block
    // partials and publishes the result itself -- no second kernel, no kernel boundary on the per-iteration critical path
    unsigned long long *granules;   // nblocks x SRL_ROW_GRANULES tagged 8-byte granules {epoch, 32-bit payload}: the published rows (null = not fused)
    unsigned long long *rec_granules;   // fused ORDERED CUT: per keypoint 16 tagged granules = the record {J[6], distance, weight} (else null)
    long long cut_max;              // fused ordered cut: max_num_residuals (> 0), the sequential loop's budget (optimize.cpp:107); 0 = no cut possible
    SrlMailbox *mailbox;        // host-mapped result mailbox (fused + RCCL: a device-side mailbox the all-reduce then works on)
    int mail_tagged;            // 1: the finisher reports in the mailbox's tagged form (host mailbox), 0: plain form + sequence word
    int pad_mail;
    unsigned long long seq;     // launch sequence number published with the result
    // ARMED launch (enqueued before its pose exists; null = the pose is Rn / R / t above): see assoc_body's prologue
    const unsigned long long *pose_box;   // tagged granules the host writes: 2 x 21 pose halves + the control granule + the keypoint count
    const double *alt_x, *alt_y, *alt_z;  // the context's OTHER sweep buffer (srl_sweep_prefetch / srl_sweep_swap): an armed launch fired with
                                          // SRL_ARM_ALT runs on it -- the first pass of the next sweep without a launch on its critical path
    const double *alt_aos;                // ... whose points still lie AoS (n x 3) in the prefetch's staging buffer: that pass reads them
                                          // there and files the SoA planes alt_x / alt_y / alt_z itself (see `aos`)
    unsigned long long *pose_relay;       // device memory: pose_relayed != 0 -- workgroup 0 republishes the box into it for the others; else everybody
                                          // polls the box and only a launch that gave up waiting leaves a note here (workgroups of later rounds leave at once)
    int pose_relayed;
    int pad_relay;
    unsigned pose_epoch;                  // tag of THIS launch's pose (low 32 bits of its sequence number, never 0)
    unsigned arm_linger_ticks;            // 100 MHz ticks an armed launch waits at most (safety net)
    long long *stamps;                    // debug time line of armed passes (host-mapped, 64 rows x 16 slots; null = off)
    const SrlPeerTable *peer;   // fused + direct peer exchange: the finishing workgroup exchanges its totals itself (else null)
    unsigned peer_epoch;        // tag of this exchange (exchange counter, never 0)
    int peer_slot;              // exchange counter & 1
    // NEIGHBOURHOOD BOUNDS (round 6): what every keypoint of the previous pass over THIS sweep and THIS map learnt -- its world position and
    // the exact squared distance of its K-th nearest neighbour -- as {x, y, z, tau} floats.  The K points found then are still in the map, so
    // the K nearest of the next pass lie within r = sqrt(tau) + |p_w - p_w_prev| of the new position: voxels whose box is further away than
    // r are not visited (phase 1, probe_finish) -- same neighbours, same bits, fewer candidate rounds.  bound_use: entries [0, bound_use)
    // may be read (0 on the first pass over a sweep, after a map change or other options); bound_out is written by every pass.
    const float *bound_in;
    float *bound_out;
    int bound_use;
    int pad_bound;
    // outputs
    double *rec;            // n x 8
    unsigned char *status;  // n
    double *partials;       // nblocks x 32
    SrlBlockInfo *binfo;    // nblocks
    // parity taps (may be null)
    int *tap_ids;           // n x K
    int *tap_ncand;         // n
    double *tap_normal;     // n x 3
    double *tap_a2d;        // n
    double *tap_offset;     // n
};

#define SRL_POSE_DOUBLES 24    // armed launches: the LDS pose block Rn[9] R[9] t[3] t_last[3], the control words behind it
static_assert(sizeof(SrlAssocArgs) <= 4096, "the struct travels in the kernarg segment");

struct SrlReduceArgs {
    const double *rec;
    const unsigned char *status;
    const double *partials;
    const SrlBlockInfo *binfo;
    int n;
    int nblocks;
    long long max_res;          // residual budget for THIS rank (already reduced by earlier ranks' counts) ...
    const long long *gather;    // ... or, when non-null: per-rank counts gathered on the stream (accepted residuals, or keypoints with a
    int rank;                   //     plane when max_num_residuals <= 0); the kernel derives budget and mode itself from
    int max_num_residuals;      //     max_num_residuals and the counts of ranks < rank -- no host synchronisation
    SrlDevOut *out;             // device result (multi-rank: all-reduced afterwards) ...
    SrlMailbox *mailbox;        // ... or, single rank: host-mapped mailbox written with system-scope stores (no memcpy)
    unsigned long long seq;
    int kpb;                    // keypoints per workgroup of the association pass that produced the partials
};

struct SrlSearchArgs {
    const double *q;        // n x 3 world points (AoS, device)
    int n;
    const SrlMapSlot *table;
    unsigned table_mask;
    const unsigned char *slabs;
    double size_voxel;
    int K;
    int thr_cap;
    int select_mode;
    int *ids;               // n x K
    float *nb_xyz;          // n x K x 3 or null
    int *num_found;         // n
};

// launchers (srl_kernels.hip)
hipError_t srl_launch_assoc(const SrlAssocArgs &a, int nb_voxels, int kpw, int wpb, hipStream_t s);
int srl_assoc_lds_bytes(int K, int nb_voxels, int kpw, int wpb);
// keypoints per wave for a pass over n keypoints: the largest of 4 / 8 / 16 that still yields >= ~4096 waves
static inline int srl_keypoints_per_wave(int n) { return n <= 16384 ? 4 : (n <= 32768 ? 8 : 16); }
// ... and for 16-wave workgroups that fuse the final reduction: the smallest instantiated count that still puts the sweep on
// the chip in ONE round of workgroups (one 16-wave workgroup per CU): a wave's serial chain is as short as the sweep allows
static inline int srl_keypoints_per_wave_one_round(int n, int num_cu) {
    const int set[7] = {2, 3, 4, 6, 8, 12, 16};
    for (int i = 0; i < 7; ++i) if ((long long)16 * set[i] * num_cu >= n) return set[i];
    return 16;
}
#define SRL_LDS_LIMIT (160 * 1024)
hipError_t srl_launch_reduce(const SrlReduceArgs &a, int mode, hipStream_t s);
hipError_t srl_launch_count(const SrlBlockInfo *binfo, int nblocks, int count_planes, long long *out_total, hipStream_t s);
hipError_t srl_launch_publish(const SrlDevOut *src, SrlMailbox *mb, unsigned long long seq, hipStream_t s);
// direct peer exchange as its own one-wave kernel (the pass was not fused, or the ordered cut needs the per-rank counts first):
// rows: the leading SRL_REDUCED_DOUBLES of *src summed over the ranks into the host mailbox (+ this rank's last_visited);
// counts: *count of every rank into gather_out[rank] (device memory the reduce kernel reads)
hipError_t srl_launch_peer_rows(const SrlPeerTable *peer, unsigned epoch, int slot, const SrlDevOut *src, SrlMailbox *mb, unsigned long long seq,
                                const long long *gather_check, hipStream_t s);
hipError_t srl_launch_peer_counts(const SrlPeerTable *peer, unsigned epoch, int slot, const long long *count, long long *gather_out, hipStream_t s);
hipError_t srl_launch_search(const SrlSearchArgs &a, int nb_voxels, hipStream_t s);
struct SrlXform { double R[9], t[3], R_il[9], t_il[3]; };
hipError_t srl_launch_transform(const double *raw_aos, int n, const SrlXform &X, double *out_aos, hipStream_t s);
hipError_t srl_launch_aos_to_soa(const double *aos, int n, double *x, double *y, double *z, hipStream_t s);
hipError_t srl_launch_sqrt(double *io, int n, hipStream_t s);   // debug: the device's sqrt(double), element-wise
