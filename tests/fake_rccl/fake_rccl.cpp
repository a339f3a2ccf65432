// fake_rccl.cpp -- TEST INFRASTRUCTURE ONLY: a stand-in for librccl that lets the communicator code of libsrlivo_hip.so
// (srl_comm_init_rank -> ncclAllGather / ncclAllReduce on the context's stream -> publish kernel; csrc/srl_capi.cpp) run with
// N > 1 ranks on a box with ONE GPU.  Real RCCL refuses two ranks on one device; the gpurun boxes have one.  The library is
// selected explicitly through srl_comm_set_library() (never by an environment variable on the hot path) and implements exactly
// the entry points csrc/srl_rccl.cpp resolves:
//   ncclGetVersion, ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclAllReduce, ncclAllGather, ncclGetErrorString.
// Ranks are PROCESSES; they meet in a POSIX shared-memory segment named by the unique id.  A collective is stream-ordered like
// RCCL's: D2H copy of the send buffer on the caller's stream -> host callback on that stream (posts this rank's contribution,
// waits for every rank's, combines in RANK ORDER -- the same bits on every rank) -> H2D copy into the receive buffer.  What follows
// on the stream (the publish kernel) therefore sees the reduced data exactly as behind a real ncclAllReduce.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace {
constexpr int MAX_RANKS = 8;
constexpr int SLOTS = 64;                 // collectives in flight / not yet consumed (ring)
constexpr int MAX_BYTES = 1024;           // per rank and collective (the library sends 50 doubles or one int64)

struct Slot {
    std::atomic<unsigned long long> posted[MAX_RANKS];   // = op index + 1 once rank r's bytes of that op are in `data`
    unsigned char data[MAX_RANKS][MAX_BYTES];
};
struct Shared {
    std::atomic<int> joined;
    std::atomic<int> left;
    Slot slot[SLOTS];
};

struct FakeComm {
    int nranks = 0, rank = 0;
    Shared *sh = nullptr;
    char name[64] = {};
    unsigned long long next_op = 0;       // every rank issues the same sequence of collectives
    unsigned char *h_stage = nullptr;     // pinned: SLOTS x (send | recv) staging, MAX_RANKS * MAX_BYTES each
    int timed_out = 0;
};

struct OpArgs {
    FakeComm *c;
    unsigned long long op;
    size_t bytes;                          // per rank
    int kind;                              // 0 all-reduce (sum of doubles), 1 all-gather
    size_t count;
    unsigned char *h_send, *h_recv;
};

void exchange_cb(void *p) {
    OpArgs *a = static_cast<OpArgs *>(p);
    FakeComm *c = a->c;
    Slot &s = c->sh->slot[a->op % SLOTS];
    std::memcpy(s.data[c->rank], a->h_send, a->bytes);
    s.posted[c->rank].store(a->op + 1, std::memory_order_release);
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < c->nranks; r++) {
        while (s.posted[r].load(std::memory_order_acquire) != a->op + 1) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) {
                std::fprintf(stderr, "fake_rccl: rank %d of %d, collective %llu (kind %d): rank %d has not posted after 120 s (its last post: %llu)\n", c->rank, c->nranks,
                             a->op, a->kind, r, s.posted[r].load());
                c->timed_out = 1; delete a; return;
            }
            std::this_thread::yield();
        }
    }
    if (a->kind == 0) {
        double *out = reinterpret_cast<double *>(a->h_recv);
        for (size_t i = 0; i < a->count; i++) {
            double acc = 0.0;
            for (int r = 0; r < c->nranks; r++) acc += reinterpret_cast<const double *>(s.data[r])[i];   // rank order: identical bits everywhere
            out[i] = acc;
        }
    } else {
        for (int r = 0; r < c->nranks; r++) std::memcpy(a->h_recv + (size_t)r * a->bytes, s.data[r], a->bytes);
    }
    delete a;
}

ncclResult_t collective(int kind, const void *send, void *recv, size_t count, size_t elem, ncclComm_t comm, hipStream_t stream) {
    FakeComm *c = reinterpret_cast<FakeComm *>(comm);
    const size_t bytes = count * elem;
    if (!c || bytes > MAX_BYTES) return ncclInvalidArgument;
    const unsigned long long op = c->next_op++;
    unsigned char *st = c->h_stage + (op % SLOTS) * (size_t)(2 * MAX_RANKS * MAX_BYTES);
    OpArgs *a = new OpArgs{c, op, bytes, kind, count, st, st + MAX_RANKS * MAX_BYTES};
    if (hipMemcpyAsync(a->h_send, send, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipLaunchHostFunc(stream, exchange_cb, a) != hipSuccess) return ncclUnhandledCudaError;
    const size_t out_bytes = kind == 0 ? bytes : bytes * (size_t)c->nranks;
    if (hipMemcpyAsync(recv, a->h_recv, out_bytes, hipMemcpyHostToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}
}  // namespace

extern "C" {
ncclResult_t ncclGetVersion(int *v) { if (v) *v = 99901; return ncclSuccess; }          // nobody's real version: the tests check they run on THIS

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    std::memset(id, 0, sizeof *id);
    std::snprintf(id->internal, sizeof id->internal, "/srlfake_%d_%lld", (int)getpid(),
                  (long long)std::chrono::steady_clock::now().time_since_epoch().count());
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    FakeComm *c = new FakeComm();
    c->nranks = nranks; c->rank = rank;
    std::snprintf(c->name, sizeof c->name, "%s", id.internal);
    const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, sizeof(Shared)) != 0) { delete c; return ncclSystemError; }
    c->sh = static_cast<Shared *>(mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0));   // a fresh segment is zero-filled
    close(fd);
    if (c->sh == MAP_FAILED) { delete c; return ncclSystemError; }
    if (hipHostMalloc((void **)&c->h_stage, (size_t)SLOTS * 2 * MAX_RANKS * MAX_BYTES) != hipSuccess) { delete c; return ncclUnhandledCudaError; }
    c->sh->joined.fetch_add(1);
    const auto t0 = std::chrono::steady_clock::now();
    while (c->sh->joined.load() < nranks) {                    // like ncclCommInitRank: returns when every rank has joined
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return ncclSystemError;
        std::this_thread::yield();
    }
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int *count) {
    const FakeComm *c = reinterpret_cast<const FakeComm *>(comm);
    if (!c || !count) return ncclInvalidArgument;
    *count = c->sh->joined.load();                                  // the ranks that actually met in the segment
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    FakeComm *c = reinterpret_cast<FakeComm *>(comm);
    if (!c) return ncclSuccess;
    hipDeviceSynchronize();
    if (c->sh->left.fetch_add(1) + 1 == c->nranks) shm_unlink(c->name);
    munmap(c->sh, sizeof(Shared));
    hipHostFree(c->h_stage);
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
    if (dt != ncclDouble || op != ncclSum) return ncclInvalidArgument;      // all the library ever asks for
    return collective(0, send, recv, count, 8, comm, stream);
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t sendcount, ncclDataType_t dt, ncclComm_t comm, hipStream_t stream) {
    if (dt != ncclInt64) return ncclInvalidArgument;
    return collective(1, send, recv, sendcount, 8, comm, stream);
}

const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake rccl error"; }
}
