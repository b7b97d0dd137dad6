// rq_comm.hip -- the one exchange step of a sharded deployment: an all-gather
// of each rank's per-round summary over RCCL (xGMI), issued from a side stream
// so that the next round's kernel overlaps with it.
//
// RCCL is bound at run time (dlopen of librccl.so.1) so that librq.so carries no
// load-time dependency on it: a single-GPU host never touches RCCL, and inside a
// torch process the call resolves to the RCCL copy torch already loaded.
#include "../../include/rq.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

namespace {

// The slice of the NCCL/RCCL C API used here (rccl.h, stable since NCCL 2.x).
typedef struct rq_nccl_comm *nccl_comm_t;
typedef struct { char internal[128]; } nccl_unique_id;
enum { kNcclSuccess = 0, kNcclUint8 = 1 };  // ncclUint8 == ncclChar's sibling, value 1 in nccl.h

struct nccl_api {
    void *handle = nullptr;
    int (*GetUniqueId)(nccl_unique_id *) = nullptr;
    int (*CommInitRank)(nccl_comm_t *, int, nccl_unique_id, int) = nullptr;
    int (*CommDestroy)(nccl_comm_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, nccl_comm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
};

nccl_api &api() {
    static nccl_api a;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            a.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (a.handle) break;
        }
        if (!a.handle) return;
        a.GetUniqueId = (int (*)(nccl_unique_id *))dlsym(a.handle, "ncclGetUniqueId");
        a.CommInitRank = (int (*)(nccl_comm_t *, int, nccl_unique_id, int))dlsym(a.handle, "ncclCommInitRank");
        a.CommDestroy = (int (*)(nccl_comm_t))dlsym(a.handle, "ncclCommDestroy");
        a.AllGather = (int (*)(const void *, void *, size_t, int, nccl_comm_t, hipStream_t))dlsym(a.handle, "ncclAllGather");
        a.GetErrorString = (const char *(*)(int))dlsym(a.handle, "ncclGetErrorString");
        a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllGather;
    });
    return a;
}

}  // namespace

// Enqueuing an RCCL collective costs ~15 us of host time, half of a 30 us
// round, so it is done by a worker thread: the caller's thread only records
// an event on its compute stream and posts a request.
struct rq_comm_req {
    uint32_t slot;
    const void *send;
    void *recv;
    uint64_t bytes;
};

struct rq_comm {
    int device = 0, rank = 0, world = 1;
    nccl_comm_t comm = nullptr;
    hipStream_t side = nullptr;
    std::vector<hipEvent_t> produced;     // per slot: the round that filled it has been enqueued
    std::vector<hipEvent_t> done;         // per slot: its all-gather has finished
    std::vector<char> posted;             // per slot: a request is in flight or finished, not yet waited for
    std::vector<char> issued;             // per slot: the worker has enqueued it (done[slot] is recorded)
    std::mutex mu;
    std::condition_variable cv_work, cv_issued;
    std::deque<rq_comm_req> queue;
    bool stop = false;
    std::thread worker;
    std::atomic<int> last_nccl{0}, last_hip{0}, failed{0};
};

static void comm_worker(rq_comm *c) {
    (void)hipSetDevice(c->device);
    for (;;) {
        rq_comm_req r;
        {
            std::unique_lock<std::mutex> lk(c->mu);
            c->cv_work.wait(lk, [c] { return c->stop || !c->queue.empty(); });
            if (c->queue.empty()) return;   // stop requested and nothing left to issue
            r = c->queue.front();
            c->queue.pop_front();
        }
        hipError_t he = hipStreamWaitEvent(c->side, c->produced[r.slot], 0);
        int nr = kNcclSuccess;
        if (he == hipSuccess) nr = api().AllGather(r.send, r.recv, (size_t)r.bytes, kNcclUint8, c->comm, c->side);
        if (he == hipSuccess && nr == kNcclSuccess) he = hipEventRecord(c->done[r.slot], c->side);
        if (he != hipSuccess) { c->last_hip.store((int)he); c->failed.store(1); }
        if (nr != kNcclSuccess) { c->last_nccl.store(nr); c->failed.store(1); }
        {
            std::lock_guard<std::mutex> lk(c->mu);
            c->issued[r.slot] = 1;
        }
        c->cv_issued.notify_all();
    }
}

#define RQC_HIP(c, call)                                                   \
    do {                                                                   \
        hipError_t e_ = (call);                                            \
        if (e_ != hipSuccess) { (c)->last_hip.store((int)e_); return RQ_E_HIP; } \
    } while (0)

extern "C" {

int rq_comm_unique_id(uint8_t *out128) {
    if (!out128) return RQ_E_ARG;
    nccl_api &a = api();
    if (!a.ok) return RQ_E_COMM;
    nccl_unique_id id;
    if (a.GetUniqueId(&id) != kNcclSuccess) return RQ_E_COMM;
    std::memcpy(out128, id.internal, 128);
    return RQ_OK;
}

int rq_comm_create(int device, const uint8_t *id128, int rank, int world, uint32_t slots,
                   rq_comm **out) {
    if (!out) return RQ_E_ARG;
    *out = nullptr;
    if (!id128 || world < 1 || rank < 0 || rank >= world || slots < 1 || slots > 4096) return RQ_E_ARG;
    nccl_api &a = api();
    if (!a.ok) return RQ_E_COMM;
    if (hipSetDevice(device) != hipSuccess) return RQ_E_NODEVICE;
    rq_comm *c = new (std::nothrow) rq_comm();
    if (!c) return RQ_E_NOMEM;
    c->device = device; c->rank = rank; c->world = world;
    nccl_unique_id id;
    std::memcpy(id.internal, id128, 128);
    int rc = a.CommInitRank(&c->comm, world, id, rank);
    if (rc != kNcclSuccess) { delete c; return RQ_E_COMM; }
    bool ok = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) == hipSuccess;
    c->produced.assign(slots, nullptr);
    c->done.assign(slots, nullptr);
    c->posted.assign(slots, 0);
    c->issued.assign(slots, 0);
    for (uint32_t s = 0; ok && s < slots; s++)
        ok = hipEventCreateWithFlags(&c->produced[s], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&c->done[s], hipEventDisableTiming) == hipSuccess;
    if (!ok) { rq_comm_destroy(c); return RQ_E_HIP; }
    c->worker = std::thread(comm_worker, c);
    *out = c;
    return RQ_OK;
}

int rq_comm_destroy(rq_comm *c) {
    if (!c) return RQ_E_ARG;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        c->stop = true;
    }
    c->cv_work.notify_all();
    if (c->worker.joinable()) c->worker.join();
    (void)hipSetDevice(c->device);
    if (c->side) (void)hipStreamSynchronize(c->side);
    if (c->comm) (void)api().CommDestroy(c->comm);
    for (hipEvent_t e : c->done) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->produced) if (e) (void)hipEventDestroy(e);
    if (c->side) (void)hipStreamDestroy(c->side);
    delete c;
    return RQ_OK;
}

int rq_comm_allgather(rq_comm *c, void *compute_stream, uint32_t slot, const void *send,
                      void *recv, uint64_t bytes) {
    if (!c || !send || !recv || slot >= c->done.size()) return RQ_E_ARG;
    if (c->failed.load()) return RQ_E_COMM;
    if (c->posted[slot]) return RQ_E_STATE;   // rq_comm_wait(slot) must come between two uses of a slot
    RQC_HIP(c, hipSetDevice(c->device));
    // the collective is ordered after everything enqueued so far on the compute stream
    RQC_HIP(c, hipEventRecord(c->produced[slot], (hipStream_t)compute_stream));
    {
        std::lock_guard<std::mutex> lk(c->mu);
        c->posted[slot] = 1;
        c->issued[slot] = 0;
        c->queue.push_back(rq_comm_req{slot, send, recv, bytes});
    }
    c->cv_work.notify_one();
    return RQ_OK;
}

// blocks the calling thread only until the worker has enqueued slot's collective
static void wait_issued(rq_comm *c, uint32_t slot) {
    std::unique_lock<std::mutex> lk(c->mu);
    c->cv_issued.wait(lk, [c, slot] { return c->issued[slot] != 0; });
}

int rq_comm_wait(rq_comm *c, void *compute_stream, uint32_t slot) {
    if (!c || slot >= c->done.size()) return RQ_E_ARG;
    if (!c->posted[slot]) return RQ_OK;
    wait_issued(c, slot);
    c->posted[slot] = 0;
    if (c->failed.load()) return RQ_E_COMM;
    RQC_HIP(c, hipSetDevice(c->device));
    RQC_HIP(c, hipStreamWaitEvent((hipStream_t)compute_stream, c->done[slot], 0));
    return RQ_OK;
}

int rq_comm_sync(rq_comm *c) {
    if (!c) return RQ_E_ARG;
    for (uint32_t s = 0; s < c->posted.size(); s++)
        if (c->posted[s]) wait_issued(c, s);
    if (c->failed.load()) return RQ_E_COMM;
    RQC_HIP(c, hipSetDevice(c->device));
    RQC_HIP(c, hipStreamSynchronize(c->side));
    return RQ_OK;
}

int rq_comm_last_error(const rq_comm *c, int *nccl_result, int *hip_error) {
    if (!c) return RQ_E_ARG;
    if (nccl_result) *nccl_result = c->last_nccl.load();
    if (hip_error) *hip_error = c->last_hip.load();
    return RQ_OK;
}

}  // extern "C"
