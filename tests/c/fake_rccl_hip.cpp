/* Test double of the HIP runtime and RCCL entry points libcloops_comm.so uses, on host memory with THREADS as ranks
 * (tests/test_comm_fake_world.py): cloops_amd/csrc/cloops_comm.cpp is compiled unchanged with g++ and linked against this file
 * instead of libamdhip64 / librccl, so that its multi-rank control flow (counts -> validation -> grouped send / receive) runs at
 * world sizes > 1 without GPUs.  "Device" memory is malloc'ed memory, streams are synchronous, a collective is a rendezvous of
 * the ranks' threads.  Every wait gives up after 20 s with an error: a protocol bug fails a test instead of hanging it. */
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {
struct Op { const void* send; void* recv; size_t count; };
struct Msg { const void* ptr = nullptr; size_t bytes = 0; bool ready = false; };
struct Hub {
    std::mutex m; std::condition_variable cv;
    int world = 0, arrived = 0; long gen = 0;
    std::vector<Op> slot;
    std::map<std::pair<int, int>, Msg> mail;           // (src, dst)
    bool barrier(std::unique_lock<std::mutex>& lk)
    {
        const long g = gen;
        if (++arrived == world) { arrived = 0; ++gen; cv.notify_all(); return true; }
        return cv.wait_for(lk, std::chrono::seconds(20), [&] { return gen != g; });
    }
};
struct FakeComm { Hub* hub; int rank; };
std::mutex g_m;
std::map<std::string, Hub*> g_hubs;
int g_next_id = 1;
struct Pending { bool send; void* ptr; size_t bytes; int peer; FakeComm* c; };
thread_local int t_depth = 0;
thread_local std::vector<Pending> t_ops;
size_t tsize(ncclDataType_t t) { return (t == ncclDouble || t == ncclInt64 || t == ncclUint64) ? 8 : ((t == ncclInt8 || t == ncclUint8) ? 1 : 4); }

ncclResult_t run_ops()
{
    std::vector<Pending> ops;
    ops.swap(t_ops);
    for (auto& o : ops) if (o.send) {
        std::unique_lock<std::mutex> lk(o.c->hub->m);
        Msg& ms = o.c->hub->mail[{o.c->rank, o.peer}];
        ms.ptr = o.ptr; ms.bytes = o.bytes; ms.ready = true;
        o.c->hub->cv.notify_all();
    }
    for (auto& o : ops) if (!o.send) {
        std::unique_lock<std::mutex> lk(o.c->hub->m);
        Msg& ms = o.c->hub->mail[{o.peer, o.c->rank}];
        if (!o.c->hub->cv.wait_for(lk, std::chrono::seconds(20), [&] { return ms.ready; })) return ncclInternalError;
        if (ms.bytes != o.bytes) return ncclInvalidArgument;
        memcpy(o.ptr, ms.ptr, o.bytes);
        ms.ready = false;
        o.c->hub->cv.notify_all();
    }
    for (auto& o : ops) if (o.send) {
        std::unique_lock<std::mutex> lk(o.c->hub->m);
        Msg& ms = o.c->hub->mail[{o.c->rank, o.peer}];
        if (!o.c->hub->cv.wait_for(lk, std::chrono::seconds(20), [&] { return !ms.ready; })) return ncclInternalError;
    }
    return ncclSuccess;
}

template <typename F>
ncclResult_t collective(FakeComm* c, const void* send, void* recv, size_t count, F&& apply /* (slots) -> writes into a private result */)
{
    Hub* h = c->hub;
    std::unique_lock<std::mutex> lk(h->m);
    h->slot[c->rank] = Op{send, recv, count};
    if (!h->barrier(lk)) return ncclInternalError;
    for (int r = 0; r < h->world; ++r) if (h->slot[r].count != count) return ncclInvalidArgument;
    std::vector<char> tmp;
    apply(h->slot, tmp);
    if (!h->barrier(lk)) return ncclInternalError;       // everyone has read the inputs
    if (!tmp.empty() && recv) memcpy(recv, tmp.data(), tmp.size());
    if (!h->barrier(lk)) return ncclInternalError;
    return ncclSuccess;
}
}

extern "C" {
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned int) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned int) { *s = (hipStream_t)malloc(8); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free((void*)s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "fake hip error"; }

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclInternalError ? "fake rccl: a rank waited 20 s for its peers" : "fake rccl error"; }
ncclResult_t ncclGetVersion(int* v) { *v = NCCL_VERSION_CODE; return ncclSuccess; }
ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    std::lock_guard<std::mutex> g(g_m);
    memset(id, 0, sizeof(*id));
    snprintf(id->internal, sizeof(id->internal), "fake-%d", g_next_id++);
    return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t* out, int world, ncclUniqueId id, int rank)
{
    std::lock_guard<std::mutex> g(g_m);
    Hub*& h = g_hubs[std::string(id.internal)];
    if (!h) { h = new Hub(); h->world = world; h->slot.resize(world); }
    if (h->world != world) return ncclInvalidArgument;
    *out = (ncclComm_t) new FakeComm{h, rank};
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) { delete (FakeComm*)c; return ncclSuccess; }
ncclResult_t ncclAllReduce(const void* s, void* r, size_t n, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, hipStream_t)
{
    if (t != ncclDouble) return ncclInvalidArgument;
    return collective((FakeComm*)c, s, r, n, [&](std::vector<Op>& sl, std::vector<char>& tmp) {
        tmp.resize(n * 8);
        double* o = (double*)tmp.data();
        for (size_t i = 0; i < n; ++i) {
            double a = ((const double*)sl[0].send)[i];
            for (size_t k = 1; k < sl.size(); ++k) { const double b = ((const double*)sl[k].send)[i]; a = op == ncclMax ? (b > a ? b : a) : a + b; }
            o[i] = a;
        }
    });
}
ncclResult_t ncclAllGather(const void* s, void* r, size_t n, ncclDataType_t t, ncclComm_t c, hipStream_t)
{
    const size_t b = n * tsize(t);
    return collective((FakeComm*)c, s, r, n, [&](std::vector<Op>& sl, std::vector<char>& tmp) {
        tmp.resize(b * sl.size());
        for (size_t k = 0; k < sl.size(); ++k) memcpy(tmp.data() + k * b, sl[k].send, b);
    });
}
ncclResult_t ncclGather(const void* s, void* r, size_t n, ncclDataType_t t, int root, ncclComm_t c, hipStream_t)
{
    const size_t b = n * tsize(t);
    FakeComm* fc = (FakeComm*)c;
    return collective(fc, s, r, n, [&](std::vector<Op>& sl, std::vector<char>& tmp) {
        if (fc->rank != root) return;
        tmp.resize(b * sl.size());
        for (size_t k = 0; k < sl.size(); ++k) memcpy(tmp.data() + k * b, sl[k].send, b);
    });
}
ncclResult_t ncclGroupStart(void) { ++t_depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd(void) { if (--t_depth > 0) return ncclSuccess; t_depth = 0; return run_ops(); }
ncclResult_t ncclSend(const void* p, size_t n, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t)
{
    t_ops.push_back(Pending{true, (void*)p, n * tsize(t), peer, (FakeComm*)c});
    return t_depth > 0 ? ncclSuccess : run_ops();
}
ncclResult_t ncclRecv(void* p, size_t n, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t)
{
    t_ops.push_back(Pending{false, p, n * tsize(t), peer, (FakeComm*)c});
    return t_depth > 0 ? ncclSuccess : run_ops();
}
}
