// cloops_comm.cpp -- libcloops_comm.so: RCCL collectives of the multi-GPU path behind the C ABI of include/cloops_comm.h
// (one process per GPU, no PyTorch in the process).  Host buffers are staged through one pinned buffer and one device
// buffer per communicator, both grown on demand; every call returns with its result on the host.
#include <cstring>
#include <string>
#include <algorithm>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include "../../include/cloops_comm.h"

static thread_local std::string g_cerr;
static int cfail(const char* what, const char* detail)
{
    g_cerr = what;
    if (detail) { g_cerr += ": "; g_cerr += detail; }
    return -2;
}
#define HIPC(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return cfail(#expr, hipGetErrorString(e_)); } while (0)
#define NCC(expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) return cfail(#expr, ncclGetErrorString(r_)); } while (0)

struct cl_comm {
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    int rank = 0, world = 1, device = 0;
    void* pin = nullptr; size_t pin_bytes = 0;
    void* dev = nullptr; size_t dev_bytes = 0;
};

extern "C" const char* cl_comm_last_error(void) { return g_cerr.c_str(); }

extern "C" int cl_comm_rccl_version(int* built_with, int* loaded)
{
    // the headers this library was compiled against vs the librccl the process has actually mapped (a process that imported
    // PyTorch first carries torch's bundled librccl under the same SONAME)
    if (built_with) *built_with = NCCL_VERSION_CODE;
    int v = 0;
    NCC(ncclGetVersion(&v));
    if (loaded) *loaded = v;
    return 0;
}

extern "C" int cl_comm_unique_id(void* id_out)
{
    if (!id_out) return cfail("cl_comm_unique_id", "null argument");
    static_assert(sizeof(ncclUniqueId) == CL_COMM_ID_BYTES, "unique id size");
    ncclUniqueId id;
    NCC(ncclGetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return 0;
}

extern "C" int cl_comm_init(const void* idp, int rank, int world, int device, cl_comm** out)
{
    if (!idp || !out || world < 1 || rank < 0 || rank >= world) return cfail("cl_comm_init", "bad arguments");
    *out = nullptr;
    HIPC(hipSetDevice(device));
    cl_comm* c = new cl_comm();
    c->rank = rank; c->world = world; c->device = device;
    ncclUniqueId id;
    memcpy(&id, idp, sizeof(id));
    hipError_t he = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (he != hipSuccess) { delete c; return cfail("hipStreamCreateWithFlags", hipGetErrorString(he)); }
    ncclResult_t r = ncclCommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) { (void)hipStreamDestroy(c->stream); delete c; return cfail("ncclCommInitRank", ncclGetErrorString(r)); }
    *out = c;
    return 0;
}

extern "C" void cl_comm_destroy(cl_comm* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)ncclCommDestroy(c->comm);
    if (c->pin) (void)hipHostFree(c->pin);
    if (c->dev) (void)hipFree(c->dev);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int cl_comm_rank(const cl_comm* c) { return c ? c->rank : -1; }
extern "C" int cl_comm_world(const cl_comm* c) { return c ? c->world : -1; }

static int ensure(cl_comm* c, size_t pin_bytes, size_t dev_bytes)
{
    HIPC(hipSetDevice(c->device));
    if (pin_bytes > c->pin_bytes) {
        if (c->pin) (void)hipHostFree(c->pin);
        c->pin = nullptr; c->pin_bytes = 0;
        const size_t want = pin_bytes + pin_bytes / 4 + 4096;
        HIPC(hipHostMalloc(&c->pin, want, hipHostMallocDefault));
        c->pin_bytes = want;
    }
    if (dev_bytes > c->dev_bytes) {
        if (c->dev) (void)hipFree(c->dev);
        c->dev = nullptr; c->dev_bytes = 0;
        const size_t want = dev_bytes + dev_bytes / 4 + 4096;
        HIPC(hipMalloc(&c->dev, want));
        c->dev_bytes = want;
    }
    return 0;
}

static int allreduce_f64(cl_comm* c, double* h, int64_t n, ncclRedOp_t op)
{
    if (!c || (n > 0 && !h) || n < 0) return cfail("cl_comm_allreduce", "bad arguments");
    if (n == 0) return 0;
    const size_t bytes = (size_t)n * 8;
    int rc = ensure(c, bytes, bytes);
    if (rc) return rc;
    memcpy(c->pin, h, bytes);
    HIPC(hipMemcpyAsync(c->dev, c->pin, bytes, hipMemcpyHostToDevice, c->stream));
    NCC(ncclAllReduce(c->dev, c->dev, (size_t)n, ncclDouble, op, c->comm, c->stream));
    HIPC(hipMemcpyAsync(c->pin, c->dev, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    memcpy(h, c->pin, bytes);
    return 0;
}

extern "C" int cl_comm_allreduce_f64(cl_comm* c, double* h, int64_t n) { return allreduce_f64(c, h, n, ncclSum); }
extern "C" int cl_comm_allreduce_max_f64(cl_comm* c, double* h, int64_t n) { return allreduce_f64(c, h, n, ncclMax); }

// n int32 per rank from host memory; ROOT < 0: all-gather (every rank receives world * n values in rank order), else only
// rank ROOT receives (the others neither copy back nor need host_out)
extern "C" void* cl_comm_host_alloc(int64_t bytes)
{
    void* p = nullptr;
    if (bytes <= 0 || hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault) != hipSuccess) { cfail("cl_comm_host_alloc", "hipHostMalloc failed"); return nullptr; }
    return p;
}
extern "C" void cl_comm_host_free(void* p) { if (p) (void)hipHostFree(p); }

// the same exchange for buffers the caller allocated with cl_comm_host_alloc (page-locked): the device copies read / write them
// directly, no staging copy on either side
extern "C" int cl_comm_gather_i32_pinned(cl_comm* c, const int32_t* pin_in, int64_t n, int root, int32_t* pin_out)
{
    if (!c || n < 0 || root >= c->world || (n > 0 && !pin_in)) return cfail("cl_comm_gather_i32_pinned", "bad arguments");
    const bool recv = root < 0 || root == c->rank;
    if (n > 0 && recv && !pin_out) return cfail("cl_comm_gather_i32_pinned", "null receive buffer");
    if (n == 0) return 0;
    const size_t in_bytes = (size_t)n * 4, out_bytes = in_bytes * (size_t)c->world, pad = ((in_bytes + 255) / 256) * 256;
    int rc = ensure(c, 0, pad + (recv ? out_bytes : (size_t)256));
    if (rc) return rc;
    char* dsend = (char*)c->dev;
    char* drecv = dsend + pad;
    HIPC(hipMemcpyAsync(dsend, pin_in, in_bytes, hipMemcpyHostToDevice, c->stream));
    if (root < 0) NCC(ncclAllGather(dsend, drecv, (size_t)n, ncclInt32, c->comm, c->stream));
    else NCC(ncclGather(dsend, drecv, (size_t)n, ncclInt32, root, c->comm, c->stream));
    if (recv) HIPC(hipMemcpyAsync(pin_out, drecv, out_bytes, hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    return 0;
}

static int gather_i32(cl_comm* c, const int32_t* hin, int64_t n, int root, int32_t* hout)
{
    if (!c || n < 0 || root >= c->world || (n > 0 && !hin)) return cfail("cl_comm_gather_i32", "bad arguments");
    const bool recv = root < 0 || root == c->rank;
    if (n > 0 && recv && !hout) return cfail("cl_comm_gather_i32", "null receive buffer");
    if (n == 0) return 0;
    const size_t in_bytes = (size_t)n * 4, out_bytes = in_bytes * (size_t)c->world, pad = ((in_bytes + 255) / 256) * 256;
    // (every rank hands RCCL a valid receive pointer, the non-root ranks a small dummy area: nothing is written there)
    int rc = ensure(c, std::max(in_bytes, recv ? out_bytes : (size_t)0), pad + (recv ? out_bytes : (size_t)256));
    if (rc) return rc;
    char* dsend = (char*)c->dev;
    char* drecv = dsend + pad;
    memcpy(c->pin, hin, in_bytes);
    HIPC(hipMemcpyAsync(dsend, c->pin, in_bytes, hipMemcpyHostToDevice, c->stream));
    if (root < 0) NCC(ncclAllGather(dsend, drecv, (size_t)n, ncclInt32, c->comm, c->stream));
    else NCC(ncclGather(dsend, drecv, (size_t)n, ncclInt32, root, c->comm, c->stream));
    if (recv) HIPC(hipMemcpyAsync(c->pin, drecv, out_bytes, hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    if (recv) memcpy(hout, c->pin, out_bytes);
    return 0;
}

extern "C" int cl_comm_allgather_i32(cl_comm* c, const int32_t* hin, int64_t n, int32_t* hout) { return gather_i32(c, hin, n, -1, hout); }
extern "C" int cl_comm_gather_i32(cl_comm* c, const int32_t* hin, int64_t n, int root, int32_t* hout)
{
    if (root < 0) return cfail("cl_comm_gather_i32", "bad root");
    return gather_i32(c, hin, n, root, hout);
}

// ---- device-resident exchanges ------------------------------------------------------------------------------------
extern "C" int cl_comm_allreduce_f64_device(cl_comm* c, double* dev, int64_t n, void* stream)
{
    if (!c || (n > 0 && !dev) || n < 0) return cfail("cl_comm_allreduce_f64_device", "bad arguments");
    if (n == 0) return 0;
    HIPC(hipSetDevice(c->device));
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    NCC(ncclAllReduce(dev, dev, (size_t)n, ncclDouble, ncclSum, c->comm, st));
    if (!stream) HIPC(hipStreamSynchronize(st));
    return 0;
}

// One all-gather of a small int32 record per rank through the communicator's staging buffers (pin: [0, in) send, [64 + ...) receive)
static int allgather_words(cl_comm* c, const int32_t* mine, int nwords, int32_t* all /* world * nwords */)
{
    const size_t in_bytes = (size_t)nwords * 4, out_bytes = in_bytes * (size_t)c->world, pad = 256;
    int rc = ensure(c, pad + out_bytes, pad + out_bytes);
    if (rc) return rc;
    memcpy(c->pin, mine, in_bytes);
    HIPC(hipMemcpyAsync(c->dev, c->pin, in_bytes, hipMemcpyHostToDevice, c->stream));
    NCC(ncclAllGather(c->dev, (char*)c->dev + pad, (size_t)nwords, ncclInt32, c->comm, c->stream));
    HIPC(hipMemcpyAsync((char*)c->pin + pad, (char*)c->dev + pad, out_bytes, hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    memcpy(all, (char*)c->pin + pad, out_bytes);
    return 0;
}

// The protocol has THREE phases so that no rank can leave another inside RCCL:
//   1. every rank all-gathers {its row count, and -- the root -- the capacity and presence of its receive buffer}: "pinned_out too
//      small" / "null receive buffer" are decided by EVERY rank from the same record, before any send or receive is posted;
//   2. every rank allocates its staging area, then the ranks all-gather one status word: an allocation that failed anywhere makes
//      every rank return the error together;
//   3. the tables move (grouped ncclSend / ncclRecv); a failure between ncclGroupStart and ncclGroupEnd still closes the group.
extern "C" int cl_comm_gather_device(cl_comm* c, const int32_t* const* tabs, const int64_t* rows, int32_t ntab, int32_t cols, int root,
                                     int32_t* pinned_out, int64_t cap_rows, int64_t* rank_rows)
{
    if (!c || ntab < 0 || cols <= 0 || root < 0 || root >= c->world || !rank_rows || (ntab > 0 && (!tabs || !rows)))
        return cfail("cl_comm_gather_device", "bad arguments");
    HIPC(hipSetDevice(c->device));
    // a table this rank cannot send is reported through the record (status word), not by leaving before the first collective
    long long mine = 0;
    int local_bad = 0;
    for (int k = 0; k < ntab; ++k) { if (rows[k] < 0 || (rows[k] > 0 && !tabs[k])) { local_bad = 1; break; } mine += rows[k]; }
    if (local_bad) mine = 0;
    const bool is_root = c->rank == root;
    const long long cap = is_root ? std::max<long long>(cap_rows, 0) : 0;
    // (counts and capacity as two 31-bit halves: a 200 M-PET genome stays far below 2^31 rows, a larger one may not)
    enum { NW = 6 };
    int32_t rec[NW] = {(int32_t)(mine & 0x7fffffff), (int32_t)(mine >> 31), (int32_t)(cap & 0x7fffffff), (int32_t)(cap >> 31),
                       (int32_t)(is_root && pinned_out ? 1 : 0), (int32_t)local_bad};
    std::string all_s((size_t)c->world * NW * 4, '\0');
    int32_t* all = (int32_t*)&all_s[0];
    int rc = allgather_words(c, rec, NW, all);
    if (rc) return rc;
    long long total = 0;
    bool any_bad = false;
    for (int r = 0; r < c->world; ++r) {
        rank_rows[r] = (long long)all[NW * r] + ((long long)all[NW * r + 1] << 31);
        total += rank_rows[r];
        any_bad |= all[NW * r + 5] != 0;
    }
    const long long root_cap = (long long)all[NW * root + 2] + ((long long)all[NW * root + 3] << 31);
    const bool root_has_out = all[NW * root + 4] != 0;
    // phase 1 verdicts: the same on every rank
    if (any_bad) return cfail("cl_comm_gather_device", "bad table (negative row count or null pointer) on some rank");
    if (total > root_cap) return cfail("cl_comm_gather_device", "pinned_out too small");
    if (total > 0 && !root_has_out) return cfail("cl_comm_gather_device", "null receive buffer");
    const size_t row_bytes = (size_t)cols * 4;
    if (c->world == 1) {
        // one rank: every table goes straight to its place in the caller's page-locked buffer
        size_t at = 0;
        for (int k = 0; k < ntab; ++k)
            if (rows[k] > 0) { HIPC(hipMemcpyAsync((char*)pinned_out + at, tabs[k], (size_t)rows[k] * row_bytes, hipMemcpyDeviceToHost, c->stream)); at += (size_t)rows[k] * row_bytes; }
        HIPC(hipStreamSynchronize(c->stream));
        return 0;
    }
    // phase 2: staging areas, then one status word per rank
    {
        const size_t need = is_root ? (size_t)total * row_bytes + 256 : (mine > 0 ? (size_t)mine * row_bytes + 256 : 0);
        int32_t st = need ? (ensure(c, 0, need + 1024 + (size_t)c->world * 4) != 0) : 0;
        const std::string keep = g_cerr;
        std::string sts((size_t)c->world * 4, '\0');
        rc = allgather_words(c, &st, 1, (int32_t*)&sts[0]);
        if (rc) return rc;
        for (int r = 0; r < c->world; ++r)
            if (((const int32_t*)&sts[0])[r] != 0) {
                if (st) { g_cerr = keep; return -2; }
                return cfail("cl_comm_gather_device", "staging allocation failed on another rank");
            }
    }
    // phase 3: the tables.  Device-to-device packing first (plain stream work), then ONLY RCCL calls inside the group; a failure inside
    // the group is remembered and the group is closed before returning
    char* stage = (char*)c->dev + 512 + (((size_t)c->world * 4 + 255) / 256) * 256;      // (behind the words allgather_words staged)
    if (is_root) {
        size_t at = 0;
        for (int r = 0; r < root; ++r) at += (size_t)rank_rows[r] * row_bytes;
        for (int k = 0; k < ntab; ++k)
            if (rows[k] > 0) { HIPC(hipMemcpyAsync(stage + at, tabs[k], (size_t)rows[k] * row_bytes, hipMemcpyDeviceToDevice, c->stream)); at += (size_t)rows[k] * row_bytes; }
    } else {
        size_t at = 0;
        for (int k = 0; k < ntab; ++k)
            if (rows[k] > 0) { HIPC(hipMemcpyAsync(stage + at, tabs[k], (size_t)rows[k] * row_bytes, hipMemcpyDeviceToDevice, c->stream)); at += (size_t)rows[k] * row_bytes; }
    }
    ncclResult_t bad = ncclSuccess;
    NCC(ncclGroupStart());
    if (is_root) {
        size_t at = 0;
        for (int r = 0; r < c->world; ++r) {
            if (r != root && rank_rows[r] > 0 && bad == ncclSuccess)
                bad = ncclRecv(stage + at, (size_t)rank_rows[r] * cols, ncclInt32, r, c->comm, c->stream);
            at += (size_t)rank_rows[r] * row_bytes;
        }
    } else if (mine > 0) {
        // (one message per rank: the root posts ONE receive per rank)
        bad = ncclSend(stage, (size_t)mine * cols, ncclInt32, root, c->comm, c->stream);
    }
    const ncclResult_t ge = ncclGroupEnd();
    if (bad != ncclSuccess) return cfail("ncclSend / ncclRecv", ncclGetErrorString(bad));
    if (ge != ncclSuccess) return cfail("ncclGroupEnd", ncclGetErrorString(ge));
    if (is_root && total > 0) HIPC(hipMemcpyAsync(pinned_out, stage, (size_t)total * row_bytes, hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int cl_comm_barrier(cl_comm* c)
{
    // everything this process has enqueued on the device is done, then all ranks meet
    if (!c) return cfail("cl_comm_barrier", "null communicator");
    HIPC(hipSetDevice(c->device));
    HIPC(hipDeviceSynchronize());
    double one = 1.0;
    return allreduce_f64(c, &one, 1, ncclSum);
}
