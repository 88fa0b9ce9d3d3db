/*
 * cloops_comm.h -- C ABI of libcloops_comm.so: the few RCCL collectives of the multi-GPU path of the cDBSCAN hot path,
 * one process per GPU, no PyTorch in the process.
 *
 * The reference runs its per-chromosome workers as joblib processes and merges their pickled results in the parent
 * (cLoops/pipe.py:113-127); between the steps of the sweep the parent estimates ONE distance cut from the concatenated
 * distance lists of all chromosomes (cLoops/pipe.py:247-275).  Here the ranks own chromosomes, so the path has exactly two
 * exchanges: the per-step sum of a small vector of statistics (cl_comm_allreduce_f64: counts ride as float64, exact below
 * 2^53) and, once per sweep, the gather of the candidate tables (cl_comm_allgather_i32 of the row counts, then of the
 * padded rows).  Both run over RCCL (xGMI inside a node) on a stream of the library's own.
 *
 * The 128-byte ncclUniqueId is made by rank 0 (cl_comm_unique_id) and handed to the other ranks by the HOST program
 * (cloops_amd/comm.py: a file in /tmp keyed by the launcher's pid; any other channel works as well).
 */
#ifndef CLOOPS_COMM_H
#define CLOOPS_COMM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CL_COMM_ID_BYTES 128

typedef struct cl_comm cl_comm;

/* last error text of the calling thread */
const char* cl_comm_last_error(void);

/* RCCL version codes (major * 10000 + minor * 100 + patch): the headers libcloops_comm.so was built against and the
 * librccl the process has mapped.  They can differ when another copy of librccl (PyTorch bundles one) was loaded first;
 * cloops_amd/comm.py refuses to form a communicator of more than one rank across a major.minor mismatch. */
int cl_comm_rccl_version(int* built_with, int* loaded);

/* rank 0: a fresh unique id (CL_COMM_ID_BYTES bytes) for cl_comm_init of all ranks */
int cl_comm_unique_id(void* id_out);

/* every rank: join the communicator on HIP device `device` (blocks until all `world` ranks have called it) */
int cl_comm_init(const void* id, int rank, int world, int device, cl_comm** out);
void cl_comm_destroy(cl_comm* c);
int cl_comm_rank(const cl_comm* c);
int cl_comm_world(const cl_comm* c);

/* element-wise sum over all ranks, in place, of n float64 in HOST memory (staged through a pinned buffer and the device;
 * returns when the result is in `host_inout`) -- the per-step statistics of the chained sweep (replaces the parent's
 * np.concatenate over its workers' lists, cLoops/pipe.py:119-127, 247-259) */
int cl_comm_allreduce_f64(cl_comm* c, double* host_inout, int64_t n);
/* same with max (bench.py: the slowest rank's wall time) */
int cl_comm_allreduce_max_f64(cl_comm* c, double* host_inout, int64_t n);

/* all-gather of `n` int32 per rank from HOST memory: host_out receives world * n values in rank order */
int cl_comm_allgather_i32(cl_comm* c, const int32_t* host_in, int64_t n, int32_t* host_out);

/* gather to one rank: only `root` receives world * n values (host_out may be null elsewhere) -- the candidate tables of a
 * sweep go to the rank that writes the result, like the reference's parent process (cLoops/pipe.py:119-127) */
int cl_comm_gather_i32(cl_comm* c, const int32_t* host_in, int64_t n, int root, int32_t* host_out);

/* Page-locked host buffers, and the gather (root >= 0) / all-gather (root < 0) of `n` int32 per rank between such buffers:
 * the device copies read `pinned_in` and write `pinned_out` directly, no staging copy on either side -- the candidate
 * tables of a 200 M-PET sweep are 58 MB at the root (cloops_amd/comm.py keeps one send and one receive buffer). */
void* cl_comm_host_alloc(int64_t bytes);
void cl_comm_host_free(void* p);
int cl_comm_gather_i32_pinned(cl_comm* c, const int32_t* pinned_in, int64_t n, int root, int32_t* pinned_out);

/* DEVICE-RESIDENT exchanges (exact-size send / receive of the loop tables, statistics reduced where the kernels
 * left them).
 * cl_comm_gather_device: every rank hands `ntab` int32 tables of rows[k] x cols that live in ITS device memory (the candidate
 * tables cl_cand_finish_device left there); `root` receives all of them -- ranks in order, a rank's tables in the order given,
 * exact sizes (row counts all-gathered first, then grouped ncclSend / ncclRecv; the root's own tables are device-to-device
 * copies) -- and copies the whole once to `pinned_out` (page-locked, cap_rows rows); rank_rows_out[world] = rows per rank
 * (every rank gets the counts).  The sending ranks touch neither host memory nor PCIe.
 * cl_comm_allreduce_f64_device: element-wise sum over the ranks of n float64 IN PLACE in device memory (the buffer the step's
 * last kernel wrote), on `stream` (the caller's; NULL = the communicator's own) -- no host staging; the caller copies the
 * result out once. */
int cl_comm_gather_device(cl_comm* c, const int32_t* const* dev_tables, const int64_t* rows, int32_t ntab, int32_t cols, int root,
                          int32_t* pinned_out, int64_t cap_rows, int64_t* rank_rows_out);
int cl_comm_allreduce_f64_device(cl_comm* c, double* dev_inout, int64_t n, void* stream);

/* hipDeviceSynchronize() of this rank's device, then a barrier over all ranks (an all-reduce of one element) */
int cl_comm_barrier(cl_comm* c);

#ifdef __cplusplus
}
#endif
#endif
