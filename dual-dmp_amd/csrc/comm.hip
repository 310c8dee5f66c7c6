// Multi-device exchange behind the C ABI (the reference is single-device, main.py:51).
//
//   ddmp_comm       one RCCL communicator per process (one process per GPU, xGMI between them)
//   ddmp_halo_plan  what a rank sends and receives per aggregation: the boundary rows of its shard (device index list, in
//                   the receivers' halo order) and the per-peer counts
//   ddmp_halo_exchange   pack kernel -> ONE grouped ncclSend/ncclRecv per peer pair straight into the halo rows of the
//                   feature tensor (rows [n_rows, n_cols), grouped by source rank), and -- in the same group, i.e. the same
//                   launch on the wire -- the all-reduce of the layer's 2C float64 BatchNorm column sums.  Everything is
//                   enqueued on the caller's stream (capturable; no host sync).
// RCCL is resolved at run time from the library PyTorch has already loaded (dlopen of its SONAME), so libddmp_hip.so has
// no link-time dependency on it and single-device use never touches it.  Payloads here are small (a few thousand rows x
// <= 512 channels per peer; 8 KB of sums): latency-bound point-to-point over xGMI, one grouped launch instead of an
// all_to_all_single + an all_reduce issued from Python.
#include "b16_common.h"

#include <dlfcn.h>

#include <algorithm>
#include <cstdlib>
#include <vector>

namespace {

using namespace ddmp;

typedef struct { char internal[128]; } nccl_uid;                 // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void* nccl_comm;
struct Rccl {
    int (*GetUniqueId)(nccl_uid*);
    int (*CommInitRank)(nccl_comm*, int, nccl_uid, int);
    int (*CommDestroy)(nccl_comm);
    int (*GroupStart)();
    int (*GroupEnd)();
    int (*Send)(const void*, size_t, int, int, nccl_comm, hipStream_t);
    int (*Recv)(void*, size_t, int, int, nccl_comm, hipStream_t);
    int (*AllReduce)(const void*, void*, size_t, int, int, nccl_comm, hipStream_t);
    int (*AllGather)(const void*, void*, size_t, int, nccl_comm, hipStream_t);
    bool ok = false;
};
constexpr int kNcclInt8 = 0, kNcclFloat32 = 7, kNcclFloat64 = 8, kNcclSum = 0;

Rccl& rccl() {
    static Rccl r;
    static bool tried = false;
    if (tried) return r;
    tried = true;
    void* h = nullptr;
    for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"}) {
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) return r;
#define DDMP_SYM(field, sym)                                   \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, sym)); \
    if (!r.field) return r;
    DDMP_SYM(GetUniqueId, "ncclGetUniqueId")
    DDMP_SYM(CommInitRank, "ncclCommInitRank")
    DDMP_SYM(CommDestroy, "ncclCommDestroy")
    DDMP_SYM(GroupStart, "ncclGroupStart")
    DDMP_SYM(GroupEnd, "ncclGroupEnd")
    DDMP_SYM(Send, "ncclSend")
    DDMP_SYM(Recv, "ncclRecv")
    DDMP_SYM(AllReduce, "ncclAllReduce")
    DDMP_SYM(AllGather, "ncclAllGather")
#undef DDMP_SYM
    r.ok = true;
    return r;
}

__global__ __launch_bounds__(256) void halo_pack_kernel(const uint4* __restrict__ src, int64_t ld16, const int64_t* __restrict__ idx,
                                                        int64_t n, int q, uint4* __restrict__ dst) {
    const int64_t total = n * q;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / q;
        const int c = (int)(i - r * q);
        dst[i] = src[idx[r] * ld16 + c];
    }
}

}  // namespace

struct ddmp_comm {
    nccl_comm comm;
    int rank, world;
    // DDMP_COMM_LOOPBACK=1 (read at creation): a communicator of ONE rank still goes through RCCL -- every exchange sends and
    // receives 256 bytes to / from itself inside the group and all-reduces over its one rank -- so that the send/recv symbol
    // table, the grouped launch and the capture of the partitioned iteration into a hipGraph can be exercised on a one-GPU box
    char* loop = nullptr;                    // device: 256 bytes out + 256 bytes in
};
struct ddmp_halo_plan {
    int world, rank;
    int64_t n_rows, n_cols, n_send;
    int64_t* send_idx;                       // device [n_send]: local owned rows, grouped by destination rank
    std::vector<int64_t> send_counts, recv_counts;
};

#define NCCL_TRY(expr)                      \
    do {                                    \
        int e__ = (expr);                   \
        if (e__ != 0) return 1000 + e__;    \
    } while (0)

extern "C" int ddmp_comm_unique_id(char* id128_host) {
    ARG_TRY(id128_host);
    if (!rccl().ok) return DDMP_EINVAL;
    nccl_uid id;
    NCCL_TRY(rccl().GetUniqueId(&id));
    std::copy(id.internal, id.internal + 128, id128_host);
    return DDMP_OK;
}

extern "C" int ddmp_comm_create(int rank, int world, const char* id128_host, ddmp_comm** out) {
    ARG_TRY(out && id128_host && world >= 1 && rank >= 0 && rank < world);
    if (!rccl().ok) return DDMP_EINVAL;
    nccl_uid id;
    std::copy(id128_host, id128_host + 128, id.internal);
    nccl_comm c = nullptr;
    NCCL_TRY(rccl().CommInitRank(&c, world, id, rank));
    ddmp_comm* cm = new ddmp_comm{c, rank, world};
    const char* lb = getenv("DDMP_COMM_LOOPBACK");
    if (world == 1 && lb && atoi(lb) == 1 && hipMalloc((void**)&cm->loop, 512) == hipSuccess) (void)hipMemset(cm->loop, 0, 512);
    *out = cm;
    return DDMP_OK;
}

extern "C" int ddmp_comm_destroy(ddmp_comm* c) {
    if (!c) return DDMP_OK;
    if (c->comm) (void)rccl().CommDestroy(c->comm);
    if (c->loop) (void)hipFree(c->loop);
    delete c;
    return DDMP_OK;
}

extern "C" int ddmp_halo_plan_create(int world, int rank, int64_t n_rows, int64_t n_cols, const int64_t* send_idx_host,
                                     const int64_t* send_counts_host, const int64_t* recv_counts_host,
                                     ddmp_halo_plan** out) {
    ARG_TRY(out && world >= 1 && rank >= 0 && rank < world && n_rows >= 0 && n_cols >= n_rows && send_counts_host && recv_counts_host);
    ddmp_halo_plan* p = new ddmp_halo_plan();
    p->world = world;
    p->rank = rank;
    p->n_rows = n_rows;
    p->n_cols = n_cols;
    p->send_counts.assign(send_counts_host, send_counts_host + world);
    p->recv_counts.assign(recv_counts_host, recv_counts_host + world);
    p->n_send = 0;
    int64_t n_recv = 0;
    for (int r = 0; r < world; ++r) {
        p->n_send += p->send_counts[r];
        n_recv += p->recv_counts[r];
    }
    p->send_idx = nullptr;
    if (n_recv != n_cols - n_rows || (p->n_send > 0 && !send_idx_host)) {
        delete p;
        return DDMP_EINVAL;
    }
    for (int64_t i = 0; i < p->n_send; ++i)
        if (send_idx_host[i] < 0 || send_idx_host[i] >= n_rows) {
            delete p;
            return DDMP_ERANGE;
        }
    if (p->n_send > 0) {
        hipError_t e = hipMalloc((void**)&p->send_idx, sizeof(int64_t) * (size_t)p->n_send);
        if (e == hipSuccess) e = hipMemcpy(p->send_idx, send_idx_host, sizeof(int64_t) * (size_t)p->n_send, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            if (p->send_idx) (void)hipFree(p->send_idx);
            delete p;
            return (int)e;
        }
    }
    *out = p;
    return DDMP_OK;
}

extern "C" int ddmp_halo_plan_destroy(ddmp_halo_plan* p) {
    if (!p) return DDMP_OK;
    if (p->send_idx) (void)hipFree(p->send_idx);
    delete p;
    return DDMP_OK;
}

extern "C" size_t ddmp_halo_pack_bytes(const ddmp_halo_plan* p, int C, int dtype) {
    if (!p || C <= 0) return 0;
    return (size_t)std::max<int64_t>(p->n_send, 1) * (size_t)C * (dtype == DDMP_BF16 ? 2 : 4);
}

extern "C" int ddmp_halo_exchange(ddmp_comm* comm, const ddmp_halo_plan* p, void* T, int64_t ld, int C, int dtype,
                                  void* pack_ws, size_t ws_bytes, double* sums /*nullable*/, int n_sums, ddmp_stream stream) {
    ARG_TRY(comm && p && T && C > 0 && (dtype == DDMP_F32 || dtype == DDMP_BF16) && comm->world == p->world && comm->rank == p->rank);
    const int es = dtype == DDMP_BF16 ? 2 : 4;
    ARG_TRY(ld >= C && (C * es) % 16 == 0 && (ld * es) % 16 == 0 && b16_aligned(T) && (n_sums == 0 || sums));
    if (ws_bytes < ddmp_halo_pack_bytes(p, C, dtype) || (p->n_send > 0 && (!pack_ws || !b16_aligned(pack_ws)))) return DDMP_EWORKSPACE;
    if (p->n_cols > p->n_rows && ld != C) return DDMP_EINVAL;    // halo rows are received in place: contiguous rows only
    hipStream_t st = (hipStream_t)stream;
    const size_t row_bytes = (size_t)C * es;
    if (p->n_send > 0) {
        const int q = (int)(row_bytes / 16);
        hipLaunchKernelGGL(halo_pack_kernel, dim3((unsigned)std::min<int64_t>(cdiv(p->n_send * q, 256), 2048)), dim3(256), 0, st,
                           (const uint4*)T, ld * es / 16, p->send_idx, p->n_send, q, (uint4*)pack_ws);
        LAUNCH_TRY();
    }
    if (comm->world == 1 && n_sums == 0 && !comm->loop) return DDMP_OK;
    Rccl& R = rccl();
    NCCL_TRY(R.GroupStart());
    // Inside the group nothing returns early: a return between GroupStart and GroupEnd would leave every later RCCL call
    // of this thread nested in a group that never launches (a hang instead of an error).  The first failure is kept, the
    // remaining calls are skipped, the group is always closed.
    int err = 0;
    int64_t soff = 0, roff = 0;
    for (int r = 0; r < p->world && !err; ++r) {
        const int64_t ns = p->send_counts[r], nr = p->recv_counts[r];
        if (ns > 0) err = R.Send((const char*)pack_ws + soff * row_bytes, (size_t)ns * row_bytes, kNcclInt8, r, comm->comm, st);
        if (nr > 0 && !err)
            err = R.Recv((char*)T + (size_t)(p->n_rows + roff) * row_bytes, (size_t)nr * row_bytes, kNcclInt8, r, comm->comm, st);
        soff += ns;
        roff += nr;
    }
    if (comm->loop && !err) {                                    // (one rank: to and from itself)
        err = R.Send(comm->loop, 256, kNcclInt8, 0, comm->comm, st);
        if (!err) err = R.Recv(comm->loop + 256, 256, kNcclInt8, 0, comm->comm, st);
    }
    if (n_sums > 0 && !err) err = R.AllReduce(sums, sums, (size_t)n_sums, kNcclFloat64, kNcclSum, comm->comm, st);
    const int end = R.GroupEnd();
    if (err) return 1000 + err;
    if (end) return 1000 + end;
    return DDMP_OK;
}

extern "C" int ddmp_comm_allreduce_sum(ddmp_comm* comm, void* buf, int64_t n, int is_f64, ddmp_stream stream) {
    ARG_TRY(comm && buf && n > 0);
    if (comm->world == 1 && !comm->loop) return DDMP_OK;
    NCCL_TRY(rccl().AllReduce(buf, buf, (size_t)n, is_f64 ? kNcclFloat64 : kNcclFloat32, kNcclSum, comm->comm, (hipStream_t)stream));
    return DDMP_OK;
}

extern "C" int ddmp_comm_allgather(ddmp_comm* comm, const void* send, void* recv, int64_t bytes_per_rank, ddmp_stream stream) {
    ARG_TRY(comm && send && recv && bytes_per_rank > 0);
    NCCL_TRY(rccl().AllGather(send, recv, (size_t)bytes_per_rank, kNcclInt8, comm->comm, (hipStream_t)stream));
    return DDMP_OK;
}
