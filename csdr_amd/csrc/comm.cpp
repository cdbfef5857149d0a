// comm.cpp -- the multi-GPU exchange of libcsdr_amd.so: one process per GPU, RCCL over xGMI.
//
// The only part of the hot path with a real exchange step is the fastddc bank (csdr_amd_fastddc_bank_create_sharded): the wideband input lives on
// rank 0, every rank owns a slice of the channels.  What the reference does there is one `csdr fastddc_fwd_cc | nmux` feeding one `csdr fastddc_inv_cc`
// process per client over TCP (ddcd_old.cpp:238-252, 474-492).  Here, per batch of blocks (fastddc_mfma.hip, ddc_mfma_submit):
//   1. scatter: rank 0 sends rank g the samples of ITS windows (blocks [g nbl, (g+1) nbl) plus the overlap in front): world-1 point-to-point
//      transfers in one group, each over its own xGMI link (the mesh is point-to-point: a root broadcast would push the whole input through
//      every one of the root's links: 7 x the bytes per link);
//   2. every rank transforms its blocks (1 / world of the forward work) straight into the fold's layout;
//   3. all-gather of the transposed spectra over the full mesh (every link carries 1 / world of the spectrum, in both directions);
//   4. every rank folds / inverse-transforms its channel slice.
// Per link and batch that is (1 + 1) / world of the data instead of 1: the exchange ceiling rises from one link's rate to ~world/2 x it, and
// it runs on a side stream under the previous batch's fold (submit / collect).
//
// RCCL is loaded on demand (dlopen of librccl.so.1: the copy torch already mapped when the caller is a torch process, the system one otherwise), so
// the single-GPU library and the CLI do not depend on it.
#include "fastddc.hpp"
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>
#include <mutex>

using namespace csdr_amd;

namespace {

typedef int ncclResult_t;
typedef void *ncclComm_t;
struct ncclUniqueId { char internal[128]; };
enum { ncclFloat32 = 7 };                                             // rccl.h ncclDataType_t: ncclFloat = 7

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl; std::mutex g_rccl_mu;

int load_rccl()
{
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.lib) return 0;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
    if (!h) return fail_msg(-6, "cannot load RCCL (librccl.so.1): %s", dlerror());
#define SYM(field, name) do { *(void **)(&g_rccl.field) = dlsym(h, name); if (!g_rccl.field) { dlclose(h); return fail_msg(-6, "RCCL symbol %s missing", name); } } while (0)
    SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
    SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd"); SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv");
    SYM(AllGather, "ncclAllGather"); SYM(Broadcast, "ncclBroadcast"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_rccl.lib = h;
    return 0;
}

int nccl_fail(ncclResult_t r, const char *what) { return fail_msg(-6, "RCCL error %d (%s) in %s", (int)r, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?", what); }
#define CSDR_NCCL(expr) do { ncclResult_t r__ = (expr); if (r__ != 0) return nccl_fail(r__, #expr); } while (0)

} // namespace

struct csdr_amd_comm { csdr_amd_ctx *ctx; ncclComm_t comm; DdcComm ddc; };

namespace {
ncclComm_t nc(const DdcComm *c) { return ((csdr_amd_comm *)c->impl)->comm; }
int c_group_start(const DdcComm *) { CSDR_NCCL(g_rccl.GroupStart()); return 0; }
int c_group_end(const DdcComm *) { CSDR_NCCL(g_rccl.GroupEnd()); return 0; }
int c_send(const DdcComm *c, const void *buf, size_t n, int peer, hipStream_t st) { CSDR_NCCL(g_rccl.Send(buf, n, ncclFloat32, peer, nc(c), st)); return 0; }
int c_recv(const DdcComm *c, void *buf, size_t n, int peer, hipStream_t st) { CSDR_NCCL(g_rccl.Recv(buf, n, ncclFloat32, peer, nc(c), st)); return 0; }
int c_all_gather(const DdcComm *c, void *all, size_t n, hipStream_t st)
{
    if (c->world == 1) return 0;                                      // in place: nothing moves
    CSDR_NCCL(g_rccl.AllGather((const char *)all + (size_t)c->rank * n * sizeof(float), all, n, ncclFloat32, nc(c), st));
    return 0;
}
} // namespace

namespace csdr_amd { const DdcComm *csdr_amd_comm_ddc(csdr_amd_comm *c) { return c ? &c->ddc : nullptr; } }

extern "C" {

int csdr_amd_comm_unique_id(char id128[128])
{
    int rc = load_rccl(); if (rc) return rc;
    ncclUniqueId id; CSDR_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(id128, id.internal, 128);
    return 0;
}

csdr_amd_comm *csdr_amd_comm_create(csdr_amd_ctx *ctx, const char id128[128], int rank, int world)
{
    if (world < 1 || rank < 0 || rank >= world) { fail_msg(-3, "comm: bad rank %d of %d", rank, world); return nullptr; }
    if (load_rccl()) return nullptr;
    if (hipSetDevice(ctx->device) != hipSuccess) { fail_msg(-1, "hipSetDevice(%d) failed", ctx->device); return nullptr; }
    csdr_amd_comm *c = new csdr_amd_comm();
    c->ctx = ctx; c->comm = nullptr;
    ncclUniqueId id; memcpy(id.internal, id128, 128);
    const ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
    if (r != 0) { nccl_fail(r, "ncclCommInitRank"); delete c; return nullptr; }
    c->ddc.rank = rank; c->ddc.world = world; c->ddc.impl = c;
    c->ddc.group_start = c_group_start; c->ddc.group_end = c_group_end; c->ddc.send = c_send; c->ddc.recv = c_recv; c->ddc.all_gather = c_all_gather;
    return c;
}

void csdr_amd_comm_destroy(csdr_amd_comm *c)
{
    if (!c) return;
    (void)hipStreamSynchronize(c->ctx->stream);
    if (c->comm) (void)g_rccl.CommDestroy(c->comm);
    delete c;
}

int csdr_amd_comm_rank(const csdr_amd_comm *c) { return c->ddc.rank; }
int csdr_amd_comm_world(const csdr_amd_comm *c) { return c->ddc.world; }

/* broadcast of a device buffer from `root` on the context's stream (bench / test plumbing: e.g. checking a sharded run against rank 0's data) */
int csdr_amd_comm_broadcast(csdr_amd_comm *c, void *dev_buf, size_t bytes, int root)
{
    if (c->ddc.world == 1) return 0;
    CSDR_NCCL(g_rccl.Broadcast(dev_buf, dev_buf, bytes, 1 /* ncclUint8 */, root, c->comm, c->ctx->stream));
    return 0;
}

} // extern "C"
