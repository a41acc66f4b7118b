// twilight_amd/csrc/twl_comm.inc.hip -- RCCL loaded at run time and the library's own communicator (twl_comm_*, include/twl_align.h).  Included inside the extern "C" block.
// Included by twl_align.hip (one translation unit: it shares that file's Device bookkeeping, error string and fill queue).

// ---- RCCL, loaded at run time (include/twl_align.h) ----
namespace {
struct RcclId { char b[TWL_COMM_ID_BYTES]; };      // ncclUniqueId: 128 opaque bytes, passed by value
struct Rccl {
    void *h = nullptr;
    int (*getUniqueId)(void *) = nullptr;
    int (*commInitRank)(void **, int, RcclId, int) = nullptr;
    int (*allGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*commDestroy)(void *) = nullptr;
    const char *(*errorString)(int) = nullptr;
};
Rccl g_rccl;
int rccl_load()
{
    if (g_rccl.h) return TWL_OK;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);      // a copy the process has already (PyTorch-ROCm brings its own)
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { g_err = std::string("librccl not found: ") + (dlerror() ? dlerror() : ""); return TWL_ERR_HIP; }
    Rccl r;
    r.h = h;
    r.getUniqueId = (int (*)(void *))dlsym(h, "ncclGetUniqueId");
    r.commInitRank = (int (*)(void **, int, RcclId, int))dlsym(h, "ncclCommInitRank");
    r.allGather = (int (*)(const void *, void *, size_t, int, void *, hipStream_t))dlsym(h, "ncclAllGather");
    r.commDestroy = (int (*)(void *))dlsym(h, "ncclCommDestroy");
    r.errorString = (const char *(*)(int))dlsym(h, "ncclGetErrorString");
    if (!r.getUniqueId || !r.commInitRank || !r.allGather || !r.commDestroy) { g_err = "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy"; return TWL_ERR_HIP; }
    g_rccl = r;
    return TWL_OK;
}
void comm_destroy_raw(void *comm) { if (g_rccl.commDestroy) (void)g_rccl.commDestroy(comm); }
int rccl_fail(const char *what, int rc)
{
    g_err = std::string(what) + ": " + (g_rccl.errorString ? g_rccl.errorString(rc) : "RCCL error") + " (" + std::to_string(rc) + ")";
    return TWL_ERR_HIP;
}
}  // namespace

int twl_comm_unique_id(void *id128)
{
    if (!id128) { g_err = "bad argument"; return TWL_ERR_BAD_ARGUMENT; }
    std::lock_guard<std::mutex> lk(g_mu);
    int rc = rccl_load();
    if (rc) return rc;
    const int n = g_rccl.getUniqueId(id128);
    return n == 0 ? TWL_OK : rccl_fail("ncclGetUniqueId", n);
}

int twl_comm_init(int device, int rank, int world, const void *id128)
{
    if (!g_init) { g_err = "twl_init not called"; return TWL_ERR_NOT_INITIALIZED; }
    if (!id128 || world < 1 || rank < 0 || rank >= world) { g_err = "bad argument"; return TWL_ERR_BAD_ARGUMENT; }
    Device *d = nullptr;
    int rc = find_dev(device, &d);
    if (rc) return rc;
    { std::lock_guard<std::mutex> lk(g_mu); if ((rc = rccl_load())) return rc; }
    std::lock_guard<std::mutex> dl(d->mu);
    // (one communicator per process and device: a second run of the process -- bench.py opens its handles ahead of the clock -- shares it; every rank does)
    if (d->comm) { if (d->comm_world == world && d->comm_rank == rank) return TWL_OK; g_err = "device already has a communicator of another shape"; return TWL_ERR_BAD_ARGUMENT; }
    HIP_TRY(hipSetDevice(d->id));
    RcclId id;
    memcpy(id.b, id128, sizeof id.b);
    void *comm = nullptr;
    const int n = g_rccl.commInitRank(&comm, world, id, rank);
    if (n != 0) return rccl_fail("ncclCommInitRank", n);
    d->comm = comm; d->comm_world = world; d->comm_rank = rank;
    return TWL_OK;
}

namespace {
// (the device's lock is held by the caller)
int comm_all_gather_locked(Device *d, const void *d_send, void *d_recv, int64_t bytes_per_rank)
{
    if (!d->comm || !d_send || !d_recv || bytes_per_rank <= 0) { g_err = "no communicator on this device (twl_comm_init) or bad argument"; return TWL_ERR_BAD_ARGUMENT; }
    HIP_TRY(hipSetDevice(d->id));
    // on the library's stream: ordered behind the kernels that packed the block, ahead of those that unpack the others'
    const int n = g_rccl.allGather(d_send, d_recv, (size_t)bytes_per_rank, 0 /* ncclChar */, d->comm, d->stream);
    if (n != 0) return rccl_fail("ncclAllGather", n);
    HIP_TRY(hipStreamSynchronize(d->stream));
    return TWL_OK;
}
}  // namespace

int twl_comm_all_gather(int device, const void *d_send, void *d_recv, int64_t bytes_per_rank)
{
    if (!g_init) { g_err = "twl_init not called"; return TWL_ERR_NOT_INITIALIZED; }
    Device *d = nullptr;
    int rc = find_dev(device, &d);
    if (rc) return rc;
    std::lock_guard<std::mutex> dl(d->mu);
    return comm_all_gather_locked(d, d_send, d_recv, bytes_per_rank);
}

int twl_comm_all_gather_host(int device, const void *send, void *recv, int64_t bytes_per_rank)
{
    if (!g_init) { g_err = "twl_init not called"; return TWL_ERR_NOT_INITIALIZED; }
    Device *d = nullptr;
    int rc = find_dev(device, &d);
    if (rc) return rc;
    // ONE hold of the device's lock from staging to copy-back (the staging buffers are the device's, two callers would have raced on them)
    std::lock_guard<std::mutex> dl(d->mu);
    if (!d->comm || !send || !recv || bytes_per_rank <= 0) { g_err = "no communicator on this device (twl_comm_init) or bad argument"; return TWL_ERR_BAD_ARGUMENT; }
    const int world = d->comm_world;
    HIP_TRY(hipSetDevice(d->id));
    if ((rc = d->comm_send.ensure((size_t)bytes_per_rank))) return rc;
    if ((rc = d->comm_recv.ensure((size_t)bytes_per_rank * (size_t)world))) return rc;
    HIP_TRY(hipMemcpyAsync(d->comm_send.p, send, (size_t)bytes_per_rank, hipMemcpyHostToDevice, d->stream));
    if ((rc = comm_all_gather_locked(d, d->comm_send.p, d->comm_recv.p, bytes_per_rank))) return rc;
    HIP_TRY(hipMemcpy(recv, d->comm_recv.p, (size_t)bytes_per_rank * (size_t)world, hipMemcpyDeviceToHost));
    return TWL_OK;
}

int twl_comm_destroy(int device)
{
    if (!g_init) return TWL_OK;
    Device *d = nullptr;
    int rc = find_dev(device, &d);
    if (rc) return rc;
    std::lock_guard<std::mutex> dl(d->mu);
    if (!d->comm) return TWL_OK;
    (void)hipSetDevice(d->id);
    (void)hipStreamSynchronize(d->stream);
    const int n = g_rccl.commDestroy(d->comm);
    d->comm = nullptr; d->comm_world = 0;
    return n == 0 ? TWL_OK : rccl_fail("ncclCommDestroy", n);
}
