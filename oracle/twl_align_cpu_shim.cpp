// oracle/twl_align_cpu_shim.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// The entry points of include/twl_align.h that the host-staged level kernel calls, answered by the CPU oracle, so that the
// multi-process orchestration of libtwl_host (deal the pairs of a level to the ranks, align the own share, all-gather the paths,
// twilight_amd/csrc/host/align_gpu.cpp) can be driven by 2 gloo ranks on a box without a GPU (tests/test_dist_cpu.py).  Linked only
// into oracle/libtwl_host_cpucheck.so; the product libraries never see it.  The device-resident entry points (twl_level.h) have no
// CPU form: they fail.
#include "../include/twl_align.h"
#include "../include/twl_level.h"
#include "talco_oracle.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fcntl.h>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {
twl_stats g_stats{};
// shared-memory communicator (see twl_comm_* below)
constexpr int kShmMaxWorld = 8;
constexpr size_t kShmSlot = (size_t)32 << 20;      // bytes a rank may contribute to one all-gather
struct ShmHead { std::atomic<int> arrived; std::atomic<int> phase; };
struct { ShmHead *seg = nullptr; size_t bytes = 0; int rank = 0, world = 1; char name[TWL_COMM_ID_BYTES] = {0}; } g_comm;
void shm_barrier()
{
    ShmHead *h = g_comm.seg;
    const int ph = h->phase.load();
    if (h->arrived.fetch_add(1) + 1 == g_comm.world) { h->arrived.store(0); h->phase.store(ph + 1); }
    else while (h->phase.load() == ph) usleep(50);
}
const char *kNoDevice = "CPU check build: the device-resident level path needs the GPU library";
}  // namespace

extern "C" {

const char *twl_last_error(void) { return kNoDevice; }
const char *twl_version(void) { return "twilight_amd CPU check shim (oracle)"; }
int twl_init(const int *, int) { return TWL_OK; }
void twl_shutdown(void) {}

int twl_align_batch(const twl_params *p, int32_t n_pairs, int32_t seq_len, const float *freq, const float *gap_open, const float *gap_extend,
                    const int32_t *len, const int32_t *num, int8_t *aln_out, int32_t *aln_len_out, int16_t *err_out)
{
    twlo_params op;
    op.P = p->P; op.matrix = p->matrix; op.gap_open = p->gap_open; op.gap_extend = p->gap_extend; op.gap_char = p->gap_char;
    op.xdrop = p->xdrop; op.flen = p->flen; op.marker = p->marker;
    twlo_stats st;
    memset(&st, 0, sizeof st);
    twlo_align_batch(&op, n_pairs, seq_len, freq, gap_open, gap_extend, len, num, aln_out, aln_len_out, err_out, 2, &st);
    g_stats = twl_stats{};
    g_stats.band_cells = st.cells;
    g_stats.n_launches = 1;
    return TWL_OK;
}
int twl_get_stats(int, twl_stats *out) { *out = g_stats; return TWL_OK; }
int twl_align_batch_device(int, void *, const twl_params *, int32_t, int32_t, const float *, const float *, const float *, const int32_t *,
                           const int32_t *, int8_t *, int32_t *, int16_t *) { return TWL_ERR_UNSUPPORTED; }
void *twl_host_alloc(uint64_t) { return nullptr; }      // (callers fall back to malloc)
void twl_host_free(void *) {}
int twl_get_pair_cells(int, uint64_t *, int32_t) { return TWL_ERR_UNSUPPORTED; }
int twl_column_scores(const twl_params *, int32_t, const float *, const int32_t *, const int32_t *, float *) { return TWL_ERR_UNSUPPORTED; }
int twl_dp_column_scores(const twl_params *, int32_t, const float *, const float *, const float *, const int32_t *, const int32_t *, float *) { return TWL_ERR_UNSUPPORTED; }

int twl_store_create(int, char, int32_t, const char *const *, const int32_t *, twl_store **) { return TWL_ERR_UNSUPPORTED; }
void twl_store_destroy(twl_store *) {}
int twl_store_read_rows(twl_store *, char *const *, int32_t *) { return TWL_ERR_UNSUPPORTED; }
int twl_store_read_cache(twl_store *, int32_t, float *, int32_t *) { return TWL_ERR_UNSUPPORTED; }
int twl_store_drop_cache(twl_store *, int32_t) { return TWL_ERR_UNSUPPORTED; }
int twl_level_prepare(twl_store *, const twl_params *, float, int32_t, const twl_side *, const int32_t *, const float *, int32_t, int32_t *, uint8_t *) { return TWL_ERR_UNSUPPORTED; }
int twl_level_read_colinfo(twl_store *, int32_t, int32_t, uint8_t *) { return TWL_ERR_UNSUPPORTED; }
int twl_level_align(twl_store *, const twl_params *, const uint8_t *, int8_t *, int32_t *, int16_t *) { return TWL_ERR_UNSUPPORTED; }
int twl_level_align_mixed(twl_store *, const twl_params *, const uint8_t *, const uint8_t *, int8_t *, int32_t *, int16_t *) { return TWL_ERR_UNSUPPORTED; }
int twl_level_commit(twl_store *, const int8_t *, const int32_t *, int32_t) { return TWL_ERR_UNSUPPORTED; }
int twl_level_commit_from_dp(twl_store *, const int8_t *, const int32_t *, int32_t, const uint8_t *) { return TWL_ERR_UNSUPPORTED; }
int twl_level_read_path(twl_store *, int32_t, int8_t *, int32_t) { return TWL_ERR_UNSUPPORTED; }
int twl_level_read_paths(twl_store *, int32_t, const int32_t *, const int32_t *, int8_t *, int32_t) { return TWL_ERR_UNSUPPORTED; }
int twl_level_read_colinfo_many(twl_store *, int32_t, const int32_t *, uint8_t *) { return TWL_ERR_UNSUPPORTED; }
int twl_level_read_columns(twl_store *, int32_t, int32_t, float *, int32_t) { return TWL_ERR_UNSUPPORTED; }
int twl_level_timing(twl_store *, double *, double *) { return TWL_ERR_UNSUPPORTED; }
int twl_level_restore(twl_store *, const twl_params *, int32_t, const int32_t *, int32_t, int32_t *) { return TWL_ERR_UNSUPPORTED; }
int twl_level_read_final(twl_store *, int32_t, int8_t *, int32_t) { return TWL_ERR_UNSUPPORTED; }
int twl_level_exchange_buffers(twl_store *, int64_t, int64_t, void **, void **) { return TWL_ERR_UNSUPPORTED; }
int twl_level_paths_to_block(twl_store *, int32_t, const int32_t *, const int32_t *, const uint8_t *, void *, const int64_t *) { return TWL_ERR_UNSUPPORTED; }
int twl_level_paths_from_block(twl_store *, int32_t, const int32_t *, const int32_t *, const void *, const int64_t *) { return TWL_ERR_UNSUPPORTED; }
int twl_level_write_final(twl_store *, int32_t, const int8_t *, int32_t) { return TWL_ERR_UNSUPPORTED; }
int twl_copy_to_device(int, void *, const void *, uint64_t) { return TWL_ERR_UNSUPPORTED; }
int twl_copy_from_device(int, void *, const void *, uint64_t) { return TWL_ERR_UNSUPPORTED; }
int twl_copy_rows_from_device(int, void *, uint64_t, const void *, uint64_t, uint64_t, uint64_t) { return TWL_ERR_UNSUPPORTED; }
int twl_set_knob(int, int) { return TWL_ERR_UNSUPPORTED; }
int twl_store_read_rows_of(twl_store *, int32_t, const int32_t *, char *, int32_t *) { return TWL_ERR_UNSUPPORTED; }
int twl_store_write_rows(twl_store *, int32_t, const int32_t *, const char *, const int32_t *) { return TWL_ERR_UNSUPPORTED; }
int twl_store_write_cache(twl_store *, int32_t, const float *, int32_t) { return TWL_ERR_UNSUPPORTED; }
int twl_store_rows_to_block(twl_store *, int32_t, const int32_t *, void *, int32_t *) { return TWL_ERR_UNSUPPORTED; }
int twl_store_rows_from_block(twl_store *, int32_t, const int32_t *, const int32_t *, const void *) { return TWL_ERR_UNSUPPORTED; }
int twl_store_exchange_buffers(twl_store *, int64_t, int64_t, void **, void **) { return TWL_ERR_UNSUPPORTED; }
// The communicator of a sharded run (include/twl_align.h: RCCL in the product) as a POSIX shared-memory segment between the processes of ONE box: what lets the forked
// CLI (twilight_amd/csrc/host/main.cpp: fork, shared page, id hand-over, watchdog) and the "the library's own collective" branch of the host-staged level kernel run
// end to end on a box without a GPU (tests/test_abi_cpu.py).  The id is the segment's name; an all-gather is copy in, barrier, copy out, barrier.
int twl_comm_unique_id(void *id128)
{
    char *id = static_cast<char *>(id128);
    memset(id, 0, TWL_COMM_ID_BYTES);
    snprintf(id, TWL_COMM_ID_BYTES, "/twl_cpucheck_%d_%ld", (int)getpid(), (long)time(nullptr));
    return TWL_OK;
}
int twl_comm_init(int, int rank, int world, const void *id128)
{
    if (g_comm.seg) return TWL_OK;
    if (world < 1 || world > kShmMaxWorld || rank < 0 || rank >= world) return TWL_ERR_BAD_ARGUMENT;
    const char *name = static_cast<const char *>(id128);
    const size_t bytes = sizeof(ShmHead) + (size_t)world * kShmSlot;
    int fd = -1;
    if (rank == 0) {
        fd = shm_open(name, O_CREAT | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) return TWL_ERR_HIP;
    } else {
        for (int t = 0; t < 20000 && fd < 0; ++t) { fd = shm_open(name, O_RDWR, 0600); if (fd < 0) usleep(500); }      // (rank 0 creates it)
        if (fd < 0) return TWL_ERR_HIP;
        struct stat sb;
        for (int t = 0; t < 20000; ++t) { if (fstat(fd, &sb) == 0 && (size_t)sb.st_size >= bytes) break; usleep(500); }
    }
    void *m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return TWL_ERR_HIP;
    g_comm.seg = static_cast<ShmHead *>(m); g_comm.bytes = bytes; g_comm.rank = rank; g_comm.world = world;
    snprintf(g_comm.name, sizeof g_comm.name, "%s", name);
    shm_barrier();          // everybody has mapped it: the name can go
    if (rank == 0) shm_unlink(name);
    return TWL_OK;
}
int twl_comm_all_gather_host(int, const void *send, void *recv, int64_t bytes)
{
    if (!g_comm.seg || bytes < 0 || (size_t)bytes > kShmSlot) return TWL_ERR_BAD_ARGUMENT;
    // test aid: TWL_CPUCHECK_DIE_AT=<n> makes rank 1 die inside its n-th all-gather, the way a rank dies inside a collective: the others are left in the barrier
    // and only the CLI's watchdog can end the run (tests/test_abi_cpu.py)
    static int calls = 0;
    if (const char *e = getenv("TWL_CPUCHECK_DIE_AT")) { if (g_comm.rank == 1 && ++calls == atoi(e)) _exit(3); }
    char *slots = reinterpret_cast<char *>(g_comm.seg + 1);
    memcpy(slots + (size_t)g_comm.rank * kShmSlot, send, (size_t)bytes);
    shm_barrier();
    for (int r = 0; r < g_comm.world; ++r) memcpy(static_cast<char *>(recv) + (size_t)r * (size_t)bytes, slots + (size_t)r * kShmSlot, (size_t)bytes);
    shm_barrier();
    return TWL_OK;
}
int twl_comm_all_gather(int d, const void *send, void *recv, int64_t bytes) { return twl_comm_all_gather_host(d, send, recv, bytes); }      // ("device" memory is host memory here)
int twl_comm_destroy(int) { if (g_comm.seg) { munmap(g_comm.seg, g_comm.bytes); g_comm.seg = nullptr; } return TWL_OK; }
int twl_plan_describe(const twl_params *, int32_t, const int32_t *, int32_t, int32_t, int32_t, char *, int32_t) { return TWL_ERR_UNSUPPORTED; }

}  // extern "C"
