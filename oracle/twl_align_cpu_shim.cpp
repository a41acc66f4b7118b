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

#include <cstring>
#include <string>

namespace {
twl_stats g_stats{};
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
int twl_comm_unique_id(void *) { return TWL_ERR_UNSUPPORTED; }
int twl_comm_init(int, int, int, const void *) { return TWL_ERR_UNSUPPORTED; }
int twl_comm_all_gather(int, const void *, void *, int64_t) { return TWL_ERR_UNSUPPORTED; }
int twl_comm_all_gather_host(int, const void *, void *, int64_t) { return TWL_ERR_UNSUPPORTED; }
int twl_comm_destroy(int) { return TWL_ERR_UNSUPPORTED; }
int twl_plan_describe(const twl_params *, int32_t, const int32_t *, int32_t, int32_t, int32_t, char *, int32_t) { return TWL_ERR_UNSUPPORTED; }

}  // extern "C"
