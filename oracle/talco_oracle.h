/*
 * oracle/talco_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, fp32, no FMA contraction) of TWILIGHT's tiled
 * TALCO-XDrop profile-profile aligner, i.e. of
 *   Talco_xdrop::Align_freq   /root/reference/src/TALCO-XDrop.cpp:62-108
 *   Talco_xdrop::Tile         /root/reference/src/TALCO-XDrop.cpp:233-689
 *   Talco_xdrop::Traceback    /root/reference/src/TALCO-XDrop.cpp:134-231
 * in the x86 TALCO_SIMD operation order (CMakeLists.txt:24-27 enables it on
 * x86_64), which is the build whose outputs SURVEY.md/BASELINE.md record.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product path (twilight_amd/) never links it.
 *
 * PARITY STATUS: PINNED end to end.  The reference cannot be compiled in this image (msa.hpp needs
 * Boost.ProgramOptions and TBB headers, neither installed; no stand-ins are written) and ships no tests, so the
 * oracle is pinned against the only outputs of the reference on record, using the reference's
 * own sample data: oracle/e2e_oracle (host mirror + this DP) reproduces dataset/sars_20 -> 20x29705, md5 53ccbd43...,
 * 468765465 band cells, levels 8/4/4/2/1, and dataset/RNASim -> 579x3988, md5 d6a19d18..., 855516114 band cells,
 * 22 levels with the recorded pair counts, max band 685 (tests/test_e2e_pin.py, tests/golden/e2e_expected.json).
 */
#ifndef TWL_TALCO_ORACLE_H
#define TWL_TALCO_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int32_t P;             /* profile width: 6 (nucleotide) or 22 (protein); matrix is (P-1)x(P-1) */
    const float *matrix;   /* row-major scoreMatrix[l][m], l,m in [0,P-1)   (TALCO-XDrop.cpp:36-44) */
    float gap_open;        /* Talco_xdrop::Params::gapOpen      */
    float gap_extend;      /* Talco_xdrop::Params::gapExtend    */
    float gap_char;        /* Talco_xdrop::Params::gapCharScore (alignment-cpu.cpp:88 may zero it) */
    int32_t xdrop;         /* TALCO-XDrop.cpp:49 */
    int32_t flen;          /* TALCO-XDrop.cpp:50 */
    int32_t marker;        /* TALCO-XDrop.cpp:51 */
} twlo_params;

typedef struct {
    uint64_t cells;        /* executions of the body of the i-loop, TALCO-XDrop.cpp:353 */
    uint64_t diags;        /* anti-diagonals entered */
    int32_t  tiles;
    int32_t  max_width;
    uint64_t empty_reduce; /* Reduction_tree called on an all-pruned diagonal (stale read, :586-588) */
    uint64_t oob_diag;     /* ptr==0 with no valid diagonal predecessor (unguarded read, :541) */
    int32_t  err3_reason;  /* debug: which errorType-3 exit fired (1 entry lengths, 2 marker state, 3 start address, 4 negative ref step, 5 lengths after advance, 6 path overflow) */
    int32_t  err3_tile;
} twlo_stats;

/* Optional per-diagonal trace hook (debug aid for the HIP kernel). */
typedef void (*twlo_trace_fn)(void *user, int tile, int k, int L, int U, float max_score_prime);

/*
 * Align one pair.  ref: [R][P] floats, qry: [Q][P] floats, gap arrays per column.
 * aln must hold R+Q bytes.  Returns 0; *err = errorType 0/1/2/3 exactly as the
 * reference (on error *aln_len = 0).  R,Q >= 1.
 */
int twlo_align_pair(const twlo_params *p,
                    const float *ref, int32_t R, const float *qry, int32_t Q,
                    const float *gop_ref, const float *gex_ref,
                    const float *gop_qry, const float *gex_qry,
                    float ref_num, float qry_num,
                    int8_t *aln, int32_t *aln_len, int16_t *err,
                    twlo_stats *stats, twlo_trace_fn trace, void *trace_user);

/*
 * Batch form with the same flat layout as include/twl_align.h (so tests can
 * feed identical buffers to both).  Pairs are spread over `threads` OpenMP
 * threads like tbb::parallel_for at alignment-cpu.cpp:46.  stats (optional) is
 * the sum over pairs.
 */
int twlo_align_batch(const twlo_params *p, int32_t n_pairs, int32_t seq_len,
                     const float *freq, const float *gap_open, const float *gap_extend,
                     const int32_t *len, const int32_t *num,
                     int8_t *aln_out, int32_t *aln_len_out, int16_t *err_out,
                     int32_t threads, twlo_stats *stats);

/* Score of one column pair, exactly as TALCO-XDrop.cpp:373-444 (before adding S). */
float twlo_column_score(const twlo_params *p, const float *ref_col, const float *qry_col, float denom);

#ifdef __cplusplus
}
#endif
#endif
