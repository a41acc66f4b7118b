/*
 * include/twl_align.h -- C ABI of libtwl_align: the MI355X (gfx950) level-batch aligner.
 *
 * TWILIGHT has no FFI: its boundary is the C++ std::function
 *     msa::alnFunction  = void(Tree*, NodePairVec&, SequenceDB*, Option*, Params&)   (reference src/msa.hpp:175)
 * whose CPU implementation cpu::parallelAlignmentCPU (src/alignment-cpu.cpp:36-183) calls, per pair,
 *     Talco_xdrop::Align_freq(Params*, freqRef, freqQry, gapOp, gapEx, num, aln, errorType)   (src/TALCO-XDrop.hpp:56-65).
 * This header is the flat, batched equivalent of that inner call for all pairs of one guide-tree
 * level; its array shapes are the ones the reference's own GPU host code already flattens to
 * (src/hip/alignment-gpu.hip.cpp:270-293: freq[pairs][2][seqLen][P], gapOpen/gapExtend[pairs][2][seqLen],
 * len[2*pairs], num[2*pairs]; out aln[pairs][2*seqLen], alnLen[pairs]).  The host-side mirror of
 * alignmentKernel_GPU that calls it lives in twilight_amd/csrc/host/ (see INTEGRATION.md).
 *
 * Plain C types only.  All functions return 0 on success or a negative twl_status; algorithmic
 * failures of a pair are NOT errors: they travel in err_out[] as the reference's errorType so the
 * caller can apply the reference's retry/defer policy unchanged (alignment-cpu.cpp:108-129).
 *
 * Threads.  The reference calls its per-pair kernel from TBB workers (alignment-cpu.cpp:46) and its level kernel with one pair
 * per level from the deferred / merge paths (progressive.cpp:286-291), so every entry point of this header and of twl_level.h
 * may be called from any host thread, concurrently.  What is serialised: calls that touch the same device take that device's
 * lock for their whole duration (a call is a whole level batch and fills the device by itself; twl_level_* calls of different
 * stores on one device interleave call by call, in the order the lock grants); calls on different devices run side by side.
 * What is per thread: twl_last_error().  What is per device and therefore belongs to the LAST call of any thread:
 * twl_get_stats / twl_get_pair_cells -- concurrent callers read their results from their own output arrays.  twl_init /
 * twl_shutdown / twl_set_knob are not meant to race with running calls.  tests/test_gpu_threads.py.
 */
#ifndef TWL_ALIGN_H
#define TWL_ALIGN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TWL_MAX_MATRIX 21          /* matrixSize: 5 nucleotide, 21 protein (msa.hpp:98-109) */
#define TWL_MAX_MARKER 1024        /* Talco_xdrop::Params::marker default, TALCO-XDrop.cpp:51 */

enum twl_status {
    TWL_OK = 0,
    TWL_ERR_NOT_INITIALIZED = -1,
    TWL_ERR_BAD_ARGUMENT = -2,
    TWL_ERR_HIP = -3,              /* a HIP runtime call failed; see twl_last_error() */
    TWL_ERR_UNSUPPORTED = -4       /* parameter outside what the kernels implement */
};

/* Replaces Talco_xdrop::Params (TALCO-XDrop.hpp:37-54, ctor TALCO-XDrop.cpp:36-53). */
typedef struct twl_params {
    int32_t P;                                       /* profile width = matrixSize+1: 6 or 22 */
    float   matrix[TWL_MAX_MATRIX * TWL_MAX_MATRIX]; /* scoreMatrix[l][m] row-major, (P-1)x(P-1) used */
    float   gap_open;                                /* gapOpen */
    float   gap_extend;                              /* gapExtend */
    float   gap_boundary;                            /* gapBoundary (unused by Align_freq when alnType==0; kept for layout parity) */
    float   gap_char;                                /* gapCharScore: gapExtend, or 0 per alignment-cpu.cpp:88 */
    int32_t xdrop;                                   /* 1000*-gapExtend by default (TALCO-XDrop.cpp:49) */
    int32_t flen;                                    /* max anti-diagonal width, default 4096 (:50) */
    int32_t marker;                                  /* tile marker, default 1024 (:51); 2 <= marker <= TWL_MAX_MARKER */
} twl_params;

/* Counters of the most recent twl_align_batch* call on the calling device. */
typedef struct twl_stats {
    uint64_t band_cells;      /* executions of the i-loop body, TALCO-XDrop.cpp:353, summed over pairs/tiles/relaunches */
    uint64_t nominal_cells;   /* sum of R*Q */
    double   kernel_ms;       /* HIP-event time of the DP kernel launches only */
    double   pack_ms;         /* HIP-event time of the column-packing kernel */
    double   total_ms;        /* HIP-event time first enqueue -> last completion (incl. H2D/D2H for the host form) */
    int32_t  n_launches;      /* DP kernel launches (1 + relaunches for pairs that overflowed the fast window) */
    int32_t  n_relaunched;    /* pairs re-run with the wide-window kernel */
    int32_t  window;          /* rows of the fast-path window used */
    int32_t  grid;            /* persistent workgroups launched */
    int32_t  matrix_mode;     /* column-score mode of the first DP launch: nucleotide 0 / 1 / 2 / 5 (one-letter query rows), protein 3 / 4 */
    int32_t  speculative;     /* two workgroups per pair with speculative tile start: 1 on a CU each (16 waves), 2 two to a CU (8 waves x 2 blocks);
                                 3 tile-parallel (all tiles of all pairs side by side from predicted starts); 0 none */
    int32_t  mt_tiles_predicted;   /* tile-parallel launches: tiles whose predicted start was the true one (taken from the parallel launch) */
    int32_t  mt_tiles_inline;      /* ... tiles computed in line by the stitch launch (prediction missing or wrong) */
    int32_t  mt_scouts_failed;     /* ... scouts that produced no path sample */
    int32_t  reserved;
    char     kernel[160];          /* the DP kernel of the first launch, as the profiler names it (template arguments spelled out) */
} twl_stats;

/* Select devices (HIP ordinals).  n_devices==0 or device_ids==NULL -> device 0 only.
   With several devices the pairs of a batch are dealt to them in cost order (no collective). */
int  twl_init(const int *device_ids, int n_devices);
/* Releases every device buffer and stream.  Stores of twl_level.h must have been destroyed before: with any of them alive the call
   is refused (a message on stderr) and the library stays initialised. */
void twl_shutdown(void);
const char *twl_last_error(void);
const char *twl_version(void);

/*
 * Align every pair of one level batch.  Host pointers; the library stages through its own device
 * buffers.  Replaces the per-pair Align_freq calls inside the tbb::parallel_for at
 * alignment-cpu.cpp:46-134.
 *
 *   freq        [n_pairs][2][seq_len][P]  weighted letter counts, 0 = reference, 1 = query, zero padded
 *   gap_open    [n_pairs][2][seq_len]     position-specific gap open   (calculatePSGP)
 *   gap_extend  [n_pairs][2][seq_len]     position-specific gap extend
 *   len         [n_pairs][2]              R, Q (after gappy-column removal)
 *   num         [n_pairs][2]              member-sequence counts; denominator = float(R_num)*float(Q_num)
 *   aln_out     [n_pairs][2*seq_len]      path codes 0 (both) / 1 (gap in ref, consumes query) / 2 (gap in query), forward order;
 *                                         bytes past aln_len_out[i] in row i are unspecified
 *   aln_len_out [n_pairs]                 path length; 0 if an input side is empty (caller emits the all-gap path,
 *                                         alignment-cpu.cpp:89-90) or if err_out != 0
 *   err_out     [n_pairs]                 errorType 0 ok / 1 X-drop emptied the band / 2 band wider than flen / 3 inconsistency
 */
int twl_align_batch(const twl_params *p, int32_t n_pairs, int32_t seq_len,
                    const float *freq, const float *gap_open, const float *gap_extend,
                    const int32_t *len, const int32_t *num,
                    int8_t *aln_out, int32_t *aln_len_out, int16_t *err_out);

/*
 * Same, with every array already resident in device memory of `device` (HBM in, HBM out; no PCIe
 * traffic).  `stream` is a hipStream_t (NULL = the library's stream for that device).  The call
 * returns after the work completed.  The library orders its work on `stream` only: with stream == NULL
 * every producer of the input buffers must have completed before the call (e.g. torch.cuda.synchronize()
 * after building them with torch); with the producers' own stream passed in, stream order suffices.
 */
int twl_align_batch_device(int device, void *stream, const twl_params *p, int32_t n_pairs, int32_t seq_len,
                           const float *d_freq, const float *d_gap_open, const float *d_gap_extend,
                           const int32_t *d_len, const int32_t *d_num,
                           int8_t *d_aln_out, int32_t *d_aln_len_out, int16_t *d_err_out);

/* Page-locked host memory for a caller's staging buffers (paths out, final paths in): copies to and from it run at link speed instead of
   through the runtime's bounce buffers.  twl_host_alloc returns NULL when the allocation fails (the caller may fall back to malloc). */
void *twl_host_alloc(uint64_t bytes);
void  twl_host_free(void *p);

/* Plain synchronous copies between host memory and memory of `device` (for callers that hold device pointers of the library -- the exchange
   blocks of twl_level.h -- but do not link the HIP runtime themselves).  2-D form: `rows` rows of `width` bytes, pitches in bytes. */
int twl_copy_to_device(int device, void *dst_dev, const void *src, uint64_t bytes);
int twl_copy_from_device(int device, void *dst, const void *src_dev, uint64_t bytes);
int twl_copy_rows_from_device(int device, void *dst, uint64_t dst_pitch, const void *src_dev, uint64_t src_pitch, uint64_t width, uint64_t rows);

int twl_get_stats(int device, twl_stats *out);

/* Per-pair band-cell counts of the last call on `device` (n entries, host buffer). */
int twl_get_pair_cells(int device, uint64_t *cells_out, int32_t n);

/* Diagnostics: the column scores similarScore(i, j) (TALCO-XDrop.cpp:444: profile x matrix sum of pairs divided by refNum*qryNum) of ONE
   pair for every query row i and reference column j, row-major out[len[1]][len[0]], as the kernels compute them.
   freq is [2][seq_len][P] (0 = reference, 1 = query), len = {R, Q}, num = {refNum, qryNum}. */
int twl_column_scores(const twl_params *p, int32_t seq_len, const float *freq, const int32_t *len, const int32_t *num, float *out);

/* Diagnostics: the same column scores as the DP KERNEL ITSELF evaluated them while aligning this ONE pair -- a diagnostics instantiation of
   the round-2 kernel (nucleotide: the matrix mode the parameters select, 0 general 5x5 / 1 zero N row and column / 2 match-transition-
   transversion; protein: the sparse in-kernel loop) that also stores the score of every cell the band visits.  out[len[1]][len[0]]; cells
   the band never visited hold NaN.  Arrays as one pair of twl_align_batch. */
int twl_dp_column_scores(const twl_params *p, int32_t seq_len, const float *freq, const float *gap_open, const float *gap_extend,
                         const int32_t *len, const int32_t *num, float *out);

/* ---- RCCL from the library's own C++ (pairs of a level sharded over the GPUs of a node, one process per GPU) ----
   The collective of a sharded run is ONE all-gather of equal byte blocks per guide-tree level, HBM to HBM over xGMI, on the library's stream of
   `device`.  librccl is loaded at run time (dlopen "librccl.so.1"; a copy already in the process, e.g. PyTorch's, is reused), so the library
   has no link-time dependency on it and single-GPU use never touches it.  Replaces the batch dealing of the reference's GPU host code
   (/root/reference/src/hip/alignment-gpu.hip.cpp:239-254) across processes.
     twl_comm_unique_id   ncclGetUniqueId: call on ONE rank, hand the 128 bytes to the others by any means (a pipe, a file, a launcher's store)
     twl_comm_init        ncclCommInitRank for this process's `device` (selected in twl_init); collective: every rank calls it
     twl_comm_all_gather  d_recv[world][bytes_per_rank] <- every rank's d_send[bytes_per_rank] (device pointers); returns when the blocks are there
     twl_comm_all_gather_host  the same for host blocks (staged through the library's device buffers)
     twl_comm_destroy     ncclCommDestroy
   All return TWL_OK or TWL_ERR_HIP with twl_last_error() set (the RCCL error string). */
#define TWL_COMM_ID_BYTES 128
int twl_comm_unique_id(void *id128);
int twl_comm_init(int device, int rank, int world, const void *id128);
int twl_comm_all_gather(int device, const void *d_send, void *d_recv, int64_t bytes_per_rank);
int twl_comm_all_gather_host(int device, const void *send, void *recv, int64_t bytes_per_rank);
int twl_comm_destroy(int device);

/* Development / test knobs of the launch policy (process-wide; not needed in production).  Returns TWL_ERR_BAD_ARGUMENT for an unknown key.
     TWL_KNOB_MT_PERTURB     n > 0: spoil every n-th predicted tile start of the tile-parallel path, so that its stitch launch has tiles to
                             compute in line (tests of that path); 0 = off (default)
     TWL_KNOB_MT_MAX_PAIRS   levels with at most this many pairs may take the tile-parallel path (default 1024; 0 = never): all levels of up to CUs/2 pairs, larger
                             ones (up to 2 * CUs, one round of the throughput kernel) when they would fill that round badly
     TWL_KNOB_MT_MIN_MARKER  ... and only with marker >= this (default 512)
     TWL_KNOB_MT_LEAD        anti-diagonals a scout starts ahead of its tile boundary (default 320)
     TWL_KNOB_MT_MARGIN      anti-diagonals a scout runs past its tile boundary (default 40)
     TWL_KNOB_MT_ROUNDS      rounds of predict / run / verify before the remaining tiles are computed in line (default 2, at most 7)
     TWL_KNOB_MT_THR_JOBS    levels with more tiles than this run scouts and tiles on the throughput geometry (default 256 = the CUs)
     TWL_KNOB_FAIL_ROW_ALLOCS  the next n device allocations for the row planes of a store (include/twl_level.h) fail: tests of the
                             fallback to the minimal pitch
     TWL_KNOB_PROT_MODE      force a protein kernel variant: 0 auto (default), 1 dense, 2 sparse, 3 precomputed scores, 4 the round-1 kernel,
                             5 lean sparse, 6 lean precomputed -- every variant computes the same sums (tests hold each to the oracle)
     TWL_KNOB_ASSUME_ONEHOT_QUERY  1: the caller promises that every query row of twl_align_batch has at most one non-zero letter
                             (single sequences), which selects the four-product column score; the device-resident level path
                             (twl_level.h) knows this by itself
     TWL_KNOB_MT_TAIL_PCT    levels of more than 2 * CUs pairs: a last round of the throughput kernel that would be filled to at most this
                             share (percent, default 70) runs on the tile-parallel path instead, behind the full rounds, when its pairs have 8 or more
                             tiles each (0 = never)
     TWL_KNOB_MT_WIDE        1 (default): pairs whose band outgrew the 1024-row window re-run with all their tiles at once on the 3072-row geometry;
                             0: tile after tile on the 2048-row kernel (the path before round 4; tests hold the two to each other)
     TWL_KNOB_NO_SPEC        1: no speculative two-workgroup teams for levels of a few pairs outside the tile-parallel path (timing tools)
     TWL_KNOB_SCOUT_XDROP_PCT  X-drop of the pair scouts of wide re-runs in percent of the call's (default 100; 10..100): they only predict tile starts
     TWL_KNOB_THR_SMALL      0 (default): a level of more pairs than CUs starts on the 512-row throughput geometry (4 waves x 2 blocks, FIVE workgroups per CU) unless
                             an earlier level of the pass outgrew it -- large levels of short pairs ask a sample of their own pairs first; pairs that outgrow it re-run
                             (768-row geometry, a few long pairs tile-parallel); scouts and, while the pass fits it, tile jobs of tile-parallel launches take it too.
                             1: never; 2: every throughput level does (tests, tools).  Setting the knob also forgets what earlier levels found
     TWL_KNOB_FORCE_GLOBAL   1: every pair of every call runs on the global-memory kernel, the last stage of the re-run chain (bands of any width; normally only
                             reached by bands beyond 4480 rows, i.e. fLen > 4608 in a retry of the deferred pass): tests of that kernel on small cases
     TWL_KNOB_LEAF_STEP      1 (default): device-resident levels (twl_level.h) whose pairs are all single sequences on both sides run the throughput kernels' step
                             without the gap-letter terms, the division and their per-block tests; 0: the general step (tests hold the two to each other)
     TWL_KNOB_POISON_TB      1: the traceback scratch is filled with 0xFF bytes in front of every DP launch.  A block writes its traceback word only when it can have
                             held band cells in the group of 8 diagonals; a word wrongly skipped would read as the zeros of a fresh allocation (a "match" pointer,
                             often right) -- with the knob it reads as garbage and the parity tests see it.  tests/conftest.py sets it for every GPU test
     TWL_KNOB_MT_ANCHOR      1 (default): the scouts of the tile-parallel path (nucleotide and protein: the most frequent letter of a column) start from the cell on which the consensus letters of the two profiles agree
                             (one small kernel per level finds it for every tile boundary), TWL_KNOB_MT_LEAD2 (default 96) anti-diagonals ahead of the boundary, where
                             that cell is trusted; elsewhere, and with 0 everywhere, from the straight line between the corners TWL_KNOB_MT_LEAD ahead.  Predictions
                             only: the results are the same either way (tests hold the two to each other)
     TWL_KNOB_PROT_CORRIDOR  rows (default 448): protein levels of few pairs precompute their column scores (matrix mode 4) only within this many rows of the straight line
                             between the corners of a pair's matrix; a pair whose band leaves that corridor is re-run by the kernel that scores in line.  0: the whole matrix */
enum twl_knob { TWL_KNOB_MT_PERTURB = 1, TWL_KNOB_MT_MAX_PAIRS = 2, TWL_KNOB_MT_MIN_MARKER = 3, TWL_KNOB_MT_LEAD = 4, TWL_KNOB_MT_MARGIN = 5,
                TWL_KNOB_MT_ROUNDS = 6, TWL_KNOB_MT_THR_JOBS = 7, TWL_KNOB_FAIL_ROW_ALLOCS = 8,
                TWL_KNOB_PROT_MODE = 9, TWL_KNOB_ASSUME_ONEHOT_QUERY = 10, TWL_KNOB_MT_TAIL_PCT = 11, TWL_KNOB_MT_WIDE = 12, TWL_KNOB_NO_SPEC = 13, TWL_KNOB_SCOUT_XDROP_PCT = 14, TWL_KNOB_THR_SMALL = 15, TWL_KNOB_FORCE_GLOBAL = 16, TWL_KNOB_LEAF_STEP = 17, TWL_KNOB_POISON_TB = 18, TWL_KNOB_MT_ANCHOR = 19, TWL_KNOB_MT_LEAD2 = 20, TWL_KNOB_PROT_CORRIDOR = 21 };
int twl_set_knob(int key, int value);
/* The launch plan of a nucleotide call in words ("throughput; mode 2; window 768; bulk 1024 tail 277"), made by the very function the launch path
   uses, without touching a device: len[n_pairs][2] as twl_align_batch, num_cu / qry_onehot / wide_streak the facts the device would supply. */
int twl_plan_describe(const twl_params *p, int32_t n_pairs, const int32_t *len, int32_t num_cu, int32_t qry_onehot, int32_t wide_streak, char *out, int32_t cap);

#ifdef __cplusplus
}
#endif
#endif
