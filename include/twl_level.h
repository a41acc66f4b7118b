/*
 * include/twl_level.h -- C ABI of libtwl_align, part 2: device-resident level processing.
 *
 * twl_align.h replaces the per-pair Talco_xdrop::Align_freq calls of a level.  This header moves the work the
 * reference does AROUND those calls for every pair (reference src/alignment-cpu.cpp:50-93 before, :136-175
 * after) onto the device as well, with the sequences kept resident in HBM for the whole progressive pass, so
 * that per level only paths and one byte per profile column cross PCIe:
 *
 *   twl_level_prepare  alignment_helper::calculateProfile      (src/alignment-helper.cpp:8-72)
 *                      alignment_helper::getConsensus          (:221-241)
 *                      alignment_helper::removeGappyColumns    (:74-166)
 *                      alignment_helper::calculatePSGP         (:168-219)
 *   twl_level_align    the Align_freq calls                    (src/alignment-cpu.cpp:95-130), on the prepared columns
 *   twl_level_commit   alignment_helper::updateFrequency       (:506-539)
 *                      alignment_helper::updateAlignment       (:377-503), row rewriting part
 *
 * What stays with the caller (small, serial, policy): the retry/defer policy, addGappyColumnsBack + pairwiseGlobal
 * on the returned paths (:243-375), and the Node bookkeeping (alnNum/alnLen/alnWeight/seqsIncluded).
 *
 * A store lives on ONE device (several devices: one replica per device) and must be destroyed before
 * twl_shutdown().  Same conventions as twl_align.h: plain C types, 0 or a negative twl_status,
 * twl_last_error() for the text; algorithmic failures of a pair travel in err_out[].
 */
#ifndef TWL_LEVEL_H
#define TWL_LEVEL_H

#include "twl_align.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct twl_store twl_store;     /* opaque: the aligned rows of every sequence + cached node profiles, in HBM */

/* One side (reference or query node) of one pair.  Mirrors what calculateProfile reads from a Node
   (src/phylogeny.hpp:40-44): seqsIncluded, alnLen, alnNum, alnWeight, msaFreq. */
typedef struct twl_side {
    int32_t n_members;      /* sequences under this node */
    int32_t member_off;     /* offset of its first entry in members[] / member_weight[] */
    int32_t len;            /* aligned length of those rows (Node::alnLen) */
    int32_t num;            /* Node::alnNum */
    float   weight;         /* Node::alnWeight */
    int32_t cache_id;       /* >= 0: this node has a cached profile (Node::msaFreq non-empty) under this id: use it (:16-21) */
    int32_t store_id;       /* >= 0 (and cache_id < 0): also cache the un-normalised profile under this new id (:35-40) */
    int32_t reserved;
} twl_side;

/* type 'n' or 'p'.  seqs[i] need not be NUL terminated; lens[i] >= 0.  Rows are stored as given (letter case kept, like
   SequenceDB; the profile kernels upper-case on the fly exactly as alignment-helper.cpp:29 does). */
int  twl_store_create(int device, char type, int32_t n_seqs, const char *const *seqs, const int32_t *lens, twl_store **out);
void twl_store_destroy(twl_store *s);
/* Current aligned row of every sequence: rows_out[i] must hold lens_out[i] bytes; call with rows_out == NULL to get the lengths. */
int  twl_store_read_rows(twl_store *s, char *const *rows_out, int32_t *lens_out);
/* The rows of a LIST of sequences, back to back in `out` (row t at the prefix sum of lens_out; out NULL: the lengths only), and the inverse: rows that
   become the current rows of these sequences.  A sharded run whose ranks aligned their own subtrees alone exchanges them once, where the subtrees
   meet (twilight_amd/csrc/host/progressive.cpp, ownership).  twl_store_write_cache: a cached profile (float[len][P]) under an id new to this store. */
int  twl_store_read_rows_of(twl_store *s, int32_t n_ids, const int32_t *ids, char *out, int32_t *lens_out);
int  twl_store_write_rows(twl_store *s, int32_t n_ids, const int32_t *ids, const char *in, const int32_t *lens);
int  twl_store_write_cache(twl_store *s, int32_t id, const float *data, int32_t len);
/* The same two on DEVICE blocks, for collectives that move device memory (RCCL over xGMI): rows_to_block packs the current rows of `ids` back to back at
   dev_block (lens_out their lengths), rows_from_block makes the rows found at dev_block the current rows of `ids`; exchange_buffers hands out two device
   buffers of the store (send, recv) of at least the given sizes. */
int  twl_store_rows_to_block(twl_store *s, int32_t n_ids, const int32_t *ids, void *dev_block, int32_t *lens_out);
int  twl_store_rows_from_block(twl_store *s, int32_t n_ids, const int32_t *ids, const int32_t *lens, const void *dev_block);
int  twl_store_exchange_buffers(twl_store *s, int64_t send_bytes, int64_t recv_bytes, void **send_dev, void **recv_dev);
/* Cached profile `id` as float[len][P]; len_out receives its length; out may be NULL to query the length. */
int  twl_store_read_cache(twl_store *s, int32_t id, float *out, int32_t *len_out);
int  twl_store_drop_cache(twl_store *s, int32_t id);

/*
 * Build the DP inputs of all pairs of a level on the device.
 *   p               gap_open / gap_extend feed calculatePSGP; P selects nucleotide (6) or protein (22)
 *   gappy_threshold option->gappyVertical; 1.0 disables the removal (:77)
 *   sides           [n_pairs][2]   (0 = reference node, 1 = query node)
 *   members         sequence ids, member_weight = seq.weight / groupWeight * num evaluated in fp32 (:27)
 *   seq_len         column stride of the level, >= every side's len
 *   len_out         [n_pairs][2]   lengths after gappy-column removal
 *   colinfo_out     [n_pairs][2][seq_len]  per ORIGINAL column: consensus letter index (getConsensus) in the low 7 bits,
 *                   0x80 set when the column was removed as gappy; the caller derives the (start,length) runs and the
 *                   consensus strings addGappyColumnsBack needs.  May be NULL.
 */
int twl_level_prepare(twl_store *s, const twl_params *p, float gappy_threshold, int32_t n_pairs, const twl_side *sides,
                      const int32_t *members, const float *member_weight, int32_t seq_len, int32_t *len_out, uint8_t *colinfo_out);

/* The column info of ONE side of the prepared level (its original `len` bytes, format as colinfo_out of twl_level_prepare), or with
   pair < 0 of the whole level ([n_pairs][2][seq_len]): for callers that pass colinfo_out = NULL and fetch it only when some side's
   length shrank (len_out < len: a column was removed). */
int twl_level_read_colinfo(twl_store *s, int32_t pair, int32_t side, uint8_t *out);
/* The column info of BOTH sides of the n_sel pairs listed in pairs[]: out[n_sel][2][seq_len], one synchronisation for all of them. */
int twl_level_read_colinfo_many(twl_store *s, int32_t n_sel, const int32_t *pairs, uint8_t *out);

/*
 * Run the DP on the prepared level.  run_mask[i] != 0 selects pair i (NULL = all); other pairs get aln_len 0, err 0.
 * May be called repeatedly (other gap_char group, retries with a larger xdrop/flen).  Outputs as in twl_align_batch,
 * aln_out is [n_pairs][2*seq_len]; aln_out == NULL leaves the paths in HBM (lengths and error codes still come back): a caller
 * fetches the few it has to edit with twl_level_read_path and commits the others in place with twl_level_commit_from_dp.
 */
int twl_level_align(twl_store *s, const twl_params *p, const uint8_t *run_mask, int8_t *aln_out, int32_t *aln_len_out, int16_t *err_out);
/* The same with the gap-character score chosen per pair, as the reference chooses it (src/alignment-cpu.cpp:88: gapCharScore is 0 for a pair with
   more than 10 000 sequences on a side, gapExtend otherwise): zero_gap[i] != 0 gives pair i the score 0, the others p->gap_char (NULL = all
   p->gap_char).  One launch instead of one per group: the top levels of a 100 000-leaf tree hold a few pairs of each kind. */
int twl_level_align_mixed(twl_store *s, const twl_params *p, const uint8_t *run_mask, const uint8_t *zero_gap, int8_t *aln_out, int32_t *aln_len_out, int16_t *err_out);

/* The first `len` path codes of pair `pair` as the level's last DP run over that pair left them. */
int twl_level_read_path(twl_store *s, int32_t pair, int8_t *out, int32_t len);
/* The same for the n_sel pairs listed in pairs[] (lens[t] codes of pair pairs[t] into out + t * out_stride), one synchronisation. */
int twl_level_read_paths(twl_store *s, int32_t n_sel, const int32_t *pairs, const int32_t *lens, int8_t *out, int32_t out_stride);

/*
 * alignment_helper::addGappyColumnsBack (+ pairwiseGlobal for runs removed on both sides at one step; alignment-helper.cpp:324-375,
 * :243-322) on the device, for the n_sel pairs listed in pairs[] of the prepared and aligned level: the columns removeGappyColumns took
 * out go back into their DP paths.  The level's DP paths (every pair's) are staged in the level's path buffer at row pitch out_stride
 * (>= the longest final path: refLen + qryLen before removal), the restored ones replace their rows there, and
 * final_len_out[t] = the final length of pairs[t] -- or -1 when that pair has a two-sided run too large for the device's per-thread
 * scratch ((m + 1) x (n + 1) > 4096 cells or n > 127): the caller then restores it on the host (twl_level_read_paths /
 * twl_level_read_colinfo_many) and hands the result to the commit as a host row.  The commit that follows must be
 * twl_level_commit_from_dp with path_stride == out_stride and from_dp[i] == 2 for the pairs restored here (1 still means: final as the
 * DP left it).  May be called several times per level (after each twl_level_align of a gap-character group), always with one out_stride.
 */
int twl_level_restore(twl_store *s, const twl_params *p, int32_t n_sel, const int32_t *pairs, int32_t out_stride, int32_t *final_len_out);
/*
 * Final paths between processes without leaving HBM (several processes align one family, include/twl_msa.h): after twl_level_restore
 * (n_sel may be 0: it fixes the row pitch of the level's path buffer) a rank packs the final paths of ITS pairs into one contiguous device
 * block, hands the block to a device all-gather, and unpacks the other ranks' paths into their rows of the path buffer; the commit then
 * takes every row from HBM (from_dp[i] == 2).
 *   twl_level_exchange_buffers  device scratch for the blocks (grow-only, owned by the level): send_bytes, recv_bytes
 *   twl_level_paths_to_block    row of pair pairs[t] (where[t] == 1: as the DP left it, 2: from the path buffer), lens[t] bytes -> blk + blk_off[t]
 *   twl_level_paths_from_block  blk + blk_off[t], lens[t] bytes -> the path-buffer row of pair pairs[t]
 *   twl_level_write_final       a path the host restored (twl_level_restore returned -1 for it) into its row of the path buffer
 */
int twl_level_exchange_buffers(twl_store *s, int64_t send_bytes, int64_t recv_bytes, void **send_dev, void **recv_dev);
int twl_level_paths_to_block(twl_store *s, int32_t n_sel, const int32_t *pairs, const int32_t *lens, const uint8_t *where, void *blk_dev, const int64_t *blk_off);
int twl_level_paths_from_block(twl_store *s, int32_t n_sel, const int32_t *pairs, const int32_t *lens, const void *blk_dev, const int64_t *blk_off);
int twl_level_write_final(twl_store *s, int32_t pair, const int8_t *path, int32_t len);
/* Diagnostics: the first `len` codes of pair `pair`'s row of the staged path buffer (after twl_level_restore). */
int twl_level_read_final(twl_store *s, int32_t pair, int8_t *out, int32_t len);

/*
 * Apply the final paths (gappy columns restored) to the rows of both nodes of every pair and merge cached profiles.
 *   paths     [n_pairs][path_stride]   codes 0/1/2;  path_len[i] == 0 leaves pair i untouched (deferred pair)
 * After the call every member row of pair i has length path_len[i]; if both sides had a cache (cache_id or store_id) the
 * merged profile replaces the reference side's cache and the query side's id is dropped (updateFrequency :506-539).
 */
int twl_level_commit(twl_store *s, const int8_t *paths, const int32_t *path_len, int32_t path_stride);

/*
 * Same, with the paths of the pairs marked in from_dp[] taken from the level's own DP output in HBM (path_len[i] = the length
 * twl_level_align returned): for pairs in which no gappy column was removed addGappyColumnsBack (alignment-helper.cpp:243-289) is the
 * identity, so their paths never have to leave the device.  Rows of `paths` of marked pairs are ignored; paths may be NULL when every
 * pair with path_len > 0 is marked.  from_dp[i] == 2: the row was put there by twl_level_restore.  from_dp == NULL is twl_level_commit.
 */
int twl_level_commit_from_dp(twl_store *s, const int8_t *paths, const int32_t *path_len, int32_t path_stride, const uint8_t *from_dp);

/* Diagnostics: the packed DP columns [len][P+2] (P frequencies, gapOpen, gapExtend) of one side of the prepared level. */
int twl_level_read_columns(twl_store *s, int32_t pair, int32_t side, float *out, int32_t max_cols);

/* HIP-event time (ms) of the last prepare / commit kernels, for the per-level report.  A commit returns while its kernels run (the
   next call on the store is queued behind them); this call waits for them. */
int twl_level_timing(twl_store *s, double *prepare_ms, double *commit_ms);

#ifdef __cplusplus
}
#endif
#endif /* TWL_LEVEL_H */
