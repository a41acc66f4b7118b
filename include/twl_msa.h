/*
 * include/twl_msa.h -- C ABI of libtwl_host: TWILIGHT's tree + sequences alignment flow with the MI355X level kernel.
 *
 * libtwl_align (twl_align.h / twl_level.h) replaces the reference's per-level DP; this header exposes the CALLER of that path --
 * the host flow of /root/reference/src/twilight-main.cpp:115-176 (DEFAULT_ALN, single partition): read the guide tree and the
 * sequences, msa::progressive::msaOnSubtree (progressive.cpp:232-299) over the level schedule with the GPU level kernel injected
 * as msa::alnFunction (msa.hpp:175), write the MSA -- in steps, so that a caller can keep the inputs resident in HBM, time the
 * alignment alone, repeat it on fresh handles, and run it on several processes (one per GPU) that align ONE family together.
 * The product CLI `twilight-mi355x` is this same flow in one go.
 *
 * Plain C types only.  Functions return 0 or a negative code; twl_msa_last_error() describes the failure.  Errors the
 * reference treats as fatal (unsupported flags, unreadable files, inconsistent trees) end the process with a message, as there.
 */
#ifndef TWL_MSA_H
#define TWL_MSA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct twl_msa twl_msa;

/* Development / test switches -- flags of the argv given to twl_msa_open (and of the CLI and the CPU checker, which parse the same way), not options of the product:
     --test-cal-profile-th n, --test-update-seq-th n   the reference's _CAL_PROFILE_TH / _UPDATE_SEQ_TH (msa.hpp:179-180, both 1000): lowered by tests so
                                that small trees reach the cached-profile and compressed-group branches
     --test-virtual-devices n   n replicas of the store on ONE device: the several-replica code path of the resident level kernel on a one-GPU box
     --test-no-ownership        a sharded run deals and exchanges every level (the design before subtree ownership; tests hold the two to each other)
     --test-fork-host-staged    (CLI) one forked process per listed device for the host-staged kernel as well, the paths of every level all-gathered through the library's
                                communicator: the forked flow (shared page, id hand-over, watchdog) on the CPU-check build, whose communicator is a shared-memory segment
   Environment: TWL_OMP_THREADS (host threads of the library; default: what OpenMP picks) and TWL_DEBUG (traces the launches of libtwl_align on stderr).
   Nothing in the environment changes a result or a launch. */

/* One level-kernel call (= one line of the reference's per-level report, progressive.cpp:178-189). */
typedef struct twl_msa_level {
    int32_t  pairs;         /* sibling pairs of the level */
    int32_t  task;          /* 0 main pass, 1 deferred pass.  A deferred-pass level holds exactly ONE pair, as the reference's does (progressive.cpp:283-291: each
                               profile is aligned to the root the previous one was merged into); the level kernel refuses a task-1 level of any other size
                               with TWL_ERR_UNSUPPORTED (its retries of alignment-cpu.cpp:116-129 re-run the level under a one-pair mask) */
    uint64_t band_cells;    /* DP band cells of the level, all ranks */
    uint64_t relaunched;    /* pairs re-run in a wider window */
    double   kernel_ms;     /* DP kernel time (HIP events; max over the ranks / devices that ran concurrently) */
    double   level_ms;      /* host wall time of the level-kernel call */
    double   exchange_ms;   /* of which: all-gather of the paths between processes */
    int32_t  matrix_mode;   /* column-score mode of the level's DP kernel (twl_stats.matrix_mode; -1 unknown) */
    int32_t  speculative;   /* the speculative two-workgroup kernel ran the level: 1 a CU per workgroup, 2 two workgroups per CU; 3 tile-parallel (twl_stats.speculative) */
    int32_t  mt_tiles_predicted;   /* tile-parallel levels: tiles whose predicted start cell was the true one */
    int32_t  mt_tiles_inline;      /* ... tiles the stitch launch computed itself */
    char     kernel[160];   /* the level's DP kernel as the profiler names it (twl_stats.kernel; device 0 / this rank) */
} twl_msa_level;

typedef struct twl_msa_totals {
    int32_t  n_levels, aln_len, n_sequences, reserved;
    uint64_t pairs, band_cells, relaunched;
    double   kernel_ms, exchange_ms, align_s;
    uint64_t nominal_cells;   /* sum of R*Q over the pairs of the level calls THIS process made (one GPU: every pair of the run): nominal cells of the classic GCUPS figure */
} twl_msa_totals;

/* All-gather of equal-sized host blocks between the processes of a sharded run: send = this rank's block of bytes_per_rank
   bytes, recv = [world][bytes_per_rank].  Returns 0 on success. */
typedef int (*twl_msa_exchange_fn)(void *user, const void *send, int64_t bytes_per_rank, void *recv);

/* argv: the flags of the CLI (-t tree -i sequences -o output [-r ...] [--type n|p] [--gpu-index k] ...; argv[0] is ignored).
   Parses the options, reads the tree and the sequences, builds nothing on the device yet. */
int  twl_msa_open(int argc, const char *const *argv, twl_msa **out);
/* Several processes, one GPU each, align this family together: every process opens the same inputs.  Below a cut of the guide tree a rank aligns
   the subtrees it owns alone (device-resident kernel: no exchange there, one exchange of rows / cached profiles / node bookkeeping where the subtrees
   meet; twilight_amd/csrc/host/align_owned.cpp); above it -- and on every level with the host-staged kernel -- it aligns the pairs dealt to `rank`
   and gets the others' paths through `exchange` once per level (twilight_amd/dist.py: torch.distributed all_gather, backend nccl = RCCL over xGMI on
   GPUs, gloo in CPU tests).  Call before twl_msa_align; world == 1 is the default. */
int  twl_msa_shard(twl_msa *m, int rank, int world, twl_msa_exchange_fn exchange, void *user);
/* The same with an all-gather of DEVICE blocks (send_dev / recv_dev live in the HBM of this process's GPU; same signature): the
   device-resident level kernel then keeps every path in HBM from the DP to the write-back -- one collective per level and no host
   staging (twilight_amd/dist.py: make_device_exchange, RCCL over xGMI).  The library has synchronised its stream before the call; the
   collective must have completed when the function returns.  `exchange` (host blocks) may be given too: the host-staged level kernel
   (--host-staged) and the bookkeeping of the subtree exchange use it. */
int  twl_msa_shard_device(twl_msa *m, int rank, int world, twl_msa_exchange_fn exchange_dev, void *user_dev, twl_msa_exchange_fn exchange, void *user);
/* The same sharded run with the collective made by the library itself: RCCL from C++ (include/twl_align.h, twl_comm_*), one ncclAllGather per
   level on the library's stream, no callback into the caller's runtime.  id128 = the 128 bytes twl_msa_rccl_unique_id gave ONE rank, handed to
   all ranks by the launcher (bench.py: torch.distributed's store; twilight-mi355x --gpu-index a,b,...: a shared page of the processes it forks).
   Collective: every rank calls it, after twl_msa_open.  This is how `twilight-mi355x --gpu-index 0,1,...,7` runs: one process per GPU.
   -3: RCCL could not make the communicator (twl_msa_last_error says why); the handle is unsharded again and may be sharded through the caller's own
   collective (twl_msa_shard_device) -- every rank has to take the same way, which is the caller's to agree on (bench.py does, with one all-reduce). */
int  twl_msa_rccl_unique_id(void *id128);
int  twl_msa_shard_rccl(twl_msa *m, int rank, int world, const void *id128);
/* Device-resident path: put the sequences into HBM now (otherwise the first level does it). */
int  twl_msa_upload(twl_msa *m);
/* The progressive alignment: every level of the main pass and the deferred pass. */
int  twl_msa_align(twl_msa *m);
/* Totals and the per-level records of the run (levels may be NULL; at most max_levels records are written). */
int  twl_msa_report(twl_msa *m, twl_msa_totals *totals, twl_msa_level *levels, int32_t max_levels);
/* Write the MSA (path NULL: the -o of twl_msa_open), FASTA as the reference writes it (io.cpp:512-525). */
int  twl_msa_write(twl_msa *m, const char *path);
void twl_msa_close(twl_msa *m);
const char *twl_msa_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
