// twilight_amd/csrc/twl_knobs.inc.hip -- the development / test knobs of the dispatch (twl_set_knob, include/twl_align.h): their variables and defaults.
// Included by twl_align.hip (one translation unit: it shares that file's Device bookkeeping, error string and fill queue).

int g_prot_corridor = 448;  // twl_set_knob(TWL_KNOB_PROT_CORRIDOR, rows): half-width of the corridor score_matrix_kernel fills for protein levels of few pairs (0: the whole R x Q matrix, as until round 5)
int g_mt_perturb = 0;       // twl_set_knob(TWL_KNOB_MT_PERTURB, n): spoil every n-th predicted tile start (tests of the later rounds and of the in-line path)
int g_mt_lead = 320, g_mt_marg = 40;
// ... of the levels of at most 2048 tile jobs (a round or two of the device's workgroups): there a scout costs ~0.1 ms of the level and a missed start a whole second round (+1.5-1.9 ms), so the
// margin behind the boundary range and the lead of an anchored scout are longer (round 6: levels 15 and 19 of 10 000 x 10 kbp lost a start each with 40 / 96, none with 64 / 128;
// on the wide levels the longer scouts cost more than the two rounds, profiles/r06/exp_scout_margins.txt).  TWL_KNOB_MT_MARGIN / TWL_KNOB_MT_LEAD2 set both kinds
int g_mt_marg_lat = 64, g_mt_lead2_lat = 128;
int g_mt_anchor = 1, g_mt_lead2 = 96;      // twl_set_knob(TWL_KNOB_MT_ANCHOR / TWL_KNOB_MT_LEAD2): scouts (either alphabet) start from the cell the profiles' consensus letters point at, this many anti-diagonals ahead (talco_nuc.hip.h, mt_anchor_kernel)
int g_mt_max_pairs = 1024, g_mt_min_marker = 512, g_mt_rounds = 2, g_mt_thr_jobs = 256;
int g_scout_xdrop_pct = 100;         // twl_set_knob(TWL_KNOB_SCOUT_XDROP_PCT): X-drop of the pair scouts in percent of the call's (they only predict: a narrower band is a cheaper scout)
int g_no_spec = 0;                   // twl_set_knob(TWL_KNOB_NO_SPEC): no speculative two-workgroup teams (tools that time the plain tile loop)
int g_thr_small = 0;                 // twl_set_knob(TWL_KNOB_THR_SMALL): 0 = the 512-row throughput geometry for levels of short pairs (plan_nucleotide), 1 never, 2 whenever the throughput kernel runs (tests)
int g_mt_wide = 1;                   // twl_set_knob(TWL_KNOB_MT_WIDE): 0 = pairs that outgrew the 1024-row window run tile after tile (the path before round 4; tests compare the two)
int g_mt_tail_pct = 70;              // twl_set_knob(TWL_KNOB_MT_TAIL_PCT): a last round filled up to this share of 2 * CUs workgroups goes through the tile-parallel path (0 = never)
int g_prot_mode = 0;                 // twl_set_knob(TWL_KNOB_PROT_MODE): force a protein kernel variant (tests of every variant)
int g_force_global = 0;             // twl_set_knob(TWL_KNOB_FORCE_GLOBAL): every pair of every call runs on the global-memory kernel (tests of that kernel on small cases)
int g_assume_onehot_query = 0;       // twl_set_knob(TWL_KNOB_ASSUME_ONEHOT_QUERY): the host form too takes the one-letter-query kernels
int g_leaf_step = 1;                 // twl_set_knob(TWL_KNOB_LEAF_STEP): 0 = leaf x leaf levels too run the general step (talco_lean_kernel, SP 0; tests hold the two to each other)
int g_poison_tb = 0;                 // twl_set_knob(TWL_KNOB_POISON_TB): the traceback scratch is filled with 0xFF bytes in front of every DP launch (tests: a word whose store was wrongly skipped then reads as garbage, not as the zeros of a fresh allocation)
struct Knobs;
