// twilight_amd/csrc/twl_policy.inc.hip -- the launch policy of a nucleotide call as a PURE function of the call's facts (plan_nucleotide; twl_plan_describe prints it without a device, tests/test_policy_cpu.py holds its cases).
// Included by twl_align.hip (one translation unit: it shares that file's Device bookkeeping, error string and fill queue).

// ---- launch policy of the nucleotide path: a pure function of the call's facts (unit-tested without a GPU through twl_plan_describe) ----
struct Knobs { int mt_max_pairs, mt_min_marker, mt_tail_pct, mt_wide, assume_onehot_query, no_spec, thr_small; };
struct NucFacts {
    int n_run = 0, num_cu = 0, marker = 0;
    const float *M = nullptr;             // 5 x 5 matrix
    float gap_char = 0;
    bool qry_onehot = false, dump = false;
    int shape = 0;                        // what the caller knows about every pair: 0 nothing, 2 leaf x leaf (single sequences on both sides)
    int wide_streak = 0, last_wide_pct = 0, wide_calls = 0;
    int small_state = 0;                  // the 512-row throughput window (NucPlan::small) on the earlier levels of this pass: 1 they fitted it, -1 one outgrew it, 0 nothing known
    const int32_t *h_len = nullptr;       // [pair][2]
    const int32_t *order = nullptr;       // the pairs that run, longest first
};
enum class NucFirst { Dump, WideMt, Mt, SpecShared, Spec16, Few16, Throughput, General };
struct NucPlan {
    NucFirst first = NucFirst::General;
    int mm = 0;                           // matrix mode 0 general / 1 zero N row and column / 2 match-transition-transversion
    bool mm5 = false;                     // ... in its one-letter-query form (mode 5)
    bool lean = false;                    // the round-2 kernels (scores within fast_div's range)
    bool four = false;                    // throughput launch on 4 waves x 3 blocks, four workgroups per CU (768-row window)
    bool small = false;                   // ... on 4 waves x 2 blocks, FIVE workgroups per CU (512-row window): levels of short pairs
    bool probe = false;                   // ... to be decided by a sample of the level's pairs (run_device): levels of 8+ rounds with nothing remembered
    bool held_back = false;               // ... not taken because a recent level outgrew it
    int bulk = 0, tail = 0;               // throughput: pairs in full rounds / remainder through the tile-parallel path
    int sp = 0;                           // the specialised step of the throughput geometries (talco_lean_kernel, SP): 1 leaf x leaf
};
NucPlan plan_nucleotide(const NucFacts &f, const Knobs &k)
{
    NucPlan pl;
    const float *M = f.M;
    // matrix mode (see talco_kernel): 2 = default match/transition/transversion structure with a zero N row/column
    bool nz = true, st3 = true;
    for (int t = 0; t < 5; ++t) nz = nz && M[20 + t] == 0.0f && M[5 * t + 4] == 0.0f;
    for (int l = 0; l < 4; ++l)
        for (int m = 0; m < 4; ++m) st3 = st3 && M[5 * l + m] == ((l == m) ? M[0] : (((l ^ m) == 2) ? M[2] : M[1]));
    pl.mm = nz ? (st3 ? 2 : 1) : 0;
    // fast_div's guard (talco_nuc.hip.h): non-zero scores within [2^-10, 2^10]; anything else takes the IEEE-division kernel
    bool divOk = true;
    auto inRange = [](float x) { const float ax = std::fabs(x); return x == 0.0f || (ax >= 0.0009765625f && ax <= 1024.0f); };
    for (int t = 0; t < 25; ++t) divOk = divOk && inRange(M[t]);
    pl.lean = divOk && inRange(f.gap_char);
    const int mm = pl.mm, n_run = f.n_run;
    // few pairs: one 64-row block per wave (16 waves) for the shortest diagonal step
    const bool few = n_run <= f.num_cu;
    int32_t maxLen = 0;
    long long sumLen = 0;
    for (int32_t t = 0; t < n_run; ++t) {
        const int32_t R = f.h_len[2 * f.order[t]], Q = f.h_len[2 * f.order[t] + 1];
        maxLen = std::max(maxLen, std::max(R, Q)); sumLen += (long long)R + Q;
    }
    // single-sequence query sides and no score for N: matrix mode 5 (the one-letter form of modes 1 and 2)
    pl.mm5 = pl.lean && mm >= 1 && (f.qry_onehot || k.assume_onehot_query);
    // the step without the per-block tests (talco_lean_kernel, SP 1): leaf x leaf, on the one-letter-query score
    pl.sp = (pl.lean && mm == 2 && f.shape == 2 && pl.mm5) ? 1 : 0;
    // very few pairs: two workgroups per pair take the tiles in turn (the mailbox words of that start carry absolute positions in 16 bits each)
    const bool spec = pl.lean && few && (mm == 2 || pl.mm5) && 2 * n_run <= f.num_cu && maxLen <= 65535 && !k.no_spec;
    // Tile-parallel path: always for levels of up to CUs/2 pairs (a pair's tile chain is what they wait for); beyond that when the pairs fill the
    // ONE round of the throughput kernel badly -- tiles spread evenly, at the price of the scouts (~1.2x the work).  Levels of several rounds: the remainder rule below.
    pl.four = pl.lean && (pl.mm5 || mm == 2);
    // With X-drop 5000 a band is ~440 rows wide whatever the length of the pair: most pairs fit a 512-row window, and at 29 KB of LDS and 96 registers FIVE
    // workgroups of 4 waves x 2 blocks share a CU -- five independent anti-diagonal chains per SIMD instead of four (16 384 pairs of 1.6 kbp: 95.7 -> 82.6 ms,
    // leaf x leaf 76.2 -> 65.3 ms, tools/exp_thr.py).  A level whose pairs outgrow the window pays for it twice (they re-run on the 768-row geometry), so the
    // outcome is remembered for the rest of the pass (run_device keeps small_state; bands widen up the tree, and a level LARGER than the one before it is the
    // start of another pass or family: nothing is known again): after a level that fitted the next ones start there, after one that sent more than 1 % of its
    // pairs on the rest of the pass stays off it (the window is worth ~16 % of a level's time; the pairs that outgrow it run twice AND their re-run is a launch
    // of its own that takes a pair's full latency, ~3.5 ms for 1.6 kbp pairs, however few they are: on 100 000 x 1.6 kbp levels of 3-5 % lost 2-11 %), and a
    // level that finds nothing remembered asks ITS OWN pairs when it is large -- eight or more rounds: one pair per CU, spread over the cost order, runs on the
    // small window first (they are part of the level: nothing is computed twice but what outgrows the window; ~3 ms) and the share of them that outgrew it
    // decides for the rest -- and simply tries when it is small.
    const long long longest = n_run > 0 ? (long long)f.h_len[2 * f.order[0]] + f.h_len[2 * f.order[0] + 1] : 0;
    // LONG pairs are eligible too (late round 4: on 10 000 x 10 kbp no pair of any level outgrows 512 rows, and the five workgroups are worth 97.8 against 110 ms
    // on its leaf level, 442 against 469 ms per pass): what made a lost bet expensive there -- the re-run of a FEW 10 kbp pairs, one after the other, a pair's
    // full latency of ~18 ms -- goes through the tile-parallel path instead (run_device: ~3 ms).  They are not sampled (a sample would cost that latency): the
    // first level of a pass pairs sibling leaves, the most similar sequences of the family, and simply tries; the levels above it do as it fared.
    const bool eligible = pl.four && n_run > f.num_cu && k.thr_small == 0;
    pl.probe = eligible && f.small_state == 0 && n_run >= 8 * f.num_cu && longest <= 4096;
    pl.small = (pl.four && n_run > f.num_cu && k.thr_small == 2) || (eligible && f.small_state >= 0);
    pl.held_back = eligible && f.small_state < 0;
    const int perRound = (pl.small ? 5 : (pl.four ? 4 : 2)) * f.num_cu;
    const double roundsThr = (double)n_run / (double)perRound;
    const bool mtOk = pl.lean && mm == 2 && !pl.mm5 && !f.dump && n_run <= k.mt_max_pairs && f.marker >= k.mt_min_marker &&
                      sumLen >= 3ll * f.marker * n_run && (2 * n_run <= f.num_cu || (roundsThr <= 1.0 && std::ceil(roundsThr) >= 1.2 * roundsThr));
    // the last calls' pairs all outgrew the fast window (the deferred pass: one pair per level against the same growing root): no point in finding
    // that out again -- straight to the 3072-row geometry; every 8th such call tries the fast window again
    // ... and so for a level of up to CUs pairs when three quarters of the previous narrow-first level's pairs went on to the wide window (the upper levels
    // of a family whose pairs outgrow the fast window: their narrow attempts cost 40-80 ms each in tiles computed in line up to the overflow); every 6th probes
    const bool wideFirst = k.mt_wide && (n_run <= 8 ? (f.wide_streak >= 2 && (f.wide_streak & 7) != 7)
                                                    : (n_run <= f.num_cu && f.last_wide_pct >= 75 && (f.wide_calls % 6) != 5));
    if (f.dump) pl.first = NucFirst::Dump;
    else if (mtOk && wideFirst) pl.first = NucFirst::WideMt;
    else if (mtOk) pl.first = NucFirst::Mt;
    // CUs/2 < pairs <= CUs: two workgroups per pair taking the tiles in turn, of the 8-wave geometry, two to a CU (all 2n resident at once, as the teams
    // wait for each other).  250 pairs of 10 kbp: 27.6 -> 20.1 ms against one 16-wave workgroup per pair
    else if (pl.lean && mm == 2 && n_run <= f.num_cu && 2 * n_run > f.num_cu && maxLen <= 65535 && !k.no_spec) pl.first = NucFirst::SpecShared;
    else if (spec) pl.first = NucFirst::Spec16;
    else if (pl.lean && few) pl.first = NucFirst::Few16;
    else if (pl.lean) {
        // Many pairs: persistent workgroups take them in rounds.  A last round that is badly filled costs a whole round: when the remainder is small enough
        // its pairs (the shortest ones, the order is longest first) go through the tile-parallel path instead, where they spread over all CUs.
        pl.first = NucFirst::Throughput;
        int tail = n_run % perRound;
        long long tailLen = 0;
        for (int32_t t = n_run - tail; t < n_run; ++t) tailLen += (long long)f.h_len[2 * f.order[t]] + f.h_len[2 * f.order[t] + 1];
        // (pairs of 8+ tiles: with fewer the scouts and extra launches cost more than the idle workgroups)
        if (!(n_run > perRound && tail > 0 && tail * 100 <= k.mt_tail_pct * perRound && tail <= k.mt_max_pairs && pl.four && f.marker >= k.mt_min_marker && tailLen >= 8ll * f.marker * tail)) tail = 0;
        pl.tail = tail; pl.bulk = n_run - tail;
    }
    else pl.first = NucFirst::General;
    if (pl.first != NucFirst::Throughput) pl.small = pl.probe = pl.held_back = false;
    return pl;
}
const char *nuc_first_name(NucFirst f)
{
    switch (f) {
    case NucFirst::Dump: return "dump";
    case NucFirst::WideMt: return "tile-parallel, 3072-row window";
    case NucFirst::Mt: return "tile-parallel";
    case NucFirst::SpecShared: return "speculative teams, 8 waves x 2 blocks";
    case NucFirst::Spec16: return "speculative teams, 16 waves";
    case NucFirst::Few16: return "16 waves x 1 block";
    case NucFirst::Throughput: return "throughput";
    default: return "general (IEEE division)";
    }
}

Knobs current_knobs() { return Knobs{g_mt_max_pairs, g_mt_min_marker, g_mt_tail_pct, g_mt_wide, g_assume_onehot_query, g_no_spec, g_thr_small}; }
