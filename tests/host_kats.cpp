// tests/host_kats.cpp -- known-answer tests of the host mirror's helpers, written from the reference source
// (/root/reference/src/alignment-helper.cpp, scoring-matrix.cpp, tree.cpp, progressive.cpp).  Prints "OK <name>" / "FAIL <name>".
#include "../twilight_amd/csrc/host/twl_host.hpp"

#include <random>
#include <functional>
#include <unordered_map>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

static int g_fail = 0;
#define CHECK(name, cond) do { if (cond) printf("OK %s\n", name); else { printf("FAIL %s\n", name); ++g_fail; } } while (0)

using namespace msa;

static std::string pathStr(const alnPath &p) { std::string s; for (auto c : p) s += char('0' + c); return s; }

int main(int argc, char **argv)
{
    const std::string tmp = argc > 1 ? argv[1] : "/tmp";
    Option opt;
    // ---- letterIdx (scoring-matrix.cpp:26-79) ----
    CHECK("letterIdx_nuc", letterIdx('n', 'A') == 0 && letterIdx('n', 'C') == 1 && letterIdx('n', 'G') == 2 && letterIdx('n', 'T') == 3 &&
                               letterIdx('n', 'U') == 3 && letterIdx('n', 'N') == 4 && letterIdx('n', 'R') == 4 && letterIdx('n', '-') == 5 && letterIdx('n', '.') == 5);
    CHECK("letterIdx_prot", letterIdx('p', 'A') == 0 && letterIdx('p', 'Y') == 19 && letterIdx('p', 'W') == 18 && letterIdx('p', 'X') == 20 &&
                                letterIdx('p', 'B') == 20 && letterIdx('p', 'U') == 20 && letterIdx('p', '-') == 21);
    // ---- msa::Params defaults (scoring-matrix.cpp:81-110; twilight-main.cpp:65-73) ----
    {
        Params p(opt, 'n');
        bool ok = p.matrixSize == 5 && p.gapOpen == -50 && p.gapExtend == -5 && p.gapBoundary == -5 && p.xdrop == 3000;
        for (int i = 0; i < 5; ++i)
            for (int j = 0; j < 5; ++j) {
                float want = (i == 4 || j == 4) ? 0 : (i == j ? 18 : (std::abs(i - j) == 2 ? -4 : -8));
                ok = ok && p.scoringMatrix[i][j] == want;
            }
        CHECK("params_nucleotide_default", ok);
        Option w = opt; w.wildcard = true;
        Params pw(w, 'n');
        CHECK("params_wildcard_N_is_match", pw.scoringMatrix[4][0] == 18 && pw.scoringMatrix[2][4] == 18);
        Params pp(opt, 'p');
        CHECK("params_protein_5xblosum62", pp.matrixSize == 21 && pp.scoringMatrix[0][0] == 20 && pp.scoringMatrix[18][18] == 55 && pp.scoringMatrix[1][3] == -20 &&
                                               pp.scoringMatrix[20][5] == 0 && pp.scoringMatrix[5][20] == 0);
    }
    // ---- msa::Params: the other built-in tables and the user matrix file (scoring-matrix.cpp:112-199, blosum.hpp:31-79) ----
    {
        Option o80 = opt; o80.blosum = 80;
        Params p80(o80, 'p');
        // BLOSUM80 as shipped is not symmetric: [I][V] = 3, [V][I] = 1 (blosum.hpp:65,75)
        CHECK("params_blosum80_asymmetric_entry", p80.scoringMatrix[7][17] == 15 && p80.scoringMatrix[17][7] == 5 && p80.scoringMatrix[0][0] == 35 && p80.scoringMatrix[18][18] == 75);
        Option o45 = opt; o45.blosum = 45; o45.wildcard = true;
        Params p45(o45, 'p');
        // the X score under -w is 5 x the mean BLOSUM62 diagonal whichever table is selected (:120-122): 5 * 5.8 = 29
        CHECK("params_blosum45_and_wildcard_X", p45.scoringMatrix[0][0] == 25 && p45.scoringMatrix[1][1] == 60 && std::fabs(p45.scoringMatrix[20][3] - 29.0f) < 1e-4f && std::fabs(p45.scoringMatrix[3][20] - 29.0f) < 1e-4f);
        Option ob = opt; ob.blosum = 50;
        Params pb(ob, 'p');
        CHECK("params_invalid_blosum_falls_back_to_62", pb.scoringMatrix[0][0] == 20 && pb.scoringMatrix[18][18] == 55);
        {   // dump every built-in protein matrix for the Python side, which holds the reference's numbers as a fixture
            std::ofstream f(tmp + "/blosum_dump.txt");
            for (int b : {45, 62, 80}) {
                Option ox = opt; ox.blosum = b;
                Params px(ox, 'p');
                for (int i = 0; i < 20; ++i) for (int j = 0; j < 20; ++j) f << px.scoringMatrix[i][j] << (j == 19 ? '\n' : ' ');
            }
        }
        if (argc > 2) {      // argv[2]: the reference's own example, dataset/substitution.txt (4 letters, no N)
            Option ou = opt; ou.matrixFile = argv[2];
            Params pu(ou, 'n');
            bool ok = pu.matrixSize == 5;
            for (int i = 0; i < 5; ++i) for (int j = 0; j < 5; ++j) ok = ok && pu.scoringMatrix[i][j] == ((i == 4 || j == 4) ? 0.0f : (i == j ? 4.0f : -1.0f));
            CHECK("params_user_matrix_reference_example", ok);
            ou.wildcard = true;
            Params pw2(ou, 'n');
            CHECK("params_user_matrix_wildcard_mean_diagonal", pw2.scoringMatrix[4][1] == 4.0f && pw2.scoringMatrix[3][4] == 4.0f && pw2.scoringMatrix[0][1] == -1.0f);
        }
        {   // letters in another order, ambiguity letter listed, asymmetric values: scoringMatrix[row letter][column letter]
            std::ofstream f(tmp + "/m5.txt");
            f << "T G C A N\n 1 2 3 4 5\n 6 7 8 9 10\n 11 12 13 14 15\n 16 17 18 19 20\n 21 22 23 24 25\n";
            f.close();
            Option ou = opt; ou.matrixFile = tmp + "/m5.txt";
            Params pu(ou, 'n');
            // row T = {T:1, G:2, C:3, A:4, N:5} -> M[3][3] = 1, M[3][2] = 2, M[3][1] = 3, M[3][0] = 4, M[3][4] = 5; row A = 16..20
            CHECK("params_user_matrix_letter_order_and_N", pu.scoringMatrix[3][3] == 1 && pu.scoringMatrix[3][2] == 2 && pu.scoringMatrix[3][0] == 4 && pu.scoringMatrix[3][4] == 5 &&
                                                           pu.scoringMatrix[0][3] == 16 && pu.scoringMatrix[0][0] == 19 && pu.scoringMatrix[4][4] == 25 && pu.scoringMatrix[2][3] == 6);
        }
    }
    // ---- getConsensus (alignment-helper.cpp:221-241): first strict maximum; all-zero -> N ----
    {
        float prof[3 * 6] = {0, 2, 2, 0, 0, 1, /*col1: C and G tie -> C*/ 0, 0, 0, 0, 0, 3, /*all zero -> N*/ 1, 0, 0, 0.5f, 9, 0 /*N count ignored: argmax over first 4 -> A*/};
        std::string c;
        alignment_helper::getConsensus(&opt, prof, c, 3);
        CHECK("consensus_ties_and_empty", c == "CNA");
    }
    // ---- calculatePSGP (alignment-helper.cpp:168-219) ----
    {
        Params p(opt, 'n');
        Node a("a", 0), b("b", 0);
        a.alnNum = 4; b.alnNum = 1;
        NodePair np{&a, &b};
        SequenceDB db;
        const int memLen = 3;
        float freq[6 * 2 * 3] = {0};
        freq[0 * 6 + 5] = 0;       // ref col0: no gaps
        freq[1 * 6 + 5] = 1;       // ref col1: 1 of 4 gaps
        freq[2 * 6 + 5] = 3.999f;  // ref col2: almost all gaps
        float go[6], ge[6];
        alignment_helper::calculatePSGP(freq, go, ge, np, &db, &opt, memLen, {0, 0}, {3, 2}, p);
        bool ok = go[0] == -50 && ge[0] == -5;
        ok = ok && go[1] == std::min(-5.0f, (float)(-50.0f * 0.5f * ((4 - 1.0f) * 1.0 / 4))) && ge[1] == std::min(-1.0f, (float)(-5.0f * ((4 - 1.0f) * 1.0 / 4)));
        ok = ok && go[2] == -5.0f && ge[2] == -1.0f;                 // clamps: 0.1*gapOpen, 0.2*gapExtend
        ok = ok && go[3] == -50 && go[4] == -50 && go[5] == 0 && ge[5] == 0;   // query: 2 columns, then padding
        CHECK("psgp_formula_and_clamps", ok);
    }
    // ---- removeGappyColumns / addGappyColumnsBack round trip (alignment-helper.cpp:74-166,324-375) ----
    {
        Params p(opt, 'n');
        Node a("a", 0), b("b", 0);
        a.alnNum = 10; b.alnNum = 10; a.alnLen = 6; b.alnLen = 5;
        NodePair np{&a, &b};
        const int memLen = 6;
        std::vector<float> f(6 * 2 * memLen, 0.0f);
        auto set = [&](int side, int col, int letter, float gaps) { f[6 * (side * memLen + col) + letter] = 10 - gaps; f[6 * (side * memLen + col) + 5] = gaps; };
        // ref: cols 2,3 gappy (>95%); query: col 0 gappy and col 4 gappy
        for (int c = 0; c < 6; ++c) set(0, c, c % 4, (c == 2 || c == 3) ? 9.8f : 0.0f);
        for (int c = 0; c < 5; ++c) set(1, c, (c + 1) % 4, (c == 0 || c == 4) ? 10.0f : 1.0f);
        std::pair<IntPairVec, IntPairVec> g;
        IntPair lens{6, 5};
        alignment_helper::removeGappyColumns(f.data(), np, &opt, g, memLen, lens, 0);
        bool ok = lens.first == 4 && lens.second == 3 && g.first.size() == 1 && g.first[0] == IntPair(2, 2) && g.second.size() == 2 &&
                  g.second[0] == IntPair(0, 1) && g.second[1] == IntPair(4, 1);
        ok = ok && f[6 * 2 + 0] == 10 && f[6 * 3 + 1] == 10 && f[6 * 4 + 0] == 0;      // ref cols 4,5 moved up, tail zeroed
        CHECK("remove_gappy_columns", ok);
        alnPath before = {0, 0, 0, 2}, after;                        // reduced path: 4 ref cols x 3 query cols
        alignment_helper::addGappyColumnsBack(before, after, g, p, {4, 3}, {"ACGTAC", "CGTAC"});
        // query col0 is gappy at the very start -> 1; two matches; ref gappy run (2,3) -> 22; match; ref col; query col4 gappy at the end -> 1
        int r = 0, q = 0;
        for (auto c : after) { r += (c != 1); q += (c != 2); }
        CHECK("add_gappy_columns_back_consumes_original_lengths", r == 6 && q == 5 && pathStr(after) == "10022012");
    }
    // ---- addGappyColumnsBack against the reference's one-step-at-a-time walk (alignment-helper.cpp:324-375) on random inputs ----
    {
        Params p(opt, 'n');
        auto walk = [&](const alnPath &before, const std::pair<IntPairVec, IntPairVec> &g, const stringPair &org) {
            alnPath after;
            int rIdx = 0, qIdx = 0;
            size_t gr = 0, gq = 0;
            for (size_t a = 0; a < before.size() + 1; ++a) {
                const bool gapR = gr < g.first.size() && rIdx == g.first[gr].first;
                const bool gapQ = gq < g.second.size() && qIdx == g.second[gq].first;
                if (gapR && gapQ) {
                    alnPath sub;
                    alignment_helper::pairwiseGlobal(org.first.substr(rIdx, g.first[gr].second), org.second.substr(qIdx, g.second[gq].second), sub, p);
                    after.insert(after.end(), sub.begin(), sub.end());
                    rIdx += g.first[gr].second; qIdx += g.second[gq].second; ++gr; ++gq;
                } else {
                    if (gapR) { after.insert(after.end(), g.first[gr].second, 2); rIdx += g.first[gr].second; ++gr; }
                    if (gapQ) { after.insert(after.end(), g.second[gq].second, 1); qIdx += g.second[gq].second; ++gq; }
                }
                if (a < before.size()) { after.push_back(before[a]); if (before[a] != 1) ++rIdx; if (before[a] != 2) ++qIdx; }
            }
            return after;
        };
        std::mt19937 rng(12345);
        bool ok = true;
        int bothEvents = 0;
        for (int t = 0; t < 400 && ok; ++t) {
            // a reduced path, then random runs of removed columns placed into the original coordinates of either side
            const int len = (t < 20) ? (int)(rng() % 6) : (int)(rng() % 300);
            alnPath before(len);
            for (auto &c : before) c = (int8_t)(rng() % 8 < 6 ? 0 : (rng() % 2 ? 1 : 2));
            int nr = 0, nq = 0;
            for (auto c : before) { nr += (c != 1); nq += (c != 2); }
            auto runs = [&](int compactLen, bool sameAs, const IntPairVec *other, const std::vector<int> *otherCompact) {
                IntPairVec out; std::vector<int> compact;
                int inserted = 0;
                for (int c = 0; c <= compactLen; ++c) {
                    const bool force = sameAs && other && !otherCompact->empty() && rng() % 3 == 0;      // (both-sides events are made below, by position)
                    (void)force;
                    if (rng() % 9 == 0) { const int l = 1 + (int)(rng() % 5); out.push_back({c + inserted, l}); compact.push_back(c); inserted += l; }
                }
                return std::make_pair(out, compact);
            };
            auto R = runs(nr, false, nullptr, nullptr);
            auto Q = runs(nq, false, nullptr, nullptr);
            std::pair<IntPairVec, IntPairVec> g{R.first, Q.first};
            int totR = nr, totQ = nq;
            for (auto &x : g.first) totR += x.second;
            for (auto &x : g.second) totQ += x.second;
            stringPair org{std::string(totR + 8, 'A'), std::string(totQ + 8, 'A')};
            static const char L[] = "ACGT";
            for (auto &c : org.first) c = L[rng() % 4];
            for (auto &c : org.second) c = L[rng() % 4];
            const alnPath want = walk(before, g, org);
            alnPath got;
            alnPath b2 = before;
            alignment_helper::addGappyColumnsBack(b2, got, g, p, {nr, nq}, org);
            ok = ok && got == want;
            // count the steps at which both sides had a run (the pairwiseGlobal branch)
            { int rIdx = 0, qIdx = 0; size_t gr = 0, gq = 0;
              for (size_t a = 0; a < before.size() + 1; ++a) {
                  const bool gapR = gr < g.first.size() && rIdx == g.first[gr].first, gapQ = gq < g.second.size() && qIdx == g.second[gq].first;
                  if (gapR && gapQ) ++bothEvents;
                  if (gapR) { rIdx += g.first[gr].second; ++gr; }
                  if (gapQ) { qIdx += g.second[gq].second; ++gq; }
                  if (a < before.size()) { if (before[a] != 1) ++rIdx; if (before[a] != 2) ++qIdx; } } }
        }
        CHECK("add_gappy_columns_back_equals_the_walk_on_random_inputs", ok && bothEvents > 50);
    }
    // ---- pairwiseGlobal tie-breaks (alignment-helper.cpp:243-322) ----
    {
        Params p(opt, 'n');
        alnPath a;
        alignment_helper::pairwiseGlobal("ACGT", "ACGT", a, p);
        bool ok = pathStr(a) == "0000";
        alignment_helper::pairwiseGlobal("AC", "ACGT", a, p);        // leading gaps are free (row/column 0 are zero), trailing ones are not:
        ok = ok && pathStr(a) == "1100";                             // two transitions (-8) beat two matches plus a trailing gap (36-55)
        alignment_helper::pairwiseGlobal("", "ACG", a, p);
        ok = ok && pathStr(a) == "111";
        alignment_helper::pairwiseGlobal("AC", "", a, p);
        ok = ok && pathStr(a) == "22";
        CHECK("pairwise_global_paths", ok);
    }
    // ---- updateFrequency (alignment-helper.cpp:506-539) ----
    {
        Node a("a", 0), b("b", 0);
        a.msaFreq = {{1, 0, 0, 0, 0, 0}, {0, 2, 0, 0, 0, 0}};
        b.msaFreq = {{0, 0, 3, 0, 0, 0}, {0, 0, 0, 4, 0, 1}};
        NodePair np{&a, &b};
        alnPath path = {0, 1, 2};
        SequenceDB db;
        alignment_helper::updateFrequency(np, &db, path, {0.5f, 0.25f});
        bool ok = a.msaFreq.size() == 3 && b.msaFreq.empty() && a.alnLen == 3;
        ok = ok && a.msaFreq[0][0] == 1 && a.msaFreq[0][2] == 3;                        // column from both
        ok = ok && a.msaFreq[1][3] == 4 && a.msaFreq[1][5] == 1.5f;                      // query column + refWeight in the gap bin
        ok = ok && a.msaFreq[2][1] == 2 && a.msaFreq[2][5] == 0.25f;                     // ref column + qryWeight in the gap bin
        CHECK("update_frequency", ok);
    }
    // ---- Newick quirks (tree.cpp:59-223) and the level schedule (progressive.cpp:10-124) ----
    {
        const std::string f = tmp + "/kat.nwk";
        { std::ofstream o(f); o << "((A:0.1,B:0.2):0.05,(C,D:0.4),E:0.3);\n"; }
        Tree T(f);
        bool ok = T.allNodes.size() == 8 && T.root->identifier == "node_1" && T.root->children.size() == 3 && T.m_numLeaves == 5;
        ok = ok && T.allNodes["node_2"]->branchLength == 0.05f && T.allNodes["A"]->branchLength == 0.1f;
        ok = ok && T.allNodes["node_3"]->branchLength == 0.4f;      // ")" without ":" reuses the piece's previous number (D:0.4)
        ok = ok && T.allNodes["C"]->branchLength == 0.05f;          // missing length followed by ',' -> 0 -> smallest positive length
        ok = ok && T.allNodes["E"]->weight > 0 && T.root->branchLength == 0;
        float maxw = 0; for (auto &kv : T.allNodes) if (kv.second->is_leaf()) maxw = std::max(maxw, kv.second->weight);
        ok = ok && maxw == 1.0f;                                     // calSeqWeight normalises to max 1
        CHECK("newick_parse_quirks", ok);
        phylogeny::assignSinglePartition(T.root);
        Tree *sub = new Tree(T.root, true);                          // binarise + reroot at the diameter centre
        bool bin = true; size_t leaves = 0;
        std::vector<Node *> st{sub->root};                           // (allNodes may keep spliced-out unary nodes; walk the tree itself)
        while (!st.empty()) {
            Node *n = st.back(); st.pop_back();
            if (n->is_leaf()) ++leaves; else bin = bin && n->children.size() == 2;
            for (Node *c : n->children) { bin = bin && c->parent == n; st.push_back(c); }
        }
        CHECK("reroot_gives_binary_tree_with_all_leaves", bin && leaves == 5);
        std::vector<NodePairVec> levels;
        progressive::scheduling(sub->root, levels, 0);
        size_t pairs = 0; for (auto &l : levels) pairs += l.size();
        CHECK("schedule_has_n_minus_1_pairs", pairs == 4 && !levels.empty() && !levels[0].empty());
        delete sub;
    }
    // ---- reroot (tree.cpp:588-696), derived by hand: the longest path of ((((A,B),C),D),(E,F)) has 7 nodes whichever leaf the search starts from
    // (the start leaf is the first leaf of an unordered_map: the quirk the level batches of real trees depend on), its middle node is X = (((A,B),C),D).
    // Every edge of the path X -> old root turns over: the old root hangs under X with the edge's length 7 and keeps (E,F) as its only child, so it is
    // spliced out and (E,F) carries 10 + 7; X now has three children [((A,B),C), D, (E,F)], of which the first two get a joint of length 0; the root
    // keeps the NAME of the old root.
    {
        const std::string f = tmp + "/reroot.nwk";
        { std::ofstream o(f); o << "((((A:1,B:2):3,C:4):5,D:6):7,(E:8,F:9):10);\n"; }
        Tree T(f);
        const std::string rootName = T.root->identifier;
        phylogeny::assignSinglePartition(T.root);
        Tree *sub = new Tree(T.root, true);
        std::function<std::string(Node *)> show = [&](Node *n) -> std::string {
            std::string s;
            if (n->is_leaf()) s = n->identifier;
            else { s = "("; for (size_t c = 0; c < n->children.size(); ++c) { if (c) s += ","; s += show(n->children[c]); if (n->children[c]->parent != n) s += "!"; } s += ")"; }
            char b[32]; snprintf(b, sizeof b, ":%g", (double)n->branchLength);
            return s + b;
        };
        const std::string got = show(sub->root);
        bool ok = got == "((((A:1,B:2):3,C:4):5,D:6):0,(E:8,F:9):17):0" && sub->root->identifier == rootName && sub->root->parent == nullptr;
        ok = ok && sub->allNodes.at(rootName) == sub->root && sub->m_numLeaves == 6 && sub->root->level == 1;
        for (auto &kv : sub->allNodes) ok = ok && kv.second->identifier == kv.first;       // every key still names its node
        if (!ok) printf("  reroot gave %s\n", got.c_str());
        CHECK("reroot_turns_the_path_over_and_keeps_the_root_s_name", ok);
        delete sub;
    }
    // ---- subtree ownership of a sharded run (progressive::planOwnership; used by gpu/align_owned.cpp) ----
    {
        // a random binary tree of 600 leaves with uniform splits (as twilight_amd/synth.py makes them: unbalanced on purpose)
        std::mt19937 rng(12345);
        int leafNo = 0;
        std::function<std::string(int)> grow = [&](int n) -> std::string {
            if (n == 1) return "s" + std::to_string(leafNo++) + ":0.01";
            const int left = 1 + (int)(rng() % (unsigned)(n - 1));
            return "(" + grow(left) + "," + grow(n - left) + "):0.01";
        };
        const std::string f = tmp + "/own.nwk";
        { std::ofstream o(f); o << grow(600) << ";\n"; }
        Tree T(f);
        phylogeny::assignSinglePartition(T.root);
        Tree *sub = new Tree(T.root, true);
        std::vector<NodePairVec> levels;
        progressive::scheduling(sub->root, levels, 0);
        for (int world : {2, 4, 8}) {
            const progressive::OwnershipPlan plan = progressive::planOwnership(sub, levels, world);
            bool ok = plan.cut >= 0 && plan.cut + 1 < (int)levels.size() && (int)plan.owner.size() == plan.cut + 1 && (int)plan.load.size() == world;
            long long below = 0, aboveP = 0;
            for (int l = 0; l < (int)levels.size(); ++l) (l <= plan.cut ? below : aboveP) += (long long)levels[l].size();
            ok = ok && aboveP + 1 >= 8 * world;                                   // the cut leaves 8 subtrees per rank ...
            long long aboveNext = aboveP - (long long)levels[plan.cut + 1].size();
            ok = ok && (plan.cut + 2 >= (int)levels.size() || aboveNext + 1 < 8 * world);   // ... and is the highest level that does
            long long sum = 0, mx = 0;
            for (long long x : plan.load) { sum += x; mx = std::max(mx, x); }
            ok = ok && sum == below && plan.subtrees >= 4 * world;                  // (leaves that only take part above the cut are subtrees without a pair: not counted)
            ok = ok && mx * world <= (world <= 4 ? 2 : 3) * sum + 8 * world;      // the longest-first deal of >= 8 subtrees per rank keeps the loads together (a 10 000-leaf tree: within 0.3 %)
            // data flow stays inside an owner: whoever produced an operand (or the child an internal operand adopts from) owns the pair that consumes it
            std::unordered_map<const Node *, int> producer;
            for (int l = 0; l <= plan.cut && ok; ++l)
                for (size_t i = 0; i < levels[l].size() && ok; ++i) {
                    const int me = plan.owner[l][i];
                    ok = ok && me >= 0 && me < world;
                    for (const Node *x : {levels[l][i].first, levels[l][i].second}) {
                        if (producer.count(x)) ok = ok && producer[x] == me;
                        if (!x->is_leaf()) for (const Node *c : x->children) if (producer.count(c)) ok = ok && producer[c] == me;
                    }
                    producer[levels[l][i].first] = me; producer[levels[l][i].second] = me;
                }
            CHECK(("ownership_plan_world_" + std::to_string(world)).c_str(), ok);
        }
        const progressive::OwnershipPlan none = progressive::planOwnership(sub, levels, 200);      // 8 x 200 subtrees do not exist in a 600-leaf tree
        CHECK("ownership_plan_none_for_too_many_ranks", none.cut == -1 && none.owner.empty());
        CHECK("ownership_plan_none_for_one_rank", progressive::planOwnership(sub, levels, 1).cut == -1);
        delete sub;
    }
    return g_fail ? 1 : 0;
}
