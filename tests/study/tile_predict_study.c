/* tests/study/tile_predict_study.c -- STUDY AID (test infrastructure, uses the oracle): how well can the start cells of
 * all tiles of a pair be predicted without running the tiles one after the other?
 *
 * For one dumped pair (oracle/e2e_oracle with TWLO_DUMP_PAIRS) it runs the true tile chain (Align_freq, TALCO-XDrop.cpp:62-108),
 * then for every tile boundary starts a "scout" tile LEAD diagonals earlier from a cell DELTA rows off the true path and asks
 * whether the scout's traceback path runs through the true start cell of the next tile.
 *   cc -O2 -ffp-contract=off -fopenmp -o /tmp/tps tests/study/tile_predict_study.c -lm
 *   /tmp/tps pair.bin [lead] [tailmarg] [scout xdrop] [1 = stop at the marker like the kernel's scouts]
 *   /tmp/tps pair.bin lead tailmarg xdrop 1 W OFF [fallback lead] [min %] [gap %] [interpolate over n boundaries] [max jump per boundary] [windows on either side]
 *       the anchored scouts of round 5 (talco_nuc.hip.h, mt_anchor_kernel): the start cell moved to the diagonal offset (-OFF .. OFF-1) on which the consensus letters of
 *       the two profiles agree most often within +-W columns; the kernel's setting is  96 40 5000 1 32 128 320 55 12 3 12 4.  TPS_VERBOSE=1 prints every boundary.
 *   Pairs come from  TWLO_DUMP_PAIRS=<dir> [TWLO_DUMP_MAX_PAIRS=n] [TWLO_DUMP_EVERY=k] oracle/e2e_oracle -t tree -i fasta -o out.aln
 */
/* scouts that do not wait for convergence: after the marker diagonal take the better of the best cells of diagonals marker-1 / marker
 * (the rule of the speculative tile start, talco_nuc.hip.h) and trace back from there */
static int g_stop_at_marker = 0;
#define TWLO_TILE_HOOK                                                                                              \
    if (g_stop_at_marker && tile == 7777 && k == marker) {                                                          \
        float b0 = -inf, b1 = -inf; int i0 = -1, i1 = -1;                                                           \
        for (int32_t i = L[c1]; i <= U[c1]; ++i) if (S[c1][i - L[c1]] > b0) { b0 = S[c1][i - L[c1]]; i0 = i; }     \
        for (int32_t i = Lk; i <= Uk; ++i) if (S[c0][i - Lk] > b1) { b1 = S[c0][i - Lk]; i1 = i; }                 \
        if (i0 >= 0 || i1 >= 0) { conv_value = (b1 >= b0) ? (i1 & 0xFFFF) : ((3 << 16) | (i0 & 0xFFFF)); conv_logic = 1; break; } \
    }
#include "../../oracle/talco_oracle.c"
#include <stdio.h>

static float M5[25];

typedef struct { int32_t r, q; } cell_t;

int main(int argc, char **argv)
{
    if (argc < 2) return 1;
    const int lead = argc > 2 ? atoi(argv[2]) : 512;
    const int marg = argc > 3 ? atoi(argv[3]) : 64;
    const int sxdrop = argc > 4 ? atoi(argv[4]) : 5000;
    g_stop_at_marker = argc > 5 ? atoi(argv[5]) : 0;
    FILE *f = fopen(argv[1], "rb");
    int32_t h[6];
    if (!f || fread(h, sizeof h, 1, f) != 1) return 2;
    const int P = h[0], R = h[1], Q = h[2];
    float *ref = malloc(sizeof(float) * P * R), *qry = malloc(sizeof(float) * P * Q);
    float *gor = malloc(4 * R), *ger = malloc(4 * R), *goq = malloc(4 * Q), *geq = malloc(4 * Q);
    if (fread(ref, 4, (size_t)P * R, f) != (size_t)P * R || fread(qry, 4, (size_t)P * Q, f) != (size_t)P * Q || fread(gor, 4, R, f) != (size_t)R ||
        fread(ger, 4, R, f) != (size_t)R || fread(goq, 4, Q, f) != (size_t)Q || fread(geq, 4, Q, f) != (size_t)Q) return 3;
    fclose(f);
    for (int i = 0; i < 5; ++i) for (int j = 0; j < 5; ++j) M5[5 * i + j] = (i == 4 || j == 4) ? 0.f : (i == j ? 18.f : ((i ^ j) == 2 ? -4.f : -8.f));
    twlo_params p = {P, M5, -50.f, -5.f, -5.f, 5000, 4096, 1024};
    ctx_t c = {&p, ref, qry, R, Q, gor, ger, goq, geq, (float)h[3], (float)h[4], NULL, NULL, NULL};

    /* true chain */
    cell_t starts[256]; int nt = 0;
    int32_t ri = 0, qi = 0; int last = 0, tile = 0; int16_t err = 0;
    bytes_t seg = {0, 0, 0};
    int8_t *full = malloc(R + Q + 8); int n = 0;
    while (!last) {
        starts[nt].r = ri; starts[nt].q = qi; nt++;
        seg.n = 0;
        tile_run(&c, &ri, &qi, &seg, &last, tile, &err);
        if (seg.n == 0) { printf("pair failed err %d\n", err); return 0; }
        for (long i = (long)seg.n - 1; i >= 0; --i) { if (i == (long)seg.n - 1 && tile > 0) continue; full[n++] = seg.d[i]; }
        tile++;
    }
    /* true path cells: cell index after each column; pathq[d] = q of the path cell on diagonal d (r+q), -1 if skipped */
    int32_t *pathq = malloc(4 * (R + Q + 2));
    for (int d = 0; d < R + Q + 2; ++d) pathq[d] = -1;
    { int r = -1, q = -1; for (int t = 0; t < n; ++t) { if (full[t] == 0) { r++; q++; } else if (full[t] == 1) q++; else r++; if (r >= 0 && q >= 0) pathq[r + q] = q; } }
    /* drift of the true path from the proportional diagonal */
    int maxdev = 0;
    for (int t = 1; t < nt; ++t) { int d = starts[t].r + starts[t].q; int pq = (int)((double)d * Q / (R + Q)); int dev = abs(pq - starts[t].q); if (dev > maxdev) maxdev = dev; }
    printf("pair R %d Q %d num %d %d tiles %d maxdev_from_proportional %d\n", R, Q, h[3], h[4], nt, maxdev);

    /* ---- what the kernel can do without knowing the path: the cell of diagonal d0 on the straight line between the corners, moved to the
       diagonal offset on which the consensus letters of the two profiles agree most often (argument 6: window half-width, 0 = off) ---- */
    const int seedW = argc > 6 ? atoi(argv[6]) : 0;
    const int seedOff = argc > 7 ? atoi(argv[7]) : 256;
    const int fbLead = argc > 8 ? atoi(argv[8]) : 0;        /* an anchor that is not trusted: the straight line with this lead (0 = always trust) */
    const int minPct = argc > 9 ? atoi(argv[9]) : 60;       /* trusted: matches >= minPct % of the window and the runner-up (not a neighbour offset) at least gapPct % of the window behind */
    const int gapPct = argc > 10 ? atoi(argv[10]) : 15;
    if (argc > 6) {
        unsigned char *cr = malloc(R), *cq = malloc(Q);
        for (int side = 0; side < 2; ++side) {
            const float *pf = side ? qry : ref; unsigned char *cc = side ? cq : cr; const int n = side ? Q : R;
            for (int i = 0; i < n; ++i) {
                int best = 0; float bc = pf[(size_t)P * i];
                for (int j = 1; j < P - 2; ++j) if (pf[(size_t)P * i + j] > bc) { bc = pf[(size_t)P * i + j]; best = j; }
                cc[i] = (bc > 0.0f && bc >= pf[(size_t)P * i + P - 1]) ? (unsigned char)best : (unsigned char)(100 + side);   /* gap-dominated / empty: matches nothing */
            }
        }
        /* pass 1: the anchor of every boundary (offset of q - r with the most agreeing consensus letters) and whether it is trusted;
           pass 2: an untrusted one takes the interpolation of its trusted neighbours (the path's drift from the straight line is smooth:
           a few rows per tile) when they are near, else the straight line with the long lead; pass 3: the scouts */
        const int interp = argc > 11 ? atoi(argv[11]) : 0;      /* 0 = no interpolation; n = neighbours up to n boundaries away */
        int *aO = calloc(nt + 1, sizeof(int)), *aOk = calloc(nt + 1, sizeof(int)), *aLead = calloc(nt + 1, sizeof(int));
#pragma omp parallel for schedule(dynamic, 1)
        for (int t = 1; t < nt; ++t) {
            const int dT = starts[t].r + starts[t].q;
            int d0 = dT - lead; if (d0 < 2) d0 = 2;
            if (getenv("TPS_NOMINAL")) { d0 = (p.marker - 1) * t - 1 - lead; if (d0 < 0) d0 = 0; }      /* as the kernel: the boundary's nominal range, not the true start */
            const int q0 = (int)((long long)d0 * Q / (R + Q)), r0 = d0 - q0;
            int bestO = 0, bestC = -1, secondC = -1;
            /* the window around the straight line's cell first; when its verdict is not trusted, windows moved along the diagonal (argument 13: how many
               on either side, 24 columns apart) -- the path's offset is the same a few columns on unless an indel lies between */
            const int nShift = argc > 13 ? atoi(argv[13]) : 0;
            int trusted = 0;
            for (int sh = 0; sh <= 2 * nShift && !trusted && seedW > 0; ++sh) {
                const int shift = (sh == 0) ? 0 : ((sh & 1) ? 24 * ((sh + 1) / 2) : -24 * (sh / 2));
                int bO = 0, bC = -1, sC = -1;
                for (int oo = 0; oo < 2 * seedOff; ++oo) {
                    const int o = (oo & 1) ? -((oo + 1) / 2) : oo / 2;       /* 0, -1, 1, -2, 2, ...: ties go to the smallest |o| */
                    int c = 0;
                    for (int i = -seedW; i < seedW; ++i) {
                        const int rr = r0 + shift + i, qq2 = q0 + shift + i + o;
                        if (rr >= 0 && rr < R && qq2 >= 0 && qq2 < Q && cr[rr] == cq[qq2]) ++c;
                    }
                    if (c > bC) { bC = c; bO = o; }
                }
                for (int o = -seedOff; o < seedOff; ++o) {
                    if (abs(o - bO) <= 2) continue;
                    int c = 0;
                    for (int i = -seedW; i < seedW; ++i) {
                        const int rr = r0 + shift + i, qq2 = q0 + shift + i + o;
                        if (rr >= 0 && rr < R && qq2 >= 0 && qq2 < Q && cr[rr] == cq[qq2]) ++c;
                    }
                    if (c > sC) sC = c;
                }
                trusted = !(bC * 100 < minPct * 2 * seedW || (bC - sC) * 100 < gapPct * 2 * seedW);
                if (sh == 0 || trusted) { bestO = bO; bestC = bC; secondC = sC; }
            }
            aO[t] = bestO;
            aOk[t] = !(fbLead > 0 && (bestC * 100 < minPct * 2 * seedW || (bestC - secondC) * 100 < gapPct * 2 * seedW));
        }
        int nint = 0;
        for (int t = 1; t < nt; ++t) {
            aLead[t] = lead;
            if (aOk[t]) continue;
            int l = -1, r = -1;
            for (int u = t - 1; u >= 1 && t - u <= interp; --u) if (aOk[u] == 1) { l = u; break; }
            for (int u = t + 1; u < nt && u - t <= interp; ++u) if (aOk[u] == 1) { r = u; break; }
            const int maxJump = argc > 12 ? atoi(argv[12]) : 1000;      /* neighbours whose offsets differ by more than this per boundary say nothing about what lies between */
            if (l > 0 && r > 0 && abs(aO[r] - aO[l]) <= maxJump * (r - l)) { aO[t] = aO[l] + (aO[r] - aO[l]) * (t - l) / (r - l); aOk[t] = 2; nint++; }
            else if (l > 0 && r > 0) { aO[t] = 0; aLead[t] = fbLead; }
            else { aO[t] = 0; aLead[t] = fbLead; }
        }
        if (getenv("TPS_VERBOSE")) for (int t = 1; t < nt; ++t) { const int dd = starts[t].r + starts[t].q - lead; int tq0 = pathq[dd]; if (tq0 < 0) tq0 = pathq[dd - 1]; printf("    t %d kind %d offset %d (true %d)\n", t, aOk[t], aO[t], 2 * (tq0 - (int)((long long)dd * Q / (R + Q)))); }
        int shits = 0, stot = 0, maxd = 0, nfb = 0; long sumd = 0, sdiag = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : shits, stot, sumd, nfb, sdiag) reduction(max : maxd)
        for (int t = 1; t < nt; ++t) {
            const int dT = starts[t].r + starts[t].q;
            int d0 = dT - aLead[t]; if (d0 < 2) d0 = 2;
            int smarker = 0;
            if (getenv("TPS_NOMINAL")) { d0 = (p.marker - 1) * t - 1 - aLead[t]; if (d0 < 0) d0 = 0; smarker = p.marker * t + 1 + marg - d0; }
            const int q0 = (int)((long long)d0 * Q / (R + Q));
            const int bestO = aO[t];
            if (!aOk[t]) nfb++;
            sdiag += dT - d0 + marg;
            /* keep the anti-diagonal: q - r changes by bestO (rounded to even) */
            int gq = q0 + (bestO >= 0 ? (bestO + 1) / 2 : -((-bestO + 1) / 2)) , gr = d0 - gq;
            if (gq < 0) { gq = 0; gr = d0; } if (gr < 0) { gr = 0; gq = d0; }
            if (gq >= Q || gr >= R) continue;
            int tq = pathq[d0]; if (tq < 0) tq = pathq[d0 - 1];
            const int dev = abs(gq - tq);
            sumd += dev; if (dev > maxd) maxd = dev;
            twlo_params sp = p; sp.marker = smarker ? smarker : dT - d0 + marg; sp.xdrop = sxdrop;
            ctx_t sc = c; sc.p = &sp;
            int32_t sr = gr, sq = gq; int sl = 0; int16_t se = 0; bytes_t sg = {0, 0, 0};
            tile_run(&sc, &sr, &sq, &sg, &sl, g_stop_at_marker ? 7777 : 1, &se);
            stot++;
            int hit = 0;
            if (sg.n) {
                int r = gr, q = gq;
                for (long i = (long)sg.n - 2; i >= 0; --i) {
                    if (sg.d[i] == 0) { r++; q++; } else if (sg.d[i] == 1) q++; else r++;
                    if (r == starts[t].r && q == starts[t].q) { hit = 1; break; }
                    if (r + q > dT) break;
                }
            }
            shits += hit;
            if (!hit && getenv("TPS_VERBOSE")) printf("    miss t %d kind %d offset %d dev %d rows lead %d\n", t, aOk[t], bestO, dev, aLead[t]);
            free(sg.d);
        }
        printf("  interpolated %d;", nint);
        printf("  seeded (window +-%d, offsets +-%d): %d / %d hit; rough start off the path by avg %.1f max %d rows; %d on the fallback lead; %ld scout diagonals\n", seedW, seedOff, shits, stot, stot ? (double)sumd / stot : 0.0, maxd, nfb, sdiag);
        return 0;
    }
    const int deltas[] = {0, 8, -8, 40, -40, 120, -120, 250, -250};
    int hits[9] = {0}, tot[9] = {0}, fail[9] = {0};
#pragma omp parallel for schedule(dynamic, 1)
    for (int t = 1; t < nt; ++t) {
        const int dT = starts[t].r + starts[t].q;
        int d0 = dT - lead; if (d0 < 2) d0 = 2;
        /* true path cell on d0 (or d0 - 1) */
        int qq = pathq[d0]; if (qq < 0) { d0--; qq = pathq[d0]; }
        if (qq < 0) continue;
        for (int di = 0; di < 9; ++di) {
            int gq = qq + deltas[di], gr = d0 - gq;
            if (gq < 0 || gr < 0 || gq >= Q || gr >= R) continue;
            twlo_params sp = p; sp.marker = dT - d0 + marg; sp.xdrop = sxdrop;
            ctx_t sc = c; sc.p = &sp;
            int32_t sr = gr, sq = gq; int sl = 0; int16_t se = 0; bytes_t sg = {0, 0, 0};
            tile_run(&sc, &sr, &sq, &sg, &sl, g_stop_at_marker ? 7777 : 1, &se);
#pragma omp atomic
            tot[di]++;
            if (sg.n == 0) {
#pragma omp atomic
                fail[di]++;
                free(sg.d); continue; }
            /* forward walk: seg reversed; its last element is the start cell's own column */
            int r = gr, q = gq, hit = 0;
            for (long i = (long)sg.n - 2; i >= 0; --i) {
                if (sg.d[i] == 0) { r++; q++; } else if (sg.d[i] == 1) q++; else r++;
                if (r == starts[t].r && q == starts[t].q) { hit = 1; break; }
                if (r + q > dT) break;
            }
            if (hit) {
#pragma omp atomic
                hits[di]++;
            }
            free(sg.d);
        }
    }
    for (int di = 0; di < 9; ++di) printf("  delta %5d: %d / %d hit (%d scout failures)\n", deltas[di], hits[di], tot[di], fail[di]);
    return 0;
}
