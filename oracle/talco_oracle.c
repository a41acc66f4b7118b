/*
 * oracle/talco_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see talco_oracle.h).
 *
 * Plain-C restatement of the reference CPU aligner.  Every block cites the
 * reference lines it follows (paths relative to /root/reference/src/).  The
 * data layout deliberately mirrors the reference's (offset-addressed rotating
 * rows, rows allocated and initialised per tile) so that reads of cells just
 * outside a stored band return what the reference's arrays would hold.
 *
 * Build: gcc -O2 -ffp-contract=off (no -ffast-math): every float operation is
 * one IEEE-754 binary32 operation in the order written.
 */
#include "talco_oracle.h"

#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define I_BOUNDARY (-2)   /* TALCO-XDrop.cpp:33 */
#define D_BOUNDARY (-3)   /* TALCO-XDrop.cpp:34 */

typedef struct {
    int8_t *d;
    size_t n, cap;
} bytes_t;

static void bytes_push(bytes_t *b, int8_t v)
{
    if (b->n == b->cap) {
        size_t nc = b->cap ? b->cap * 2 : 4096;
        b->d = (int8_t *)realloc(b->d, nc);
        b->cap = nc;
    }
    b->d[b->n++] = v;
}

/* ---- column score: TALCO-XDrop.cpp:373-444 (TALCO_SIMD branches) ---------------------------- */
float twlo_column_score(const twlo_params *p, const float *r, const float *q, float denom)
{
    float numerator = 0.0f;
    const float gc = p->gap_char;
    if (p->P == 6) {
        /* :378-393  five masked 8-lane rounds; lanes 0..4 live; sumvec = 0 + (q*M)*r */
        for (int l = 0; l < 5; ++l) {
            const float *row = p->matrix + 5 * l;
            const float rl = r[l];
            float t0 = 0.0f + (q[0] * row[0]) * rl;
            float t1 = 0.0f + (q[1] * row[1]) * rl;
            float t2 = 0.0f + (q[2] * row[2]) * rl;
            float t3 = 0.0f + (q[3] * row[3]) * rl;
            float t4 = 0.0f + (q[4] * row[4]) * rl;
            numerator += (t0 + t1 + t2 + t3 + t4);
        }
        /* :394-395 */
        for (int l = 0; l < 5; ++l) numerator += r[l] * q[5] * gc;
        for (int m = 0; m < 5; ++m) numerator += r[5] * q[m] * gc;
    } else {
        /* :409-430 */
        for (int l = 0; l < 21; ++l) {
            const float *row = p->matrix + 21 * l;
            const float rl = r[l];
            float v[8];
            for (int t = 0; t < 8; ++t) {
                float s = 0.0f + (q[t] * row[t]) * rl;      /* m = 0 block  */
                s = s + (q[8 + t] * row[8 + t]) * rl;       /* m = 8 block  */
                v[t] = s;
            }
            for (int m = 16; m < 21; ++m) numerator += rl * q[m] * row[m];   /* :423-425 scalar tail first */
            numerator += (v[0] + v[1] + v[2] + v[3] + v[4] + v[5] + v[6] + v[7]);
        }
        /* :432-433 */
        for (int l = 0; l < 21; ++l) numerator += r[l] * q[21] * gc;
        for (int m = 0; m < 21; ++m) numerator += r[21] * q[m] * gc;
    }
    return numerator / denom;   /* :444 */
}

/* ---- Reduction_tree: TALCO-XDrop.cpp:110-119 ------------------------------------------------ */
static int32_t uniform_or_minus1(const int32_t *C, int32_t start, int32_t length)
{
    int32_t conv = C[start];
    for (int32_t i = start + 1; i <= start + length; ++i)
        if (C[i] != conv) return -1;
    return conv;
}

/* ---- Traceback: TALCO-XDrop.cpp:134-231 ----------------------------------------------------- */
static void traceback(const int32_t *ftr_len, const int32_t *ftr_low, int32_t start_addr, int32_t start_ftr,
                      int8_t start_state, int32_t start_qidx, int32_t start_ridx, const int8_t *tb, size_t tb_n,
                      bytes_t *out, int first_tile)
{
    int32_t addr = start_addr;
    int32_t ftr = (int16_t)start_ftr;          /* the reference narrows these three to int16 (:138-141) */
    int32_t idx = (int16_t)start_qidx;
    int32_t qidx = (int16_t)start_qidx;
    int32_t ridx = (int16_t)start_ridx;
    int8_t state = start_state;
    while (ftr >= 0) {
        int8_t v = (addr >= 0 && (size_t)addr < tb_n) ? tb[addr] : 0;
        int8_t dir;
        if (state == 0) {                       /* :161-185 */
            state = v & 0x03;
            if (state == 0) dir = 0;
            else if (state == 1) { dir = 1; state = (v & 0x04) ? 1 : 0; }
            else                 { dir = 2; state = (v & 0x08) ? 2 : 0; }
        } else if (state == 1) {                /* :186-192 */
            dir = 1; state = (v & 0x04) ? 1 : 0;
        } else {                                /* :193-200 */
            dir = 2; state = (v & 0x08) ? 2 : 0;
        }
        /* :201-217 ragged-address update, all terms use the pre-move idx/ftr */
        if (ftr > 0) addr = addr - (idx - ftr_low[ftr] + 1) - ftr_len[ftr - 1];
        if (dir == 0) {
            if (ftr > 1) addr = addr - ftr_len[ftr - 2] + (idx - ftr_low[ftr - 2]);
            ftr -= 2; idx -= 1; qidx--; ridx--;
        } else if (dir == 1) {
            if (ftr > 0) addr = addr + (idx - ftr_low[ftr - 1]);
            ftr -= 1; idx -= 1; qidx--;
        } else {
            if (ftr > 0) addr = addr + (idx - ftr_low[ftr - 1] + 1);
            ftr -= 1; ridx--;
        }
        bytes_push(out, dir);
        if (first_tile && (ridx < 0 || qidx < 0)) break;   /* :219 */
    }
    if (first_tile) {                           /* :221-230 */
        while (ridx > -1) { bytes_push(out, 2); ridx--; }
        while (qidx > -1) { bytes_push(out, 1); qidx--; }
    }
}

typedef struct {
    const twlo_params *p;
    const float *ref, *qry;
    int32_t R, Q;
    const float *gop_r, *gex_r, *gop_q, *gex_q;
    float ref_num, qry_num;
    twlo_stats *st;
    twlo_trace_fn trace;
    void *trace_user;
} ctx_t;

/* ---- Tile: TALCO-XDrop.cpp:233-689 ---------------------------------------------------------- */
static void tile_run(const ctx_t *c, int32_t *reference_idx, int32_t *query_idx, bytes_t *aln,
                     int *last_tile, int tile, int16_t *err)
{
    const twlo_params *p = c->p;
    const int P = p->P;
    const float inf = 2.0 * p->xdrop + 1.0;                 /* :252 (double arithmetic narrowed to float) */
    const int32_t marker = p->marker;
    int converged = 0, conv_logic = 0;
    int32_t reference_length = c->R - *reference_idx;        /* :256-257 */
    int32_t query_length = c->Q - *query_idx;
    int32_t mn = reference_length < query_length ? reference_length : query_length;
    const int32_t fLen = p->flen < mn ? p->flen : mn;       /* :258 */
    float max_score = 0, max_score_prime = -inf;            /* :259 */
    float conv_score = 0;
    int32_t conv_value = 0, conv_ref_idx = 0, conv_query_idx = 0;
    int32_t tb_start_addr = 0, tb_start_ftr = 0;
    int32_t tb_state = 0;
    const float denominator = c->ref_num * c->qry_num;       /* :269 */
    const float gapOpenAtEnds = p->gap_open;                 /* :272-275 with alnType == 0 */
    const float gapExtendAtEnds = p->gap_extend;

    if (reference_length < 0 || query_length < 0) {          /* :313-320 */
        *err = 3; aln->n = 0; if (c->st) { c->st->err3_reason = 1; c->st->err3_tile = tile; } return;
    }

    /* :277-311 rotating rows; one guard element so that the reference's read at
       offset == width (possible when width == fLen) stays inside our buffer. */
    const size_t rowlen = (size_t)(fLen > 0 ? fLen : 1) + 1;
    float *fbuf = (float *)malloc(sizeof(float) * rowlen * 7);
    int32_t *ibuf = (int32_t *)malloc(sizeof(int32_t) * rowlen * 7);
    float *S[3], *I[2], *D[2];
    int32_t *CS[3], *CI[2], *CD[2];
    for (int s = 0; s < 3; ++s) { S[s] = fbuf + rowlen * s; CS[s] = ibuf + rowlen * s; }
    for (int s = 0; s < 2; ++s) {
        I[s] = fbuf + rowlen * (3 + s); D[s] = fbuf + rowlen * (5 + s);
        CI[s] = ibuf + rowlen * (3 + s); CD[s] = ibuf + rowlen * (5 + s);
    }
    for (size_t t = 0; t < rowlen; ++t) {
        for (int s = 0; s < 3; ++s) { S[s][t] = -1; CS[s][t] = -1; }
        for (int s = 0; s < 2; ++s) { I[s][t] = -1; D[s][t] = -1; CI[s][t] = I_BOUNDARY; CD[s][t] = D_BOUNDARY; }
    }
    int32_t L[3] = {0, 1, 2}, U[3] = {0, -1, -2};           /* :296-297 */

    bytes_t tb = {0, 0, 0};
    int32_t *ftr_len = (int32_t *)malloc(sizeof(int32_t) * (size_t)(marker + 2));
    int32_t *ftr_low = (int32_t *)malloc(sizeof(int32_t) * (size_t)(marker + 2));
    int32_t n_ftr = 0, ftr_addr = 0, last_k = 0, prev_conv_s = -1;

#define TILE_FREE() do { free(fbuf); free(ibuf); free(tb.d); free(ftr_len); free(ftr_low); } while (0)

    for (int32_t k = 0; k < reference_length + query_length - 1; ++k) {   /* :321 */
        const int c0 = k % 3, c1 = (k + 2) % 3, c2 = (k + 1) % 3;         /* rows of k, k-1, k-2 */
        const int b0 = k % 2, b1 = (k + 1) % 2;
        const int32_t Lk = L[c0], Uk = U[c0];
        if (Lk >= Uk + 1) {                                   /* :323-329 */
            *last_tile = 1; *err = 1; aln->n = 0; TILE_FREE(); return;
        }
        if (Uk - Lk + 1 > fLen) {                             /* :331-338 */
            *last_tile = 1; *err = 2; aln->n = 0; TILE_FREE(); return;
        }
        if (k <= marker) {                                    /* :340-344 */
            ftr_len[n_ftr] = Uk - Lk + 1;
            ftr_low[n_ftr] = Lk;
            ++n_ftr;
            ftr_addr += Uk - Lk + 1;
        }
        if (c->st) {
            c->st->cells += (uint64_t)(Uk - Lk + 1);
            c->st->diags += 1;
            if (Uk - Lk + 1 > c->st->max_width) c->st->max_width = Uk - Lk + 1;
        }
        const int32_t w1 = U[c1] - L[c1];                     /* U-L of diagonal k-1 */
        const int32_t w2 = U[c2] - L[c2];                     /* U-L of diagonal k-2 */

        for (int32_t i = Lk; i < Uk + 1; ++i) {               /* :353  i: query index, j: reference index */
            int8_t ptr = 0;
            int Iptr = 0, Dptr = 0;
            const int32_t j = k - i;                          /* :358-359 reduce to k - i */
            float match = -inf, insOp = -inf, delOp = -inf, insExt = -inf, delExt = -inf;   /* :364 */
            const int32_t offset = i - Lk;                    /* :365-368 */
            const int32_t offsetDiag = Lk - L[c2] + offset - 1;
            const int32_t offsetUp = Lk - L[c1] + offset;
            const int32_t offsetLeft = offsetUp - 1;
            const int diag_ok = (offsetDiag >= 0) && (offsetDiag <= w2);
            const int edge0 = (tile == 0) && (i == 0 || j == 0);
            if (k == 0 || diag_ok || edge0) {                 /* :369-371 */
                const float sim = twlo_column_score(p, c->ref + (size_t)P * (size_t)(*reference_idx + j),
                                                    c->qry + (size_t)P * (size_t)(*query_idx + i), denominator);
                if (edge0) {                                  /* :445-448 */
                    if (i == 0 && j == 0) match = sim;
                    else {
                        int32_t a = *reference_idx + j, b = *query_idx + i;
                        int32_t far = (a > b ? a : b) - 1;
                        if (far < 0) far = 0;
                        match = sim + gapOpenAtEnds + gapExtendAtEnds * far;
                    }
                } else if (offsetDiag < 0) match = sim;       /* :449 */
                else match = S[c2][offsetDiag] + sim;         /* :450 */
            }
            const float gop_ref = c->gop_r[*reference_idx + j];   /* :452-455 */
            const float gop_qry = c->gop_q[*query_idx + i];
            const float gex_ref = c->gex_r[*reference_idx + j];
            const float gex_qry = c->gex_q[*query_idx + i];
            if (offsetUp >= 0 && offsetUp <= w1) {            /* :456-459 */
                delOp = S[c1][offsetUp] + gop_ref;
                delExt = D[b1][offsetUp] + gex_ref;
            }
            if (offsetLeft >= 0 && offsetLeft <= w1) {        /* :460-463 */
                insOp = S[c1][offsetLeft] + gop_qry;
                insExt = I[b1][offsetLeft] + gex_qry;
            }
            float Iv = insOp, Dv = delOp;                     /* :464-475 */
            if (insExt >= insOp) { Iv = insExt; Iptr = 1; }
            if (delExt >= delOp) { Dv = delExt; Dptr = 1; }
            float Sv;                                         /* :477-494 */
            if (match >= Iv) {
                if (match >= Dv) { Sv = match; ptr = 0; }
                else { Sv = Dv; ptr = 2; }
            } else if (Iv > Dv) { Sv = Iv; ptr = 1; }
            else { Sv = Dv; ptr = 2; }
            if (Sv < max_score - p->xdrop) Sv = -inf;         /* :495-497 */
            I[b0][offset] = Iv; D[b0][offset] = Dv; S[c0][offset] = Sv;
            if (max_score_prime < Sv) max_score_prime = Sv;   /* :501-503 */

            if (k == marker - 1) {                            /* :520-526 */
                CS[c0][offset] = (3 << 16) | (i & 0xFFFF);
            } else if (k == marker) {
                CS[c0][offset] = (0 << 16) | (i & 0xFFFF);
                CI[b0][offset] = (1 << 16) | (i & 0xFFFF);
                CD[b0][offset] = (2 << 16) | (i & 0xFFFF);
            } else if (k >= marker + 1) {                     /* :527-547 */
                if (Iptr) CI[b0][offset] = (offsetLeft >= 0) ? CI[b1][offsetLeft] : I_BOUNDARY;
                else CI[b0][offset] = (offsetLeft >= 0 && CS[c1][offsetLeft] != -1) ? CS[c1][offsetLeft] : I_BOUNDARY;
                if (Dptr) CD[b0][offset] = (offsetUp >= 0) ? CD[b1][offsetUp] : D_BOUNDARY;
                else CD[b0][offset] = (offsetUp >= 0 && CS[c1][offsetUp] != -1) ? CS[c1][offsetUp] : D_BOUNDARY;
                if (ptr == 0) {
                    if (!diag_ok && c->st) c->st->oob_diag++;
                    /* The reference reads CS[(k+1)%3][offsetDiag] unguarded (:541).  Without a diagonal predecessor that offset is
                       negative or past the stored band (memory outside the row, or a stale slot): undefined in the reference.
                       Defined here, and identically in the HIP kernel, as "unset" (-1).  It arises for pruned edge cells (trimmed
                       on the same diagonal, value never used) and for first-row/column cells of tile 0 that are still in the band
                       at the marker (tiny markers, or one sequence far shorter than the other). */
                    CS[c0][offset] = diag_ok ? CS[c2][offsetDiag] : -1;
                } else if (ptr == 1) CS[c0][offset] = CI[b0][offset];
                else CS[c0][offset] = CD[b0][offset];
            }
            if (Iptr) ptr |= 0x04;                            /* :548-553 */
            if (Dptr) ptr |= 0x08;
            if (k <= marker) bytes_push(&tb, ptr);            /* :554-557 */
        }

        int32_t newL = Lk, newU = Uk;                         /* :563-583 */
        while (newL <= Uk && S[c0][newL - Lk] <= -inf) newL++;
        while (newU >= Lk && S[c0][newU - Lk] <= -inf) newU--;

        if (!converged && k < reference_length + query_length - 2) {   /* :585-595 */
            if (newU < newL && c->st) c->st->empty_reduce++;
            int32_t conv_I = uniform_or_minus1(CI[b0], newL - Lk, newU - newL);
            int32_t conv_D = uniform_or_minus1(CD[b0], newL - Lk, newU - newL);
            int32_t conv_S = uniform_or_minus1(CS[c0], newL - Lk, newU - newL);
            if (conv_I == conv_D && conv_I == conv_S && prev_conv_s == conv_S && conv_I != -1) {
                converged = 1;
                conv_value = prev_conv_s;
                conv_score = max_score_prime;
            }
            prev_conv_s = conv_S;
        }
        if (c->trace) c->trace(c->trace_user, tile, k, Lk, Uk, max_score_prime);

        {                                                     /* :597-604 */
            int32_t v1 = query_length - 1;
            int32_t v2 = k + 2 - reference_length;
            int32_t v3 = newU + 1;
            int32_t Lprime = v2 > 0 ? v2 : 0;
            L[c2] = newL > Lprime ? newL : Lprime;            /* (k+1)%3 == c2 */
            U[c2] = v1 < v3 ? v1 : v3;
        }
        max_score = (max_score_prime < 0) ? 0 : max_score_prime;   /* :607 */
        last_k = k;
#ifdef TWLO_TILE_HOOK      /* study builds only (tests/study/): never defined for the checker library */
        TWLO_TILE_HOOK
#endif
        if (converged && max_score > conv_score) { conv_logic = 1; break; }   /* :609-612 */
    }

    int bad_conv = 0;
    if (conv_logic) {                                         /* :615-622 */
        conv_query_idx = conv_value & 0xFFFF;
        tb_state = (conv_value >> 16) & 0xFFFF;
    } else if (last_k < marker) {                             /* :625-632 global end cell */
        conv_query_idx = query_length - 1;
        conv_ref_idx = reference_length - 1;
        tb_start_addr = ftr_addr - 1;
        tb_start_ftr = last_k;
        tb_state = 0;
        *last_tile = 1;
    } else {                                                  /* :633-642 */
        conv_query_idx = CS[last_k % 3][0] & 0xFFFF;
        tb_state = (CS[last_k % 3][0] >> 16) & 0xFFFF;
    }
    if (conv_logic || last_k >= marker) {
        /* :618-622 == :636-641.  A boundary sentinel (-2/-3) or an unset -1 here makes the
           reference index its traceback store far out of range (undefined behaviour); we
           report that as errorType 3 instead of reproducing the fault. */
        if (tb_state > 3 || n_ftr < 2) bad_conv = 2;
        else {
            conv_ref_idx = marker - conv_query_idx - ((tb_state == 3) ? 1 : 0);
            tb_start_addr = ftr_addr - ftr_len[n_ftr - 1];
            tb_start_addr = (tb_state == 3)
                ? tb_start_addr - ftr_len[n_ftr - 2] + (conv_query_idx - ftr_low[n_ftr - 2])
                : tb_start_addr + (conv_query_idx - ftr_low[n_ftr - 1]);
            tb_start_ftr = (tb_state == 3) ? n_ftr - 2 : n_ftr - 1;
            if (tb_start_addr < 0 || (size_t)tb_start_addr >= tb.n) bad_conv = 3;
            else if (conv_ref_idx < 0) bad_conv = 4;
        }
    }
    if (bad_conv) { *err = 3; *last_tile = 1; aln->n = 0; if (c->st) { c->st->err3_reason = bad_conv; c->st->err3_tile = tile; } TILE_FREE(); return; }

    *reference_idx += conv_ref_idx;                           /* :654-655 */
    *query_idx += conv_query_idx;
    reference_length = c->R - *reference_idx;
    query_length = c->Q - *query_idx;
    if (reference_length < 0 || query_length < 0) {           /* :659-668 */
        *err = 3; aln->n = 0; if (c->st) { c->st->err3_reason = 5; c->st->err3_tile = tile; } TILE_FREE(); return;
    }
    if (*reference_idx == c->R - 1 && *query_idx < c->Q - 1) {    /* :671-674 */
        for (int32_t q = 0; q < c->Q - *query_idx - 1; ++q) bytes_push(aln, 1);
        *last_tile = 1;
    }
    if (*query_idx == c->Q - 1 && *reference_idx < c->R - 1) {    /* :675-678 */
        for (int32_t r = 0; r < c->R - *reference_idx - 1; ++r) bytes_push(aln, 2);
        *last_tile = 1;
    }
    if (*reference_idx == c->R - 1 && *query_idx == c->Q - 1) *last_tile = 1;   /* :679 */

    traceback(ftr_len, ftr_low, tb_start_addr, tb_start_ftr, (int8_t)(tb_state % 3), conv_query_idx, conv_ref_idx,
              tb.d, tb.n, aln, tile == 0);                     /* :681-682 */
    TILE_FREE();
#undef TILE_FREE
}

/* ---- Align_freq: TALCO-XDrop.cpp:62-108 ----------------------------------------------------- */
int twlo_align_pair(const twlo_params *p, const float *ref, int32_t R, const float *qry, int32_t Q,
                    const float *gop_ref, const float *gex_ref, const float *gop_qry, const float *gex_qry,
                    float ref_num, float qry_num, int8_t *aln, int32_t *aln_len, int16_t *err,
                    twlo_stats *stats, twlo_trace_fn trace, void *trace_user)
{
    ctx_t c = {p, ref, qry, R, Q, gop_ref, gex_ref, gop_qry, gex_qry, ref_num, qry_num, stats, trace, trace_user};
    int32_t reference_idx = 0, query_idx = 0;
    int last_tile = 0, tile = 0;
    int32_t n = 0;
    bytes_t tile_aln = {0, 0, 0};
    *err = 0;
    *aln_len = 0;
    if (R < 1 || Q < 1 || p->marker < 2 || p->marker > 32767) { *err = 3; return 0; }
    while (!last_tile) {
        tile_aln.n = 0;
        tile_run(&c, &reference_idx, &query_idx, &tile_aln, &last_tile, tile, err);
        if (stats) stats->tiles += 1;
        if (tile_aln.n == 0) { free(tile_aln.d); *aln_len = 0; return 0; }      /* :94-97 */
        for (long i = (long)tile_aln.n - 1; i >= 0; --i) {                     /* :98-102 */
            if (i == (long)tile_aln.n - 1 && tile > 0) continue;
            if (n >= R + Q) { free(tile_aln.d); *err = 3; *aln_len = 0; if (stats) stats->err3_reason = 6; return 0; }
            aln[n++] = tile_aln.d[i];
        }
        tile++;
    }
    free(tile_aln.d);
    *aln_len = n;
    return 0;
}

int twlo_align_batch(const twlo_params *p, int32_t n_pairs, int32_t seq_len, const float *freq,
                     const float *gap_open, const float *gap_extend, const int32_t *len, const int32_t *num,
                     int8_t *aln_out, int32_t *aln_len_out, int16_t *err_out, int32_t threads, twlo_stats *stats)
{
    const size_t P = (size_t)p->P;
    twlo_stats total;
    memset(&total, 0, sizeof total);
#ifdef _OPENMP
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
#endif
    for (int32_t n = 0; n < n_pairs; ++n) {
        twlo_stats st;
        memset(&st, 0, sizeof st);
        const float *fr = freq + ((size_t)n * 2 + 0) * (size_t)seq_len * P;
        const float *fq = freq + ((size_t)n * 2 + 1) * (size_t)seq_len * P;
        const float *go = gap_open + (size_t)n * 2 * (size_t)seq_len;
        const float *ge = gap_extend + (size_t)n * 2 * (size_t)seq_len;
        const int32_t R = len[2 * n], Q = len[2 * n + 1];
        int8_t *out = aln_out + (size_t)n * 2 * (size_t)seq_len;
        if (R <= 0 || Q <= 0) { aln_len_out[n] = 0; err_out[n] = 0; continue; }
        twlo_align_pair(p, fr, R, fq, Q, go, ge, go + seq_len, ge + seq_len, (float)num[2 * n], (float)num[2 * n + 1],
                        out, &aln_len_out[n], &err_out[n], &st, NULL, NULL);
#ifdef _OPENMP
#pragma omp critical
#endif
        {
            total.cells += st.cells; total.diags += st.diags; total.tiles += st.tiles;
            if (st.max_width > total.max_width) total.max_width = st.max_width;
            total.empty_reduce += st.empty_reduce; total.oob_diag += st.oob_diag;
            if (st.err3_reason) { total.err3_reason = st.err3_reason; total.err3_tile = st.err3_tile; }
        }
    }
    if (stats) *stats = total;
    (void)threads;
    return 0;
}
