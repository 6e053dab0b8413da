/*
 * plslam_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 * See plslam_oracle.h for the parity status ("parity unpinned" for kNN order / ratio /
 * mutual semantics; distance pinned to the reference's in-tree popcount code).
 *
 * Build with -ffp-contract=off: the fp64 rows must not be FMA-contracted so that the
 * operation order written below (the reference's source order) is what executes.
 */
#include "plslam_oracle.h"

#include <limits.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------ */
/* distance                                                                             */
/* ------------------------------------------------------------------------------------ */

/* bitops_custom.hpp:83-96: for each 16-byte step, four u32 XOR + __builtin_popcount. */
int plo_hamming256(const uint8_t* a, const uint8_t* b)
{
    int out = 0;
    for (int i = 0; i <= PLO_DESC_BYTES - 16; i += 16) {
        uint32_t a0, a1, a2, a3, b0, b1, b2, b3;
        memcpy(&a0, a + i, 4);      memcpy(&b0, b + i, 4);
        memcpy(&a1, a + i + 4, 4);  memcpy(&b1, b + i + 4, 4);
        memcpy(&a2, a + i + 8, 4);  memcpy(&b2, b + i + 8, 4);
        memcpy(&a3, a + i + 12, 4); memcpy(&b3, b + i + 12, 4);
        out += __builtin_popcount(a0 ^ b0) + __builtin_popcount(a1 ^ b1) +
               __builtin_popcount(a2 ^ b2) + __builtin_popcount(a3 ^ b3);
    }
    return out;
}

/* FORB.cpp:78-101: SWAR popcount of each of the four u64 words of a^b. */
int plo_hamming256_swar(const uint8_t* a, const uint8_t* b)
{
    uint64_t ret = 0;
    for (int i = 0; i < PLO_DESC_BYTES / 8; ++i) {
        uint64_t pa, pb, v;
        memcpy(&pa, a + 8 * i, 8);
        memcpy(&pb, b + 8 * i, 8);
        v = pa ^ pb;
        v = v - ((v >> 1) & (uint64_t)~(uint64_t)0 / 3);
        v = (v & (uint64_t)~(uint64_t)0 / 15 * 3) + ((v >> 2) & (uint64_t)~(uint64_t)0 / 15 * 3);
        v = (v + (v >> 4)) & (uint64_t)~(uint64_t)0 / 255 * 15;
        ret += (uint64_t)(v * ((uint64_t)~(uint64_t)0 / 255)) >> (sizeof(uint64_t) - 1) * CHAR_BIT;
    }
    return (int)ret;
}

/* bitops_custom.hpp:58-76 lookup[] semantics, generated instead of tabulated. */
int plo_hamming256_lut(const uint8_t* a, const uint8_t* b)
{
    static int lut[256];
    static int init = 0;
    if (!init) {
        for (int v = 0; v < 256; ++v) {
            int c = 0;
            for (int k = 0; k < 8; ++k) c += (v >> k) & 1;
            lut[v] = c;
        }
        init = 1;
    }
    int out = 0;
    for (int i = 0; i < PLO_DESC_BYTES; ++i) out += lut[a[i] ^ b[i]];
    return out;
}

/* fast inner distance for the scans (same value as the three above; u64 builtin) */
static inline int ham_fast(const uint8_t* a, const uint8_t* b)
{
    uint64_t x[4], y[4];
    memcpy(x, a, 32);
    memcpy(y, b, 32);
    return __builtin_popcountll(x[0] ^ y[0]) + __builtin_popcountll(x[1] ^ y[1]) +
           __builtin_popcountll(x[2] ^ y[2]) + __builtin_popcountll(x[3] ^ y[3]);
}

/* ------------------------------------------------------------------------------------ */
/* kNN-2: OpenCV batchDistance(K=2) insertion semantics                                  */
/* ------------------------------------------------------------------------------------ */
void plo_knn2(const uint8_t* q, int32_t nq, const uint8_t* t, int32_t nt, int32_t* idx,
              int32_t* dist)
{
    for (int32_t i = 0; i < nq; ++i) {
        int32_t bd[2] = {INT32_MAX, INT32_MAX};
        int32_t bi[2] = {-1, -1};
        const uint8_t* qi = q + (size_t)i * PLO_DESC_BYTES;
        for (int32_t j = 0; j < nt; ++j) {
            const int32_t d = ham_fast(qi, t + (size_t)j * PLO_DESC_BYTES);
            if (d < bd[1]) {            /* strict '<' against the worst kept slot         */
                int k = 0;              /* K-2 = 0: shift slot 0 down while it is worse   */
                if (bd[0] > d) {        /* strict '>' => equal distance keeps lower index */
                    bd[1] = bd[0];
                    bi[1] = bi[0];
                    k = -1;
                }
                bd[k + 1] = d;
                bi[k + 1] = j;
            }
        }
        idx[2 * i] = bi[0];
        idx[2 * i + 1] = bi[1];
        dist[2 * i] = bd[0];
        dist[2 * i + 1] = bd[1];
    }
}

/* ------------------------------------------------------------------------------------ */
/* stvo-pl matchNNR / match                                                              */
/* ------------------------------------------------------------------------------------ */
/* keep_prior: [RECALL] stvo-pl matchNNR opens with `matches_12.resize(desc1.rows, -1)` -- a vector that already holds
 * desc1.rows entries (the matchGrid result of src/mapHandler.cpp:271 handed on to match() at :277, likewise :418/:424,
 * :591/:597, :706/:712) KEEPS them, and only rows that pass the ratio test are overwritten.  A fresh vector gives the
 * all -1 start, which is keep_prior = 0. */
static int32_t match_nnr_impl(const uint8_t* q, int32_t nq, const uint8_t* t, int32_t nt, float nnr, int32_t* m12,
                              int keep_prior)
{
    int32_t matches = 0;
    for (int32_t i = 0; i < nq; ++i) {
        int32_t idx[2], dist[2];
        plo_knn2(q + (size_t)i * PLO_DESC_BYTES, 1, t, nt, idx, dist);
        if (!keep_prior) m12[i] = -1;
        if (idx[1] < 0) continue;       /* nt < 2: defined as "no match" */
        const volatile float d0 = (float)dist[0];
        const volatile float d1n = (float)dist[1] * nnr; /* one fp32 multiply, then compare */
        if (d0 < d1n) {
            m12[i] = idx[0];
            ++matches;
        }
    }
    return matches;
}

int32_t plo_match_nnr(const uint8_t* q, int32_t nq, const uint8_t* t, int32_t nt, float nnr,
                      int32_t* m12)
{
    return match_nnr_impl(q, nq, t, nt, nnr, m12, 0);
}

static int32_t match_impl(const uint8_t* d1, int32_t n1, const uint8_t* d2, int32_t n2, float nnr,
                          int mutual, int32_t* m12, int keep_prior)
{
    int32_t matches = match_nnr_impl(d1, n1, d2, n2, nnr, m12, keep_prior);
    if (!mutual) return matches;
    int32_t* m21 = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n2 > 0 ? n2 : 1));
    plo_match_nnr(d2, n2, d1, n1, nnr, m21);
    /* the consistency loop runs over EVERY entry >= 0, kept ones included: a kept entry that fails it is cleared and
     * decrements a count it never incremented (the reference's arithmetic; an entry >= n2 would be an out-of-bounds
     * read upstream and is defined here as "fails") */
    for (int32_t i1 = 0; i1 < n1; ++i1) {
        const int32_t i2 = m12[i1];
        if (i2 >= 0 && (i2 >= n2 || m21[i2] != i1)) {
            m12[i1] = -1;
            --matches;
        }
    }
    free(m21);
    return matches;
}

int32_t plo_match(const uint8_t* d1, int32_t n1, const uint8_t* d2, int32_t n2, float nnr,
                  int mutual, int32_t* m12)
{
    return match_impl(d1, n1, d2, n2, nnr, mutual, m12, 0);
}

int32_t plo_match_prior(const uint8_t* d1, int32_t n1, const uint8_t* d2, int32_t n2, float nnr,
                        int mutual, int32_t* m12)
{
    return match_impl(d1, n1, d2, n2, nnr, mutual, m12, 1);
}

void plo_match_batched(const uint8_t* d1, const int32_t* off1, const uint8_t* d2,
                       const int32_t* off2, int32_t B, float nnr, int mutual, int32_t* m12,
                       int32_t* n_matches)
{
    for (int32_t b = 0; b < B; ++b) {
        const int32_t n = plo_match(d1 + (size_t)off1[b] * PLO_DESC_BYTES, off1[b + 1] - off1[b],
                                    d2 + (size_t)off2[b] * PLO_DESC_BYTES, off2[b + 1] - off2[b],
                                    nnr, mutual, m12 + off1[b]);
        if (n_matches) n_matches[b] = n;
    }
}

typedef struct {
    const uint8_t *d1, *d2;
    const int32_t *off1, *off2;
    int32_t b0, b1;
    float nnr;
    int mutual;
    int32_t *m12, *n_matches;
} mt_job;

static void* mt_worker(void* p)
{
    mt_job* j = (mt_job*)p;
    for (int32_t b = j->b0; b < j->b1; ++b) {
        const int32_t n =
            plo_match(j->d1 + (size_t)j->off1[b] * PLO_DESC_BYTES, j->off1[b + 1] - j->off1[b],
                      j->d2 + (size_t)j->off2[b] * PLO_DESC_BYTES, j->off2[b + 1] - j->off2[b],
                      j->nnr, j->mutual, j->m12 + j->off1[b]);
        if (j->n_matches) j->n_matches[b] = n;
    }
    return NULL;
}

void plo_match_batched_mt(const uint8_t* d1, const int32_t* off1, const uint8_t* d2,
                          const int32_t* off2, int32_t B, float nnr, int mutual, int32_t* m12,
                          int32_t* n_matches, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > B) nthreads = B > 0 ? B : 1;
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)nthreads);
    mt_job* jobs = (mt_job*)malloc(sizeof(mt_job) * (size_t)nthreads);
    for (int k = 0; k < nthreads; ++k) {
        jobs[k] = (mt_job){d1, d2, off1, off2, (int32_t)((int64_t)B * k / nthreads),
                           (int32_t)((int64_t)B * (k + 1) / nthreads), nnr, mutual, m12, n_matches};
        pthread_create(&th[k], NULL, mt_worker, &jobs[k]);
    }
    for (int k = 0; k < nthreads; ++k) pthread_join(th[k], NULL);
    free(jobs);
    free(th);
}

/* ------------------------------------------------------------------------------------ */
/* representative descriptor: src/mapFeatures.cpp:51-93                                  */
/* ------------------------------------------------------------------------------------ */
static int cmp_int(const void* a, const void* b)
{
    const int x = *(const int*)a, y = *(const int*)b;
    return (x > y) - (x < y);
}

int32_t plo_median_desc(const uint8_t* descs, int32_t n)
{
    if (n <= 1) return 0; /* ctor path (:28-38): a single observation is its own median */
    int* conf = (int*)malloc(sizeof(int) * (size_t)n * (size_t)n);
    int* row = (int*)malloc(sizeof(int) * (size_t)n);
    for (int i = 0; i < n; ++i) {
        conf[i * n + i] = 0;
        for (int j = i + 1; j < n; ++j) {
            const int d = plo_hamming256(descs + (size_t)i * 32, descs + (size_t)j * 32);
            conf[i * n + j] = d;
            conf[j * n + i] = d;
        }
    }
    int max_dist = 99999, max_idx = 0;
    const int med = (int)(1 + 0.5 * (n - 1));
    for (int i = 0; i < n; ++i) {
        memcpy(row, conf + (size_t)i * n, sizeof(int) * (size_t)n);
        qsort(row, (size_t)n, sizeof(int), cmp_int);
        if (row[med] < max_dist) { /* strict '<': first minimum wins */
            max_dist = row[med];
            max_idx = i;
        }
    }
    free(row);
    free(conf);
    return max_idx;
}

void plo_median_desc_batched(const uint8_t* descs, const int32_t* off, int32_t n_lm, int32_t* med_idx,
                             uint8_t* med_desc)
{
    for (int32_t l = 0; l < n_lm; ++l) {
        const int32_t n = off[l + 1] - off[l];
        const uint8_t* list = descs + (size_t)off[l] * 32;
        med_idx[l] = n > 0 ? plo_median_desc(list, n) : -1;
        if (med_desc) {
            if (n > 0) memcpy(med_desc + (size_t)l * 32, list + (size_t)med_idx[l] * 32, 32); /* :84 */
            else memset(med_desc + (size_t)l * 32, 0, 32);
        }
    }
}

/* ------------------------------------------------------------------------------------ */
/* SE(3) helpers (stvo-pl auxiliar.cpp [RECALL]); 4x4 row-major                          */
/* ------------------------------------------------------------------------------------ */
void plo_inverse_se3(const double T[16], double Ti[16])
{
    double o[16];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) o[4 * i + j] = T[4 * j + i];
    for (int i = 0; i < 3; ++i)
        o[4 * i + 3] = (-T[4 * 0 + i]) * T[3] + (-T[4 * 1 + i]) * T[7] + (-T[4 * 2 + i]) * T[11];
    o[12] = 0.0; o[13] = 0.0; o[14] = 0.0; o[15] = 1.0;
    memcpy(Ti, o, sizeof(o));
}

static void skew3(const double w[3], double s[9])
{
    s[0] = 0.0;   s[1] = -w[2]; s[2] = w[1];
    s[3] = w[2];  s[4] = 0.0;   s[5] = -w[0];
    s[6] = -w[1]; s[7] = w[0];  s[8] = 0.0;
}

static void mat3_mul(const double a[9], const double b[9], double c[9])
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            c[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}

void plo_expmap_se3(const double x[6], double T[16])
{
    const double* t = x;
    const double* w = x + 3;
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    double tt[3] = {t[0], t[1], t[2]};
    const double theta = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    if (!(theta < 0.000001)) {
        double s[9], s2[9], wn[3] = {w[0] / theta, w[1] / theta, w[2] / theta};
        skew3(wn, s);
        mat3_mul(s, s, s2);
        const double sn = sin(theta), cs = cos(theta);
        double V[9];
        for (int k = 0; k < 9; ++k) {
            const double id = (k % 4 == 0) ? 1.0 : 0.0;
            R[k] = id + s[k] * sn + s2[k] * (1.0 - cs);
            V[k] = id + s[k] * (1.0 - cs) / theta + s2[k] * (theta - sn) / theta;
        }
        for (int i = 0; i < 3; ++i)
            tt[i] = V[3 * i] * t[0] + V[3 * i + 1] * t[1] + V[3 * i + 2] * t[2];
    }
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) T[4 * i + j] = R[3 * i + j];
        T[4 * i + 3] = tt[i];
    }
    T[12] = 0.0; T[13] = 0.0; T[14] = 0.0; T[15] = 1.0;
}

static int mat3_inv(const double m[9], double o[9])
{
    const double c0 = m[4] * m[8] - m[5] * m[7];
    const double c1 = m[5] * m[6] - m[3] * m[8];
    const double c2 = m[3] * m[7] - m[4] * m[6];
    const double det = m[0] * c0 + m[1] * c1 + m[2] * c2;
    if (det == 0.0) return -1;
    const double id = 1.0 / det;
    o[0] = c0 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = c1 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = c2 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
    return 0;
}

void plo_logmap_se3(const double T[16], double x[6])
{
    double R[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, w[3] = {0, 0, 0};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R[3 * i + j] = T[4 * i + j];
    double cosine = (R[0] + R[4] + R[8] - 1.0) / 2.0;
    if (cosine > 1.0) cosine = 1.0; else if (cosine < -1.0) cosine = -1.0;
    double sine = sqrt(1.0 - cosine * cosine);
    if (sine > 1.0) sine = 1.0; else if (sine < -1.0) sine = -1.0;
    const double theta = acos(cosine);
    if (theta > 0.000001) {
        /* w_hat = theta * (R - R^T) / (2 sine); w = skewcoords(w_hat) */
        w[0] = theta * (R[7] - R[5]) / (2.0 * sine);
        w[1] = theta * (R[2] - R[6]) / (2.0 * sine);
        w[2] = theta * (R[3] - R[1]) / (2.0 * sine);
        double s[9], s2[9], wn[3] = {w[0] / theta, w[1] / theta, w[2] / theta};
        skew3(wn, s);
        mat3_mul(s, s, s2);
        for (int k = 0; k < 9; ++k) {
            const double id = (k % 4 == 0) ? 1.0 : 0.0;
            V[k] = id + s[k] * (1.0 - cosine) / theta + s2[k] * (theta - sine) / theta;
        }
    }
    double Vi[9];
    if (mat3_inv(V, Vi) != 0) memcpy(Vi, V, sizeof(Vi));
    for (int i = 0; i < 3; ++i)
        x[i] = Vi[3 * i] * T[3] + Vi[3 * i + 1] * T[7] + Vi[3 * i + 2] * T[11];
    x[3] = w[0]; x[4] = w[1]; x[5] = w[2];
}

/* ------------------------------------------------------------------------------------ */
/* LBA rows                                                                              */
/* ------------------------------------------------------------------------------------ */
static inline double dmax(double a, double b) { return a > b ? a : b; } /* std::max(a,b) */

/* stvo-pl PinholeStereoCamera::projection [RECALL]: u = cx + fx*X/Z, v = cy + fy*Y/Z */
static inline void project(const plo_cam* K, const double P[3], double uv[2])
{
    uv[0] = K->cx + K->fx * P[0] / P[2];
    uv[1] = K->cy + K->fy * P[1] / P[2];
}

/* Tiw = inverse_se3(T_kf_w) (:1372); returns R (row-major 3x3) and t */
static inline void inv_pose(const double* T, double R[9], double t[3])
{
    double Ti[16];
    plo_inverse_se3(T, Ti);
    for (int i = 0; i < 3; ++i) {
        R[3 * i] = Ti[4 * i]; R[3 * i + 1] = Ti[4 * i + 1]; R[3 * i + 2] = Ti[4 * i + 2];
        t[i] = Ti[4 * i + 3];
    }
}

static inline void xform(const double R[9], const double t[3], const double X[3], double o[3])
{
    for (int i = 0; i < 3; ++i)
        o[i] = (R[3 * i] * X[0] + R[3 * i + 1] * X[1] + R[3 * i + 2] * X[2]) + t[i];
}

/* the 6-vector of :1392-1397 / :1475-1480 (before any normalisation) */
static inline void jac6(double a, double b, double k, const double G[3], double J[6])
{
    const double gx = G[0], gy = G[1], gz = G[2];
    J[0] = +k * a * gz;
    J[1] = +k * b * gz;
    J[2] = -k * (a * gx + b * gy);
    J[3] = -k * (a * gx * gy + b * gy * gy + b * gz * gz);
    J[4] = +k * (a * gx * gx + a * gz * gz + b * gx * gy);
    J[5] = +k * (b * gx * gz - a * gy * gz);
}

void plo_lba_point_rows(const plo_cam* K, double th, const double* T_kf_w, const double* Xw,
                        const double* obs_uv, const int32_t* lm_loc, const int32_t* kf_slot,
                        int32_t nobs, double* J_pose, double* J_lm, double* r, double* w)
{
    for (int32_t o = 0; o < nobs; ++o) {
        double R[9], t[3], G[3], p[2], Jc[6];
        inv_pose(T_kf_w + 16 * (size_t)kf_slot[o], R, t);        /* :1370-1372 */
        xform(R, t, Xw + 3 * (size_t)lm_loc[o], G);              /* :1373 */
        project(K, G, p);                                        /* :1374 */
        const double dx = obs_uv[2 * o] - p[0];                  /* :1376 */
        const double dy = obs_uv[2 * o + 1] - p[1];
        const double nrm = sqrt(dx * dx + dy * dy);              /* :1377 */
        const double k = 1.0 / dmax(th, G[2] * G[2]);            /* :1382-1383 */
        const double a = K->fx * dx, b = K->fy * dy;             /* :1388-1389 */
        jac6(a, b, k, G, Jc);                                    /* :1392-1397 */
        const double den = dmax(th, nrm);
        for (int c = 0; c < 6; ++c) J_pose[6 * (size_t)o + c] = Jc[c] / den;   /* :1398 */
        for (int j = 0; j < 3; ++j)                                            /* :1404 */
            J_lm[3 * (size_t)o + j] = (Jc[0] * R[j] + Jc[1] * R[3 + j] + Jc[2] * R[6 + j]) / den;
        r[o] = nrm;
        w[o] = 1.0 / (1.0 + nrm * nrm);                          /* robustWeightCauchy :1407 */
    }
}

void plo_lba_line_rows(const plo_cam* K, double th_in, int compat, const double* T_kf_w,
                       const double* Lw, const double* l_obs, const int32_t* lm_loc,
                       const int32_t* kf_slot, int32_t nobs, double* J_pose, double* J_lm,
                       double* r, double* w)
{
    const double th = compat ? 0.0000001 : th_in;               /* :1698 literal */
    for (int32_t o = 0; o < nobs; ++o) {
        double R[9], t[3], P[3], Q[3], p[2], q[2], JP[6], JQ[6];
        const double* Pw = compat ? Lw + 3 * (size_t)lm_loc[o] : Lw + 6 * (size_t)lm_loc[o];
        const double* Qw = compat ? Lw + 3 * (size_t)lm_loc[o] : Lw + 6 * (size_t)lm_loc[o] + 3;
        inv_pose(T_kf_w + 16 * (size_t)kf_slot[o], R, t);        /* :1449-1451 */
        xform(R, t, Pw, P);                                      /* :1452 */
        xform(R, t, Qw, Q);                                      /* :1453 */
        project(K, P, p);
        project(K, Q, q);
        const double* l = l_obs + 3 * (size_t)o;
        const double e0 = l[0] * p[0] + l[1] * p[1] + l[2];      /* :1458 */
        const double e1 = l[0] * q[0] + l[1] * q[1] + l[2];      /* :1459 */
        const double nrm = sqrt(e0 * e0 + e1 * e1);              /* :1460 */
        const double a = K->fx * e0, b = K->fy * e1;             /* :1469-1472 (sic: l_err) */
        const double kP = 1.0 / dmax(th, P[2] * P[2]);
        const double kQ = 1.0 / dmax(th, Q[2] * Q[2]);
        jac6(a, b, kP, P, JP);                                   /* :1475-1480 */
        jac6(a, b, kQ, Q, JQ);                                   /* :1495-1500 */
        const double den = dmax(th, nrm);
        for (int j = 0; j < 3; ++j) {
            const double vp = JP[0] * R[j] + JP[1] * R[3 + j] + JP[2] * R[6 + j];
            const double vq = JQ[0] * R[j] + JQ[1] * R[3 + j] + JQ[2] * R[6 + j];
            J_lm[6 * (size_t)o + j] = vp * e0 / den;             /* :1486 */
            J_lm[6 * (size_t)o + 3 + j] = vq * e1 / den;         /* :1506 */
        }
        for (int c = 0; c < 6; ++c)
            J_pose[6 * (size_t)o + c] = (JP[c] * e0 + JQ[c] * e1) / den;       /* :1509 */
        r[o] = nrm;
        w[o] = 1.0 / (1.0 + nrm * nrm);                          /* :1516 */
    }
}

/* H/g accumulation :1410-1429 (dl = 3) and :1519-1538 (dl = 6) */
static void accumulate(int32_t N, int32_t lm_base, int dl, const int32_t* lm_loc,
                       const int32_t* kf_loc, int32_t nobs, const double* J_pose,
                       const double* J_lm, const double* r, const double* w, double* H,
                       double* g, double* err)
{
    for (int32_t o = 0; o < nobs; ++o) {
        const double* Jp = J_pose + 6 * (size_t)o;
        const double* Jl = J_lm + (size_t)dl * o;
        const int32_t idx = 6 * kf_loc[o];
        const int32_t jdx = lm_base + dl * lm_loc[o];
        const double rr = r[o], ww = w[o];
        for (int a = 0; a < dl; ++a) g[jdx + a] += Jl[a] * rr * ww;
        *err += rr * rr * ww;
        for (int a = 0; a < dl; ++a)
            for (int b = 0; b < dl; ++b) H[(size_t)(jdx + a) * N + jdx + b] += Jl[a] * Jl[b] * ww;
        if (kf_loc[o] == -1) continue;
        for (int a = 0; a < 6; ++a) g[idx + a] += Jp[a] * rr * ww;
        for (int a = 0; a < 6; ++a)
            for (int b = 0; b < 6; ++b) H[(size_t)(idx + a) * N + idx + b] += Jp[a] * Jp[b] * ww;
        for (int a = 0; a < dl; ++a)
            for (int b = 0; b < 6; ++b) {
                const double h = Jl[a] * Jp[b] * ww;            /* Haux */
                H[(size_t)(jdx + a) * N + idx + b] += h;
                H[(size_t)(idx + b) * N + jdx + a] += h;
            }
    }
}

void plo_lba_accumulate_points(int32_t nkf, int32_t npt, int32_t nls, const int32_t* lm_loc,
                               const int32_t* kf_loc, int32_t nobs, const double* J_pose,
                               const double* J_lm, const double* r, const double* w, double* H,
                               double* g, double* err)
{
    const int32_t N = 6 * nkf + 3 * npt + 6 * nls;
    accumulate(N, 6 * nkf, 3, lm_loc, kf_loc, nobs, J_pose, J_lm, r, w, H, g, err);
}

void plo_lba_accumulate_lines(int32_t nkf, int32_t npt, int32_t nls, const int32_t* lm_loc,
                              const int32_t* kf_loc, int32_t nobs, const double* J_pose,
                              const double* J_lm, const double* r, const double* w, double* H,
                              double* g, double* err)
{
    const int32_t N = 6 * nkf + 3 * npt + 6 * nls;
    accumulate(N, 6 * nkf + 3 * npt, 6, lm_loc, kf_loc, nobs, J_pose, J_lm, r, w, H, g, err);
}

/* ------------------------------------------------------------------------------------ */
/* map<->KF gates                                                                        */
/* ------------------------------------------------------------------------------------ */
static inline void xform44(const double T[16], const double X[3], double o[3])
{
    for (int i = 0; i < 3; ++i)
        o[i] = (T[4 * i] * X[0] + T[4 * i + 1] * X[1] + T[4 * i + 2] * X[2]) + T[4 * i + 3];
}

int32_t plo_map2kf_point_gate(const plo_cam* K, const double Twf[16], const double* Xw,
                              const int32_t* m12, int32_t nq, const double* pl, double max_epip,
                              uint8_t* mask)
{
    int32_t n = 0;
    for (int32_t i = 0; i < nq; ++i) {
        mask[i] = 0;
        const int32_t i2 = m12[i];
        if (i2 < 0) continue;                                    /* :603 */
        double Pf[3], pf[2];
        xform44(Twf, Xw + 3 * (size_t)i, Pf);                    /* :605 */
        project(K, Pf, pf);                                      /* :610 */
        const double ex = pf[0] - pl[2 * (size_t)i2], ey = pf[1] - pl[2 * (size_t)i2 + 1];
        if (sqrt(ex * ex + ey * ey) < max_epip) {                /* :612-613 */
            mask[i] = 1;
            ++n;
        }
    }
    return n;
}

int32_t plo_map2kf_line_gate(const plo_cam* K, const double Twf[16], const double* Lw,
                             const int32_t* m12, int32_t nq, const double* le, double max_epip,
                             uint8_t* mask)
{
    int32_t n = 0;
    for (int32_t i = 0; i < nq; ++i) {
        mask[i] = 0;
        const int32_t i2 = m12[i];
        if (i2 < 0) continue;                                    /* :718 */
        double sP[3], eP[3], sp[2], ep[2];
        xform44(Twf, Lw + 6 * (size_t)i, sP);                    /* :720 */
        project(K, sP, sp);
        xform44(Twf, Lw + 6 * (size_t)i + 3, eP);                /* :722 */
        project(K, eP, ep);
        const double* l = le + 3 * (size_t)i2;
        const double e0 = l[0] * sp[0] + l[1] * sp[1] + l[2];    /* :727 */
        const double e1 = l[0] * ep[0] + l[1] * ep[1] + l[2];    /* :728 */
        if (e0 < max_epip && e1 < max_epip) {                    /* :729 signed, no abs() */
            mask[i] = 1;
            ++n;
        }
    }
    return n;
}

static inline int inside(const plo_cam* K, const double P[3])
{
    double p[2];
    project(K, P, p);
    return p[0] > 0 && p[0] < K->width && p[1] > 0 && p[1] < K->height && P[2] > 0.0;
}

void plo_map_point_visible(const plo_cam* K, const double Twf[16], const double* Xw, int32_t n,
                           uint8_t* vis)
{
    for (int32_t i = 0; i < n; ++i) {
        double Pf[3];
        xform44(Twf, Xw + 3 * (size_t)i, Pf);                    /* :549-551 */
        vis[i] = (uint8_t)inside(K, Pf);
    }
}

void plo_map_line_visible(const plo_cam* K, const double Twf[16], const double* Lw, int32_t n,
                          uint8_t* vis)
{
    for (int32_t i = 0; i < n; ++i) {
        double sP[3], eP[3];
        xform44(Twf, Lw + 6 * (size_t)i, sP);                    /* :650-655 */
        xform44(Twf, Lw + 6 * (size_t)i + 3, eP);
        vis[i] = (uint8_t)(inside(K, sP) && inside(K, eP));
    }
}

/* ------------------------------------------------------------------------------------ */
/* map <-> keyframe drivers: src/mapHandler.cpp:532-632 (points), :634-752 (lines)        */
/* ------------------------------------------------------------------------------------ */
/* ------------------------------------------------------------------------------------ */
/* pose-only GN of the loop closure: src/mapHandler.cpp:3324-3424                          */
/* ------------------------------------------------------------------------------------ */
static void gn_jac6(double fgz2, double a, double b, double gx, double gy, double gz, double J[6])
{
    J[0] = +fgz2 * a * gz;
    J[1] = +fgz2 * b * gz;
    J[2] = -fgz2 * (gx * a + gy * b);
    J[3] = -fgz2 * (gx * gy * a + gy * gy * b + gz * gz * b);
    J[4] = +fgz2 * (gx * gx * a + gz * gz * a + gx * gy * b);
    J[5] = +fgz2 * (gx * gz * b - gy * gz * a);
}

void plo_pose_gn_accumulate(const plo_cam* K, double homog_th, const double T_inc[16], const double* P,
                            const double* pl_obs, const uint8_t* pt_inlier, int32_t npt, const double* sPeP,
                            const double* le_obs, const uint8_t* ls_inlier, int32_t nls, double* H, double* g, double* e,
                            int32_t* n_obs)
{
    double Hp[36] = {0}, Hl[36] = {0}, gp[6] = {0}, gl[6] = {0}, ep = 0.0, el = 0.0;
    int32_t Np = 0, Nl = 0;
    for (int32_t i = 0; i < npt; ++i) {
        if (!pt_inlier[i]) continue;
        double G[3], p[2], J[6];
        xform44(T_inc, P + 3 * (size_t)i, G);
        project(K, G, p);
        const double dx = p[0] - pl_obs[2 * (size_t)i], dy = p[1] - pl_obs[2 * (size_t)i + 1];
        const double r = sqrt(dx * dx + dy * dy);
        const double fgz2 = K->fx / dmax(homog_th, G[2] * G[2]);
        gn_jac6(fgz2, dx, dy, G[0], G[1], G[2], J);
        const double den = dmax(homog_th, r);
        for (int k = 0; k < 6; ++k) J[k] = J[k] / den;
        const double w = 1.0 / (1.0 + r * r);                         /* robustWeightCauchy */
        for (int a = 0; a < 6; ++a) {
            for (int b = 0; b < 6; ++b) Hp[6 * a + b] += J[a] * J[b] * w;
            gp[a] += J[a] * r * w;
        }
        ep += r * r * w;
        ++Np;
    }
    for (int32_t i = 0; i < nls; ++i) {
        if (!ls_inlier[i]) continue;
        double S[3], E[3], ps[2], pe[2], Js[6], Je[6], J[6];
        xform44(T_inc, sPeP + 6 * (size_t)i, S);
        project(K, S, ps);
        xform44(T_inc, sPeP + 6 * (size_t)i + 3, E);
        project(K, E, pe);
        const double lx = le_obs[3 * (size_t)i], ly = le_obs[3 * (size_t)i + 1], lz = le_obs[3 * (size_t)i + 2];
        const double ds = lx * ps[0] + ly * ps[1] + lz, de = lx * pe[0] + ly * pe[1] + lz;
        const double r = sqrt(ds * ds + de * de);
        gn_jac6(K->fx / dmax(homog_th, S[2] * S[2]), lx, ly, S[0], S[1], S[2], Js);
        gn_jac6(K->fx / dmax(homog_th, E[2] * E[2]), lx, ly, E[0], E[1], E[2], Je);
        const double den = dmax(homog_th, r);
        for (int k = 0; k < 6; ++k) J[k] = (Js[k] * ds + Je[k] * de) / den;
        const double w = 1.0 / (1.0 + r * r);
        for (int a = 0; a < 6; ++a) {
            for (int b = 0; b < 6; ++b) Hl[6 * a + b] += J[a] * J[b] * w;
            gl[a] += J[a] * r * w;
        }
        el += r * r * w;
        ++Nl;
    }
    for (int k = 0; k < 36; ++k) H[k] = Hp[k] + Hl[k];
    for (int k = 0; k < 6; ++k) g[k] = gp[k] + gl[k];
    *e = ep + el;
    if (n_obs) { n_obs[0] = Np; n_obs[1] = Nl; }
}

/* double -> int as the reference's x86 build does it (cvttsd2si): truncation toward zero; NaN and values outside
 * int32 give INT_MIN ("integer indefinite") */
static inline int32_t cvtt_x86(double v)
{
    return (v > -2147483649.0 && v < 2147483648.0) ? (int32_t)v : INT32_MIN;
}

/* grid of keyframe features for matchGrid: points -> their cell; lines -> Bresenham cells + unit directions.
 * feat: per item 2 (points: pl) or 4 (lines: spl, epl) doubles, item b = feat[sel ? sel[b] : b].
 * cs: cols*rows+1; returns malloc'ed items (caller frees); dir2: n x 2 or NULL */
static int32_t* fill_feature_grid(int lines, const double* feat, const int32_t* sel, int32_t n, int32_t cols,
                                  int32_t rows, double inv_w, double inv_h, int32_t* cs, double* dir2)
{
    int32_t* items;
    if (!lines) {
        int32_t* xy = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)(n > 0 ? n : 1));
        for (int32_t b = 0; b < n; ++b) {
            const double* p = feat + 2 * (size_t)(sel ? sel[b] : b);
            xy[2 * b] = cvtt_x86(p[0] * inv_w);
            xy[2 * b + 1] = cvtt_x86(p[1] * inv_h);
        }
        items = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
        plo_grid_fill_points(xy, n, cols, rows, cs, items);
        free(xy);
        return items;
    }
    int32_t** cell_of = (int32_t**)malloc(sizeof(int32_t*) * (size_t)(n > 0 ? n : 1));
    int32_t* ncell_of = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    size_t total = 0;
    for (size_t c = 0; c <= (size_t)cols * rows; ++c) cs[c] = 0;
    for (int32_t b = 0; b < n; ++b) {
        const double* sg = feat + 4 * (size_t)(sel ? sel[b] : b);
        double v[2] = {(sg[2] - sg[0]) * inv_w, (sg[3] - sg[1]) * inv_h};
        plo_normalize2(v);
        dir2[2 * b] = v[0];
        dir2[2 * b + 1] = v[1];
        const double x1 = sg[0] * inv_w, y1 = sg[1] * inv_h, x2 = sg[2] * inv_w, y2 = sg[3] * inv_h;
        const int32_t m = plo_get_line_coords(x1, y1, x2, y2, NULL, 0);
        cell_of[b] = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)(m > 0 ? m : 1));
        plo_get_line_coords(x1, y1, x2, y2, cell_of[b], m);
        ncell_of[b] = m;
        for (int32_t k = 0; k < m; ++k) {
            const int32_t x = cell_of[b][2 * k], y = cell_of[b][2 * k + 1];
            if (x >= 0 && x < cols && y >= 0 && y < rows) { ++cs[(size_t)x * rows + y + 1]; ++total; }
        }
    }
    for (size_t c = 0; c < (size_t)cols * rows; ++c) cs[c + 1] += cs[c];
    items = (int32_t*)malloc(sizeof(int32_t) * (total > 0 ? total : 1));
    int32_t* fill = (int32_t*)malloc(sizeof(int32_t) * (size_t)cols * rows);
    for (size_t c = 0; c < (size_t)cols * rows; ++c) fill[c] = cs[c];
    for (int32_t b = 0; b < n; ++b) {                                     /* push_back order: idx ascending */
        for (int32_t k = 0; k < ncell_of[b]; ++k) {
            const int32_t x = cell_of[b][2 * k], y = cell_of[b][2 * k + 1];
            if (x >= 0 && x < cols && y >= 0 && y < rows) items[fill[(size_t)x * rows + y]++] = b;
        }
        free(cell_of[b]);
    }
    free(fill); free(ncell_of); free(cell_of);
    return items;
}

/* matchGrid over projected 3D features: cen = cells of proj(T * X) * (sx, sy), dir1 for lines */
static int32_t grid_match_projected(int lines, const plo_cam* K, const double T[16], const double* X3, int32_t nq,
                                    double sx, double sy, const uint8_t* Q, const int32_t* cs, const int32_t* items,
                                    int32_t cols, int32_t rows, const uint8_t* Tdesc, int32_t nt, const double* dir2,
                                    const plo_fast_matching* fm, int mutual, int32_t* m12)
{
    const int nc = lines ? 2 : 1;
    const int32_t w[4] = {fm->ws, fm->ws, fm->ws, fm->ws};
    int32_t* cen = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)nc * (size_t)(nq > 0 ? nq : 1));
    double* dir1 = lines ? (double*)malloc(sizeof(double) * 2 * (size_t)(nq > 0 ? nq : 1)) : NULL;
    for (int32_t a = 0; a < nq; ++a) {
        for (int c = 0; c < nc; ++c) {
            double Pf[3], pf[2];
            xform44(T, X3 + (size_t)a * 3 * nc + 3 * c, Pf);
            project(K, Pf, pf);
            cen[((size_t)a * nc + c) * 2] = cvtt_x86(pf[0] * sx);
            cen[((size_t)a * nc + c) * 2 + 1] = cvtt_x86(pf[1] * sy);
        }
        if (lines) {   /* matchGrid derives the query direction from the integer end points */
            double v[2] = {(double)cen[4 * (size_t)a + 2] - (double)cen[4 * (size_t)a],
                           (double)cen[4 * (size_t)a + 3] - (double)cen[4 * (size_t)a + 1]};
            plo_normalize2(v);
            dir1[2 * a] = v[0];
            dir1[2 * a + 1] = v[1];
        }
    }
    const int32_t n = plo_match_grid(cen, nc, Q, nq, cs, items, cols, rows, Tdesc, nt, dir1, dir2, fm->line_sim_th, w,
                                     fm->nnr_grid, mutual, m12);
    free(dir1); free(cen);
    return n;
}

static int32_t kf2kf_driver(int lines, const plo_cam* K, const double DT[16], const double* X_prev,
                            const uint8_t* desc_prev, int32_t n_prev, const double* feat_curr, const uint8_t* desc_curr,
                            int32_t n_curr, float nnr, int mutual, int32_t min_matches, const plo_fast_matching* fm,
                            int32_t* m12, int32_t* used_match)
{
    for (int32_t i = 0; i < n_prev; ++i) m12[i] = -1;
    if (used_match) *used_match = 0;
    if (n_prev <= 0 || n_curr <= 0) return 0;                                         /* :243 / :368 */
    int32_t matches = 0;
    if (fm && fm->enabled) {
        int32_t* cs = (int32_t*)malloc(sizeof(int32_t) * ((size_t)fm->grid_cols * fm->grid_rows + 1));
        double* dir2 = lines ? (double*)malloc(sizeof(double) * 2 * (size_t)n_curr) : NULL;
        int32_t* items = fill_feature_grid(lines, feat_curr, NULL, n_curr, fm->grid_cols, fm->grid_rows, fm->inv_width,
                                           fm->inv_height, cs, dir2);
        /* points: pixels * inv (:256); lines: the projected pixels themselves (:392-393) */
        matches = grid_match_projected(lines, K, DT, X_prev, n_prev, lines ? 1.0 : fm->inv_width,
                                       lines ? 1.0 : fm->inv_height, desc_prev, cs, items, fm->grid_cols, fm->grid_rows,
                                       desc_curr, n_curr, dir2, fm, mutual, m12);
        free(items); free(dir2); free(cs);
    }
    if (n_curr > min_matches && n_prev > min_matches && matches < min_matches) {    /* :274-278 / :421-425 */
        /* the vector matchGrid filled is handed on: its entries survive where the ratio test rejects (plo_match_prior) */
        matches = (fm && fm->enabled) ? plo_match_prior(desc_prev, n_prev, desc_curr, n_curr, nnr, mutual, m12)
                                      : plo_match(desc_prev, n_prev, desc_curr, n_curr, nnr, mutual, m12);
        if (used_match) *used_match = 1;
    }
    return matches;
}

int32_t plo_kf2kf_match_points(const plo_cam* K, const double DT[16], const double* P_prev, const uint8_t* desc_prev,
                               int32_t n_prev, const double* pl_curr, const uint8_t* desc_curr, int32_t n_curr,
                               float nnr, int mutual, int32_t min_matches, const plo_fast_matching* fm, int32_t* m12,
                               int32_t* used_match)
{
    return kf2kf_driver(0, K, DT, P_prev, desc_prev, n_prev, pl_curr, desc_curr, n_curr, nnr, mutual, min_matches, fm,
                        m12, used_match);
}

int32_t plo_kf2kf_match_lines(const plo_cam* K, const double DT[16], const double* sPeP_prev,
                              const uint8_t* desc_prev, int32_t n_prev, const double* seg_curr,
                              const uint8_t* desc_curr, int32_t n_curr, float nnr, int mutual, int32_t min_matches,
                              const plo_fast_matching* fm, int32_t* m12, int32_t* used_match)
{
    return kf2kf_driver(1, K, DT, sPeP_prev, desc_prev, n_prev, seg_curr, desc_curr, n_curr, nnr, mutual, min_matches,
                        fm, m12, used_match);
}

static int32_t map2kf_driver(int lines, const plo_cam* K, const double Twf[16], const double* LM,
                             const uint8_t* med_desc, const uint8_t* candidate, int32_t n_map,
                             const uint8_t* kf_desc, const double* kf_feat, const double* kf_seg,
                             const int32_t* kf_idx, int32_t n_kf, float nnr, int mutual, double max_epip,
                             int32_t min_matches, const plo_fast_matching* fm, int32_t* map_to_kf,
                             int32_t* used_match)
{
    const int lw = lines ? 6 : 3, fw = lines ? 3 : 2;
    for (int32_t i = 0; i < n_map; ++i) map_to_kf[i] = -1;
    if (used_match) *used_match = 0;
    uint8_t* vis = (uint8_t*)malloc((size_t)(n_map > 0 ? n_map : 1));
    if (lines) plo_map_line_visible(K, Twf, LM, n_map, vis); else plo_map_point_visible(K, Twf, LM, n_map, vis);
    int32_t* qi = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n_map > 0 ? n_map : 1));
    int32_t* ti = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n_kf > 0 ? n_kf : 1));
    int32_t nq = 0, nt = 0;
    for (int32_t i = 0; i < n_map; ++i) if (candidate[i] && vis[i]) qi[nq++] = i;      /* :545-558 / :647-663 */
    for (int32_t i = 0; i < n_kf; ++i) if (kf_idx[i] == -1) ti[nt++] = i;              /* :563-569 / :668-674 */
    int32_t matches = 0;
    if (nq > 0 && nt > 0) {                                                             /* :571 / :676 */
        uint8_t* Q = (uint8_t*)malloc((size_t)nq * 32);
        uint8_t* T = (uint8_t*)malloc((size_t)nt * 32);
        double* QL = (double*)malloc(sizeof(double) * (size_t)nq * lw);
        double* TF = (double*)malloc(sizeof(double) * (size_t)nt * fw);
        for (int32_t a = 0; a < nq; ++a) {
            memcpy(Q + (size_t)a * 32, med_desc + (size_t)qi[a] * 32, 32);
            memcpy(QL + (size_t)a * lw, LM + (size_t)qi[a] * lw, sizeof(double) * lw);
        }
        for (int32_t b = 0; b < nt; ++b) {
            memcpy(T + (size_t)b * 32, kf_desc + (size_t)ti[b] * 32, 32);
            memcpy(TF + (size_t)b * fw, kf_feat + (size_t)ti[b] * fw, sizeof(double) * fw);
        }
        int32_t* m12 = (int32_t*)malloc(sizeof(int32_t) * (size_t)nq);
        int32_t n_m12 = 0;                         /* matches_12.size(): 0 until a matcher ran */
        if (fm && fm->enabled) {                   /* :578-592 / :681-707 */
            int32_t* cs = (int32_t*)malloc(sizeof(int32_t) * ((size_t)fm->grid_cols * fm->grid_rows + 1));
            double* dir2 = lines ? (double*)malloc(sizeof(double) * 2 * (size_t)nt) : NULL;
            /* the grid of the unmatched keyframe features: points :581-584, lines :686-698 (kf_seg = spl, epl) */
            int32_t* items = fill_feature_grid(lines, lines ? kf_seg : kf_feat, ti, nt, fm->grid_cols, fm->grid_rows,
                                               fm->inv_width, fm->inv_height, cs, dir2);
            /* pj_points / pj_lines: projections * inv_width / inv_height, make_pair<int,int> (:555, :659) */
            matches = grid_match_projected(lines, K, Twf, QL, nq, fm->inv_width, fm->inv_height, Q, cs, items,
                                           fm->grid_cols, fm->grid_rows, T, nt, dir2, fm, mutual, m12);
            n_m12 = nq;
            free(items); free(dir2); free(cs);
        }
        if (nq > min_matches && matches < min_matches) {                               /* :594-598 / :709-713 */
            matches = n_m12 ? plo_match_prior(Q, nq, T, nt, nnr, mutual, m12)     /* matchGrid's entries survive */
                            : plo_match(Q, nq, T, nt, nnr, mutual, m12);
            n_m12 = nq;
            if (used_match) *used_match = 1;
        }
        if (n_m12) {
            uint8_t* mask = (uint8_t*)malloc((size_t)nq);
            matches = lines ? plo_map2kf_line_gate(K, Twf, QL, m12, nq, TF, max_epip, mask)
                            : plo_map2kf_point_gate(K, Twf, QL, m12, nq, TF, max_epip, mask);
            for (int32_t a = 0; a < nq; ++a)
                if (mask[a]) map_to_kf[qi[a]] = ti[m12[a]];
            free(mask);
        }
        free(m12); free(TF); free(QL); free(T); free(Q);
    }
    free(ti); free(qi); free(vis);
    return matches;
}

int32_t plo_map2kf_match_points(const plo_cam* K, const double Twf[16], const double* Xw,
                                const uint8_t* med_desc, const uint8_t* candidate, int32_t n_map,
                                const uint8_t* kf_desc, const double* kf_pl, const int32_t* kf_idx,
                                int32_t n_kf, float nnr, int mutual, double max_epip,
                                int32_t min_matches, int32_t* map_to_kf)
{
    return map2kf_driver(0, K, Twf, Xw, med_desc, candidate, n_map, kf_desc, kf_pl, NULL, kf_idx, n_kf, nnr,
                         mutual, max_epip, min_matches, NULL, map_to_kf, NULL);
}

int32_t plo_map2kf_match_lines(const plo_cam* K, const double Twf[16], const double* Lw,
                               const uint8_t* med_desc, const uint8_t* candidate, int32_t n_map,
                               const uint8_t* kf_desc, const double* kf_le, const int32_t* kf_idx,
                               int32_t n_kf, float nnr, int mutual, double max_epip,
                               int32_t min_matches, int32_t* map_to_kf)
{
    return map2kf_driver(1, K, Twf, Lw, med_desc, candidate, n_map, kf_desc, kf_le, NULL, kf_idx, n_kf, nnr,
                         mutual, max_epip, min_matches, NULL, map_to_kf, NULL);
}

int32_t plo_map2kf_match_points_fast(const plo_cam* K, const double Twf[16], const double* Xw,
                                     const uint8_t* med_desc, const uint8_t* candidate, int32_t n_map,
                                     const uint8_t* kf_desc, const double* kf_pl, const int32_t* kf_idx,
                                     int32_t n_kf, float nnr, int mutual, double max_epip, int32_t min_matches,
                                     const plo_fast_matching* fm, int32_t* map_to_kf, int32_t* used_match)
{
    return map2kf_driver(0, K, Twf, Xw, med_desc, candidate, n_map, kf_desc, kf_pl, NULL, kf_idx, n_kf, nnr,
                         mutual, max_epip, min_matches, fm, map_to_kf, used_match);
}

int32_t plo_map2kf_match_lines_fast(const plo_cam* K, const double Twf[16], const double* Lw,
                                    const uint8_t* med_desc, const uint8_t* candidate, int32_t n_map,
                                    const uint8_t* kf_desc, const double* kf_le, const double* kf_seg,
                                    const int32_t* kf_idx, int32_t n_kf, float nnr, int mutual, double max_epip,
                                    int32_t min_matches, const plo_fast_matching* fm, int32_t* map_to_kf,
                                    int32_t* used_match)
{
    return map2kf_driver(1, K, Twf, Lw, med_desc, candidate, n_map, kf_desc, kf_le, kf_seg, kf_idx, n_kf, nnr,
                         mutual, max_epip, min_matches, fm, map_to_kf, used_match);
}

/* ------------------------------------------------------------------------------------ */
/* stereo gates of StVO::StereoFrame (stvo-pl stereoFrame.cpp) -- [RECALL]                */
/* ------------------------------------------------------------------------------------ */
static inline double dmin2(double a, double b) { return b < a ? b : a; }   /* std::min */
static inline double dmax2(double a, double b) { return a < b ? b : a; }   /* std::max */

double plo_line_segment_overlap_stereo(double spl_obs, double epl_obs, double spl_proj, double epl_proj,
                                       double line_horiz_th)
{
    double overlap = 1.f;
    if (fabs(epl_obs - spl_obs) > line_horiz_th) {  /* normal lines (verticals included) */
        const double sln = dmin2(spl_obs, epl_obs);
        const double eln = dmax2(spl_obs, epl_obs);
        const double spn = dmin2(spl_proj, epl_proj);
        const double epn = dmax2(spl_proj, epl_proj);
        const double length = eln - spn;
        if ((epn < sln) || (spn > eln))
            overlap = 0.f;
        else {
            if ((epn > eln) && (spn < sln))
                overlap = eln - sln;
            else
                overlap = dmin2(eln, epn) - dmax2(sln, spn);
        }
        if (length > 0.01f)
            overlap = overlap / length;
        else
            overlap = 0.f;
        if (overlap > 1.f) overlap = 1.f;
    }
    return overlap;
}

int32_t plo_stereo_point_gate(const int32_t* m12, int32_t n_l, const float* kp_l, const float* kp_r, int32_t n_r,
                              double max_dist_epip, double min_disp, int32_t* stereo_12, double* disp)
{
    int32_t n = 0;
    for (int32_t i1 = 0; i1 < n_l; ++i1) {
        stereo_12[i1] = -1;
        disp[i1] = 0.0;
        const int32_t i2 = m12[i1];
        if (i2 < 0 || i2 >= n_r) continue;
        const float dy = kp_l[2 * (size_t)i1 + 1] - kp_r[2 * (size_t)i2 + 1];
        if ((double)fabsf(dy) <= max_dist_epip) {                  /* check stereo epipolar constraint */
            const float dx = kp_l[2 * (size_t)i1] - kp_r[2 * (size_t)i2];
            const double disp_ = (double)dx;
            if (disp_ >= min_disp) {                               /* check minimal disparity */
                stereo_12[i1] = i2;
                disp[i1] = disp_;
                ++n;
            }
        }
    }
    return n;
}

int32_t plo_stereo_line_gate(const int32_t* m12, int32_t n_l, const float* seg_l, const float* seg_r, int32_t n_r,
                             double min_disp, double line_horiz_th, double stereo_overlap_th,
                             double ls_min_disp_ratio, int32_t* stereo_12, double* disp_se)
{
    int32_t n = 0;
    for (int32_t i1 = 0; i1 < n_l; ++i1) {
        stereo_12[i1] = -1;
        disp_se[2 * (size_t)i1] = disp_se[2 * (size_t)i1 + 1] = 0.0;
        const int32_t i2 = m12[i1];
        if (i2 < 0 || i2 >= n_r) continue;
        const double sp_l[2] = {seg_l[4 * (size_t)i1], seg_l[4 * (size_t)i1 + 1]};
        const double ep_l[2] = {seg_l[4 * (size_t)i1 + 2], seg_l[4 * (size_t)i1 + 3]};
        double sp_r[2] = {seg_r[4 * (size_t)i2], seg_r[4 * (size_t)i2 + 1]};
        double ep_r[2] = {seg_r[4 * (size_t)i2 + 2], seg_r[4 * (size_t)i2 + 3]};
        const double overlap = plo_line_segment_overlap_stereo(sp_l[1], ep_l[1], sp_r[1], ep_r[1], line_horiz_th);
        /* estimate the disparity of the endpoints: the right end points slide along the right line */
        const double sx = (sp_r[0] * (sp_l[1] - ep_r[1]) + ep_r[0] * (sp_r[1] - sp_l[1])) / (sp_r[1] - ep_r[1]);
        sp_r[0] = sx;
        sp_r[1] = sp_l[1];
        const double ex = (sp_r[0] * (ep_l[1] - ep_r[1]) + ep_r[0] * (sp_r[1] - ep_l[1])) / (sp_r[1] - ep_r[1]);
        ep_r[0] = ex;
        ep_r[1] = ep_l[1];
        /* filterLineSegmentDisparity */
        double disp_s = sp_l[0] - sp_r[0];
        double disp_e = ep_l[0] - ep_r[0];
        if (dmin2(disp_s, disp_e) / dmax2(disp_s, disp_e) < ls_min_disp_ratio) {
            disp_s = -1.0;
            disp_e = -1.0;
        }
        if (disp_s >= min_disp && disp_e >= min_disp && fabs(sp_l[1] - ep_l[1]) > line_horiz_th &&
            fabs(sp_r[1] - ep_r[1]) > line_horiz_th && overlap > stereo_overlap_th) {
            stereo_12[i1] = i2;
            disp_se[2 * (size_t)i1] = disp_s;
            disp_se[2 * (size_t)i1 + 1] = disp_e;
            ++n;
        }
    }
    return n;
}

/* ------------------------------------------------------------------------------------ */
/* stvo-pl matchGrid (matching.cpp) + GridStructure (gridStructure.cpp) -- [RECALL]       */
/* call sites: src/mapHandler.cpp:271, :418, :591, :706                                   */
/* ------------------------------------------------------------------------------------ */
static void grid_get(int32_t x, int32_t y, const int32_t w[4], const int32_t* cell_start,
                     const int32_t* cell_items, int32_t cols, int32_t rows, int32_t** cand, size_t* n,
                     size_t* cap)
{
    /* GridStructure::get(x, y, w, indices) */
    const int64_t min_x = (int64_t)x - w[0] > 0 ? (int64_t)x - w[0] : 0;
    const int64_t max_x = (int64_t)x + w[1] + 1 < cols ? (int64_t)x + w[1] + 1 : cols;
    const int64_t min_y = (int64_t)y - w[2] > 0 ? (int64_t)y - w[2] : 0;
    const int64_t max_y = (int64_t)y + w[3] + 1 < rows ? (int64_t)y + w[3] + 1 : rows;
    for (int64_t x_ = min_x; x_ < max_x; ++x_)
        for (int64_t y_ = min_y; y_ < max_y; ++y_) {
            const int64_t id = x_ * rows + y_;
            for (int32_t k = cell_start[id]; k < cell_start[id + 1]; ++k) {
                if (*n == *cap) {
                    *cap = *cap ? *cap * 2 : 64;
                    *cand = (int32_t*)realloc(*cand, *cap * sizeof(int32_t));
                }
                (*cand)[(*n)++] = cell_items[k];
            }
        }
}

int32_t plo_match_grid(const int32_t* centres, int32_t n_centres, const uint8_t* d1, int32_t n1,
                       const int32_t* cell_start, const int32_t* cell_items, int32_t cols, int32_t rows,
                       const uint8_t* d2, int32_t n2, const double* dir1, const double* dir2,
                       double sim_th, const int32_t w[4], double nnr, int mutual, int32_t* m12)
{
    int32_t matches = 0;
    for (int32_t i = 0; i < n1; ++i) m12[i] = -1;                 /* matches_12.resize(desc1.rows, -1) */
    int32_t* m21 = NULL;
    int* distances = NULL;
    if (mutual) {
        m21 = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n2 > 0 ? n2 : 1));
        distances = (int*)malloc(sizeof(int) * (size_t)(n2 > 0 ? n2 : 1));
        for (int32_t j = 0; j < n2; ++j) { m21[j] = -1; distances[j] = INT_MAX; }
    }
    int32_t* cand = NULL;
    size_t cap = 0;
    for (int32_t i1 = 0; i1 < n1; ++i1) {
        int best_d = INT_MAX, best_d2 = INT_MAX;
        int32_t best_idx = -1;
        size_t nc = 0;
        for (int32_t c = 0; c < n_centres; ++c) {
            const int32_t* p = centres + ((size_t)i1 * (size_t)n_centres + (size_t)c) * 2;
            grid_get(p[0], p[1], w, cell_start, cell_items, cols, rows, &cand, &nc, &cap);
        }
        if (nc == 0) continue;                                      /* candidates.empty() */
        /* the unordered_set: unique values; visiting order defined here as ascending */
        qsort(cand, nc, sizeof(int32_t), cmp_int);
        size_t nu = 0;
        for (size_t k = 0; k < nc; ++k)
            if (nu == 0 || cand[nu - 1] != cand[k]) cand[nu++] = cand[k];
        for (size_t k = 0; k < nu; ++k) {
            const int32_t i2 = cand[k];
            if (i2 < 0 || i2 >= n2) continue;
            if (dir1 && dir2) {
                const double dot = dir1[2 * i1] * dir2[2 * i2] + dir1[2 * i1 + 1] * dir2[2 * i2 + 1];
                if (fabs(dot) < sim_th) continue;
            }
            const int d = plo_hamming256(d1 + (size_t)i1 * PLO_DESC_BYTES, d2 + (size_t)i2 * PLO_DESC_BYTES);
            if (mutual) {
                if (d < distances[i2]) {
                    distances[i2] = d;
                    m21[i2] = i1;
                } else
                    continue;
            }
            if (d < best_d) {
                best_d2 = best_d;
                best_d = d;
                best_idx = i2;
            } else if (d < best_d2)
                best_d2 = d;
        }
        if ((double)best_d < (double)best_d2 * nnr) {
            m12[i1] = best_idx;
            ++matches;
        }
    }
    if (mutual) {
        for (int32_t i1 = 0; i1 < n1; ++i1) {
            const int32_t i2 = m12[i1];
            if (i2 >= 0 && m21[i2] != i1) {
                m12[i1] = -1;
                --matches;
            }
        }
    }
    free(cand); free(distances); free(m21);
    return matches;
}

void plo_grid_fill_points(const int32_t* xy, int32_t n, int32_t cols, int32_t rows, int32_t* cell_start,
                          int32_t* cell_items)
{
    const int64_t ncell = (int64_t)cols * rows;
    for (int64_t c = 0; c <= ncell; ++c) cell_start[c] = 0;
    for (int32_t i = 0; i < n; ++i) {
        const int32_t x = xy[2 * i], y = xy[2 * i + 1];
        if (x >= 0 && x < cols && y >= 0 && y < rows) ++cell_start[(int64_t)x * rows + y + 1];
    }
    for (int64_t c = 0; c < ncell; ++c) cell_start[c + 1] += cell_start[c];
    int32_t* fill = (int32_t*)malloc(sizeof(int32_t) * (size_t)(ncell > 0 ? ncell : 1));
    for (int64_t c = 0; c < ncell; ++c) fill[c] = cell_start[c];
    for (int32_t i = 0; i < n; ++i) {                               /* push_back order = ascending idx */
        const int32_t x = xy[2 * i], y = xy[2 * i + 1];
        if (x >= 0 && x < cols && y >= 0 && y < rows) cell_items[fill[(int64_t)x * rows + y]++] = i;
    }
    free(fill);
}

int32_t plo_get_line_coords(double x1, double y1, double x2, double y2, int32_t* out_xy, int32_t cap)
{
    /* Bresenham as in stvo-pl gridStructure.cpp::getLineCoords */
    const int steep = fabs(y2 - y1) > fabs(x2 - x1);
    double t;
    if (steep) { t = x1; x1 = y1; y1 = t; t = x2; x2 = y2; y2 = t; }
    if (x1 > x2) { t = x1; x1 = x2; x2 = t; t = y1; y1 = y2; y2 = t; }
    const double dx = x2 - x1;
    const double dy = fabs(y2 - y1);
    double error = dx / 2.0;
    const int ystep = (y1 < y2) ? 1 : -1;
    int y = (int)y1;
    const int maxX = (int)x2;
    int32_t n = 0;
    for (int x = (int)x1; x < maxX; x++) {
        if (n < cap) {
            out_xy[2 * n] = steep ? y : x;
            out_xy[2 * n + 1] = steep ? x : y;
        }
        ++n;
        error -= dy;
        if (error < 0) {
            y += ystep;
            error += dx;
        }
    }
    return n;
}

void plo_normalize2(double v[2])
{
    const double magnitude = sqrt(v[0] * v[0] + v[1] * v[1]);
    v[0] /= magnitude;
    v[1] /= magnitude;
}

/* ---- LBD float descriptor: binary_descriptor_custom.cpp:1026-1372 ------------------------------------- */
#define PLO_NUM_OF_BANDS 9
void plo_lbd_gauss_tables(int32_t w, double* coef_l, double* coef_g)
{
    /* :146-176 (the constructor), integer divisions as written */
    double u = (w * 3 - 1) / 2;
    double sigma = (w * 2 + 1) / 2;
    double invsigma2 = -1 / (2 * sigma * sigma);
    for (int i = 0; i < w * 3; i++) {
        const double dis = i - u;
        coef_l[i] = exp(dis * dis * invsigma2);
    }
    u = (PLO_NUM_OF_BANDS * w - 1) / 2;
    sigma = u;
    invsigma2 = -1 / (2 * sigma * sigma);
    for (int i = 0; i < PLO_NUM_OF_BANDS * w; i++) {
        const double dis = i - u;
        coef_g[i] = exp(dis * dis * invsigma2);
    }
}

void plo_lbd_compute(const int16_t* pdxImg, const int16_t* pdyImg, int32_t width, int32_t height, const plo_lbd_line* lines,
                     int32_t n, int32_t widthOfBand, float* lbd)
{
    double* gaussCoefL = (double*)malloc(sizeof(double) * 3 * (size_t)widthOfBand);
    double* gaussCoefG = (double*)malloc(sizeof(double) * PLO_NUM_OF_BANDS * (size_t)widthOfBand);
    plo_lbd_gauss_tables(widthOfBand, gaussCoefL, gaussCoefG);
    const short heightOfLSP = (short)(widthOfBand * PLO_NUM_OF_BANDS);
    const short descriptor_size = PLO_NUM_OF_BANDS * 8;
    const short halfHeight = (heightOfLSP - 1) / 2;
    const short realWidth = (short)width;
    const short imageWidth = realWidth - 1;
    const short imageHeight = (short)(height - 1);
    for (int32_t li = 0; li < n; ++li) {
        const plo_lbd_line* L = &lines[li];
        float pgdLBandSum[PLO_NUM_OF_BANDS] = {0}, ngdLBandSum[PLO_NUM_OF_BANDS] = {0}, pgdL2BandSum[PLO_NUM_OF_BANDS] = {0},
              ngdL2BandSum[PLO_NUM_OF_BANDS] = {0}, pgdOBandSum[PLO_NUM_OF_BANDS] = {0}, ngdOBandSum[PLO_NUM_OF_BANDS] = {0},
              pgdO2BandSum[PLO_NUM_OF_BANDS] = {0}, ngdO2BandSum[PLO_NUM_OF_BANDS] = {0};
        const short lengthOfLSP = (short)L->num_pixels;
        const short halfWidth = (lengthOfLSP - 1) / 2;
        const float lineMiddlePointX = (float)(0.5 * (L->sx + L->ex));
        const float lineMiddlePointY = (float)(0.5 * (L->sy + L->ey));
        float dL[2], dO[2];
        /* `dL[0] = cos( pSingleLine->direction )` with a float argument: under libstdc++'s <math.h> (GCC >= 6;
         * bitarray_custom.hpp:52 includes it) the unqualified call binds to the float overload = cosf, which is what
         * the reference compiled in this image does (oracle/_ref, test_lbd_compute_pinned_to_reference_code).  A
         * toolchain whose <math.h> is the plain C header evaluates in double and rounds: ~2 % of directions then
         * differ by one ulp, ~2e-7 in the descriptor. */
        dL[0] = cosf(L->direction);
        dL[1] = sinf(L->direction);
        dO[0] = -dL[1];
        dO[1] = dL[0];
        float sCorX0 = -dL[0] * halfWidth + dL[1] * halfHeight + lineMiddlePointX;
        float sCorY0 = -dL[1] * halfWidth - dL[0] * halfHeight + lineMiddlePointY;
        for (short hID = 0; hID < heightOfLSP; hID++) {
            float sCorX = sCorX0, sCorY = sCorY0;
            float pgdLRowSum = 0, ngdLRowSum = 0, pgdORowSum = 0, ngdORowSum = 0;
            for (short wID = 0; wID < lengthOfLSP; wID++) {
                short tempCor = (short)round(sCorX);
                const short xCor = (tempCor < 0) ? 0 : (tempCor > imageWidth) ? imageWidth : tempCor;
                tempCor = (short)round(sCorY);
                const short yCor = (tempCor < 0) ? 0 : (tempCor > imageHeight) ? imageHeight : tempCor;
                const short dx = pdxImg[yCor * realWidth + xCor];
                const short dy = pdyImg[yCor * realWidth + xCor];
                const float gDL = dx * dL[0] + dy * dL[1];
                const float gDO = dx * dO[0] + dy * dO[1];
                if (gDL > 0) pgdLRowSum += gDL; else ngdLRowSum -= gDL;
                if (gDO > 0) pgdORowSum += gDO; else ngdORowSum -= gDO;
                sCorX += dL[0];
                sCorY += dL[1];
            }
            sCorX0 -= dL[1];
            sCorY0 += dL[0];
            float coefInGaussion = (float)gaussCoefG[hID];
            pgdLRowSum = coefInGaussion * pgdLRowSum;
            ngdLRowSum = coefInGaussion * ngdLRowSum;
            const float pgdL2RowSum = pgdLRowSum * pgdLRowSum;
            const float ngdL2RowSum = ngdLRowSum * ngdLRowSum;
            pgdORowSum = coefInGaussion * pgdORowSum;
            ngdORowSum = coefInGaussion * ngdORowSum;
            const float pgdO2RowSum = pgdORowSum * pgdORowSum;
            const float ngdO2RowSum = ngdORowSum * ngdORowSum;
            short bandID = (short)(hID / widthOfBand);
            for (int nb = 0; nb < 3; ++nb) {        /* own band (:1201-1210), the one above (:1215-1227), the one below (:1228-1239) */
                int b, off;
                if (nb == 0) { b = bandID; off = widthOfBand; }
                else if (nb == 1) { b = bandID - 1; off = 2 * widthOfBand; if (b < 0) continue; }
                else { b = bandID + 1; off = 0; if (b >= PLO_NUM_OF_BANDS) continue; }
                coefInGaussion = (float)(gaussCoefL[hID % widthOfBand + off]);
                pgdLBandSum[b] += coefInGaussion * pgdLRowSum;
                ngdLBandSum[b] += coefInGaussion * ngdLRowSum;
                pgdL2BandSum[b] += coefInGaussion * coefInGaussion * pgdL2RowSum;
                ngdL2BandSum[b] += coefInGaussion * coefInGaussion * ngdL2RowSum;
                pgdOBandSum[b] += coefInGaussion * pgdORowSum;
                ngdOBandSum[b] += coefInGaussion * ngdORowSum;
                pgdO2BandSum[b] += coefInGaussion * coefInGaussion * pgdO2RowSum;
                ngdO2BandSum[b] += coefInGaussion * coefInGaussion * ngdO2RowSum;
            }
        }
        float* desVec = lbd + (size_t)li * descriptor_size;
        const float invN2 = (float)(1.0 / (widthOfBand * 2.0));
        const float invN3 = (float)(1.0 / (widthOfBand * 3.0));
        for (short bandID = 0; bandID < PLO_NUM_OF_BANDS; bandID++) {
            const float invN = (bandID == 0 || bandID == PLO_NUM_OF_BANDS - 1) ? invN2 : invN3;
            const short desID = bandID * 8;
            float temp = pgdLBandSum[bandID] * invN;
            desVec[desID] = temp;
            desVec[desID + 4] = sqrtf(pgdL2BandSum[bandID] * invN - temp * temp);
            temp = ngdLBandSum[bandID] * invN;
            desVec[desID + 1] = temp;
            desVec[desID + 5] = sqrtf(ngdL2BandSum[bandID] * invN - temp * temp);
            temp = pgdOBandSum[bandID] * invN;
            desVec[desID + 2] = temp;
            desVec[desID + 6] = sqrtf(pgdO2BandSum[bandID] * invN - temp * temp);
            temp = ngdOBandSum[bandID] * invN;
            desVec[desID + 3] = temp;
            desVec[desID + 7] = sqrtf(ngdO2BandSum[bandID] * invN - temp * temp);
        }
        float tempM = 0, tempS = 0;
        for (int i = 0; i < descriptor_size; i += 8) {
            tempM += desVec[i] * desVec[i];
            tempM += desVec[i + 1] * desVec[i + 1];
            tempM += desVec[i + 2] * desVec[i + 2];
            tempM += desVec[i + 3] * desVec[i + 3];
            tempS += desVec[i + 4] * desVec[i + 4];
            tempS += desVec[i + 5] * desVec[i + 5];
            tempS += desVec[i + 6] * desVec[i + 6];
            tempS += desVec[i + 7] * desVec[i + 7];
        }
        tempM = 1 / sqrtf(tempM);
        tempS = 1 / sqrtf(tempS);
        for (int i = 0; i < descriptor_size; i += 8) {
            desVec[i] = desVec[i] * tempM;
            desVec[i + 1] = desVec[i + 1] * tempM;
            desVec[i + 2] = desVec[i + 2] * tempM;
            desVec[i + 3] = desVec[i + 3] * tempM;
            desVec[i + 4] = desVec[i + 4] * tempS;
            desVec[i + 5] = desVec[i + 5] * tempS;
            desVec[i + 6] = desVec[i + 6] * tempS;
            desVec[i + 7] = desVec[i + 7] * tempS;
        }
        for (short i = 0; i < descriptor_size; i++)
            if (desVec[i] > 0.4) desVec[i] = (float)0.4;
        float temp = 0;
        for (short i = 0; i < descriptor_size; i++) temp += desVec[i] * desVec[i];
        temp = 1 / sqrtf(temp);
        for (short i = 0; i < descriptor_size; i++) desVec[i] = desVec[i] * temp;
    }
    free(gaussCoefG);
    free(gaussCoefL);
}

/* ---- LBD float -> binary line descriptor -------------------------------------------------- */
/* binary_descriptor_custom.cpp:74-107 -- band index pairs of the 32 output bytes */
const int plo_lbd_pairs[32][2] = {
    {0, 1}, {0, 2}, {0, 3}, {0, 4}, {0, 5}, {0, 6},
    {1, 2}, {1, 3}, {1, 4}, {1, 5}, {1, 6},
    {2, 3}, {2, 4}, {2, 5}, {2, 6}, {2, 7}, {2, 8},
    {3, 4}, {3, 5}, {3, 6}, {3, 7}, {3, 8},
    {4, 5}, {4, 6}, {4, 7}, {4, 8},
    {5, 6}, {5, 7}, {5, 8},
    {6, 7}, {6, 8},
    {7, 8}};

/* binary_descriptor_custom.cpp:401-412 (get2Pow(i) = 2^i for 0<=i<=7, :338-341) */
uint8_t plo_lbd_binary_conversion(const float* f1, const float* f2)
{
    uint8_t result = 0;
    for (int i = 0; i < 8; ++i)
        if (f1[i] > f2[i]) result = (uint8_t)(result + (1u << i));
    return result;
}

/* binary_descriptor_custom.cpp:653-668 */
void plo_lbd_binarise(const float* lbd, int32_t n, uint8_t* desc)
{
    for (int32_t l = 0; l < n; ++l) {
        const float* v = lbd + (size_t)l * PLO_LBD_FLOATS;
        uint8_t* row = desc + (size_t)l * 32;
        for (int comb = 0; comb < 32; ++comb)
            row[comb] = plo_lbd_binary_conversion(&v[8 * plo_lbd_pairs[comb][0]],
                                                  &v[8 * plo_lbd_pairs[comb][1]]);
    }
}
