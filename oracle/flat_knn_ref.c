/*
 * oracle/flat_knn_ref.c -- CPU restatement of the exact flat k-NN search that the reference runs
 * at retrieve/retrieve_faiss.py:62-74 (faiss.IndexFlatL2(d) :65, index.add :66, index.search :71).
 *
 * THIS IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load it.  The product path (textreact_amd/) never links, imports or calls it.
 *
 * PARITY UNPINNED.  FAISS itself is a third-party dependency that is absent from /root/reference
 * (not vendored, not pinned in environment.yml:27-129, not installable here), and the reference
 * holds no tests, golden vectors or fixtures for this path.  What follows
 * restates the *published* algorithm of faiss (>= 1.7.3) IndexFlat::search from its upstream
 * sources -- faiss/IndexFlat.cpp (IndexFlat::search -> knn_inner_product / knn_L2sqr),
 * faiss/utils/distances.cpp (exhaustive_{inner_product,L2sqr}_{seq,blas}: query block 4096,
 * database block 1024, BLAS path when nx >= 20, L2 = |x|^2 + |y|^2 - 2 x.y clamped at 0),
 * faiss/impl/ResultHandler.h (HeapBlockResultHandler: scan ids ascending, admit iff the heap top
 * compares strictly worse), faiss/utils/Heap.h + ordered_key_value.h (binary heap ordered by
 * (value, id), final heap_reorder) -- in my own words; no FAISS source was available to check it
 * against, so the restatement is anchored on the reference's call site and on properties that any
 * exact flat search must satisfy (tests/test_oracle.py).
 *
 * Two scoring modes are provided:
 *
 *  trxo_knn_faiss()      the literal restatement: fp32 inner products in (4096 x 1024) blocks,
 *                        fp32 norms trick for L2, strict-admission heap with (value,id) ordering.
 *                        Its fp32 accumulation ORDER is this file's own (k ascending, 8 interleaved
 *                        partial sums -- an "sgemm" stand-in); real FAISS inherits the order of
 *                        whatever BLAS it was linked with, so on inputs whose partial sums are not
 *                        exactly representable the last bits of its distances are not defined by
 *                        FAISS either.
 *
 *  trxo_knn_canonical()  the order-independent definition the HIP path is held to bit-for-bit:
 *                        score = fp64 fused multiply-add chain over k = 0..d-1
 *                          IP : s = fma((double)x[k], (double)y[k], s)
 *                          L2 : t = (double)x[k] - (double)y[k]; s = fma(t, t, s)
 *                        D = (float)s, neighbours = the k best by the total order
 *                        (score best first on the fp64 value, then id ascending).
 *                        On inputs whose fp32 partial sums are all exact (integer fingerprints --
 *                        the reference's real input class -- and the 2^-3 grid set) both modes
 *                        give identical D and, for L2, identical I including ties; for IP they
 *                        give identical I wherever no exact score tie occurs (the FAISS heap's
 *                        tie order for the min-heap is an artefact of heap mechanics; see
 *                        DESIGN.md "Tie rule").
 *
 * Build: make -C oracle   (gcc -O2 -fopenmp -shared; see oracle/Makefile)
 */
#include <float.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define TRXO_METRIC_IP 0
#define TRXO_METRIC_L2 1

/* ------------------------------------------------------------------------------------------ */
/* Heap with (value, id) ordering.  "worse" is the relation that puts an element nearer the top:
 * for L2 (keep the k smallest) the top is the LARGEST (dist, id); for IP (keep the k largest)
 * the top is the SMALLEST (score, id).  [faiss/utils/ordered_key_value.h: CMax/CMin::cmp2]      */

typedef struct {
    int keep_smallest; /* 1 = L2 (max-heap on top), 0 = IP (min-heap on top) */
} heap_kind;

static inline int worse2(heap_kind h, float va, int64_t ia, float vb, int64_t ib) {
    /* is (va, ia) nearer the top than (vb, ib)? */
    if (h.keep_smallest) return (va > vb) || (va == vb && ia > ib);
    return (va < vb) || (va == vb && ia < ib);
}

static inline int strictly_better_than_top(heap_kind h, float top, float v) {
    /* admission test of HeapBlockResultHandler::add_results: C::cmp(thresh, dis), values only */
    return h.keep_smallest ? (top > v) : (top < v);
}

static inline float neutral(heap_kind h) { return h.keep_smallest ? FLT_MAX : -FLT_MAX; }

/* replace the top element by (v, id) and sift it down  [Heap.h: heap_replace_top] */
static void heap_replace_top(heap_kind h, int k, float* hv, int64_t* hi, float v, int64_t id) {
    int i = 0; /* 0-based: children of i are 2i+1, 2i+2 */
    for (;;) {
        int c1 = 2 * i + 1, c2 = c1 + 1, c;
        if (c1 >= k) break;
        if (c2 >= k || worse2(h, hv[c1], hi[c1], hv[c2], hi[c2])) c = c1; else c = c2;
        if (worse2(h, v, id, hv[c], hi[c])) break; /* new element already nearer the top */
        hv[i] = hv[c]; hi[i] = hi[c];
        i = c;
    }
    hv[i] = v; hi[i] = id;
}

/* pop the top of a heap of n elements  [Heap.h: heap_pop] */
static void heap_pop(heap_kind h, int n, float* hv, int64_t* hi) {
    float v = hv[n - 1]; int64_t id = hi[n - 1];
    int i = 0;
    for (;;) {
        int c1 = 2 * i + 1, c2 = c1 + 1, c;
        if (c1 >= n) break;
        if (c2 >= n || worse2(h, hv[c1], hi[c1], hv[c2], hi[c2])) c = c1; else c = c2;
        if (worse2(h, v, id, hv[c], hi[c])) break;
        hv[i] = hv[c]; hi[i] = hi[c];
        i = c;
    }
    hv[i] = v; hi[i] = id;
}

/* sort the heap content best-first in place, unfilled slots (id == -1) last  [Heap.h: heap_reorder] */
static void heap_reorder(heap_kind h, int k, float* hv, int64_t* hi) {
    int filled = 0;
    for (int i = 0; i < k; i++) {
        float v = hv[0]; int64_t id = hi[0];
        heap_pop(h, k - i, hv, hi);
        hv[k - filled - 1] = v; hi[k - filled - 1] = id;
        if (id != -1) filled++;
    }
    memmove(hv, hv + k - filled, (size_t)filled * sizeof(float));
    memmove(hi, hi + k - filled, (size_t)filled * sizeof(int64_t));
    for (int i = filled; i < k; i++) { hv[i] = neutral(h); hi[i] = -1; }
}

/* ------------------------------------------------------------------------------------------ */
/* fp32 kernels standing in for fvec_inner_product / fvec_L2sqr / sgemm_                         */

static float ip_f32(const float* x, const float* y, int d) {
    float p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int k = 0;
    for (; k + 8 <= d; k += 8)
        for (int u = 0; u < 8; u++) p[u] += x[k + u] * y[k + u];
    float s = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
    for (; k < d; k++) s += x[k] * y[k];
    return s;
}

static float l2_f32(const float* x, const float* y, int d) {
    float p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int k = 0;
    for (; k + 8 <= d; k += 8)
        for (int u = 0; u < 8; u++) { float t = x[k + u] - y[k + u]; p[u] += t * t; }
    float s = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
    for (; k < d; k++) { float t = x[k] - y[k]; s += t * t; }
    return s;
}

/* Add a block of results dis[(i - i0) * (j1 - j0) + (j - j0)] to the per-query heaps.
 * Exported so that bench.py can time "host BLAS sgemm + this handler" (FAISS's own structure).
 * [ResultHandler.h: HeapBlockResultHandler::add_results]                                        */
void trxo_heap_add_block(int metric, int k, int64_t i0, int64_t i1, int64_t j0, int64_t j1,
                         const float* dis, float* D, int64_t* I) {
    heap_kind h = {metric == TRXO_METRIC_L2};
#pragma omp parallel for schedule(static)
    for (int64_t i = i0; i < i1; i++) {
        float* hv = D + i * k; int64_t* hi = I + i * k;
        const float* row = dis + (i - i0) * (j1 - j0);
        float top = hv[0];
        for (int64_t j = j0; j < j1; j++) {
            float v = row[j - j0];
            if (strictly_better_than_top(h, top, v)) {
                heap_replace_top(h, k, hv, hi, v, j);
                top = hv[0];
            }
        }
    }
}

/* OpenMP team size of the CALLING thread (its own ICV): a host thread that is itself one of many workers -- the
 * block-parallel driver of flat_knn.knn_faiss_blas_mt -- sets 1, so that the handlers above run inline in it. */
void trxo_set_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n > 0 ? n : 1);
#else
    (void)n;
#endif
}

/* [ResultHandler.h: begin_multiple -> heap_heapify with neutral values and ids -1] */
void trxo_heap_begin(int metric, int k, int64_t nq, float* D, int64_t* I) {
    heap_kind h = {metric == TRXO_METRIC_L2};
    for (int64_t t = 0; t < nq * k; t++) { D[t] = neutral(h); I[t] = -1; }
}

/* [ResultHandler.h: end_multiple -> heap_reorder per query] */
void trxo_heap_end(int metric, int k, int64_t nq, float* D, int64_t* I) {
    heap_kind h = {metric == TRXO_METRIC_L2};
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nq; i++) heap_reorder(h, k, D + i * k, I + i * k);
}

/* L2 block fix-up of exhaustive_L2sqr_blas: dis = |x|^2 + |y|^2 - 2 ip, negative -> 0 */
void trxo_l2_from_ip_block(int64_t i0, int64_t i1, int64_t j0, int64_t j1, const float* xn,
                           const float* yn, float* blk) {
#pragma omp parallel for schedule(static)
    for (int64_t i = i0; i < i1; i++) {
        float* row = blk + (i - i0) * (j1 - j0);
        for (int64_t j = j0; j < j1; j++) {
            float dis = xn[i] + yn[j] - 2 * row[j - j0];
            if (dis < 0) dis = 0;
            row[j - j0] = dis;
        }
    }
}

void trxo_norms_f32(const float* x, int64_t n, int d, float* out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) out[i] = ip_f32(x + i * d, x + i * d, d);
}

/* The literal restatement.  x: nq x d queries, y: nb x d database, both fp32 row-major.
 * [IndexFlat.cpp: IndexFlat::search; distances.cpp: knn_inner_product / knn_L2sqr]              */
int trxo_knn_faiss(int metric, const float* x, int64_t nq, const float* y, int64_t nb, int d,
                   int k, float* D, int64_t* I) {
    if (k <= 0 || d <= 0 || nq < 0 || nb < 0) return -1;
    heap_kind h = {metric == TRXO_METRIC_L2};
    trxo_heap_begin(metric, k, nq, D, I);
    if (nq == 0 || nb == 0) { trxo_heap_end(metric, k, nq, D, I); return 0; }

    const int64_t blas_threshold = 20; /* distance_compute_blas_threshold */
    if (nq < blas_threshold) {
        /* exhaustive_*_seq: one query at a time, direct distance, same admission rule */
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < nq; i++) {
            float* hv = D + i * k; int64_t* hi = I + i * k;
            float top = hv[0];
            for (int64_t j = 0; j < nb; j++) {
                float v = (metric == TRXO_METRIC_L2) ? l2_f32(x + i * d, y + j * d, d)
                                                     : ip_f32(x + i * d, y + j * d, d);
                if (strictly_better_than_top(h, top, v)) {
                    heap_replace_top(h, k, hv, hi, v, j);
                    top = hv[0];
                }
            }
            heap_reorder(h, k, hv, hi);
        }
        return 0;
    }

    const int64_t bs_x = 4096, bs_y = 1024; /* distance_compute_blas_{query,database}_bs */
    float* blk = (float*)malloc((size_t)bs_x * bs_y * sizeof(float));
    float* xn = NULL; float* yn = NULL;
    if (!blk) return -2;
    if (metric == TRXO_METRIC_L2) {
        xn = (float*)malloc((size_t)nq * sizeof(float));
        yn = (float*)malloc((size_t)nb * sizeof(float));
        if (!xn || !yn) { free(blk); free(xn); free(yn); return -2; }
        trxo_norms_f32(x, nq, d, xn);
        trxo_norms_f32(y, nb, d, yn);
    }
    for (int64_t i0 = 0; i0 < nq; i0 += bs_x) {
        int64_t i1 = i0 + bs_x < nq ? i0 + bs_x : nq;
        for (int64_t j0 = 0; j0 < nb; j0 += bs_y) {
            int64_t j1 = j0 + bs_y < nb ? j0 + bs_y : nb;
            /* the sgemm_ call: blk[i][j] = x_i . y_j */
#pragma omp parallel for schedule(static)
            for (int64_t i = i0; i < i1; i++)
                for (int64_t j = j0; j < j1; j++)
                    blk[(i - i0) * (j1 - j0) + (j - j0)] = ip_f32(x + i * d, y + j * d, d);
            if (metric == TRXO_METRIC_L2) trxo_l2_from_ip_block(i0, i1, j0, j1, xn, yn, blk);
            trxo_heap_add_block(metric, k, i0, i1, j0, j1, blk, D, I);
        }
    }
    trxo_heap_end(metric, k, nq, D, I);
    free(blk); free(xn); free(yn);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Canonical definition (what the HIP path must reproduce bit-for-bit).                         */

double trxo_score_canonical(int metric, const float* x, const float* y, int d) {
    double s = 0.0;
    if (metric == TRXO_METRIC_L2) {
        for (int k = 0; k < d; k++) { double t = (double)x[k] - (double)y[k]; s = fma(t, t, s); }
    } else {
        for (int k = 0; k < d; k++) s = fma((double)x[k], (double)y[k], s);
    }
    return s;
}

/* is (sa, ia) ranked before (sb, ib)?  best score first, then smaller id */
static inline int before(int metric, double sa, int64_t ia, double sb, int64_t ib) {
    if (sa != sb) return metric == TRXO_METRIC_L2 ? (sa < sb) : (sa > sb);
    return ia < ib;
}

int trxo_knn_canonical(int metric, const float* x, int64_t nq, const float* y, int64_t nb, int d,
                       int k, float* D, int64_t* I) {
    if (k <= 0 || d <= 0 || nq < 0 || nb < 0) return -1;
    int err = 0;
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t i = 0; i < nq; i++) {
        double* bs = (double*)malloc((size_t)k * sizeof(double));
        int64_t* bi = (int64_t*)malloc((size_t)k * sizeof(int64_t));
        if (!bs || !bi) { err = 1; free(bs); free(bi); continue; }
        int n = 0; /* sorted best-first insertion list of the k best so far */
        for (int64_t j = 0; j < nb; j++) {
            double s = trxo_score_canonical(metric, x + i * d, y + j * d, d);
            if (s != s) continue; /* NaN never ranks */
            if (n == k && !before(metric, s, j, bs[k - 1], bi[k - 1])) continue;
            int p = n < k ? n : k - 1;
            while (p > 0 && before(metric, s, j, bs[p - 1], bi[p - 1])) {
                bs[p] = bs[p - 1]; bi[p] = bi[p - 1]; p--;
            }
            bs[p] = s; bi[p] = j;
            if (n < k) n++;
        }
        for (int t = 0; t < k; t++) {
            if (t < n) { D[i * k + t] = (float)bs[t]; I[i * k + t] = bi[t]; }
            else { D[i * k + t] = metric == TRXO_METRIC_L2 ? FLT_MAX : -FLT_MAX; I[i * k + t] = -1; }
        }
        free(bs); free(bi);
    }
    return err ? -2 : 0;
}

/* Canonical scores of given (query, id) pairs: out[i*k+t] = score(x_i, y_{I[i*k+t]}) as fp64.
 * Used by the near-tie audit in the tests.                                                     */
void trxo_scores_at(int metric, const float* x, int64_t nq, const float* y, int d, int k,
                    const int64_t* I, double* out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nq; i++)
        for (int t = 0; t < k; t++) {
            int64_t j = I[i * k + t];
            out[i * k + t] = j < 0 ? NAN : trxo_score_canonical(metric, x + i * d, y + j * d, d);
        }
}

/* Merge nlists sorted result lists per query (row-sharded search: ids already global) with the
 * canonical total order.  Layout Dl/Il: [nlists][nq][k].  D values are fp32 here, so the order
 * is (D best first, id ascending); entries with id < 0 are padding.                            */
void trxo_merge_lists(int metric, int nlists, int64_t nq, int k, const float* Dl,
                      const int64_t* Il, float* D, int64_t* I) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nq; i++) {
        int* pos = (int*)calloc((size_t)nlists, sizeof(int));
        for (int t = 0; t < k; t++) {
            int best = -1;
            for (int l = 0; l < nlists; l++) {
                if (pos[l] >= k) continue;
                int64_t off = ((int64_t)l * nq + i) * k + pos[l];
                if (Il[off] < 0) { pos[l] = k; continue; }
                if (best < 0) { best = l; continue; }
                int64_t boff = ((int64_t)best * nq + i) * k + pos[best];
                if (before(metric, Dl[off], Il[off], Dl[boff], Il[boff])) best = l;
            }
            if (best < 0) { D[i * k + t] = metric == TRXO_METRIC_L2 ? FLT_MAX : -FLT_MAX; I[i * k + t] = -1; }
            else {
                int64_t boff = ((int64_t)best * nq + i) * k + pos[best];
                D[i * k + t] = Dl[boff]; I[i * k + t] = Il[boff]; pos[best]++;
            }
        }
        free(pos);
    }
}
