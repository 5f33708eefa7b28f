/*
 * trx_knn.h -- C ABI of libtrxknn.so: exact brute-force k-NN ("flat index") on one MI355X.
 *
 * This is the drop-in boundary for the retrieval hot path of thomas0809/textreact.  The reference
 * has no native code; what it binds is the FAISS Python object protocol, used at
 *     retrieve/retrieve_faiss.py:65   index = faiss.IndexFlatL2(d)        -> trx_index_create
 *     retrieve/retrieve_faiss.py:66   index.add(train_fps)                -> trx_index_add
 *     retrieve/retrieve_faiss.py:71   distance, rank = index.search(q, k) -> trx_index_search
 * (the inner-product form of the same protocol, faiss.IndexFlatIP, is what the external dense
 * retriever named at README.md:44-47 runs; BASELINE.json configs[0..2]).  Each entry point below
 * names the call it replaces.  Plain pointers and sizes only: no torch / HIP types in signatures
 * (streams travel as void*).  Functions return 0 on success or a negative TRX_E* code and never
 * throw; trx_last_error() gives the message for the calling thread.
 *
 * Threading: thread-compatible -- one index per thread at a time.  The *_device variants take their
 * inputs and outputs in device memory and run their kernels on the caller's `stream` (so they order
 * against the caller's other work on that stream).  trx_index_add_device synchronises the stream
 * before it returns (the caller may free x).  A search is stream-ordered work followed by ONE read-back:
 * trx_index_search_device_begin enqueues everything -- query statistics, the exact-class decision (a
 * device-side flag), scan, select and the exact re-scan of up to 4 queries per 65,536-query batch whose
 * certificate failed (a device-side count decides) -- and returns without waiting for the GPU;
 * trx_index_search_finish waits for the stream, reads the certificate counts back and, when more
 * queries failed than the enqueued re-scan covers (near-duplicate clusters; never on the benchmark
 * inputs), completes them.  Work the caller enqueues on the same stream between the two calls (the
 * all-gather and merge of the row-sharded search) overlaps nothing of the search but costs no host
 * round trip either; stats.late_fallback says whether it consumed outputs that finish then changed.
 * Memory: the large search workspaces (2.3 GiB for a 65,536-query batch at 4 corpus splits) are shared by all indexes of
 * the process on one device and never shrink while an index exists there; searches of different indexes serialise on
 * the GPU where they use them.  Per index: the data, 256 KiB of flags per batch, and the exact fall-back's rows
 * (4 x n doubles: 32 MB per million vectors, reserved with the first fast-path search since the device decides
 * whether the rows are needed).
 * trx_index_search_device / _s64 / trx_index_search are begin + finish: on return the outputs are
 * final.  (One exception to "no wait in begin": fp32 queries against an index that so far holds only
 * bf16-exact data read their statistics back, because inexact queries re-lay the index out.)
 *
 * Results (both metrics): neighbours are the k best by the total order
 *   (score best-first, then id ascending), score = fp64 fma chain over the d components,
 *   IP: sum x*y;  L2: sum (x-y)^2;  D = (float)score.  When fewer than k vectors are indexed the
 *   trailing slots are I = -1, D = +FLT_MAX (L2) / -FLT_MAX (IP) -- the FAISS convention.
 * See DESIGN.md "Exactness" and oracle/flat_knn_ref.c (trxo_knn_canonical).
 */
#ifndef TRX_KNN_H
#define TRX_KNN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct trx_index trx_index;

enum { TRX_METRIC_IP = 0, TRX_METRIC_L2 = 1 };
enum { TRX_DTYPE_F32 = 0, TRX_DTYPE_BF16 = 1,
       /* HOST entry points only (trx_index_add, trx_index_search) -- the array types the reference hands FAISS: */
       TRX_DTYPE_I8 = 2,  /* int8 rows -- the Morgan bit vectors of retrieve_faiss.py:36-44 -- cross PCIe as bytes and are
                             widened on the device; same values, same results as their float32 conversion */
       TRX_DTYPE_I64 = 3, /* int64 rows -- the reaction difference fingerprints of retrieve_faiss.py:24-27 (numpy's default
                             integer).  FAISS' Python wrapper turns such an array into float32 on one thread before the index
                             sees it; here worker threads (TRX_HOST_THREADS, default min(cores, 32)) narrow it to int8 while
                             every value of a block fits a signed byte -- an eighth of the bytes over PCIe -- and convert a
                             block that holds a larger value to the float32 FAISS would have seen.  Same values, same results */
       TRX_DTYPE_I32 = 4, TRX_DTYPE_I16 = 5, TRX_DTYPE_U8 = 6, /* likewise */
       TRX_DTYPE_F64 = 7  /* float64 rows: rounded to float32 (nearest even) by the worker threads, as the wrapper's astype does */ };

enum {
    TRX_OK = 0,
    TRX_EINVAL = -1,   /* bad argument (d mismatch, k out of range, null pointer, ...) */
    TRX_ENOMEM = -2,   /* host or device allocation failed */
    TRX_EHIP = -3,     /* a HIP runtime call or kernel launch failed */
    TRX_ENODEV = -4    /* no usable gfx950 device */
};

#define TRX_MAX_K 2048   /* largest k any path accepts (FAISS GPU's own limit) */
#define TRX_FAST_MAX_K 24 /* k up to this: one scan (thresholds from the k' = 16 / 32 best keys seen, candidates certified) */
#define TRX_WIDE_MAX_K 256 /* k up to this: two scans -- the first ranks 24 rows per query exactly, the second lists every row
                             above a threshold extrapolated from them, and the exact scores of those rows prove the answer;
                             a query left with fewer than k rows (stats.n_rescored counts them) gets a better threshold from
                             the rows it did find and a third scan (n_rescanned), then the exact scan (n_uncertified).
                             k above this: the exact fp64 scan of the whole index for every query (0.17 ms per query at
                             1,000,000 x 768 rows) */

/* faiss.IndexFlatIP(d) / faiss.IndexFlatL2(d)  [retrieve_faiss.py:65].
 * device = HIP device ordinal this index lives on (one process per GPU: pass LOCAL_RANK). */
int trx_index_create(int d, int metric, int device, trx_index** out);

/* index.add(x)  [retrieve_faiss.py:66]: append n vectors (row-major, contiguous, d components
 * each) from HOST memory; the caller keeps ownership of x.  Ids are assigned sequentially.  Any TRX_DTYPE_*: the rows reach
 * the device through pinned double buffers filled by worker threads, a chunk crossing PCIe while the next is prepared. */
int trx_index_add(trx_index* idx, const void* x, int64_t n, int dtype);

/* Same, x already in DEVICE memory of the index's device; runs on `stream` (hipStream_t) and
 * synchronises it before returning. */
int trx_index_add_device(trx_index* idx, const void* x, int64_t n, int dtype, void* stream);

/* index.ntotal */
int64_t trx_index_ntotal(const trx_index* idx);

/* index.d */
int trx_index_dim(const trx_index* idx);

/* index.reset(): drop all vectors, keep d / metric / device. */
int trx_index_reset(trx_index* idx);

/* del index */
void trx_index_destroy(trx_index* idx);

/* distance, rank = index.search(q, k)  [retrieve_faiss.py:70-71]: HOST in, HOST out.
 * D: float[nq*k], I: int64[nq*k], caller-allocated.  Queries are taken in blocks of 65,536: the next block's copy to the
 * device runs beside the current block's search; trx_index_last_stats afterwards covers all blocks.  Any TRX_DTYPE_* (the
 * integer and float64 types are accepted here and by trx_index_add only: see the enum). */
int trx_index_search(trx_index* idx, const void* q, int64_t nq, int dtype, int k, float* D,
                     int64_t* I);

/* The storage-type change the two host entry points above apply to the caller's array, alone and host to host (no GPU is
 * touched: CPU tests pin it to numpy's astype): `count` components of `dtype` -> to_dtype TRX_DTYPE_I8 (integer dtypes only)
 * or TRX_DTYPE_F32 (the float32 FAISS' wrapper makes of every array), on the library's worker threads.  Returns 0, 1 when
 * to_dtype is I8 and some value does not fit a signed byte (dst is then undefined), or a negative TRX_E* code. */
int trx_host_convert(const void* src, int dtype, int64_t count, void* dst, int to_dtype);

/* Worker threads of the host entry points: TRX_HOST_THREADS, default min(cores in the affinity mask, 32). */
int trx_host_threads(void);

/* Same with q, D, I in DEVICE memory; runs on `stream`, blocks until D and I are final (see
 * Threading above).  This is the form bench.py times (inputs resident in HBM) and the one the
 * row-sharded multi-GPU path uses. */
int trx_index_search_device(trx_index* idx, const void* q, int64_t nq, int dtype, int k, float* D,
                            int64_t* I, void* stream);

/* Same as trx_index_search_device, and additionally writes the fp64 canonical scores S[nq*k]
 * (pads: +-FLT_MAX as double).  The row-sharded search all-gathers (S, I + shard offset) so that
 * the merge below orders by exactly the values a single unsharded index orders by. */
int trx_index_search_device_s64(trx_index* idx, const void* q, int64_t nq, int dtype, int k, float* D,
                                int64_t* I, double* S, void* stream);

/* Stream-ordered form of the two calls above (S may be null): see Threading.  D, I, S are final only
 * after trx_index_search_finish(idx) has returned 0.  One search per index may be in flight. */
int trx_index_search_device_begin(trx_index* idx, const void* q, int64_t nq, int dtype, int k, float* D,
                                  int64_t* I, double* S, void* stream);
int trx_index_search_finish(trx_index* idx);

/* Merge step of the row-sharded search: nlists result lists per query,
 * ids already global, DEVICE memory, layout S_lists/I_lists [nlists][nq][k] exactly as an
 * all-gather of per-shard (S, I) leaves them (nlists <= 16).  Same total order as a single
 * unsharded index: score best first on the fp64 value, then id ascending; D = (float)S. */
int trx_merge_topk_device(int metric, int nlists, int64_t nq, int k, const double* S_lists,
                          const int64_t* I_lists, float* D, int64_t* I, void* stream);

/* The same merge, and additionally the merged fp64 scores S[nq*k] (may be null): what trx_faiss_tie_order_device reads. */
int trx_merge_topk_device_s64(int metric, int nlists, int64_t nq, int k, const double* S_lists,
                              const int64_t* I_lists, float* D, int64_t* I, double* S, void* stream);

/* Order among EXACT score ties.  TRX_TIES_BY_ID (default): the total order above -- score best first, then id ascending,
 * both metrics.  TRX_TIES_FAISS: what faiss.IndexFlatIP / IndexFlatL2 [retrieve_faiss.py:65, :71] return.  For L2 that IS
 * the total order (FAISS' max-heap ordered by (distance, id) with strict admission keeps the k smallest (distance, id));
 * for the inner product FAISS' min-heap gives another deterministic answer -- of the rows tied at the k-th score, those
 * admitted while the heap was not yet full minus the smallest ids evicted by better rows that arrived later, and the output
 * ordered (score descending, id DESCENDING) -- which this mode reproduces exactly: the heap of FAISS >= 1.7.3, whose sift compares
 * (value, id) pairs (faiss/utils/ordered_key_value.h: cmp2); older releases break ties by the heap's structure and are not replayed
 * (closed form in knn_select.hip:
 * faiss_tie_kernel; checked against the heap replay of oracle/flat_knn_ref.c).  Ties are ties of the canonical fp64 score:
 * duplicates and exact-arithmetic inputs (integer fingerprints, the 2^-3 grid); on inputs whose fp32 sums round, FAISS'
 * own near-tie order depends on its BLAS and is not defined by FAISS either.  Cost: an inner-product search runs for the
 * canonical top 2k (k <= 1024).  Set before searching; not while a search is in flight. */
enum { TRX_TIES_BY_ID = 0, TRX_TIES_FAISS = 1 };
int trx_index_set_tie_rule(trx_index* idx, int rule);

/* The FAISS order from a canonical list: S2 / I2 [nq][k2] hold each query's canonical top k2 >= 2k (fp64 scores best first,
 * ids ascending among equal scores, pads I = -1 last) of an INNER-PRODUCT search, DEVICE memory; writes FAISS' top k to
 * D / I [nq][k].  The row-sharded search applies it after its merge (trx_merge_topk_device_s64 at k2), so that a sharded
 * index answers as one FAISS index over all rows would. */
int trx_faiss_tie_order_device(int64_t nq, int k2, int k, const double* S2, const int64_t* I2, float* D, int64_t* I,
                               void* stream);

/* Counters of the most recent search on this index (all zero before the first one). */
typedef struct trx_search_stats {
    int64_t nq;            /* queries in the call */
    int64_t n_uncertified; /* queries whose candidate lists could not be certified exact and were
                              re-done by the exact fp64 scan of the whole index (0 on benign inputs) */
    int32_t k_split;       /* K of the bf16 MFMA contraction: d (inputs exact in bf16; since round 4 also fp32 inputs, through
                              their bf16 rounding and a listing slack) or 3d (fp32 inputs as a three-term split: TRX_FP32_SPLIT=1) */
    int32_t n_splits;      /* corpus column-splits per query tile in the scan kernel */
    int32_t exact_class;   /* 1 = all partial sums exactly representable (integer inputs) */
    int32_t scan_launches; /* scan-kernel launches in the call */
    float scan_ms;         /* HIP-event time of the scan kernel(s), valid when timing is enabled */
    float total_ms;        /* HIP-event time of the whole call on its stream, ditto */
    int32_t late_fallback; /* 1 = trx_index_search_finish re-did queries AFTER the enqueued work (more
                              certificate failures than the inline re-scan covers): anything the caller
                              computed from D / I / S between begin and finish must be redone */
    int32_t n_rescored;    /* queries the select kernel flagged and the wide re-score looked at (all their listed rows
                              re-scored exactly) */
    int32_t n_rescanned;   /* of those, queries the wide re-score could not certify either and the fixed-threshold re-scan
                              took; n_uncertified of them went on to the exact scan */
    int32_t int8_scan;     /* 1 = the scan ran in its int8 form (integer inputs that fit a signed byte -- the L2 scan stages
                              the doubled query --, d >= 256: v_mfma_i32_16x16x64_i8, twice the MACs per instruction);
                              2 = in its fp4 form (every value on both sides one of 0, +-1, +-2, +-3, +-4, +-6 -- Morgan bit
                              vectors: v_mfma_f32_16x16x128_f8f6f4 on E2M1 operands, four times the MACs per instruction);
                              same keys, same lists, same results as the bf16 form */
} trx_search_stats;

int trx_index_last_stats(const trx_index* idx, trx_search_stats* out);

/* sizeof(trx_search_stats) as this library was built: a caller compiled against another revision of the struct checks
 * it before handing trx_index_last_stats a buffer (the struct grew in 0.2: `late_fallback`, `n_rescored`, `n_rescanned`). */
int trx_search_stats_size(void);

/* Enable (1) / disable (0) HIP-event timing of the scan kernel inside search calls.  Timing makes
 * begin synchronise after every batch; leave it off for overlapped pipelines. */
int trx_index_set_timing(trx_index* idx, int enabled);

/* Message of the last error on this thread ("" if none). */
const char* trx_last_error(void);

/* Library version string, e.g. "trxknn 0.2 (gfx950)". */
const char* trx_version(void);

#ifdef __cplusplus
}
#endif
#endif /* TRX_KNN_H */
