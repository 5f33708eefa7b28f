/* trx_tanimoto.h -- C ABI of libtrxtani.so: brute-force Tanimoto similarity of count fingerprints on MI355X.
 *
 * Replaces the scoring loop of the reference's retrieve/retrieve.py:
 *   :18-31  reaction_similarity(...)  -> rdkit DataStructs.TanimotoSimilarity(fp1, fp2) on the difference
 *           fingerprints of two reactions (rdChemReactions.CreateDifferenceFingerprintForReaction: a sparse vector of
 *           signed counts over 2048 positions -- the same vectors retrieve_faiss.py:24-27 turns into dense int arrays)
 *   :34-40  compute_reaction_similarities(test_smiles, train_smiles_list): one query against every train row
 *           (a 64-process pool in the reference)
 *   :55-62  ranks = np.argsort(similarities)[::-1][:100]; {'rank': ranks, 'similarity': [...]}
 *
 * RDKit (third party, absent from this image; the reference pins no version) computes, for two sparse count vectors
 * (SparseIntVect.h: TanimotoSimilarity -> TverskySimilarity(a = b = 1) -> calcVectParams):
 *      |v| = sum_i |v_i|        and = sum_i min(|v1_i|, |v2_i|)        sim = and / (|v1| + |v2| - and),  0 if the
 * denominator is < 1e-6.  Here the vectors are dense arrays of d counts (d % 4 == 0); magnitudes are stored as bytes,
 * so |count| <= 255 and sum_i |v_i| < 32768 are required (the entry points report a violation instead of clamping).
 *
 * Layout of a packed corpus: rows in blocks of 64; dword j (counts 4j .. 4j+3, one byte each, little endian) of row r
 * lives at packed[((r / 64) * (d / 4) + j) * 64 + r % 64], so that a wave reads dword j of 64 rows with one coalesced
 * 256-byte load; rows past n in the last block are zero.  Size: trx_tanimoto_packed_bytes(n, d).
 *
 * All pointers are DEVICE pointers; every call is asynchronous on `stream`.  Return value: 0, or a negative code with
 * the text in trx_tanimoto_last_error().
 */
#ifndef TRX_TANIMOTO_H
#define TRX_TANIMOTO_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TRX_TANI_I64 0
#define TRX_TANI_I32 1
#define TRX_TANI_I8 2          /* signed bytes (what DataStructs.ConvertToNumpyArray gives for bit vectors) */
#define TRX_TANI_QUERY_GROUP 16 /* queries are handed over in groups of 16 (see trx_tanimoto_scores) */
#define TRX_TANI_KEY_ID_BITS 27 /* row numbers < 2^27 ride in the low bits of a key */

/* bytes of a packed corpus of n rows of d counts */
int64_t trx_tanimoto_packed_bytes(int64_t n, int d);

/* fps [n, ld] counts of type `dtype` (row-major, ld >= d) -> packed magnitudes + row_sum[n] = sum |count| (int32).
 * row_sum must be zeroed by the caller.  flags[0] (int32, zeroed by the caller) gets bit 0 set if a magnitude
 * exceeded 255.  With first_row > 0 the rows are appended after `first_row` existing rows (first_row % 64 == 0). */
int trx_tanimoto_pack(const void* fps, int dtype, int64_t n, int d, int64_t ld, int64_t first_row, void* packed, int32_t* row_sum,
                      int32_t* flags, void* stream);

/* queries: q_t [d / 4, nq_pad] dwords of packed magnitudes, TRANSPOSED (dword j of query q at q_t[j * nq_pad + q];
 * nq_pad = nq rounded up to a multiple of 16, padding zero) and q_sum[nq_pad].
 * -> and_out[q * ld_out + r] = sum_i min(|q_i|, |row_r,i|)  (int16: the sums are < 32768), from which
 *    sim(q, r) = and / (q_sum[q] + row_sum[r] - and), and, if block_max != NULL,
 *    block_max[q * ceil(n / 64) + b] = the best fp32 APPROXIMATION of sim among rows 64 b .. 64 b + 63
 *    ((float)and * rcp((float)den): relative error < 2^-21).  Requires n < 2^27. */
int trx_tanimoto_scores(const void* packed, const int32_t* row_sum, int64_t n, int d, const uint32_t* q_t, const int32_t* q_sum,
                        int nq, int16_t* and_out, int64_t ld_out, float* block_max, void* stream);

/* Selection without sorting all N similarities of a query (retrieve.py:59 argsorts them).
 * Exact order: key = floor(sim * (2^36 - 1)) << 27 | r with sim = (double)and / (double)den (the double the reference's
 * Python float holds).  Two different similarities of vectors with sums < 32768 differ by more than 2^-32 (> 15 key
 * units), equal rationals give the same double: comparing keys as integers IS the order (similarity, then row number),
 * ties won by the larger row number -- a stable ascending argsort read backwards.
 * Bound: let t = the k-th largest entry of block_max[q] (k blocks each hold a row whose approximate similarity is >= t).
 * Then the k-th best exact similarity is >= t (1 - 2^-21), and every one of the k best rows has an approximate
 * similarity >= t (1 - 2^-21)^2 > thr = t (1 - 2^-20).
 * trx_tanimoto_filter: for slot s = 0 .. nsel-1 and query q = q_ids ? q_ids[s] : s, appends the exact keys of the rows
 * whose approximate similarity is >= thr[q] (thr == NULL: of every row) to out[s * cap ...] in any order; counts[s]
 * (zeroed by the caller) ends as the number found -- above cap the list is incomplete and the caller repeats that
 * query with thr == NULL and cap >= n.  The k largest keys of a complete list are the k best matches, in order. */
int trx_tanimoto_filter(const int16_t* and_in, int64_t ld, const int32_t* row_sum, const int32_t* q_sum, const int32_t* q_ids, int nsel,
                        int64_t n, const float* thr, int64_t cap, int64_t* out, int32_t* counts, void* stream);

const char* trx_tanimoto_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
