/* C ABI of libcst_host.so: the HOST side of the pretrain stage's content-distance labels
 * (consistent__style_transfer_amd/csrc/host_wmd.cpp; plain C++17, no GPU).
 *
 * Replaces, for the three-stage training path only:
 *   src/wmd.py:31-45   WMDdistance.cal_wmd / cal_wmd_label  (gensim KeyedVectors.wmdistance -> pyemd, third-party, not vendored)
 *   src/loader.py:60   the call site inside collate_pretrain: one label per sentence of the batch
 *
 * A maintainer of the reference binds it with ctypes (INTEGRATION.md shows the stub); this build's binding is
 * consistent__style_transfer_amd/wmd.py.  Conventions: plain pointers and sizes, int status (0 ok, 1 bad argument), no exceptions
 * cross the boundary, the caller owns every buffer, thread-safe (no global state). */
#ifndef CST_HOST_H
#define CST_HOST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int cst_host_abi_version(void);

/* Exact earth mover's distance between histograms a[n] and b[m] of equal total mass (all entries > 0) under cost[n*m] (row-major):
 * min sum F_ij cost_ij, F >= 0, F 1 = a, F^T 1 = b.  What pyemd.emd computes for gensim's wmdistance (src/wmd.py:31-32). */
int cst_host_emd(int n, int m, const double* a, const double* b, const double* cost, double* out);

/* labels[p], p in [p_lo, p_hi), of n_pairs sentence pairs (src/wmd.py:34-45 cal_wmd_label over noised_1 / noised_2 of
 * src/loader.py:49-60).  Sentences are ragged int32 token-id lists: pair p = ids1[off1[p] .. off1[p+1]) and ids2[off2[p] .. off2[p+1]).
 * row_of_id[n_ids] maps a vocabulary id to its row of `vectors` [rows, dim] (float64, L2-normalised: src/wmd.py:54) or to -1 when the
 * token has no word vector (gensim drops such tokens).  Per pair: an empty id list -> max(len_1, len_2) (wmd.py:37-38); Word Mover's
 * Distance inf -> (len_1 + len_2) / 2 (wmd.py:41-42); else the distance.  Pairs outside [p_lo, p_hi) are left untouched: a
 * data-parallel rank computes the labels of its own rows only.  nthreads >= 1 worker threads split the range. */
int cst_host_wmd_labels(const int32_t* ids1, const int64_t* off1, const int32_t* ids2, const int64_t* off2, int64_t n_pairs,
                        int64_t p_lo, int64_t p_hi, const int32_t* row_of_id, int32_t n_ids, const double* vectors, int32_t dim,
                        int32_t nthreads, double* labels);

#ifdef __cplusplus
}
#endif
#endif /* CST_HOST_H */
