/* tt_hip.h -- C ABI of libtt_hip.so, the MI355X (gfx950) implementation of
 * tensor-truth's retrieval hot path (embed -> exact top-k scan -> rerank).
 *
 * The reference (ljubobratovicrelja/tensor-truth) is pure Python and has no FFI
 * for this path; its "plugin boundary" is three LlamaIndex interfaces (SURVEY.md
 * section 8b).  Each entry point below names the reference call site whose
 * arithmetic it replaces.  The Python classes in tensor_truth_amd/ mirror the
 * reference's interfaces and call these functions through ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - the caller owns all buffers (torch tensors); nothing is allocated or freed
 *     inside a launch function, nothing synchronises the device, so every call
 *     is stream-ordered and graph-capturable;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - return 0 on success, a negative TT_E_* code on failure;
 *     tt_last_error() returns a thread-local message for the last failure;
 *   - re-entrant: no global mutable state, safe to call from the reference's
 *     executor threads (rag_engine.py:420-424) on distinct streams/workspaces;
 *   - bf16 tensors are raw uint16 storage, row-major, 16-byte aligned.
 */
#ifndef TT_HIP_H
#define TT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TT_OK 0
#define TT_E_INVALID (-1)   /* bad argument (shape, alignment, null pointer)  */
#define TT_E_UNSUPPORTED (-2) /* shape outside the compiled kernel set         */
#define TT_E_WORKSPACE (-3) /* workspace too small                            */
#define TT_E_HIP (-4)       /* a HIP runtime call failed                      */

/* ---- library info ------------------------------------------------------- */
int tt_version(void);              /* ABI version, currently 1                */
const char* tt_arch(void);         /* "gfx950"                                */
const char* tt_last_error(void);   /* thread-local, never NULL                */
int tt_device_cu_count(void);      /* compute units of the current device, <0 on error */

/* ---- similarity scan + top-k --------------------------------------------
 * Replaces the vector search inside VectorIndexRetriever.retrieve
 * (reference: src/tensortruth/rag_engine.py:639,674 -> ChromaVectorStore.query,
 * collection created at rag_engine.py:628-630; upstream HNSW is approximate,
 * this is the exact scan BASELINE.json specifies):
 *     S[q][n] = sum_d Q[q][d] * C[n][d]      bf16 inputs, fp32 accumulate
 *     out = top-k per query by (score desc, row index asc)
 * Fewer than k valid rows -> trailing entries are (score -inf, index -1).
 * `dim` must be a multiple of 128 and <= 1024; 1 <= k <= 1024.
 * out_idx[q][j] = idx_base + local row index.
 * status_flag (device int32, may be NULL): set non-zero iff a candidate buffer
 * overflowed, in which case results for this call are NOT exact and the caller
 * must re-run through tt_scan_topk_exact (adversarial score distributions only).
 */
size_t tt_scan_workspace_bytes(int64_t n_rows, int dim, int n_queries, int k);

int tt_scan_topk(const void* corpus_bf16, int64_t n_rows, int dim,
                 const void* queries_bf16, int n_queries, int k, int32_t idx_base,
                 float* out_scores, int32_t* out_idx,
                 void* workspace, size_t workspace_bytes,
                 int32_t* status_flag, void* stream);

/* Always-exact variant: dense scores for every row + selection.  Needs
 * n_queries * n_rows * 4 bytes of workspace; meant for small shards and as the
 * overflow fallback. */
size_t tt_scan_exact_workspace_bytes(int64_t n_rows, int dim, int n_queries, int k);
int tt_scan_topk_exact(const void* corpus_bf16, int64_t n_rows, int dim,
                       const void* queries_bf16, int n_queries, int k, int32_t idx_base,
                       float* out_scores, int32_t* out_idx,
                       void* workspace, size_t workspace_bytes, void* stream);

/* Merge per-shard partial top-k lists (the step after the RCCL all-gather of
 * SURVEY.md section 8e; also MultiIndexRetriever's concatenate+sort,
 * rag_engine.py:463-507, when indexes live in one matrix).
 * in_scores/in_idx: [n_queries][n_lists * k_in] (per query, lists concatenated);
 * entries with idx < 0 are padding.  Output ordered by (score desc, idx asc). */
int tt_topk_merge(const float* in_scores, const int32_t* in_idx,
                  int n_queries, int n_candidates, int k_out,
                  float* out_scores, int32_t* out_idx, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TT_HIP_H */
