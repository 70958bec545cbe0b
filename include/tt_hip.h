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

/* The library is built with -fvisibility=hidden: the declarations of this header -- and nothing else -- are its dynamic
 * symbols (`nm -D libtt_hip.so`; tests/test_lib_abi.py). */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
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

/* Segmented variant for several index modules living in ONE matrix (replaces the thread-pool fan-out
 * over per-module retrievers, src/tensortruth/rag_engine.py:420-424, and the n_indexes separate vector
 * searches behind it): rows [seg_offsets[s], seg_offsets[s+1]) are module s (host array of
 * n_segments + 1 non-decreasing offsets within [0, n_rows], 1 <= n_segments <= 64).  One pass over the
 * matrix scores every row once; every (query, segment) then gets its own exact top-k, ordered by
 * (score desc, row asc):  out_scores / out_idx are [n_queries][n_segments][k], indices are
 * SEGMENT-LOCAL rows, short or empty segments are padded with (-inf, -1).
 * Workspace: n_queries * rows * 4 bytes -- meant for the handful of queries of an
 * interactive retrieve(); large query batches over one big index go through tt_scan_topk. */
size_t tt_scan_segmented_workspace_bytes(int64_t n_rows, int dim, int n_queries, int k);
int tt_scan_topk_segmented(const void* corpus_bf16, int64_t n_rows, int dim,
                           const void* queries_bf16, int n_queries, int k,
                           const int64_t* seg_offsets, int n_segments,
                           float* out_scores, int32_t* out_idx,
                           void* workspace, size_t workspace_bytes, void* stream);

/* fp8 shadow of the corpus: an EXACT prefilter for the scan of a lone caller (the reference's own usage: one un-batched
 * retrieve() per query, src/tensortruth/rag_engine.py:420-424, README.md:13), where tt_scan_topk is one full read of the bf16
 * matrix (20.5 GB at 10 M x 1024: 3 ms).  The shadow block holds, for a capacity of cap_rows rows, the rows as e4m3 bytes
 * (x 256) and two floats per row -- the norm of what the rounding took and the norm of what it kept -- so that
 * |q.c - q8.c8| <= ||q|| be + ||q - q8|| dn bounds every row's exact score from above (csrc/shadow.hip).
 * tt_scan_topk_shadow: threshold = k-th best exact score of a row sample (as tt_scan_topk); one pass over the SHADOW lists
 * the rows whose bound reaches it; those rows are re-scored from the bf16 matrix with tt_scan_topk's own arithmetic and the
 * same exact selection.  Scores and indices are bit-identical to tt_scan_topk's.  n_queries <= 4; rows [0, n_rows) of the
 * shadow must have been built from the same corpus rows (tt_scan_shadow_build, any sub-range at a time: appended rows only
 * need their own range).  status_flag as in tt_scan_topk (a survivor list that overflowed: re-run tt_scan_topk). */
size_t tt_scan_shadow_bytes(int64_t cap_rows, int dim);
int tt_scan_shadow_build(const void* corpus_bf16, int dim, int64_t row_lo, int64_t row_hi,
                         void* shadow, int64_t cap_rows, void* stream);
size_t tt_scan_shadow_workspace_bytes(int64_t n_rows, int dim, int n_queries, int k);
int tt_scan_topk_shadow(const void* corpus_bf16, const void* shadow, int64_t cap_rows, int64_t n_rows, int dim,
                        const void* queries_bf16, int n_queries, int k, int32_t idx_base,
                        float* out_scores, int32_t* out_idx,
                        void* workspace, size_t workspace_bytes, int32_t* status_flag, void* stream);

/* Merge per-shard partial top-k lists (the step after the RCCL all-gather of
 * SURVEY.md section 8e; also MultiIndexRetriever's concatenate+sort,
 * rag_engine.py:463-507, when indexes live in one matrix).
 * in_scores/in_idx: [n_queries][n_lists * k_in] (per query, lists concatenated);
 * entries with idx < 0 are padding.  Output ordered by (score desc, idx asc). */
int tt_topk_merge(const float* in_scores, const int32_t* in_idx,
                  int n_queries, int n_candidates, int k_out,
                  float* out_scores, int32_t* out_idx, void* stream);


/* ---- encoder (bi-encoder embedder / cross-encoder reranker) --------------------
 * Replaces the transformer forward the reference reaches through
 *   HuggingFaceEmbedding(...)       src/tensortruth/services/model_manager.py:254-260,
 *                                   src/tensortruth/indexing/builder.py:146-152
 *   SentenceTransformerRerank(...)  src/tensortruth/services/model_manager.py:333-337
 * (upstream: sentence-transformers over transformers XLMRobertaModel / BertModel /
 * XLMRobertaForSequenceClassification).
 *
 * Token layout: sequences are PACKED (no padding tokens are computed): token rows
 * [seq_start[b], seq_start[b] + seq_len[b]) belong to sequence b (any start row, sequences may
 * follow each other without a gap: the attention kernels mask the 8-row token groups two
 * sequences share), and the row count n_rows (>= last start + len) is a multiple of 128 -- or 64 / 192:
 * up to 256 rows (one query, a handful of short texts) the projections run as weight-streaming skinny GEMMs;
 * rows that belong to no sequence are computed but never read.
 * All weights are bf16 [out][in] (nn.Linear layout), biases / LayerNorm parameters fp32.
 * The structs below hold DEVICE pointers but live in HOST memory.
 */
typedef struct tt_layer_weights {
    const void* qkv_w;   /* [3H][H]  query, key, value rows concatenated */
    const float* qkv_b;  /* [3H] */
    const void* o_w;     /* [H][H]   attention.output.dense */
    const float* o_b;
    const float* ln1_g;  /* attention.output.LayerNorm */
    const float* ln1_b;
    const void* ffn1_w;  /* [F][H]   intermediate.dense (GELU-erf) */
    const float* ffn1_b;
    const void* ffn2_w;  /* [H][F]   output.dense */
    const float* ffn2_b;
    const float* ln2_g;  /* output.LayerNorm */
    const float* ln2_b;
    /* Optional fp8 (OCP e4m3) copies of the two projections whose input is a LayerNorm output, with one
     * fp32 scale per output row: w ~= w8 * wscale[out].  When every layer has them (and hidden, ffn and
     * n_rows are multiples of 256) the forward runs those GEMMs on the fp8 matrix cores, quantising the
     * LayerNorm outputs per token; NULL = bf16 (BASELINE.json config 5, "fp8 MFMA reranker"). */
    const void* qkv_w8;      /* [3H][H] bytes */
    const float* qkv_wscale; /* [3H] */
    const void* ffn1_w8;     /* [F][H] bytes */
    const float* ffn1_wscale;/* [F] */
    /* ... and of the other two: the attention output (its input, the attention context, is quantised per
     * token by a row pass) and the FFN output projection, whose input -- the GELU output, rows of F values
     * spread over F/256 tiles -- is written as e4m3 by the FFN-up epilogue with ONE static scale per layer,
     * ffn_act_scale (> 0; from a calibration forward: tt_encoder_weights.ffn_absmax_out). */
    const void* o_w8;        /* [H][H] bytes */
    const float* o_wscale;   /* [H] */
    const void* ffn2_w8;     /* [H][F] bytes */
    const float* ffn2_wscale;/* [H] */
    float ffn_act_scale;     /* 0 = FFN output projection stays bf16 */
} tt_layer_weights;

typedef struct tt_encoder_weights {
    int32_t hidden, layers, heads, ffn, vocab, max_pos, type_vocab;
    float ln_eps;
    const void* word_emb;  /* [vocab][H] bf16 */
    const void* pos_emb;   /* [max_pos][H] bf16 */
    const void* type_emb;  /* [type_vocab][H] bf16 */
    const float* emb_ln_g;
    const float* emb_ln_b;
    const tt_layer_weights* layer; /* host array [layers] */
    const void* cls_dense_w;  /* [H][H] bf16 or NULL (no classification head) */
    const float* cls_dense_b;
    const void* cls_out_w;    /* [1][H] bf16 */
    const float* cls_out_b;   /* [1] */
    /* calibration hook: device float[layers] (zero it first); every forward that produces the bf16 FFN
     * intermediate raises entry l to max |GELU output| of layer l.  NULL = off. */
    float* ffn_absmax_out;
} tt_encoder_weights;

size_t tt_encoder_workspace_bytes(const tt_encoder_weights* w, int n_rows);

/* ids/pos/type_ids: [n_rows] int32 (type_ids may be NULL = all zero); hidden_out:
 * [n_rows][H] bf16 last hidden state.  max_len = longest sequence (host value). */
int tt_encoder_forward(const tt_encoder_weights* w, const int32_t* ids, const int32_t* pos,
                       const int32_t* type_ids, const int32_t* seq_start, const int32_t* seq_len,
                       int n_seq, int n_rows, int max_len, void* hidden_out,
                       void* workspace, size_t workspace_bytes, void* stream);

/* Same forward, but the LAST layer is evaluated for the CLS row of every sequence only (the one row the
 * pooling and the classification head read): cls_out is [round_up(n_seq, 256)][H] bf16 (rows beyond round_up(n_seq, 64)
 * are not written when n_seq <= 256), row b = final hidden state of token seq_start[b].  Identical arithmetic for those rows up to the attention kernel used for the
 * single query row (fp32 probabilities instead of bf16); saves 1/24 of the encoder work at 24 layers. */
size_t tt_encoder_cls_workspace_bytes(const tt_encoder_weights* w, int n_rows, int n_seq);
int tt_encoder_forward_cls(const tt_encoder_weights* w, const int32_t* ids, const int32_t* pos,
                           const int32_t* type_ids, const int32_t* seq_start, const int32_t* seq_len,
                           int n_seq, int n_rows, int max_len, void* cls_out,
                           void* workspace, size_t workspace_bytes, void* stream);

/* sentence-transformers Pooling(cls) + Normalize: out[b] = h[rows[b]] / ||h[rows[b]]||_2.
 * out_bf16 (optional) is the same vector rounded to bf16, ready to be a scan query. */
int tt_embed_pool(const void* hidden_bf16, int ld, const int32_t* rows, int n_seq, int hidden,
                  float* out_f32, void* out_bf16, void* stream);

/* sentence-transformers Pooling(mean) + Normalize for checkpoints whose 1_Pooling/config.json says so (e5, all-MiniLM, gte ...;
 * the reference loads any HuggingFace embedding model its config names, services/model_manager.py:188-272):
 * out[b] = mean over the sequence's rows [seq_start[b], seq_start[b] + seq_len[b]) of h, L2-normalised.  hidden: the FULL last
 * hidden state (tt_encoder_forward / _f32 / _x3), bf16 or fp32. */
int tt_embed_pool_mean(const void* hidden_bf16, int ld, const int32_t* seq_start, const int32_t* seq_len, int n_seq, int hidden,
                       float* out_f32, void* out_bf16, void* stream);
int tt_embed_pool_mean_f32(const float* hidden_f32, int ld, const int32_t* seq_start, const int32_t* seq_len, int n_seq, int hidden,
                           float* out_f32, void* out_bf16, void* stream);

/* XLMRobertaClassificationHead + CrossEncoder sigmoid on the CLS rows:
 * scores[b] = sigmoid(out_proj(tanh(dense(h[rows[b]])))) ; logits optional.
 * workspace >= 2 * round_up(n_seq,128) * H * 2 bytes. */
int tt_rerank_head(const tt_encoder_weights* w, const void* hidden_bf16, const int32_t* rows, int n_seq,
                   float* scores, float* logits, void* workspace, size_t workspace_bytes, void* stream);

/* Semantic splitter distances (reference: SemanticSplitterNodeParser built at
 * src/tensortruth/indexing/builder.py:393-407): out_dist[i] = 1 - cos(e[i], e[i+1]) for the
 * n consecutive sentence-group embeddings e (fp32 [n][hidden], e.g. tt_embed_pool's output). */
int tt_adjacent_cosine(const float* emb_f32, int n, int hidden, float* out_dist, void* stream);

/* Building blocks, exported for the parity tests (same kernels the forward uses).
 * tt_attention_varlen: Q and K are row-major [rows][ld_qk] at column offsets q_col0 / k_col0;
 * V is passed in the token-blocked transposed layout the QKV GEMM epilogue writes,
 * vt[(row / 8) * ldvt + feature * 8 + row % 8] with ldvt = 8 * heads * head_dim. */
/* tt_gemm_bf16: m, n multiples of 128 and k of 64 (tiled kernels), or m a multiple of 64 up to 256 with n % 16 == 0 and
 * k % 32 == 0 (weight-streaming skinny kernel; same bits as the tiled kernels for the same rows). */
int tt_gemm_bf16(const void* a, const void* w, const float* bias, const void* residual, void* c,
                 int m, int n, int k, int epilogue /*0 bias,1 gelu,2 +residual,3 tanh*/, void* stream);
int tt_layernorm_bf16(const void* in, void* out, const float* gamma, const float* beta, int rows, int hidden,
                      float eps, void* stream);
int tt_attention_varlen(const void* qk, int ld_qk, int q_col0, int k_col0, const void* vt, int ldvt, void* out,
                        int ld_out, const int32_t* seq_start, const int32_t* seq_len, int n_seq, int heads,
                        int head_dim, int max_len, void* stream);

/* fp8 building blocks (same kernels the fp8 forward uses).
 * tt_quantize_rows_fp8: q[r][c] = e4m3(x[r][c] * 448 / absmax_r), scale[r] = absmax_r / 448 (1 for a zero row).
 * tt_layernorm_bf16_fp8: tt_layernorm_bf16 that also emits that quantisation of its bf16 output.
 * tt_gemm_fp8: c = epi((a8 . w8^T) * a_scale[m] * w_scale[n] + bias), m, n, k multiples of 256, epilogue 0 / 1. */
int tt_quantize_rows_fp8(const void* in_bf16, int rows, int cols, void* out_fp8, float* out_scale, void* stream);
int tt_layernorm_bf16_fp8(const void* in, void* out, const float* gamma, const float* beta, int rows, int hidden,
                          float eps, void* out_fp8, float* out_scale, void* stream);
int tt_gemm_fp8(const void* a8, const float* a_scale, const void* w8, const float* w_scale, const float* bias, void* c,
                int m, int n, int k, int epilogue /*0 bias, 1 gelu*/, void* stream);
/* ... with the residual epilogue (epilogue 2: + residual[m][n], bf16) or, epilogues 0 / 1, an e4m3 result instead of
 * the bf16 one: c_fp8[m][n] = e4m3(bf16(result) * c_fp8_inv_scale) (c_bf16 is then not written and may be NULL). */
int tt_gemm_fp8_ex(const void* a8, const float* a_scale, const void* w8, const float* w_scale, const float* bias,
                   const void* residual, void* c_bf16, void* c_fp8, float c_fp8_inv_scale, int m, int n, int k,
                   int epilogue, void* stream);

/* ---- reference-precision (fp32) forward -------------------------------------------------------------------
 * The reference's default embedder / reranker dtype is fp32 (src/tensortruth/app_utils/config_schema.py:66-76:
 * torch_dtype None; services/model_manager.py:218-229 passes torch_dtype only when configured).  These entry points
 * run the same encoder with fp32 weights, fp32 activations and fp32 MFMA (v_mfma_f32_32x32x2_f32, exact f32
 * arithmetic at 1/16 of the bf16 matrix rate) for callers that ask for model_kwargs={"torch_dtype": "float32"}:
 * scores within 1e-3 relative of the CPU reference, at interactive batch sizes (one query's candidate pairs).
 * Same packed token layout as tt_encoder_forward; n_rows is any count >= last start + len.
 * All weight tensors fp32, [out][in] for matrices. */
typedef struct tt_layer_weights_f32 {
    const float* qkv_w;  /* [3H][H] */
    const float* qkv_b;
    const float* o_w;    /* [H][H] */
    const float* o_b;
    const float* ln1_g;
    const float* ln1_b;
    const float* ffn1_w; /* [F][H] */
    const float* ffn1_b;
    const float* ffn2_w; /* [H][F] */
    const float* ffn2_b;
    const float* ln2_g;
    const float* ln2_b;
} tt_layer_weights_f32;

typedef struct tt_encoder_weights_f32 {
    int32_t hidden, layers, heads, ffn, vocab, max_pos, type_vocab;
    float ln_eps;
    const float* word_emb;  /* [vocab][H] */
    const float* pos_emb;   /* [max_pos][H] */
    const float* type_emb;  /* [type_vocab][H] */
    const float* emb_ln_g;
    const float* emb_ln_b;
    const tt_layer_weights_f32* layer; /* host array [layers] */
    const float* cls_dense_w;  /* [H][H] or NULL */
    const float* cls_dense_b;
    const float* cls_out_w;    /* [1][H] */
    const float* cls_out_b;    /* [1] */
} tt_encoder_weights_f32;

size_t tt_encoder_f32_workspace_bytes(const tt_encoder_weights_f32* w, int n_rows);
/* hidden_out: [n_rows][H] fp32 last hidden state */
int tt_encoder_forward_f32(const tt_encoder_weights_f32* w, const int32_t* ids, const int32_t* pos,
                           const int32_t* type_ids, const int32_t* seq_start, const int32_t* seq_len,
                           int n_seq, int n_rows, int max_len, float* hidden_out,
                           void* workspace, size_t workspace_bytes, void* stream);
/* tt_embed_pool / tt_rerank_head on an fp32 hidden state (head workspace >= 2 * round_up(n_seq,128) * H * 4 bytes) */
int tt_embed_pool_f32(const float* hidden_f32, int ld, const int32_t* rows, int n_seq, int hidden,
                      float* out_f32, void* out_bf16, void* stream);
int tt_rerank_head_f32(const tt_encoder_weights_f32* w, const float* hidden_f32, const int32_t* rows, int n_seq,
                       float* scores, float* logits, void* workspace, size_t workspace_bytes, void* stream);
/* building block (parity tests): c = epi(a . w^T + bias), fp32; any m, n % 128 == 0, k % 32 == 0;
 * epilogue 0 bias, 1 exact-erf GELU, 2 + residual, 3 tanh */
int tt_gemm_f32(const float* a, const float* w, const float* bias, const float* residual, float* c,
                int m, int n, int k, int epilogue, void* stream);

/* ---- fp16 compute mode (round 3): the same forward with IEEE fp16 activations / weights and v_mfma_*_f16 ---------------------
 * Same weight struct (the matrices then point at fp16 data: word / position / type tables, all projections, the head), same
 * workspace sizes, same argument meaning as the functions they twin; hidden states and V^T are fp16.  Values beyond +-65504
 * saturate at the GEMM / LayerNorm outputs instead of becoming infinite.  No fp8 projections, no split planes in this mode.
 * (FlagEmbedding loads these very models with use_fp16=True by default; the reference's torch_dtype: "float16",
 * app_utils/config_schema.py:66-76, lands here instead of being mapped to bf16.) */
size_t tt_encoder_workspace_bytes_f16(const tt_encoder_weights* w, int n_rows);
size_t tt_encoder_cls_workspace_bytes_f16(const tt_encoder_weights* w, int n_rows, int n_seq);
int tt_encoder_forward_f16(const tt_encoder_weights* w, const int32_t* ids, const int32_t* pos,
                           const int32_t* type_ids, const int32_t* seq_start, const int32_t* seq_len,
                           int n_seq, int n_rows, int max_len, void* hidden_out,
                           void* workspace, size_t workspace_bytes, void* stream);
int tt_encoder_forward_cls_f16(const tt_encoder_weights* w, const int32_t* ids, const int32_t* pos,
                               const int32_t* type_ids, const int32_t* seq_start, const int32_t* seq_len,
                               int n_seq, int n_rows, int max_len, void* cls_out,
                               void* workspace, size_t workspace_bytes, void* stream);
int tt_embed_pool_f16(const void* hidden_f16, int ld, const int32_t* rows, int n_seq, int hidden, float* out_f32,
                      void* out_f16, void* stream);
int tt_embed_pool_mean_f16(const void* hidden_f16, int ld, const int32_t* seq_start, const int32_t* seq_len, int n_seq, int hidden,
                           float* out_f32, void* out_f16, void* stream);
int tt_rerank_head_f16(const tt_encoder_weights* w, const void* hidden_f16, const int32_t* rows, int n_seq,
                       float* scores, float* logits, void* workspace, size_t workspace_bytes, void* stream);
/* building blocks (parity tests) */
int tt_gemm_f16(const void* a, const void* w, const float* bias, const void* residual, void* c,
                int m, int n, int k, int epilogue, void* stream);
int tt_layernorm_f16(const void* in, void* out, const float* gamma, const float* beta, int rows, int hidden,
                     float eps, void* stream);
int tt_attention_varlen_f16(const void* qk, int ld_qk, int q_col0, int k_col0, const void* vt, int ldvt, void* out,
                            int ld_out, const int32_t* seq_start, const int32_t* seq_len, int n_seq, int heads,
                            int head_dim, int max_len, void* stream);

/* ---- reference precision on the bf16 matrix cores: split-bf16 ("bf16x3") forward -------------------------------------
 * Same contract as the fp32 forward above (the unchanged reference call SentenceTransformerRerank(model=, top_n=, device=),
 * src/tensortruth/services/model_manager.py:333-337, and HuggingFaceEmbedding with torch_dtype None,
 * app_utils/config_schema.py:66-76, are fp32), at a third of the bf16 path's rate instead of a sixteenth: every matrix
 * product runs on v_mfma_*_bf16 with both operands split into two bf16 planes (x = hi + lo, hi = bf16(x),
 * lo = bf16(x - hi); a.b ~= a_hi.b_hi + a_hi.b_lo + a_lo.b_hi, fp32 accumulate), the residual stream, LayerNorm, softmax
 * and the exact-erf GELU stay fp32.  Scores within 1e-3 relative of the CPU reference (measured ~1e-5).
 * "Planes" = bf16 [rows][2 W]: columns [0, W) hold hi, [W, 2 W) hold lo.  Weight matrices are planes [out][2 in];
 * embedding tables, biases, LayerNorm parameters and the classification head are fp32.  hidden a multiple of 256 (<= 1024)
 * with head_dim 64, ffn a multiple of 256, n_rows a multiple of 256. */
typedef struct tt_layer_weights_x3 {
    const void* qkv_w;   /* planes [3H][2H] */
    const float* qkv_b;
    const void* o_w;     /* planes [H][2H] */
    const float* o_b;
    const float* ln1_g;
    const float* ln1_b;
    const void* ffn1_w;  /* planes [F][2H] */
    const float* ffn1_b;
    const void* ffn2_w;  /* planes [H][2F] */
    const float* ffn2_b;
    const float* ln2_g;
    const float* ln2_b;
} tt_layer_weights_x3;

typedef struct tt_encoder_weights_x3 {
    int32_t hidden, layers, heads, ffn, vocab, max_pos, type_vocab;
    float ln_eps;
    const float* word_emb;  /* [vocab][H] fp32 */
    const float* pos_emb;
    const float* type_emb;
    const float* emb_ln_g;
    const float* emb_ln_b;
    const tt_layer_weights_x3* layer; /* host array [layers] */
    const float* cls_dense_w;  /* [H][H] fp32 or NULL */
    const float* cls_dense_b;
    const float* cls_out_w;
    const float* cls_out_b;
} tt_encoder_weights_x3;

size_t tt_encoder_x3_workspace_bytes(const tt_encoder_weights_x3* w, int n_rows);
/* hidden_out: [n_rows][H] fp32 last hidden state (pool it with tt_embed_pool_f32) */
int tt_encoder_forward_x3(const tt_encoder_weights_x3* w, const int32_t* ids, const int32_t* pos,
                          const int32_t* type_ids, const int32_t* seq_start, const int32_t* seq_len,
                          int n_seq, int n_rows, int max_len, float* hidden_out,
                          void* workspace, size_t workspace_bytes, void* stream);
/* classification head on the fp32 hidden state (workspace as tt_rerank_head_f32) */
/* Same forward with the LAST layer evaluated for the first row (CLS) of every sequence only -- the split-bf16 twin of
 * tt_encoder_forward_cls: cls_out [pad(n_seq)][hidden] fp32, pad = n_seq rounded up to 64 (<= 256 sequences) or to 256. */
size_t tt_encoder_x3_cls_workspace_bytes(const tt_encoder_weights_x3* w, int n_rows, int n_seq);
int tt_encoder_forward_x3_cls(const tt_encoder_weights_x3* w, const int32_t* ids, const int32_t* pos, const int32_t* type_ids,
                              const int32_t* seq_start, const int32_t* seq_len, int n_seq, int n_rows, int max_len,
                              float* cls_out, void* workspace, size_t workspace_bytes, void* stream);
int tt_rerank_head_x3(const tt_encoder_weights_x3* w, const float* hidden_f32, const int32_t* rows, int n_seq,
                      float* scores, float* logits, void* workspace, size_t workspace_bytes, void* stream);
/* building blocks (parity tests).  tt_split_planes: fp32 [rows][cols] -> planes [rows][2 cols], cols % 4 == 0.
 * tt_gemm_x3: a planes [m][2k], w planes [n][2k]; m, n multiples of 256, k of 64; epilogue 0 bias / 1 exact-erf GELU ->
 * c_planes [m][2n]; epilogue 2 -> c_f32 [m][n] = a.w^T + bias + residual_f32 [m][n].
 * tt_attention_x3: Q / K planes in one buffer (hi at q_col0 / k_col0 + head * 64, lo lo_off columns further), V in the
 * token-blocked transposed layout of tt_attention_varlen, once per plane; context planes out (hi at head * 64, lo at
 * out_lo_off + head * 64). */
int tt_split_planes(const float* in_f32, int64_t rows, int cols, void* out_planes, void* stream);
int tt_gemm_x3(const void* a_planes, const void* w_planes, const float* bias, const float* residual_f32, void* c_planes,
               float* c_f32, int m, int n, int k, int epilogue, void* stream);
int tt_attention_x3(const void* qk_planes, int ld_qk, int q_col0, int k_col0, int lo_off, const void* vt_hi,
                    const void* vt_lo, int ldvt, void* out_planes, int ld_out, int out_lo_off,
                    const int32_t* seq_start, const int32_t* seq_len, int n_seq, int heads, int max_len, void* stream);

/* ---- reference precision on TWO matrix-time units: the "f16c" forward (csrc/f16c_path.hip, round 4) --------------------
 * What the reference's unchanged calls compute -- fp32 semantics (services/model_manager.py:333-337 passes no dtype;
 * app_utils/config_schema.py:66-76: torch_dtype None) -- at half the bf16 matrix rate instead of split-bf16's third: every
 * GEMM operand is carried as "c-planes", hi = fp16(x) plus two OCP e4m3 planes (x and x - hi) with one E8M0 block exponent
 * per 32 elements, and a product runs as  hi.hi (fp16 MFMA) + e4m3(a).e4m3(w_lo) + e4m3(a_lo).e4m3(w)  (block-scaled MFMA at
 * twice the rate; the cross terms are 2^-12 of the result).  Attention on single fp16 products, fp32 softmax; residual
 * stream, LayerNorm, exact-erf GELU and the head in fp32.  Scores within 1e-3 relative of the CPU path (measured 1e-4).
 * Weights: matrices as c-planes in the weight flavour [out][hi: 2 in | lo8: in | x8: in] bytes + tiled scales, both made by
 * tt_f16c_quantize(weight = 1) from the fp32 tensor; tables, biases, LayerNorm parameters and the head fp32.
 * hidden a multiple of 256 (<= 1024) with head_dim 64, ffn a multiple of 256, n_rows a multiple of 256. */
typedef struct tt_layer_weights_f16c {
    const void* qkv_w;   /* c-planes [3H][4H bytes] */
    const void* qkv_s;   /* tiled scales, tt_f16c_scale_bytes(3H, H, 1) bytes */
    const float* qkv_b;
    const void* o_w;     /* [H][4H bytes] */
    const void* o_s;
    const float* o_b;
    const float* ln1_g;
    const float* ln1_b;
    const void* ffn1_w;  /* [F][4H bytes] */
    const void* ffn1_s;
    const float* ffn1_b;
    const void* ffn2_w;  /* [H][4F bytes] */
    const void* ffn2_s;
    const float* ffn2_b;
    const float* ln2_g;
    const float* ln2_b;
} tt_layer_weights_f16c;

typedef struct tt_encoder_weights_f16c {
    int32_t hidden, layers, heads, ffn, vocab, max_pos, type_vocab;
    float ln_eps;
    const float* word_emb;  /* [vocab][H] fp32 */
    const float* pos_emb;
    const float* type_emb;
    const float* emb_ln_g;
    const float* emb_ln_b;
    const tt_layer_weights_f16c* layer; /* host array [layers] */
    const float* cls_dense_w;  /* [H][H] fp32 or NULL */
    const float* cls_dense_b;
    const float* cls_out_w;
    const float* cls_out_b;
} tt_encoder_weights_f16c;

size_t tt_encoder_f16c_workspace_bytes(const tt_encoder_weights_f16c* w, int n_rows);
/* hidden_out: [n_rows][H] fp32 last hidden state (pool it with tt_embed_pool_f32 / tt_embed_pool_mean_f32) */
int tt_encoder_forward_f16c(const tt_encoder_weights_f16c* w, const int32_t* ids, const int32_t* pos,
                            const int32_t* type_ids, const int32_t* seq_start, const int32_t* seq_len,
                            int n_seq, int n_rows, int max_len, float* hidden_out,
                            void* workspace, size_t workspace_bytes, void* stream);
/* the LAST layer for the first row (CLS) of every sequence only: cls_out [pad(n_seq)][hidden] fp32, pad = n_seq rounded up to 256 */
size_t tt_encoder_f16c_cls_workspace_bytes(const tt_encoder_weights_f16c* w, int n_rows, int n_seq);
int tt_encoder_forward_f16c_cls(const tt_encoder_weights_f16c* w, const int32_t* ids, const int32_t* pos, const int32_t* type_ids,
                                const int32_t* seq_start, const int32_t* seq_len, int n_seq, int n_rows, int max_len,
                                float* cls_out, void* workspace, size_t workspace_bytes, void* stream);
int tt_rerank_head_f16c(const tt_encoder_weights_f16c* w, const float* hidden_f32, const int32_t* rows, int n_seq,
                        float* scores, float* logits, void* workspace, size_t workspace_bytes, void* stream);
/* building blocks (weight preparation, parity tests).
 * tt_f16c_quantize: fp32 [rows][k] (k a multiple of 256) -> c-planes [rows][4 k bytes] + tiled scales
 *   (tt_f16c_scale_bytes(rows, k, weight) bytes); weight = 0: activation flavour [hi | x8 | lo8], 1: weight flavour
 *   [hi | lo8 | x8] with two scale parts.  Rows are padded to 256 in the scale array only.
 * tt_gemm_f16c: a c-planes [m][4k], w c-planes [n][4k]; m, n, k multiples of 256; epilogue 0: fp16 c_out [m][n] =
 *   a.w^T + bias; 1: exact-erf GELU -> c-planes c_out [m][4n] + c_scales; 2: fp32 c_out [m][n] = a.w^T + bias + residual_f32.
 * tt_attention_f16c: Q / K as two fp16 planes in one buffer (hi at q_col0 / k_col0 + head * 64, lo lo_off columns further:
 *   the score product runs on three fp16 products), V fp16 in the V8 layout of tt_attention_varlen, head_dim 64; context as c-planes. */
size_t tt_f16c_scale_bytes(int64_t rows, int k, int weight);
int tt_f16c_quantize(const float* in_f32, int64_t rows, int k, int weight, void* out_planes, void* out_scales, void* stream);
int tt_gemm_f16c(const void* a_planes, const void* a_scales, const void* w_planes, const void* w_scales, const float* bias,
                 const float* residual_f32, void* c_out, void* c_scales, int m, int n, int k, int epilogue, void* stream);
int tt_attention_f16c(const void* qk_f16, int ld_qk, int q_col0, int k_col0, int lo_off, const void* vt_f16, int ldvt, void* out_planes,
                      void* out_scales, const int32_t* seq_start, const int32_t* seq_len, int n_seq, int heads, int max_len,
                      void* stream);

/* ---- "f16x3": the split-plane forward above with fp16 planes (csrc/x3_path.hip compiled a second time, round 4) -------
 * Same entry points, same layouts, `_f16` suffix: a value is carried as hi = fp16(x), lo = fp16(x - hi) -- 22 significand
 * bits per operand instead of the two bf16 planes' 16 -- and a product as three v_mfma_*_f16 products.  It is the DEFAULT
 * implementation of the reference precision: on weights with trained-model statistics (attention logits of ~100, scores
 * down to 0.005: tests/stress_weights.py) it holds 1e-3 relative with a margin where bf16x3 sits AT the bar and the
 * two-unit f16c path is 7x outside.  Values beyond +-65504 saturate in the hi plane (the lo plane takes what is left). */
size_t tt_encoder_x3_workspace_bytes_f16(const tt_encoder_weights_x3* w, int n_rows);
int tt_encoder_forward_x3_f16(const tt_encoder_weights_x3* w, const int32_t* ids, const int32_t* pos,
                              const int32_t* type_ids, const int32_t* seq_start, const int32_t* seq_len,
                              int n_seq, int n_rows, int max_len, float* hidden_out,
                              void* workspace, size_t workspace_bytes, void* stream);
size_t tt_encoder_x3_cls_workspace_bytes_f16(const tt_encoder_weights_x3* w, int n_rows, int n_seq);
int tt_encoder_forward_x3_cls_f16(const tt_encoder_weights_x3* w, const int32_t* ids, const int32_t* pos, const int32_t* type_ids,
                                  const int32_t* seq_start, const int32_t* seq_len, int n_seq, int n_rows, int max_len,
                                  float* cls_out, void* workspace, size_t workspace_bytes, void* stream);
int tt_rerank_head_x3_f16(const tt_encoder_weights_x3* w, const float* hidden_f32, const int32_t* rows, int n_seq,
                          float* scores, float* logits, void* workspace, size_t workspace_bytes, void* stream);
int tt_split_planes_f16(const float* in_f32, int64_t rows, int cols, void* out_planes, void* stream);
int tt_gemm_x3_f16(const void* a_planes, const void* w_planes, const float* bias, const float* residual_f32, void* c_planes,
                   float* c_f32, int m, int n, int k, int epilogue, void* stream);
int tt_attention_x3_f16(const void* qk_planes, int ld_qk, int q_col0, int k_col0, int lo_off, const void* vt_hi,
                        const void* vt_lo, int ldvt, void* out_planes, int ld_out, int out_lo_off,
                        const int32_t* seq_start, const int32_t* seq_len, int n_seq, int heads, int max_len, void* stream);

/* Per-kernel device timing (HIP events on the launch stream), for bench.py's roofline leg.
 * tt_prof_enable(1) (or a mask of 1 << id, to time only some kernels) starts recording one event pair per launch of the tracked kernels on the
 * calling thread; tt_prof_read() synchronises those events and returns total milliseconds
 * and launch count for kernel id `which` since the last enable, then keeps recording.
 * ids: 1 scan filter pass, 2 scan sample pass, 3 top-k select, 4 gemm, 5 attention, 6 row ops,
 * 7 scan tail rows (the < 256 rows behind the tiled filter pass of a 65+ query batch) */
int tt_prof_enable(int on);
int tt_prof_read(int which, double* total_ms_host, int* launches_host);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif

#ifdef __cplusplus
}
#endif
#endif /* TT_HIP_H */
