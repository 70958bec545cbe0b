// Internal interfaces of the encoder kernels (gemm.hip, attention.hip, rowops.hip, encoder_api.hip).
#pragma once
#include "common.h"

enum { TT_EPI_BIAS = 0, TT_EPI_GELU = 1, TT_EPI_RESIDUAL = 2, TT_EPI_TANH = 3, TT_EPI_QKV = 4,
       TT_EPI_VT = 5 /* internal: whole output stored transposed into vt (the V third of a QKV projection) */,
       TT_EPI_SCAN = 6 /* internal: similarity scan -- A = corpus rows, W = 256 queries, nothing is stored: scores >= the
                          query's threshold (bias[n]) are appended to per-query candidate lists (scan_api.hip) */ };

struct GemmParams {
    const uint16_t* A;        // [M][lda] bf16
    const uint16_t* W;        // [N][K] bf16 (nn.Linear weight)
    const float* bias;        // [N] fp32
    const uint16_t* residual; // [M][ldr] bf16 (TT_EPI_RESIDUAL)
    uint16_t* C;              // [M][ldc] bf16
    uint16_t* vt;             // TT_EPI_QKV: columns >= vt_col0 go to the V8 buffer:
                              //   vt[(m / 8) * ldvt + (n - vt_col0) * 8 + m % 8],  ldvt = 8 * (N - vt_col0)
    int M, N, K, lda, ldc, ldr, ldvt, vt_col0;
    // fp8 (OCP e4m3) operands: A, W point to bytes, lda / K count elements = bytes; the result is
    // acc * a_scale[m] * w_scale[n] (+ bias ...).  256x256 kernels only (bias, GELU and V^T epilogues).
    const float* a_scale;     // [M] per-row dequantisation scale of A
    const float* w_scale;     // [N] per-output-channel scale of W
    int fp8;
    // fp8 OUTPUT (bias / GELU epilogues of the one-tile kernel): C8[m][ldc] = e4m3(bf16(result) * c8_inv_scale),
    // one static scale for the whole tensor (the FFN intermediate, whose rows span 16 tiles); C is not written
    uint8_t* C8;
    float c8_inv_scale;
    int nt_store;             // whole-line output stores issued as streaming stores (set by the launcher)
    int sn;                   // super-tile width in N-tiles (32 tiles per super-tile: sm = 32 / sn); set by the launcher
    int xp;                   // experiment switches of the persistent kernel (TT_GEMM_XP; see gemm.hip launch())
    // TT_EPI_SCAN: candidate lists of the filter pass (same layout as ScanParams' shared lists)
    int32_t* scan_cnt;        // [N] running candidate count per query (may exceed scan_cap)
    float* scan_scores;       // [N][scan_cap]
    int32_t* scan_idx;        // [N][scan_cap]
    int scan_cap;
    int32_t scan_idx_base;    // emitted index = scan_idx_base + A row
    float* scan_dense;        // SAMPLE form (round 4): not null -> no filtering; tile t covers A rows [t * 256 * scan_tile_stride, + 256) and
    int scan_dense_stride;    // writes the maximum of each of its eight 32-row groups per query: scan_dense[q * stride + 8 t + group]
    int scan_tile_stride;
    int scan_rows;            // rows of the shard (0 = M).  Not a multiple of 256: M = rows rounded up, and the LAST tile is the
                              // shard's last 256 rows (it overlaps its predecessor; the rows scored twice are reported once)
    // ---- split-bf16 ("bf16x3") operands: reference precision on the bf16 matrix cores (x3_path.hip) -------------------
    // A value x is carried as TWO bf16 planes, hi = bf16(x) and lo = bf16(x - hi) (x = hi + lo to 2^-17 relative), stored
    // side by side in a row: A[m][0..K) = hi, A[m][K..2K) = lo (lda >= 2K), W[n][0..K) = hi, W[n][K..2K) = lo (ldw = 2K).
    // The product runs as ONE contraction over a virtual K' = 3K: A-hi.W-hi + A-hi.W-lo + A-lo.W-hi (lo.lo, 2^-16 of the
    // result, is dropped), fp32 accumulate -- the bf16 main loop unchanged, K-tile t reading A tile (t < nk ? t : t - nk)
    // and W tile (t < 2nk ? t : t - 2nk).  256x256 kernels only.  Epilogues: TT_EPI_BIAS / TT_EPI_GELU (exact erf) write
    // planes, C[m][n] = hi, C[m][c_lo_off + n] = lo; TT_EPI_RESIDUAL writes fp32, C32[m][n] = acc + bias + res32[m][n];
    // TT_EPI_VT writes the V8 layout twice (vt = hi, vt_lo = lo).
    int x3;
    int ldw;                  // W row stride in elements (0 = K)
    int c_lo_off;             // planes output: column offset of the lo plane
    float* C32;               // fp32 output [M][ldc] (x3 residual epilogue)
    const float* res32;       // fp32 residual [M][ldr]
    const uint16_t* res_planes;   // (round 4) the residual as the two planes of the same activation instead: res = hi + lo, hi at
    int res_lo_off;               // res_planes[m * ldr + n], lo res_lo_off columns further -- the LayerNorm then writes no fp32 copy
    uint16_t* vt_lo;          // lo plane of the V8 output
    int x3_zero_lo;           // diagnostic (TT_X3_ROUND_MASK): planes outputs are written with lo = 0, i.e. rounded to bf16
    // ---- f16c operands: reference precision on TWO matrix-time units (f16c_path.hip; fp16 instantiation only) ---------------
    // A value x is carried as "c-planes": hi = fp16(x), x8 = e4m3(x 2^-s), lo8 = e4m3((x - hi) 2^-(s - 11)), with one E8M0 block
    // exponent s per 32 consecutive K elements (s = exponent of the block's absmax - 7).  A row of an operand is ONE byte stream
    //     A[m]: [ hi: 2K bytes | x8: K | lo8: K ]        W[n]: [ hi: 2K | lo8: K | x8: K ]        (lda = ldw = 2K uint16 = 4K bytes)
    // and the product runs as ONE contraction over 2 K / 64 K-tiles of 128 bytes (K a multiple of 256): K / 64 tiles of v_mfma_f32_16x16x32_f16
    // (hi.hi), then K / 64 tiles of v_mfma_scale_f32_16x16x128_f8f6f4 with the block scales (x8.w_lo8, then lo8.w_x8): the cross
    // terms are 2^-12 of the result and need 2^-4, the dropped lo.lo term is 2^-24.  Scales are stored TILED, 1 KiB per (256-row
    // block, 128-element K-tile), in the order the kernel reads them (f16c_path.hip: a_scale_at / w_scale_at).
    // Epilogues: BIAS -> plain fp16 C (Q, K), VT -> fp16 V8, GELU -> c-planes (C = hi plane base, ldc = 2N uint16, c_scales),
    // RESIDUAL -> fp32 C32 = acc + bias + res32.
    int xc;
    const uint8_t* a_scales;  // [M / 256][K / 128][1024]
    const uint8_t* w_scales;  // [N / 256][2 K / 128][1024]: the w_lo8 part's K / 128 chunks, then the w_x8 part's (exponents - 11)
    uint8_t* c_scales;        // GELU epilogue: the output's scales, tiled for a consumer with K = N
#if TT_DIAG
    // ---- LayerNorm-folding EXPERIMENT (round 6; diagnostic library only: tools/ln_fold_bench, profiles/r06_ln_folding_ab.log) ----
    // 1 = consumer epilogue (bias / GELU; the A operand is the RAW pre-LayerNorm sum, W carries gamma):
    //         out[m][n] = lnf_rows[m].rstd * acc + lnf_rows[m].nmr * lnf_c0[n] + bias[n]        (lnf_c0 = column sums of the folded W)
    // 2 = producer epilogue (residual): the residual tile is the RAW pre-LayerNorm sum y of the previous block and LayerNorm(y) is
    //     rebuilt on the fly, res = y * (rstd * gamma[n]) + (nmr * gamma[n] + beta[n])  (lnf_c0 = gamma, lnf_c1 = beta); the epilogue
    //     also emits, per output row and 64-column strip, (sum, sum of squares) of the bf16-rounded outputs -> lnf_part[m][N / 64][2]
    int lnf;
    const float* lnf_rows;    // [M][2]: rstd, -mu * rstd of the rows of the folded LayerNorm's input
    const float* lnf_c0;      // [N]
    const float* lnf_c1;      // [N]
    float* lnf_part;          // [M][N / 64][2]
#endif
};
// Filter pass of the similarity scan for 65..256 queries per pass as a 256x256x64-tiled MFMA contraction (gemm.hip):
// corpus [rows][dim] bf16 with rows a multiple of 256, queries256 [256][dim] bf16 (rows beyond the batch zero),
// thr256 [256] fp32 (+inf for rows beyond the batch).  ONE pass over the corpus whatever the batch size.
int tt_scan_gemm_launch(const uint16_t* corpus, int64_t rows, int dim, const uint16_t* queries256, const float* thr256,
                        int32_t* cnt, float* cand_scores, int32_t* cand_idx, int cap, int32_t idx_base, hipStream_t st);
// Threshold sample of the tiled scan on the same contraction: `tiles` row tiles of 256 rows, tile t at row t * 256 * tile_stride;
// dense[q * dense_stride + 8 t + g] = max over the tile's g-th 32-row group of q . row (NaN rows ignored).  One read of the sample
// rows for all 256 queries (the streaming sample kernel reads them once per 64-query tile).
int tt_scan_gemm_sample_launch(const uint16_t* corpus, int tiles, int tile_stride, int dim, const uint16_t* queries256, const float* thr256,
                               float* dense, int dense_stride, hipStream_t st);
int tt_gemm_launch(const GemmParams& p, int epilogue, hipStream_t st);
// M <= 256 (a multiple of 64) runs on the weight-streaming skinny kernel unless TT_GEMM_SKINNY=0 (A/B switch)
bool tt_gemm_skinny_enabled();

struct AttnParams {
    const uint16_t* qk;       // [T][ld_qk] bf16: Q at column q_col0 + h*dh, K at k_col0 + h*dh
    const uint16_t* vt;       // V8 layout [T/8][heads*dh][8] bf16: 8 consecutive tokens of one feature = 16 B;
                              // ldvt = elements per 8-token group = 8 * heads*dh
    uint16_t* out;            // [T][ld_out] bf16, context at column h*dh
    const int32_t* seq_start; // [B] first token row of each sequence (multiple of 8)
    const int32_t* seq_len;   // [B]
    int n_seq, heads, head_dim, max_len;
    int total_rows;           // token rows of the batch (0 = unknown): the launcher's wave-count choice wants the MEAN length too
    int ld_qk, q_col0, k_col0, ldvt, ld_out;
    float scale;              // 1/sqrt(head_dim)
    float lazy;               // set by the launcher: log2 slack of the running softmax reference
    int n_qt;                 // set by the launcher: query tiles per sequence (grid decode)
    int rotate;               // set by the launcher: tail tiles deal their live 32-row blocks to different waves per (sequence, head)
    unsigned long long* dbg;  // diagnostic build only (tools/att_stamps): s_memtime stamps of one workgroup; NULL otherwise
    const uint16_t* q_rows;   // CLS variant only: the query row of sequence b at q_rows + b * ld_q_rows (+ h*dh); NULL = row seq_start[b] of qk
    int ld_q_rows;
    // f16c (fp16 instantiation): the context goes out as c-planes (f16c.h) -- `out` = the hi plane of rows of 4 * out_width bytes
    // (ld_out = 2 * out_width), x8 / lo8 planes behind it, block scales tiled for a consumer with K = out_width.  NULL = fp16 out.
    uint8_t* out_scales;
    int out_width;            // heads * head_dim
    int qk_lo_off;            // CLS variant, f16c: Q / K are TWO fp16 planes, the lo plane qk_lo_off columns behind the hi one (0 = one plane)
};
int tt_attention_launch(const AttnParams& p, hipStream_t st);
// CLS-only variant: one query row (seq_start[b]) per sequence; out row index = sequence index
int tt_attention_cls_launch(const AttnParams& p, hipStream_t st);

struct EmbedParams {
    const int32_t* ids;       // [T]
    const int32_t* pos;       // [T]
    const int32_t* type;      // [T] or null (=> type 0)
    const uint16_t* word;     // [vocab][H] bf16
    const uint16_t* posemb;   // [max_pos][H] bf16
    const uint16_t* typeemb;  // [type_vocab][H] bf16
    const float* gamma;
    const float* beta;
    uint16_t* out;            // [T][H] bf16
    int T, H, vocab, max_pos, type_vocab;
    float eps;
    uint8_t* q8;              // optional: [T][H] e4m3 copy of out, per-row scale q8_scale[T] (fp8 GEMM operand)
    float* q8_scale;
};
int tt_embed_ln_launch(const EmbedParams& p, hipStream_t st);

// out = LayerNorm(in) * gamma + beta, rows of H bf16
int tt_layernorm_launch(const uint16_t* in, uint16_t* out, const float* gamma, const float* beta, int rows, int H,
                        float eps, hipStream_t st, uint8_t* q8 = nullptr, float* q8_scale = nullptr);
int tt_absmax_launch(const uint16_t* x, size_t n, float* out, hipStream_t st);
int tt_quantize_rows_launch(const uint16_t* in, int ld, int rows, int cols, uint8_t* q8, float* scale, hipStream_t st);

// out_f32[b] = x[row[b]] / max(||x[row[b]]||, 1e-12); optional bf16 copy
int tt_cls_pool_l2norm_launch(const uint16_t* hidden, int ld, const int32_t* rows, int n, int H, float* out_f32,
                              uint16_t* out_bf16, hipStream_t st);

// gather rows: dst[b][:] = src[rows[b]][:]  (bf16, H elements), rows beyond n zero-filled up to n_pad
int tt_mean_pool_l2norm_launch(const void* hidden, int is_f32, int ld, const int32_t* seq_start, const int32_t* seq_len, int n, int H,
                               float* out_f32, uint16_t* out_bf16, hipStream_t st);
int tt_gather_rows_launch(const uint16_t* src, int ld, const int32_t* rows, int n, int n_pad, int H, uint16_t* dst,
                          hipStream_t st);

// score[b] = sigmoid(dot(t[b][:], w) + bias)   (t bf16 [n][ld], w bf16 [H]); logits optional
int tt_head_out_sigmoid_launch(const uint16_t* t, int ld, const uint16_t* w, const float* bias, int n, int H,
                               float* scores, float* logits, hipStream_t st);

// dist[i] = 1 - cos(e[i], e[i+1]), e fp32 [n][H]
int tt_adjacent_cosine_launch(const float* e, int n, int H, float* dist, hipStream_t st);
