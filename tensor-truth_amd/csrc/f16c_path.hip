// Reference precision on TWO matrix-time units: the "f16c" (fp16 + e4m3 corrections) encoder forward, gfx950.  Round 4.
//
// The reference's default embedder / reranker dtype is fp32 (app_utils/config_schema.py:66-76: torch_dtype None;
// services/model_manager.py:218-229, 333-337 pass no dtype); north_star's score tolerance is 1e-3 relative.  x3_path.hip meets
// it with split-bf16 operands at three bf16 MFMA products per product.  This file meets it at TWO units: an operand value is
//       x = hi + lo,   hi = fp16(x)                                  (11-bit significand)
//       a.w ~= a_hi.w_hi                                             v_mfma_f32_16x16x32_f16, exact products, fp32 accumulate: 1 unit
//            + e4m3(a).e4m3(w_lo) + e4m3(a_lo).e4m3(w)               v_mfma_scale_f32_16x16x128_f8f6f4 with E8M0 block scales per
//                                                                    32 elements: twice the bf16 rate -> 1/2 unit each
// The two cross terms are 2^-12 of the result and only need e4m3's 2^-4; the dropped lo.lo term is 2^-24.  Attention runs on
// single fp16 products (Q, K, V, P rounded to fp16; fp32 scores, softmax and accumulators: attention.hip's fp16 instantiation).
// Everything that is not a product stays fp32, as in x3_path.hip: the residual stream, LayerNorm, exact-erf GELU, the head.
// CPU emulation of exactly this scheme before any kernel was written (tools/probes/f16c_emulation.py, full depth, the committed
// fp32 fixture): scores within 8.3e-5 relative of the fp32 oracle, Kendall tau 1.000.
//
// Tensors: GEMM A operands and weights are "c-planes" (f16c.h: [hi | x8 | lo8] rows of 4 K bytes + tiled E8M0 scales); Q / K /
// V are plain fp16 (V in the V8 layout); the residual stream is fp32.  Layer schedule (post-LN block):
//   qk (fp16), V8 (fp16)  = GEMMc(x_c, Wqkv)              two launches: bias epilogue, V^T epilogue
//   ctx_c                 = attention_f16(qk, V8)          c-planes written by the attention epilogue
//   y (fp32)              = GEMMc(ctx_c, Wo) + bo + x      fp32 residual read by the epilogue
//   x1 (fp32), x1_c       = LayerNorm(y)
//   f_c                   = GELU_erf(GEMMc(x1_c, W1) + b1) c-planes written by the GEMM epilogue
//   y (fp32)              = GEMMc(f_c, W2) + b2 + x1
//   x (fp32), x_c         = LayerNorm(y)
// Roofline: MFMA-bound like the bf16 path; per GEMM launch 2 M N K flops on fp16 operands + 2 x 2 M N K on e4m3 operands at
// twice the rate = 2 matrix-time units (split-bf16: 3).
// The whole file belongs to the fp16 instantiation (common.h TT_F16): the plain compilation of it is empty.
#include "common.h"
#include "encoder.h"
#include "f16c.h"

#if TT_F16

extern "C" int tt_rerank_head_f32(const tt_encoder_weights_f32* w, const float* hidden_f32, const int32_t* rows, int n_seq,
                                  float* scores, float* logits, void* workspace, size_t workspace_bytes, void* stream);

namespace {

constexpr int kRowThreadsC = 256;   // four rows per workgroup, one wave per row
constexpr int kMaxC4c = 4;          // H <= 1024 in the LayerNorm kernels

__device__ __forceinline__ float wave_sum_c(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Four consecutive values of a row (elements e0 .. e0 + 3, e0 = 4 lane (mod 256): a scale block = 8 consecutive lanes) -> the
// three planes + the block's scale byte.  weight = the W flavour: [hi | lo8 | x8], the lo8 plane carries its OWN block exponent
// (part 0 of the tiled weight scales), the x8 plane's is stored 11 lower (part 1): it meets the activations' lo8 plane, whose
// values sit 11 binades below their block scale.
__device__ __forceinline__ void store_c4(char* row_base, int W, int e0, float4 y, int lane, uint8_t* scales, int row, int nks, bool weight) {
    float amax = fmaxf(fmaxf(fabsf(y.x), fabsf(y.y)), fmaxf(fabsf(y.z), fabsf(y.w)));
    amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 4, 64));
    int sbyte, sh;
    xc_block_scale(amax, sbyte, sh);
    if (!weight) {
        uint2 hi;
        uint32_t x8, l8;
        xc_split4(y.x, y.y, y.z, y.w, sh, sh + 11, hi, x8, l8);
        *reinterpret_cast<uint2*>(row_base + (size_t)e0 * 2) = hi;
        *reinterpret_cast<uint32_t*>(row_base + (size_t)2 * W + e0) = x8;
        *reinterpret_cast<uint32_t*>(row_base + (size_t)3 * W + e0) = l8;
        if ((lane & 7) == 0) scales[xc_a_scale_at(row, e0 >> 5, nks)] = (uint8_t)sbyte;
        return;
    }
    uint2 hi;
    hi.x = pack_e2(y.x, y.y);
    hi.y = pack_e2(y.z, y.w);
    const float l0 = y.x - elo(hi.x), l1 = y.y - ehi(hi.x), l2 = y.z - elo(hi.y), l3 = y.w - ehi(hi.y);
    float lmax = fmaxf(fmaxf(fabsf(l0), fabsf(l1)), fmaxf(fabsf(l2), fabsf(l3)));
    lmax = fmaxf(lmax, __shfl_xor(lmax, 1, 64));
    lmax = fmaxf(lmax, __shfl_xor(lmax, 2, 64));
    lmax = fmaxf(lmax, __shfl_xor(lmax, 4, 64));
    int lbyte, lsh;
    xc_block_scale(lmax, lbyte, lsh);
    const bool tiny = sbyte < 11;                  // |w| < 2^-109: the block is zero for every purpose; keeps sbyte - 11 >= 0
    const uint32_t x8 = tiny ? 0u : xc_pack4(xc_sat(ldexpf(y.x, sh)), xc_sat(ldexpf(y.y, sh)), xc_sat(ldexpf(y.z, sh)), xc_sat(ldexpf(y.w, sh)));
    const uint32_t l8 = xc_pack4(xc_sat(ldexpf(l0, lsh)), xc_sat(ldexpf(l1, lsh)), xc_sat(ldexpf(l2, lsh)), xc_sat(ldexpf(l3, lsh)));
    *reinterpret_cast<uint2*>(row_base + (size_t)e0 * 2) = hi;
    *reinterpret_cast<uint32_t*>(row_base + (size_t)2 * W + e0) = l8;
    *reinterpret_cast<uint32_t*>(row_base + (size_t)3 * W + e0) = x8;
    if ((lane & 7) == 0) {
        scales[xc_w_scale_at(row, 0, e0 >> 5, nks)] = (uint8_t)lbyte;
        scales[xc_w_scale_at(row, 1, e0 >> 5, nks)] = (uint8_t)(tiny ? 0 : sbyte - 11);
    }
}

// LayerNorm of a row held as x[c] (float4 = elements 256 c + 4 lane ...), two-pass fp32 statistics; writes the fp32 row (out32,
// may be NULL) and its c-planes (cbase = the row's first byte, may be NULL)
__device__ __forceinline__ void ln_row_c(float4 (&x)[kMaxC4c], int nc, int H, const float* gamma, const float* beta, float eps, float* out32,
                                         char* cbase, uint8_t* scales, int row, int lane) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < kMaxC4c; ++c)
        if (c < nc) s += (x[c].x + x[c].y) + (x[c].z + x[c].w);
    const float mean = wave_sum_c(s) / (float)H;
    float v = 0.f;
#pragma unroll
    for (int c = 0; c < kMaxC4c; ++c)
        if (c < nc) {
            const float a = x[c].x - mean, b = x[c].y - mean, d = x[c].z - mean, e = x[c].w - mean;
            v += (a * a + b * b) + (d * d + e * e);
        }
    const float rstd = 1.0f / sqrtf(wave_sum_c(v) / (float)H + eps);
#pragma unroll
    for (int c = 0; c < kMaxC4c; ++c)
        if (c < nc) {
            const int e0 = 256 * c + 4 * lane;
            const float4 g = *reinterpret_cast<const float4*>(gamma + e0);
            const float4 b = *reinterpret_cast<const float4*>(beta + e0);
            float4 y = float4{(x[c].x - mean) * rstd * g.x + b.x, (x[c].y - mean) * rstd * g.y + b.y,
                              (x[c].z - mean) * rstd * g.z + b.z, (x[c].w - mean) * rstd * g.w + b.w};
            asm("" : "+v"(y.x), "+v"(y.y), "+v"(y.z), "+v"(y.w));     // opaque before the split (see gemm.hip epilogue_x3)
            if (out32) *reinterpret_cast<float4*>(out32 + e0) = y;
            if (cbase) store_c4(cbase, H, e0, y, lane, scales, row, H >> 7, false);
        }
}

__global__ __launch_bounds__(kRowThreadsC) void embed_ln_c_kernel(const int32_t* ids, const int32_t* pos, const int32_t* type,
                                                                   const float* word, const float* posemb, const float* typeemb,
                                                                   const float* gamma, const float* beta, float* out32, char* planes,
                                                                   uint8_t* scales, int T, int H, int vocab, int max_pos, int type_vocab,
                                                                   float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= T) return;
    int id = ids[row], p = pos[row], t = type ? type[row] : 0;
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    p = p < 0 ? 0 : (p >= max_pos ? max_pos - 1 : p);
    t = t < 0 ? 0 : (t >= type_vocab ? type_vocab - 1 : t);
    const int nc = H / 256;
    float4 x[kMaxC4c];
#pragma unroll
    for (int c = 0; c < kMaxC4c; ++c)
        if (c < nc) {
            const int e0 = 256 * c + 4 * lane;
            const float4 a = *reinterpret_cast<const float4*>(word + (size_t)id * H + e0);
            const float4 b = *reinterpret_cast<const float4*>(posemb + (size_t)p * H + e0);
            const float4 d = *reinterpret_cast<const float4*>(typeemb + (size_t)t * H + e0);
            x[c] = float4{a.x + b.x + d.x, a.y + b.y + d.y, a.z + b.z + d.z, a.w + b.w + d.w};
        }
    ln_row_c(x, nc, H, gamma, beta, eps, out32 + (size_t)row * H, planes + (size_t)row * 4 * H, scales, row, lane);
}

__global__ __launch_bounds__(kRowThreadsC) void layernorm_c_kernel(const float* in, float* out32, char* planes, uint8_t* scales,
                                                                    const float* gamma, const float* beta, int rows, int H, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int nc = H / 256;
    float4 x[kMaxC4c];
#pragma unroll
    for (int c = 0; c < kMaxC4c; ++c)
        if (c < nc) x[c] = *reinterpret_cast<const float4*>(in + (size_t)row * H + 256 * c + 4 * lane);
    ln_row_c(x, nc, H, gamma, beta, eps, out32 ? out32 + (size_t)row * H : nullptr, planes ? planes + (size_t)row * 4 * H : nullptr, scales,
             row, lane);
}

// fp32 [rows][K] -> c-planes [rows][4 K bytes] + tiled scales (weights on load; tests).  One wave per row, K a multiple of 256.
__global__ __launch_bounds__(kRowThreadsC) void quantize_c_kernel(const float* in, char* planes, uint8_t* scales, int rows, int K, int weight) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    for (int c = 0; c < K / 256; ++c) {
        const int e0 = 256 * c + 4 * lane;
        const float4 y = *reinterpret_cast<const float4*>(in + (size_t)row * K + e0);
        store_c4(planes + (size_t)row * 4 * K, K, e0, y, lane, scales, row, K >> 7, weight != 0);
    }
}

// rows seq_start[b] of an fp32 [T][H] matrix -> dst [n_pad][H] (rows beyond n: zeros)
__global__ __launch_bounds__(256) void gather_rows_f32c_kernel(const float* src, const int32_t* rows, int n, int n_pad, int H, float* dst) {
    const int b = blockIdx.x;
    if (b >= n_pad) return;
    for (int c = threadIdx.x * 4; c < H; c += 256 * 4) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (b < n) v = *reinterpret_cast<const float4*>(src + (size_t)rows[b] * H + c);
        *reinterpret_cast<float4*>(dst + (size_t)b * H + c) = v;
    }
}

inline dim3 row_grid_c(int rows) { return dim3((unsigned)((rows + 3) / 4)); }
inline size_t scale_bytes(size_t rows256, size_t K) { return rows256 / 256 * (K / 128) * 1024; }      // activation scales
inline int cls_pad_c(int n_seq) { return (n_seq + 255) / 256 * 256; }

// ---- forward ---------------------------------------------------------------------------------------------------------
struct XcWs {
    size_t off_xa, off_xb, off_y, off_xc, off_xs, off_qk, off_vt, off_ctx, off_cs, off_ffn, off_fs, total;
    size_t off_cctx, off_ccs, off_cx, off_cy, off_cx1, off_cxc, off_cxs, off_cffn, off_cfs;     // CLS tail (n_cls > 0)
};

XcWs xc_plan(const tt_encoder_weights_f16c* w, int n_rows, int n_cls = 0) {
    XcWs e{};
    const size_t H = (size_t)w->hidden, F = (size_t)w->ffn, T = ((size_t)n_rows + 255) / 256 * 256;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += tt_align_up(bytes, 256); return o; };
    e.off_xa = take(T * H * 4);
    e.off_xb = take(T * H * 4);
    e.off_y = take(T * H * 4);
    e.off_xc = take(T * 4 * H);
    e.off_xs = take(scale_bytes(T, H));
    e.off_qk = take(T * 2 * H * 2);
    e.off_vt = take(T * H * 2);
    e.off_ctx = take(T * 4 * H);
    e.off_cs = take(scale_bytes(T, H));
    e.off_ffn = take(T * 4 * F);
    e.off_fs = take(scale_bytes(T, F));
    if (n_cls > 0) {
        const size_t B = (size_t)cls_pad_c(n_cls);
        e.off_cctx = take(B * 4 * H);
        e.off_ccs = take(scale_bytes(B, H));
        e.off_cx = take(B * H * 4);
        e.off_cy = take(B * H * 4);
        e.off_cx1 = take(B * H * 4);
        e.off_cxc = take(B * 4 * H);
        e.off_cxs = take(scale_bytes(B, H));
        e.off_cffn = take(B * 4 * F);
        e.off_cfs = take(scale_bytes(B, F));
    }
    e.total = off;
    return e;
}

int check_weights_c(const tt_encoder_weights_f16c* w) {
    TT_CHECK_ARG(w != nullptr, "null weights");
    TT_CHECK_ARG(w->hidden > 0 && w->hidden % 256 == 0 && w->hidden <= 1024, "hidden=%d: the f16c path takes multiples of 256 up to 1024", w->hidden);
    TT_CHECK_ARG(w->heads > 0 && w->hidden == w->heads * 64, "heads=%d: the f16c path is written for head_dim 64", w->heads);
    TT_CHECK_ARG(w->ffn > 0 && w->ffn % 256 == 0, "ffn=%d must be a multiple of 256", w->ffn);
    TT_CHECK_ARG(w->layers >= 0 && (w->layers == 0 || w->layer != nullptr), "layer array missing");
    TT_CHECK_ARG(w->word_emb && w->pos_emb && w->type_emb && w->emb_ln_g && w->emb_ln_b, "embedding tables missing");
    return TT_OK;
}

// one projection on c-planes operands
GemmParams gemm_c(const void* a_planes, const uint8_t* a_scales, const void* w_planes, const uint8_t* w_scales, const float* bias, int M,
                  int N, int K) {
    GemmParams g{};
    g.xc = 1;
    g.A = (const uint16_t*)a_planes; g.lda = 2 * K; g.a_scales = a_scales;
    g.W = (const uint16_t*)w_planes; g.ldw = 2 * K; g.w_scales = w_scales;
    g.bias = bias; g.M = M; g.N = N; g.K = K;
    return g;
}

// hidden_out: the last hidden state [n_rows][H]; cls_out (instead): the last hidden state of every sequence's FIRST row only,
// [cls_pad_c(n_seq)][H] -- the last layer then runs its attention, output projection, LayerNorms and FFN for those rows only
int forward_c_impl(const tt_encoder_weights_f16c* w, const int32_t* ids, const int32_t* pos, const int32_t* type_ids,
                   const int32_t* seq_start, const int32_t* seq_len, int n_seq, int n_rows, int max_len, float* hidden_out,
                   float* cls_out, void* workspace, size_t workspace_bytes, void* stream) {
    if (int rc = check_weights_c(w)) return rc;
    TT_CHECK_ARG(n_rows > 0 && n_rows % 256 == 0, "n_rows=%d must be a positive multiple of 256", n_rows);
    TT_CHECK_ARG(n_seq > 0 && max_len > 0, "n_seq=%d max_len=%d", n_seq, max_len);
    TT_CHECK_ARG(ids && pos && seq_start && seq_len && (hidden_out || cls_out), "null pointer");
    const bool cls_tail = cls_out != nullptr && w->layers > 0;
    const XcWs e = xc_plan(w, n_rows, cls_tail ? n_seq : 0);
    if (!workspace || workspace_bytes < e.total) {
        tt_set_error("tt_encoder_forward_f16c: workspace %zu < required %zu bytes", workspace_bytes, e.total);
        return TT_E_WORKSPACE;
    }
    TT_CHECK_ARG(((uintptr_t)workspace % 256) == 0, "workspace must be 256-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    const int H = w->hidden, F = w->ffn, T = n_rows;
    float* xa = (float*)(ws + e.off_xa);
    float* xb = (float*)(ws + e.off_xb);
    float* y = (float*)(ws + e.off_y);
    char* xc = ws + e.off_xc;
    uint8_t* xs = (uint8_t*)(ws + e.off_xs);
    uint16_t* qk = (uint16_t*)(ws + e.off_qk);
    uint16_t* vt = (uint16_t*)(ws + e.off_vt);
    char* ctx = ws + e.off_ctx;
    uint8_t* cs = (uint8_t*)(ws + e.off_cs);
    char* ffn = ws + e.off_ffn;
    uint8_t* fs = (uint8_t*)(ws + e.off_fs);
    // rows of no sequence (alignment gaps, the padding behind the last one) are never written by the attention kernel: their
    // context planes and scales must be finite (a NaN / garbage row only reaches its own outputs, but keep the run reproducible)
    TT_CHECK_HIP(hipMemsetAsync(ctx, 0, (size_t)T * 4 * H, st));
    TT_CHECK_HIP(hipMemsetAsync(cs, 0, scale_bytes(T, H), st));

    float* x = (w->layers == 0 && hidden_out) ? hidden_out : xa;
    {
        TtProfScope prof(TT_K_ROWOPS, st);
        hipLaunchKernelGGL(embed_ln_c_kernel, row_grid_c(T), dim3(kRowThreadsC), 0, st, ids, pos, type_ids, w->word_emb, w->pos_emb,
                           w->type_emb, w->emb_ln_g, w->emb_ln_b, x, xc, xs, T, H, w->vocab, w->max_pos, w->type_vocab, w->ln_eps);
        TT_CHECK_LAUNCH();
    }
    for (int l = 0; l < w->layers; ++l) {
        const tt_layer_weights_f16c& lw = w->layer[l];
        TT_CHECK_ARG(lw.qkv_w && lw.qkv_s && lw.qkv_b && lw.o_w && lw.o_s && lw.o_b && lw.ln1_g && lw.ln1_b && lw.ffn1_w && lw.ffn1_s &&
                         lw.ffn1_b && lw.ffn2_w && lw.ffn2_s && lw.ffn2_b && lw.ln2_g && lw.ln2_b, "layer %d has a null weight pointer", l);
        // Q, K columns -> fp16 [T][2H]; V columns -> V8 fp16
        GemmParams g = gemm_c(xc, xs, lw.qkv_w, (const uint8_t*)lw.qkv_s, lw.qkv_b, T, 2 * H, H);
        g.C = qk; g.ldc = 2 * H;
        if (int rc = tt_gemm_launch(g, TT_EPI_BIAS, st)) return rc;
        GemmParams gv = gemm_c(xc, xs, (const char*)lw.qkv_w + (size_t)2 * H * 4 * H, (const uint8_t*)lw.qkv_s + (size_t)(2 * H / 256) * 2 * (H / 128) * 1024,
                               lw.qkv_b + 2 * H, T, H, H);
        gv.vt = vt; gv.ldvt = 8 * H; gv.vt_col0 = 0;
        if (int rc = tt_gemm_launch(gv, TT_EPI_VT, st)) return rc;
        AttnParams a{};
        a.qk = qk; a.ld_qk = 2 * H; a.q_col0 = 0; a.k_col0 = H; a.vt = vt; a.ldvt = 8 * H;
        a.seq_start = seq_start; a.seq_len = seq_len; a.n_seq = n_seq; a.heads = w->heads; a.head_dim = 64; a.max_len = max_len;
        a.scale = 0.125f; a.out_width = H; a.ld_out = 2 * H;
        if (cls_tail && l == w->layers - 1) {
            // ---- last layer, first rows only: one-query attention per (sequence, head), then the output projection, the
            //      LayerNorms and the FFN on n_seq (padded to 256) rows instead of n_rows
            const int Bp = cls_pad_c(n_seq);
            char* cctx = ws + e.off_cctx;
            uint8_t* ccs = (uint8_t*)(ws + e.off_ccs);
            float* cx = (float*)(ws + e.off_cx);
            float* cy = (float*)(ws + e.off_cy);
            float* cx1 = (float*)(ws + e.off_cx1);
            char* cxc = ws + e.off_cxc;
            uint8_t* cxs = (uint8_t*)(ws + e.off_cxs);
            char* cffn = ws + e.off_cffn;
            uint8_t* cfs = (uint8_t*)(ws + e.off_cfs);
            TT_CHECK_HIP(hipMemsetAsync(cctx, 0, (size_t)Bp * 4 * H, st));
            TT_CHECK_HIP(hipMemsetAsync(ccs, 0, scale_bytes(Bp, H), st));
            a.out = (uint16_t*)cctx; a.out_scales = ccs;
            if (int rc = tt_attention_cls_launch(a, st)) return rc;
            {
                TtProfScope prof(TT_K_ROWOPS, st);
                hipLaunchKernelGGL(gather_rows_f32c_kernel, dim3(Bp), dim3(256), 0, st, x, seq_start, n_seq, Bp, H, cx);
                TT_CHECK_LAUNCH();
            }
            GemmParams go = gemm_c(cctx, ccs, lw.o_w, (const uint8_t*)lw.o_s, lw.o_b, Bp, H, H);
            go.res32 = cx; go.ldr = H; go.C32 = cy; go.ldc = H;
            if (int rc = tt_gemm_launch(go, TT_EPI_RESIDUAL, st)) return rc;
            {
                TtProfScope prof(TT_K_ROWOPS, st);
                hipLaunchKernelGGL(layernorm_c_kernel, row_grid_c(Bp), dim3(kRowThreadsC), 0, st, cy, cx1, cxc, cxs, lw.ln1_g, lw.ln1_b, Bp, H,
                                   w->ln_eps);
                TT_CHECK_LAUNCH();
            }
            GemmParams g1 = gemm_c(cxc, cxs, lw.ffn1_w, (const uint8_t*)lw.ffn1_s, lw.ffn1_b, Bp, F, H);
            g1.C = (uint16_t*)cffn; g1.ldc = 2 * F; g1.c_scales = cfs;
            if (int rc = tt_gemm_launch(g1, TT_EPI_GELU, st)) return rc;
            GemmParams g2 = gemm_c(cffn, cfs, lw.ffn2_w, (const uint8_t*)lw.ffn2_s, lw.ffn2_b, Bp, H, F);
            g2.res32 = cx1; g2.ldr = H; g2.C32 = cy; g2.ldc = H;
            if (int rc = tt_gemm_launch(g2, TT_EPI_RESIDUAL, st)) return rc;
            TtProfScope prof(TT_K_ROWOPS, st);
            hipLaunchKernelGGL(layernorm_c_kernel, row_grid_c(Bp), dim3(kRowThreadsC), 0, st, cy, cls_out, (char*)nullptr, (uint8_t*)nullptr,
                               lw.ln2_g, lw.ln2_b, Bp, H, w->ln_eps);
            TT_CHECK_LAUNCH();
            return TT_OK;
        }
        a.out = (uint16_t*)ctx; a.out_scales = cs;
        if (int rc = tt_attention_launch(a, st)) return rc;
        GemmParams go = gemm_c(ctx, cs, lw.o_w, (const uint8_t*)lw.o_s, lw.o_b, T, H, H);
        go.res32 = x; go.ldr = H; go.C32 = y; go.ldc = H;
        if (int rc = tt_gemm_launch(go, TT_EPI_RESIDUAL, st)) return rc;
        float* x1 = (x == xa) ? xb : xa;
        {
            TtProfScope prof(TT_K_ROWOPS, st);
            hipLaunchKernelGGL(layernorm_c_kernel, row_grid_c(T), dim3(kRowThreadsC), 0, st, y, x1, xc, xs, lw.ln1_g, lw.ln1_b, T, H, w->ln_eps);
            TT_CHECK_LAUNCH();
        }
        GemmParams g1 = gemm_c(xc, xs, lw.ffn1_w, (const uint8_t*)lw.ffn1_s, lw.ffn1_b, T, F, H);
        g1.C = (uint16_t*)ffn; g1.ldc = 2 * F; g1.c_scales = fs;
        if (int rc = tt_gemm_launch(g1, TT_EPI_GELU, st)) return rc;
        GemmParams g2 = gemm_c(ffn, fs, lw.ffn2_w, (const uint8_t*)lw.ffn2_s, lw.ffn2_b, T, H, F);
        g2.res32 = x1; g2.ldr = H; g2.C32 = y; g2.ldc = H;
        if (int rc = tt_gemm_launch(g2, TT_EPI_RESIDUAL, st)) return rc;
        const bool last = l == w->layers - 1;
        float* dst = last ? hidden_out : x;
        {
            TtProfScope prof(TT_K_ROWOPS, st);
            hipLaunchKernelGGL(layernorm_c_kernel, row_grid_c(T), dim3(kRowThreadsC), 0, st, y, dst, last ? (char*)nullptr : xc,
                               last ? (uint8_t*)nullptr : xs, lw.ln2_g, lw.ln2_b, T, H, w->ln_eps);
            TT_CHECK_LAUNCH();
        }
        x = dst;
    }
    if (cls_out) {     // no layers: the "last hidden state" is the embedding LayerNorm's output -- gather the first rows
        TtProfScope prof(TT_K_ROWOPS, st);
        const int Bp = cls_pad_c(n_seq);
        hipLaunchKernelGGL(gather_rows_f32c_kernel, dim3(Bp), dim3(256), 0, st, x, seq_start, n_seq, Bp, H, cls_out);
        TT_CHECK_LAUNCH();
    }
    return TT_OK;
}
}  // namespace

extern "C" {

size_t tt_f16c_scale_bytes(int64_t rows, int k, int weight) {
    if (rows <= 0 || k <= 0) return 0;
    const size_t r256 = ((size_t)rows + 255) / 256;
    return r256 * (size_t)(k / 128) * 1024 * (weight ? 2 : 1);
}

int tt_f16c_quantize(const float* in_f32, int64_t rows, int k, int weight, void* out_planes, void* out_scales, void* stream) {
    TT_CHECK_ARG(in_f32 && out_planes && out_scales && rows >= 0 && k > 0 && k % 256 == 0, "bad argument (k must be a multiple of 256)");
    TT_CHECK_ARG(rows < (int64_t)1 << 31, "rows out of range");
    if (rows == 0) return TT_OK;
    hipLaunchKernelGGL(quantize_c_kernel, row_grid_c((int)rows), dim3(kRowThreadsC), 0, (hipStream_t)stream, in_f32, (char*)out_planes,
                       (uint8_t*)out_scales, (int)rows, k, weight);
    TT_CHECK_LAUNCH();
    return TT_OK;
}

int tt_gemm_f16c(const void* a_planes, const void* a_scales, const void* w_planes, const void* w_scales, const float* bias,
                 const float* residual_f32, void* c_out, void* c_scales, int m, int n, int k, int epilogue, void* stream) {
    TT_CHECK_ARG(a_planes && a_scales && w_planes && w_scales && bias && c_out, "null pointer");
    TT_CHECK_ARG(epilogue == TT_EPI_BIAS || epilogue == TT_EPI_GELU || epilogue == TT_EPI_RESIDUAL, "epilogue %d", epilogue);
    GemmParams g = gemm_c(a_planes, (const uint8_t*)a_scales, w_planes, (const uint8_t*)w_scales, bias, m, n, k);
    if (epilogue == TT_EPI_RESIDUAL) {
        TT_CHECK_ARG(residual_f32, "residual epilogue: fp32 residual");
        g.res32 = residual_f32; g.ldr = n; g.C32 = (float*)c_out; g.ldc = n;
    } else if (epilogue == TT_EPI_GELU) {
        TT_CHECK_ARG(c_scales, "GELU epilogue: c-planes out need a scale array");
        g.C = (uint16_t*)c_out; g.ldc = 2 * n; g.c_scales = (uint8_t*)c_scales;
    } else {
        g.C = (uint16_t*)c_out; g.ldc = n;             // plain fp16 [m][n]
    }
    return tt_gemm_launch(g, epilogue, (hipStream_t)stream);
}

int tt_attention_f16c(const void* qk_f16, int ld_qk, int q_col0, int k_col0, const void* vt_f16, int ldvt, void* out_planes,
                      void* out_scales, const int32_t* seq_start, const int32_t* seq_len, int n_seq, int heads, int max_len, void* stream) {
    TT_CHECK_ARG(qk_f16 && vt_f16 && out_planes && out_scales && seq_start && seq_len, "null pointer");
    AttnParams a{};
    a.qk = (const uint16_t*)qk_f16; a.ld_qk = ld_qk; a.q_col0 = q_col0; a.k_col0 = k_col0; a.vt = (const uint16_t*)vt_f16; a.ldvt = ldvt;
    a.out = (uint16_t*)out_planes; a.out_scales = (uint8_t*)out_scales; a.out_width = heads * 64; a.ld_out = 2 * heads * 64;
    a.seq_start = seq_start; a.seq_len = seq_len; a.n_seq = n_seq; a.heads = heads; a.head_dim = 64; a.max_len = max_len; a.scale = 0.125f;
    return tt_attention_launch(a, (hipStream_t)stream);
}

size_t tt_encoder_f16c_workspace_bytes(const tt_encoder_weights_f16c* w, int n_rows) {
    if (!w || n_rows <= 0) return 0;
    return xc_plan(w, n_rows).total;
}

int tt_encoder_forward_f16c(const tt_encoder_weights_f16c* w, const int32_t* ids, const int32_t* pos, const int32_t* type_ids,
                            const int32_t* seq_start, const int32_t* seq_len, int n_seq, int n_rows, int max_len,
                            float* hidden_out, void* workspace, size_t workspace_bytes, void* stream) {
    TT_CHECK_ARG(hidden_out != nullptr, "null pointer");
    return forward_c_impl(w, ids, pos, type_ids, seq_start, seq_len, n_seq, n_rows, max_len, hidden_out, nullptr, workspace,
                          workspace_bytes, stream);
}

size_t tt_encoder_f16c_cls_workspace_bytes(const tt_encoder_weights_f16c* w, int n_rows, int n_seq) {
    if (!w || n_rows <= 0 || n_seq <= 0) return 0;
    return xc_plan(w, n_rows, n_seq).total;
}

int tt_encoder_forward_f16c_cls(const tt_encoder_weights_f16c* w, const int32_t* ids, const int32_t* pos, const int32_t* type_ids,
                                const int32_t* seq_start, const int32_t* seq_len, int n_seq, int n_rows, int max_len,
                                float* cls_out, void* workspace, size_t workspace_bytes, void* stream) {
    TT_CHECK_ARG(cls_out != nullptr, "null pointer");
    return forward_c_impl(w, ids, pos, type_ids, seq_start, seq_len, n_seq, n_rows, max_len, nullptr, cls_out, workspace,
                          workspace_bytes, stream);
}

int tt_rerank_head_f16c(const tt_encoder_weights_f16c* w, const float* hidden_f32, const int32_t* rows, int n_seq, float* scores,
                        float* logits, void* workspace, size_t workspace_bytes, void* stream) {
    TT_CHECK_ARG(w != nullptr, "null weights");
    tt_encoder_weights_f32 h{};     // the head is a [n_seq x H x H] product: the fp32 kernels (f32_path.hip) on fp32 head weights
    h.hidden = w->hidden; h.layers = 0; h.heads = w->heads; h.ffn = w->ffn; h.vocab = w->vocab; h.max_pos = w->max_pos;
    h.type_vocab = w->type_vocab; h.ln_eps = w->ln_eps;
    h.word_emb = w->word_emb; h.pos_emb = w->pos_emb; h.type_emb = w->type_emb; h.emb_ln_g = w->emb_ln_g; h.emb_ln_b = w->emb_ln_b;
    h.cls_dense_w = w->cls_dense_w; h.cls_dense_b = w->cls_dense_b; h.cls_out_w = w->cls_out_w; h.cls_out_b = w->cls_out_b;
    return tt_rerank_head_f32(&h, hidden_f32, rows, n_seq, scores, logits, workspace, workspace_bytes, stream);
}

}  // extern "C"

#endif  // TT_F16
