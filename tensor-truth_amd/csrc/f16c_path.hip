// Reference precision on TWO matrix-time units: the "f16c" (fp16 + e4m3 corrections) encoder forward, gfx950.  Round 4.
//
// The reference's default embedder / reranker dtype is fp32 (app_utils/config_schema.py:66-76: torch_dtype None;
// services/model_manager.py:218-229, 333-337 pass no dtype); north_star's score tolerance is 1e-3 relative.  x3_path.hip meets
// it with split-bf16 operands at three bf16 MFMA products per product.  This file meets it at TWO units: an operand value is
//       x = hi + lo,   hi = fp16(x)                                  (11-bit significand)
//       a.w ~= a_hi.w_hi                                             v_mfma_f32_16x16x32_f16, exact products, fp32 accumulate: 1 unit
//            + e4m3(a).e4m3(w_lo) + e4m3(a_lo).e4m3(w)               v_mfma_scale_f32_16x16x128_f8f6f4 with E8M0 block scales per
//                                                                    32 elements: twice the bf16 rate -> 1/2 unit each
// The two cross terms are 2^-12 of the result and only need e4m3's 2^-4; the dropped lo.lo term is 2^-24.  Attention: the score
// product on THREE fp16 products (Q and K as two fp16 planes: logits of tens need more than 2^-11), values on one (V and P
// rounded to fp16); fp32 scores, softmax and accumulators (attention_qk2_kernel below).
// Everything that is not a product stays fp32, as in x3_path.hip: the residual stream, LayerNorm, exact-erf GELU, the head.
// CPU emulation of exactly this scheme before any kernel was written (tools/probes/f16c_emulation.py, full depth, the committed
// fp32 fixture): scores within 8.3e-5 relative of the fp32 oracle, Kendall tau 1.000.
//
// Tensors: GEMM A operands and weights are "c-planes" (f16c.h: [hi | x8 | lo8] rows of 4 K bytes + tiled E8M0 scales); Q / K are
// two fp16 planes, V plain fp16 in the V8 layout; the residual stream is fp32.  Layer schedule (post-LN block):
//   qk (2 x fp16), V8     = GEMMc(x_c, Wqkv)              two launches: planes epilogue, V^T epilogue
//   ctx_c                 = attention_qk2(qk, V8)          c-planes written by the attention epilogue
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


// ---- attention: scores on THREE fp16 products, values on one -------------------------------------------------------------------
// The score product is the one place of the path where fp16's 2^-11 is not enough by itself: a trained model's peaked heads carry
// logits of tens to a hundred, and an operand rounding of 2^-11 on q and k moves such a logit by 0.05 -- percents of a
// probability, 3e-2 relative on the scores of the stress fixture (tests/stress_weights.py: 1.5-bit attention on half the heads;
// the split-bf16 path, whose two bf16 planes carry 16 bits, sits exactly AT 1e-3 there).  So Q and K come out of their
// projection as two fp16 planes, hi = fp16(x), lo = fp16(x - hi) (22 significand bits), and
//       S^T = K_lo.Q_hi + K_hi.Q_lo + K_hi.Q_hi          fp32 scores and softmax
//       O^T += V.P                                        V and P single fp16 (a probability's 2^-11 averages out over the keys)
// x3_path.hip's kernel structure (one workgroup = 4 waves = 128 query rows of one (sequence, head); keys in tiles of 64 through two
// LDS-DMA buffers, one barrier per tile; both products swapped so the query stays on the lane) with three planes per tile
// (K hi, K lo, V: 24 KiB) and the c-planes epilogue of attention.hip.
struct AttnQ2Params {
    const uint16_t* qk;       // fp16 planes [T][ld_qk]: Q hi at q_col0 + h*64, K hi at k_col0 + h*64; lo planes lo_off columns further
    const uint16_t* vt;       // V8 fp16 [T/8][heads*64][8]
    char* out;                // context c-planes [T][4 W bytes]
    uint8_t* out_scales;
    const int32_t* seq_start;
    const int32_t* seq_len;
    int n_seq, heads, max_len;
    int ld_qk, q_col0, k_col0, lo_off, ldvt, out_width;
    float scale, lazy;
    int n_qt;
};

typedef unsigned int u32x4q __attribute__((ext_vector_type(4)));
template <int OFF>
__device__ __forceinline__ u32x4q ldsq_read128(uint32_t addr) {
    u32x4q r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
template <int N>
__device__ __forceinline__ void ldsq_wait8(u32x4q& a, u32x4q& b, u32x4q& c, u32x4q& d, u32x4q& e, u32x4q& f, u32x4q& g, u32x4q& h) {
    asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "n"(N));
}
template <int N>
__device__ __forceinline__ void ldsq_wait4(u32x4q& a, u32x4q& b, u32x4q& c, u32x4q& d) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}

constexpr int kQKTile = 64, kQWaves = 4, kQDH = 64;
constexpr int kQPlane = kQKTile * kQDH * 2;          // 8 KiB: one plane of a K tile (64 rows x 128 B) or the V tile
constexpr int kQBuf = 3 * kQPlane;                   // K hi | K lo | V
constexpr int kQLds = 2 * kQBuf;                     // 48 KiB

__global__ __launch_bounds__(64 * kQWaves, 3) void attention_qk2_kernel(AttnQ2Params p) {
    constexpr int DH = kQDH, RB = DH * 2, CH = RB / 16, RPB = 256 / RB, KS = DH / 16, DT = DH / 32;
    constexpr int NP = kQPlane / 1024;               // 8 one-KiB copy pieces per plane and tile
    constexpr int PPW = NP / kQWaves;                // 2 pieces of every plane per wave
    extern __shared__ __attribute__((aligned(1024))) char lds[];

    const int nqt = p.n_qt;
    const int L = blockIdx.x;
    const int qt = L % nqt, pair = L / nqt;
    if (pair >= p.heads * p.n_seq) return;
    const int head = pair % p.heads, seq = pair / p.heads;
    const int len = p.seq_len[seq];
    if (qt * 32 * kQWaves >= len) return;
    const int t0 = p.seq_start[seq];
    const int t0a = t0 & ~7, off = t0 - t0a, alen = off + len;       // aligned key frame of the V8 token groups
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & 31, hh = lane >> 5;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;

    // ---- Q fragments (B operand), both planes, straight from global
    const int q_row = (qt * kQWaves + wave) * 32 + ql;
    const int q_row_c = q_row < len ? q_row : len - 1;
    const uint16_t* qp = p.qk + (size_t)(t0 + q_row_c) * p.ld_qk + p.q_col0 + head * DH + hh * 8;
    ex8 qh[KS], qlo[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        qh[s] = *reinterpret_cast<const ex8*>(qp + s * 16);
        qlo[s] = *reinterpret_cast<const ex8*>(qp + p.lo_off + s * 16);
    }

    const int n_kt = (alen + kQKTile - 1) / kQKTile;
    const int n_g8 = (alen + 7) >> 3;
    const uint16_t* kbase = p.qk + (size_t)t0a * p.ld_qk + p.k_col0 + head * DH;
    const uint16_t* vbase = p.vt + (size_t)(t0a >> 3) * p.ldvt + (size_t)head * DH * 8;
    constexpr int kRowsPerPiece = 64 / CH;           // 8 K rows per piece
    uint32_t kvoff0, vvoff0;
    {
        const int e = wave * PPW * 64 + lane, r = e / CH, pos = e % CH;
        kvoff0 = ((uint32_t)r * (uint32_t)p.ld_qk + (uint32_t)((pos ^ ((r / RPB) & (CH - 1))) << 3)) * 2u;
        const int ev = wave * PPW * 64 + lane;
        vvoff0 = ((uint32_t)(ev / DH) * (uint32_t)p.ldvt + (uint32_t)((ev % DH) * 8)) * 2u;
    }
    auto sbase = [](const void* ptr) {
        const unsigned long long b = reinterpret_cast<unsigned long long>(ptr);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
        return reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
    };
    auto glds16 = [](const char* base, uint32_t voff, uint32_t lds_addr) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds_addr) : "memory", "m0");
    };
    auto issue_tile = [&](int kt) {
        const uint32_t buf = lds0 + (uint32_t)(kt & 1) * kQBuf;
        const bool clamp = (kt + 1) * kQKTile > alen || (kt == 0 && off != 0);   // wave-uniform
        if (!clamp) {
#pragma unroll
            for (int i = 0; i < PPW; ++i) {
                const uint16_t* kb = kbase + ((size_t)kt * kQKTile + (size_t)i * kRowsPerPiece) * p.ld_qk;
                const uint32_t vo = (i & 1) ? (kvoff0 ^ 64u) : kvoff0;
                glds16(sbase(kb), vo, buf + (uint32_t)(wave * PPW + i) * 1024u);
                glds16(sbase(kb + p.lo_off), vo, buf + kQPlane + (uint32_t)(wave * PPW + i) * 1024u);
            }
#pragma unroll
            for (int i = 0; i < PPW; ++i)
                glds16(sbase(vbase + ((size_t)kt * 8 + i) * p.ldvt), vvoff0, buf + 2 * kQPlane + (uint32_t)(wave * PPW + i) * 1024u);
            return;
        }
        // first / last tile: rows / token groups outside the sequence are clamped to its nearest one (finite values, probability 0)
        const char* kb = sbase(kbase);
        const char* kbl = sbase(kbase + p.lo_off);
        const char* vb = sbase(vbase);
        int lane_c = lane;
        asm volatile("" : "+v"(lane_c));
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int e = (wave * PPW + i) * 64 + lane_c, r = e / CH, pos = e % CH;
            int row = kt * kQKTile + r;
            row = row < off ? off : (row < alen ? row : alen - 1);
            const uint32_t vo = ((uint32_t)row * (uint32_t)p.ld_qk + (uint32_t)((pos ^ ((r / RPB) & (CH - 1))) << 3)) * 2u;
            glds16(kb, vo, buf + (uint32_t)(wave * PPW + i) * 1024u);
            glds16(kbl, vo, buf + kQPlane + (uint32_t)(wave * PPW + i) * 1024u);
        }
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int e = (wave * PPW + i) * 64 + lane_c;
            int g8 = kt * 8 + e / DH;
            g8 = g8 < n_g8 ? g8 : n_g8 - 1;
            glds16(vb, ((uint32_t)g8 * (uint32_t)p.ldvt + (uint32_t)((e % DH) * 8)) * 2u, buf + 2 * kQPlane + (uint32_t)(wave * PPW + i) * 1024u);
        }
    };

    f32x16 acc_o[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[d][r] = 0.f;
    float m_run = -__builtin_inff();
    float l_run = 0.f;
    const float sc = p.scale * 1.4426950408889634f;
    const bool wave_active = (qt * kQWaves + wave) * 32 < len;

    const int krow = (ql & 0x13) | ((ql & 4) << 1) | ((ql & 8) >> 1);     // K row permutation: bits 2 and 3 swapped
    uint32_t koff[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) koff[s] = lds0 + krow * RB + (((2 * s + hh) ^ ((krow / RPB) & (CH - 1))) << 4);
    const uint32_t voff = lds0 + 2 * kQPlane + (hh * DH + ql) * 16;         // + (4 j + 2 s2) * DH*16 + 512 dt

    issue_tile(0);
    for (int kt = 0; kt < n_kt; ++kt) {
        const int k0 = kt * kQKTile;
        __builtin_amdgcn_s_waitcnt(0x0F70);          // this wave's pieces of tile kt have landed (builtin: see attention.hip)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                // ... everyone's; and tile kt - 1 is no longer read
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < n_kt) issue_tile(kt + 1);
        if (!wave_active) continue;
        const uint32_t bufo = (kt & 1) * kQBuf;

        // ---- S^T = K . Q^T, three products, two 32-key sub-tiles
        f32x16 acc_s[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            u32x4q kh[KS], kl[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (j == 0) { kh[s] = ldsq_read128<0>(koff[s] + bufo); kl[s] = ldsq_read128<kQPlane>(koff[s] + bufo); }
                else { kh[s] = ldsq_read128<32 * RB>(koff[s] + bufo); kl[s] = ldsq_read128<kQPlane + 32 * RB>(koff[s] + bufo); }
            }
            ldsq_wait8<0>(kh[0], kh[1], kh[2], kh[3], kl[0], kl[1], kl[2], kl[3]);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_s[j][r] = 0.f;
            // small terms first
#pragma unroll
            for (int s = 0; s < KS; ++s) acc_s[j] = TT_MFMA_32x32x16(__builtin_bit_cast(ex8, kl[s]), qh[s], acc_s[j]);
#pragma unroll
            for (int s = 0; s < KS; ++s) acc_s[j] = TT_MFMA_32x32x16(__builtin_bit_cast(ex8, kh[s]), qlo[s], acc_s[j]);
#pragma unroll
            for (int s = 0; s < KS; ++s) acc_s[j] = TT_MFMA_32x32x16(__builtin_bit_cast(ex8, kh[s]), qh[s], acc_s[j]);
        }
        // V fragments of the first 32 keys: in flight during the softmax
        u32x4q vf[2][2];
        const uint32_t vaddr = voff + bufo;
        vf[0][0] = ldsq_read128<0>(vaddr); vf[0][1] = ldsq_read128<512>(vaddr);
        vf[1][0] = ldsq_read128<2 * DH * 16>(vaddr); vf[1][1] = ldsq_read128<2 * DH * 16 + 512>(vaddr);

        // ---- mask, running reference, exponentials (fp32).  Register r of sub-tile j is key k0 + 32 j + 16 (r>>3) + 8 hh + (r&7)
        if (kt == 0 && off != 0) {
            if (hh == 0) {
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    if (r < off) acc_s[0][r] = -__builtin_inff();
            }
        }
        if (k0 + kQKTile > alen) {
            const int lim = alen - k0 - 8 * hh;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (32 * j + 16 * (r >> 3) + (r & 7) >= lim) acc_s[j][r] = -__builtin_inff();
        }
        float mx = -__builtin_inff();
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, acc_s[j][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mt = mx * sc;
        const float m_new = (mt > m_run + p.lazy) ? mt : m_run;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __builtin_amdgcn_exp2f(fmaf(acc_s[j][r], sc, -m_new));
                acc_s[j][r] = e;
                psum += e;
            }
        l_run = l_run * alpha + psum;
        if (kt > 0 && !__all(alpha == 1.0f)) {
#pragma unroll
            for (int d = 0; d < DT; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc_o[d][r] *= alpha;
        }

        // ---- O^T += V^T . P^T, one product
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            ldsq_wait4<0>(vf[0][0], vf[0][1], vf[1][0], vf[1][1]);
            ex8 va[2][2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int d = 0; d < DT; ++d) va[s2][d] = __builtin_bit_cast(ex8, vf[s2][d]);
            if (j == 0) {   // next 32 keys' fragments, in flight during these MFMAs
                vf[0][0] = ldsq_read128<4 * DH * 16>(vaddr); vf[0][1] = ldsq_read128<4 * DH * 16 + 512>(vaddr);
                vf[1][0] = ldsq_read128<6 * DH * 16>(vaddr); vf[1][1] = ldsq_read128<6 * DH * 16 + 512>(vaddr);
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                uint4 pb;
                pb.x = pack_e2_inrange(acc_s[j][8 * s2 + 0], acc_s[j][8 * s2 + 1]);
                pb.y = pack_e2_inrange(acc_s[j][8 * s2 + 2], acc_s[j][8 * s2 + 3]);
                pb.z = pack_e2_inrange(acc_s[j][8 * s2 + 4], acc_s[j][8 * s2 + 5]);
                pb.w = pack_e2_inrange(acc_s[j][8 * s2 + 6], acc_s[j][8 * s2 + 7]);
                const ex8 pf = __builtin_bit_cast(ex8, pb);
#pragma unroll
                for (int d = 0; d < DT; ++d) acc_o[d] = TT_MFMA_32x32x16(va[s2][d], pf, acc_o[d]);
            }
        }
    }

    // ---- normalise and store as c-planes: lane = query row, registers = 4 consecutive d (attention.hip's epilogue)
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q_row < len) {
        char* orow = p.out + (size_t)(t0 + q_row) * 4 * p.out_width;
        const int W = p.out_width, nks = W >> 7;
#pragma unroll
        for (int d = 0; d < DT; ++d) {
            float y[16];
            float amax = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                y[r] = acc_o[d][r] * inv;
                asm("" : "+v"(y[r]));
                amax = fmaxf(amax, fabsf(y[r]));
            }
            amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
            int sbyte, sh;
            xc_block_scale(amax, sbyte, sh);
            const int col0 = head * DH + 32 * d;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 hi;
                uint32_t x8, l8;
                xc_split4(y[4 * g + 0], y[4 * g + 1], y[4 * g + 2], y[4 * g + 3], sh, sh + 11, hi, x8, l8);
                const int col = col0 + 8 * g + 4 * hh;
                *reinterpret_cast<uint2*>(orow + (size_t)col * 2) = hi;
                *reinterpret_cast<uint32_t*>(orow + (size_t)2 * W + col) = x8;
                *reinterpret_cast<uint32_t*>(orow + (size_t)3 * W + col) = l8;
            }
            if (hh == 0) p.out_scales[xc_a_scale_at(t0 + q_row, col0 >> 5, nks)] = (uint8_t)sbyte;
        }
    }
}

int attention_qk2_launch(const AttnQ2Params& p, hipStream_t st) {
    if (p.n_seq <= 0 || p.max_len <= 0) return TT_OK;
    if ((p.ld_qk % 8) || (p.q_col0 % 8) || (p.k_col0 % 8) || (p.lo_off % 8) || (p.ldvt % 8) || (p.out_width % 128)) {
        tt_set_error("attention f16c: leading dimensions / column offsets must keep 16-byte alignment, out_width a multiple of 128");
        return TT_E_INVALID;
    }
    const int n_qt = (p.max_len + 32 * kQWaves - 1) / (32 * kQWaves);
    const long long pairs = (long long)p.heads * p.n_seq;
    if (pairs * n_qt > 0x7FFFFFFFLL) {
        tt_set_error("attention f16c: %lld workgroups exceed the grid limit", pairs * n_qt);
        return TT_E_UNSUPPORTED;
    }
    AttnQ2Params q = p;
    q.lazy = 8.0f;
    q.n_qt = n_qt;
    TT_SET_MAX_LDS(attention_qk2_kernel, kQLds);
    TtProfScope prof(TT_K_ATTENTION, st);
    hipLaunchKernelGGL(attention_qk2_kernel, dim3((unsigned)(pairs * n_qt)), dim3(64 * kQWaves), kQLds, st, q);
    TT_CHECK_LAUNCH();
    return TT_OK;
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
    e.off_qk = take(T * 4 * H * 2);               // Q hi | K hi | Q lo | K lo
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
        // Q, K columns -> two fp16 planes [T][4H] (hi at [0, 2H), lo at [2H, 4H)); V columns -> V8 fp16
        GemmParams g = gemm_c(xc, xs, lw.qkv_w, (const uint8_t*)lw.qkv_s, lw.qkv_b, T, 2 * H, H);
        g.C = qk; g.ldc = 4 * H; g.c_lo_off = 2 * H;
        if (int rc = tt_gemm_launch(g, TT_EPI_BIAS, st)) return rc;
        GemmParams gv = gemm_c(xc, xs, (const char*)lw.qkv_w + (size_t)2 * H * 4 * H, (const uint8_t*)lw.qkv_s + (size_t)(2 * H / 256) * 2 * (H / 128) * 1024,
                               lw.qkv_b + 2 * H, T, H, H);
        gv.vt = vt; gv.ldvt = 8 * H; gv.vt_col0 = 0;
        if (int rc = tt_gemm_launch(gv, TT_EPI_VT, st)) return rc;
        AttnParams a{};                 // (the CLS tail's one-query kernel: attention.hip, Q / K as hi + lo planes)
        a.qk = qk; a.ld_qk = 4 * H; a.q_col0 = 0; a.k_col0 = H; a.qk_lo_off = 2 * H; a.vt = vt; a.ldvt = 8 * H;
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
        AttnQ2Params a2{};
        a2.qk = qk; a2.ld_qk = 4 * H; a2.q_col0 = 0; a2.k_col0 = H; a2.lo_off = 2 * H; a2.vt = vt; a2.ldvt = 8 * H;
        a2.out = ctx; a2.out_scales = cs; a2.out_width = H; a2.seq_start = seq_start; a2.seq_len = seq_len;
        a2.n_seq = n_seq; a2.heads = w->heads; a2.max_len = max_len; a2.scale = 0.125f;
        if (int rc = attention_qk2_launch(a2, st)) return rc;
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

int tt_attention_f16c(const void* qk_f16, int ld_qk, int q_col0, int k_col0, int lo_off, const void* vt_f16, int ldvt, void* out_planes,
                      void* out_scales, const int32_t* seq_start, const int32_t* seq_len, int n_seq, int heads, int max_len, void* stream) {
    TT_CHECK_ARG(qk_f16 && vt_f16 && out_planes && out_scales && seq_start && seq_len, "null pointer");
    AttnQ2Params a{};
    a.qk = (const uint16_t*)qk_f16; a.ld_qk = ld_qk; a.q_col0 = q_col0; a.k_col0 = k_col0; a.lo_off = lo_off; a.vt = (const uint16_t*)vt_f16;
    a.ldvt = ldvt; a.out = (char*)out_planes; a.out_scales = (uint8_t*)out_scales; a.out_width = heads * 64;
    a.seq_start = seq_start; a.seq_len = seq_len; a.n_seq = n_seq; a.heads = heads; a.max_len = max_len; a.scale = 0.125f;
    return attention_qk2_launch(a, (hipStream_t)stream);
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
