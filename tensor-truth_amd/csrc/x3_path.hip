// Reference precision on the bf16 matrix cores: the "bf16x3" (split-bf16) encoder forward, gfx950.
//
// The reference's default embedder / reranker dtype is fp32 (app_utils/config_schema.py:66-76: torch_dtype None;
// services/model_manager.py:218-229, 333-337 pass no dtype), and north_star's score tolerance -- 1e-3 relative -- is an
// fp32 tolerance.  f32_path.hip meets it on the fp32 MFMA (1/16 of the bf16 matrix rate: 174 ms for one query's 50 pairs).
// This file meets it at a third of the bf16 rate: every GEMM-shaped product runs on v_mfma_*_bf16 with both operands
// split into two bf16 planes,
//       x = hi + lo,   hi = bf16(x),  lo = bf16(x - hi)            (|x - hi - lo| <= 2^-17 |x|)
//       a.b ~= a_hi.b_hi + a_hi.b_lo + a_lo.b_hi                   (the dropped lo.lo term is 2^-16 of the product)
// accumulated in fp32 -- three MFMA products per fp32-grade product instead of sixteen MFMA-f32 cycles.  Everything that is
// not a product stays fp32: the residual stream, LayerNorm statistics, softmax, exact-erf GELU.
//
// Tensors:  GEMM A operands are "planes" [rows][2W] bf16 (hi | lo side by side: gemm.hip reads them as one K stream of 3K,
// GemmParams.x3); the residual stream (LayerNorm in / out) is fp32; weights are planes [out][2 in] made on load.
// Layer schedule (post-LN block, as encoder_api.hip):
//   qk planes, V8 planes = QKV-GEMMx3(x planes)         [T][4H] = Q hi | K hi | Q lo | K lo;  V8 hi, V8 lo
//   ctx planes           = attention_x3(qk, V8)          S = K.Q^T and O = V^T.P^T as 3 MFMA products each, fp32 softmax
//   y (fp32)             = GEMMx3(ctx, Wo) + bo + x      fp32 residual read by the epilogue
//   x1 (fp32 + planes)   = LayerNorm(y)
//   f planes             = GELU_erf(GEMMx3(x1, W1) + b1) [T][2F]
//   y (fp32)             = GEMMx3(f, W2) + b2 + x1
//   x (fp32 + planes)    = LayerNorm(y)
// Roofline: MFMA-bound like the bf16 path, 3 x 2 M N K flops per GEMM launch.
#include "common.h"
#include "encoder.h"

#define TT_MFMA_32x32x16Z(a, b, c, x, y, z) TT_MFMA_32x32x16((a), (b), (c))

extern "C" int tt_rerank_head_f32(const tt_encoder_weights_f32* w, const float* hidden_f32, const int32_t* rows, int n_seq,
                                  float* scores, float* logits, void* workspace, size_t workspace_bytes, void* stream);

namespace {

// ---- row kernels: one wave per row, lane owns 4 consecutive elements per 256-element chunk ---------------------------
constexpr int kRowThreadsX = 256;
constexpr int kMaxC4 = 4;   // H <= 1024

__device__ __forceinline__ float wave_sum_x(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// LayerNorm of a row held as x[c] (float4 = elements 256 c + 4 lane ...), two-pass fp32 statistics; writes the fp32 row
// (out32, may be NULL) and its two bf16 planes (planes[0..H) = hi, planes[H..2H) = lo; may be NULL)
__device__ __forceinline__ float rbf(float v) { return elo(pack_e2(v, 0.f)); }   // round to the planes' element type, kept as fp32

// flags (diagnostic, TT_X3_ROUND_MASK): 1 = the input row is rounded to bf16 first, 2 = the output is rounded to bf16
// (H a multiple of 128, round 6: bge-small's 384 -- in the last 256-element chunk only the lanes with 256 c + 4 lane < H hold
// elements; `nc` = chunks, whole or partial; the callers load x[c] for those lanes only)
__device__ __forceinline__ bool lane_has(int c, int lane, int H) { return 256 * c + 4 * lane < H; }
__device__ __forceinline__ void ln_row_x3(float4 (&x)[kMaxC4], int nc, int H, const float* gamma, const float* beta, float eps,
                                          float* out32, uint16_t* planes, int lane, int flags = 0) {
    if (flags & 1) {
#pragma unroll
        for (int c = 0; c < kMaxC4; ++c)
            if (c < nc && lane_has(c, lane, H)) x[c] = float4{rbf(x[c].x), rbf(x[c].y), rbf(x[c].z), rbf(x[c].w)};
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < kMaxC4; ++c)
        if (c < nc && lane_has(c, lane, H)) s += (x[c].x + x[c].y) + (x[c].z + x[c].w);
    const float mean = wave_sum_x(s) / (float)H;
    float v = 0.f;
#pragma unroll
    for (int c = 0; c < kMaxC4; ++c)
        if (c < nc && lane_has(c, lane, H)) {
            const float a = x[c].x - mean, b = x[c].y - mean, d = x[c].z - mean, e = x[c].w - mean;
            v += (a * a + b * b) + (d * d + e * e);
        }
    const float rstd = 1.0f / sqrtf(wave_sum_x(v) / (float)H + eps);
#pragma unroll
    for (int c = 0; c < kMaxC4; ++c)
        if (c < nc && lane_has(c, lane, H)) {
            const int e0 = 256 * c + 4 * lane;
            const float4 g = *reinterpret_cast<const float4*>(gamma + e0);
            const float4 b = *reinterpret_cast<const float4*>(beta + e0);
            float4 y = float4{(x[c].x - mean) * rstd * g.x + b.x, (x[c].y - mean) * rstd * g.y + b.y,
                              (x[c].z - mean) * rstd * g.z + b.z, (x[c].w - mean) * rstd * g.w + b.w};
            if (flags & 2) y = float4{rbf(y.x), rbf(y.y), rbf(y.z), rbf(y.w)};
            if (out32) *reinterpret_cast<float4*>(out32 + e0) = y;
            if (planes) {
                uint2 hi, lo;
                hi.x = pack_e2(y.x, y.y);
                hi.y = pack_e2(y.z, y.w);
                lo.x = pack_e2(y.x - elo(hi.x), y.y - ehi(hi.x));
                lo.y = pack_e2(y.z - elo(hi.y), y.w - ehi(hi.y));
                *reinterpret_cast<uint2*>(planes + e0) = hi;
                *reinterpret_cast<uint2*>(planes + H + e0) = lo;
            }
        }
}

__global__ __launch_bounds__(kRowThreadsX) void embed_ln_x3_kernel(const int32_t* ids, const int32_t* pos, const int32_t* type,
                                                                    const float* word, const float* posemb, const float* typeemb,
                                                                    const float* gamma, const float* beta, float* out32,
                                                                    uint16_t* planes, int T, int H, int vocab, int max_pos,
                                                                    int type_vocab, float eps, int flags) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= T) return;
    int id = ids[row], p = pos[row], t = type ? type[row] : 0;
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    p = p < 0 ? 0 : (p >= max_pos ? max_pos - 1 : p);
    t = t < 0 ? 0 : (t >= type_vocab ? type_vocab - 1 : t);
    const int nc = (H + 255) / 256;
    float4 x[kMaxC4];
#pragma unroll
    for (int c = 0; c < kMaxC4; ++c)
        if (c < nc && lane_has(c, lane, H)) {
            const int e0 = 256 * c + 4 * lane;
            const float4 a = *reinterpret_cast<const float4*>(word + (size_t)id * H + e0);
            const float4 b = *reinterpret_cast<const float4*>(posemb + (size_t)p * H + e0);
            const float4 d = *reinterpret_cast<const float4*>(typeemb + (size_t)t * H + e0);
            x[c] = float4{a.x + b.x + d.x, a.y + b.y + d.y, a.z + b.z + d.z, a.w + b.w + d.w};
        }
    ln_row_x3(x, nc, H, gamma, beta, eps, out32 + (size_t)row * H, planes + (size_t)row * 2 * H, lane, flags & 2);
}

__global__ __launch_bounds__(kRowThreadsX) void layernorm_x3_kernel(const float* in, float* out32, uint16_t* planes, const float* gamma,
                                                                     const float* beta, int rows, int H, float eps, int flags) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int nc = (H + 255) / 256;
    float4 x[kMaxC4];
#pragma unroll
    for (int c = 0; c < kMaxC4; ++c)
        if (c < nc && lane_has(c, lane, H)) x[c] = *reinterpret_cast<const float4*>(in + (size_t)row * H + 256 * c + 4 * lane);
    ln_row_x3(x, nc, H, gamma, beta, eps, out32 ? out32 + (size_t)row * H : nullptr, planes ? planes + (size_t)row * 2 * H : nullptr, lane,
              flags);
}

// fp32 [rows][cols] -> planes [rows][2 cols] (tests, and callers that bring fp32 activations)
__global__ __launch_bounds__(256) void split_planes_kernel(const float* in, uint16_t* planes, int64_t rows, int cols) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;       // one float4 per thread
    const int c4 = cols / 4;
    if (i >= rows * c4) return;
    const int64_t r = i / c4;
    const int c = (int)(i % c4) * 4;
    const float4 y = *reinterpret_cast<const float4*>(in + r * cols + c);
    uint2 hi, lo;
    hi.x = pack_e2(y.x, y.y);
    hi.y = pack_e2(y.z, y.w);
    lo.x = pack_e2(y.x - elo(hi.x), y.y - ehi(hi.x));
    lo.y = pack_e2(y.z - elo(hi.y), y.w - ehi(hi.y));
    *reinterpret_cast<uint2*>(planes + r * 2 * cols + c) = hi;
    *reinterpret_cast<uint2*>(planes + r * 2 * cols + cols + c) = lo;
}

// ---- attention ---------------------------------------------------------------------------------------------------------
// attention.hip's structure (one workgroup = 4 waves = 128 query rows of one (sequence, head); keys in tiles of 64 through
// two LDS-DMA buffers, one barrier per tile; both products swapped on v_mfma_f32_32x32x16_bf16 so the query stays on the
// lane; K rows fed in permuted order so that a P fragment is 8 consecutive keys = one 16-byte piece of the V8 layout)
// with every operand in two planes and three MFMA products per product:
//   S^T  = K_hi.Q_hi + K_hi.Q_lo + K_lo.Q_hi                        fp32 scores, softmax in fp32 (exp2, lazy running reference)
//   O^T += V_hi.P_hi + V_lo.P_hi + V_hi.P_lo,   P_hi = bf16(p), P_lo = bf16(p - P_hi);  the row sum is taken over the fp32 p
// A tile is 32 KiB of LDS (K hi, K lo, V hi, V lo: 8 KiB each), two buffers, two workgroups per CU.
struct AttnX3Params {
    const uint16_t* qk;       // planes [T][ld_qk]: Q hi at q_col0 + h*64, K hi at k_col0 + h*64; lo planes lo_off columns further
    const uint16_t* vt;       // V8 hi [T/8][heads*64][8]
    const uint16_t* vt_lo;    // V8 lo
    uint16_t* out;            // context planes [T][ld_out]: hi at h*64, lo at out_lo_off + h*64
    const int32_t* seq_start;
    const int32_t* seq_len;
    int n_seq, heads, max_len;
    int ld_qk, q_col0, k_col0, lo_off, ldvt, ld_out, out_lo_off;
    float scale, lazy;
    int n_qt;
    int round_flags;          // diagnostic (TT_X3_ROUND_MASK): 1 = probabilities rounded to bf16 (P lo = 0), 2 = context rounded (lo = 0)
    int head_dim;             // 64 (bge-m3, bge-reranker-v2-m3 / -base) or 32 (bge-small-en-v1.5, ms-marco-MiniLM: round 6); 0 = 64
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int OFF>
__device__ __forceinline__ u32x4 ldsx_read128(uint32_t addr) {
    u32x4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
template <int N>
__device__ __forceinline__ void ldsx_wait8(u32x4& a, u32x4& b, u32x4& c, u32x4& d, u32x4& e, u32x4& f, u32x4& g, u32x4& h) {
    asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "n"(N));
}
template <int N>
__device__ __forceinline__ void ldsx_wait4(u32x4& a, u32x4& b, u32x4& c, u32x4& d) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}

constexpr int kXKTile = 64, kXWaves = 4;
// DH = head width: 64, or 32 (round 6: the 384-wide BERT models' 12 x 32 heads -- half the bytes per row and tile, two MFMA K-steps
// per score tile instead of four, one 32-feature output tile instead of two; same schedule)
template <int DH>
__global__ __launch_bounds__(64 * kXWaves, 2) void attention_x3_kernel(AttnX3Params p) {
    constexpr int kXPlane = kXKTile * DH * 2;        // 8 KiB (DH 64): one plane of a K tile (64 rows x 128 B) or of a V tile
    constexpr int kXBuf = 4 * kXPlane;               // K hi | K lo | V hi | V lo
    constexpr int RB = DH * 2, CH = RB / 16, RPB = 256 / RB, KS = DH / 16, DT = DH / 32;
    constexpr int NP = kXPlane / 1024;               // 8 (4) one-KiB copy pieces per plane and tile
    constexpr int PPW = NP / kXWaves;                // 2 (1) pieces of every plane per wave
    extern __shared__ __attribute__((aligned(1024))) char lds[];

    const int nqt = p.n_qt;
    const int L = blockIdx.x;
    const int qt = L % nqt, pair = L / nqt;
    if (pair >= p.heads * p.n_seq) return;
    const int head = pair % p.heads, seq = pair / p.heads;
    const int len = p.seq_len[seq];
    if (qt * 32 * kXWaves >= len) return;
    const int t0 = p.seq_start[seq];
    const int t0a = t0 & ~7, off = t0 - t0a, alen = off + len;       // aligned key frame of the V8 token groups
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & 31, hh = lane >> 5;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;

    // ---- Q fragments (B operand), both planes, straight from global
    const int q_row = (qt * kXWaves + wave) * 32 + ql;
    const int q_row_c = q_row < len ? q_row : len - 1;
    const uint16_t* qp = p.qk + (size_t)(t0 + q_row_c) * p.ld_qk + p.q_col0 + head * DH + hh * 8;
    ex8 qh[KS], qlo[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        qh[s] = *reinterpret_cast<const ex8*>(qp + s * 16);
        qlo[s] = *reinterpret_cast<const ex8*>(qp + p.lo_off + s * 16);
    }

    const int n_kt = (alen + kXKTile - 1) / kXKTile;
    const int n_g8 = (alen + 7) >> 3;
    const uint16_t* kbase = p.qk + (size_t)t0a * p.ld_qk + p.k_col0 + head * DH;
    const uint16_t* vbase = p.vt + (size_t)(t0a >> 3) * p.ldvt + (size_t)head * DH * 8;
    const uint16_t* vbase_lo = p.vt_lo + (size_t)(t0a >> 3) * p.ldvt + (size_t)head * DH * 8;
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
        const uint32_t buf = lds0 + (uint32_t)(kt & 1) * kXBuf;
        const bool clamp = (kt + 1) * kXKTile > alen || (kt == 0 && off != 0);   // wave-uniform
        if (!clamp) {
#pragma unroll
            for (int i = 0; i < PPW; ++i) {
                const uint16_t* kb = kbase + ((size_t)kt * kXKTile + (size_t)i * kRowsPerPiece) * p.ld_qk;
                const uint32_t vo = (i & 1) ? (kvoff0 ^ 64u) : kvoff0;
                glds16(sbase(kb), vo, buf + (uint32_t)(wave * PPW + i) * 1024u);
                glds16(sbase(kb + p.lo_off), vo, buf + kXPlane + (uint32_t)(wave * PPW + i) * 1024u);
            }
#pragma unroll
            for (int i = 0; i < PPW; ++i) {
                const size_t go = ((size_t)kt * 8 + i) * p.ldvt;
                glds16(sbase(vbase + go), vvoff0, buf + 2 * kXPlane + (uint32_t)(wave * PPW + i) * 1024u);
                glds16(sbase(vbase_lo + go), vvoff0, buf + 3 * kXPlane + (uint32_t)(wave * PPW + i) * 1024u);
            }
            return;
        }
        // first / last tile: rows / token groups outside the sequence are clamped to its nearest one (finite values, probability 0)
        const char* kb = sbase(kbase);
        const char* kbl = sbase(kbase + p.lo_off);
        const char* vb = sbase(vbase);
        const char* vbl = sbase(vbase_lo);
        int lane_c = lane;
        asm volatile("" : "+v"(lane_c));
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int e = (wave * PPW + i) * 64 + lane_c, r = e / CH, pos = e % CH;
            int row = kt * kXKTile + r;
            row = row < off ? off : (row < alen ? row : alen - 1);
            const uint32_t vo = ((uint32_t)row * (uint32_t)p.ld_qk + (uint32_t)((pos ^ ((r / RPB) & (CH - 1))) << 3)) * 2u;
            glds16(kb, vo, buf + (uint32_t)(wave * PPW + i) * 1024u);
            glds16(kbl, vo, buf + kXPlane + (uint32_t)(wave * PPW + i) * 1024u);
        }
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int e = (wave * PPW + i) * 64 + lane_c;
            int g8 = kt * 8 + e / DH;
            g8 = g8 < n_g8 ? g8 : n_g8 - 1;
            const uint32_t vo = ((uint32_t)g8 * (uint32_t)p.ldvt + (uint32_t)((e % DH) * 8)) * 2u;
            glds16(vb, vo, buf + 2 * kXPlane + (uint32_t)(wave * PPW + i) * 1024u);
            glds16(vbl, vo, buf + 3 * kXPlane + (uint32_t)(wave * PPW + i) * 1024u);
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
    const bool wave_active = (qt * kXWaves + wave) * 32 < len;

    const int krow = (ql & 0x13) | ((ql & 4) << 1) | ((ql & 8) >> 1);     // K row permutation: bits 2 and 3 swapped
    uint32_t koff[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) koff[s] = lds0 + krow * RB + (((2 * s + hh) ^ ((krow / RPB) & (CH - 1))) << 4);
    const uint32_t voff = lds0 + 2 * kXPlane + (hh * DH + ql) * 16;         // + (4 j + 2 s2) * DH*16 + 512 dt; lo plane + kXPlane

    issue_tile(0);
    for (int kt = 0; kt < n_kt; ++kt) {
        const int k0 = kt * kXKTile;
        __builtin_amdgcn_s_waitcnt(0x0F70);          // this wave's pieces of tile kt have landed (builtin: see attention.hip)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                // ... everyone's; and tile kt - 1 is no longer read
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < n_kt) issue_tile(kt + 1);
        if (!wave_active) continue;
        const uint32_t bufo = (kt & 1) * kXBuf;

        // ---- S^T = K . Q^T, three products, two 32-key sub-tiles
        f32x16 acc_s[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            u32x4 kh[KS], kl[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (j == 0) { kh[s] = ldsx_read128<0>(koff[s] + bufo); kl[s] = ldsx_read128<kXPlane>(koff[s] + bufo); }
                else { kh[s] = ldsx_read128<32 * RB>(koff[s] + bufo); kl[s] = ldsx_read128<kXPlane + 32 * RB>(koff[s] + bufo); }
            }
            if constexpr (KS == 4) ldsx_wait8<0>(kh[0], kh[1], kh[2], kh[3], kl[0], kl[1], kl[2], kl[3]);
            else ldsx_wait4<0>(kh[0], kh[1], kl[0], kl[1]);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_s[j][r] = 0.f;
            // small terms first
#pragma unroll
            for (int s = 0; s < KS; ++s)
                acc_s[j] = TT_MFMA_32x32x16Z(__builtin_bit_cast(ex8, kl[s]), qh[s], acc_s[j], 0, 0, 0);
#pragma unroll
            for (int s = 0; s < KS; ++s)
                acc_s[j] = TT_MFMA_32x32x16Z(__builtin_bit_cast(ex8, kh[s]), qlo[s], acc_s[j], 0, 0, 0);
#pragma unroll
            for (int s = 0; s < KS; ++s)
                acc_s[j] = TT_MFMA_32x32x16Z(__builtin_bit_cast(ex8, kh[s]), qh[s], acc_s[j], 0, 0, 0);
        }

        // ---- mask, running reference, exponentials (fp32).  Register r of sub-tile j is key k0 + 32 j + 16 (r>>3) + 8 hh + (r&7)
        if (kt == 0 && off != 0) {
            if (hh == 0) {
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    if (r < off) acc_s[0][r] = -__builtin_inff();
            }
        }
        if (k0 + kXKTile > alen) {
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
        // Round 5: raw v_exp_f32 on packed FMAs, as in the 16-bit kernel (libm's exp2f wraps the same instruction in ~6 more per value
        // to rescale results below 2^-126, which are zero probabilities either way: arguments are <= 2^lazy, and a p of 1e-38 adds
        // nothing to a sum that holds at least one p >= 2^-lazy) -- the softmax was the VALU half of a loop that issues 48 MFMAs per tile
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        f32x2 psum2 = f32x2{0.f, 0.f};
        const f32x2 sc2 = f32x2{sc, sc}, mn2 = f32x2{m_new, m_new};
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 t = f32x2{acc_s[j][r], acc_s[j][r + 1]} * sc2 - mn2;
                const f32x2 e = f32x2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
                acc_s[j][r] = e.x;
                acc_s[j][r + 1] = e.y;
                psum2 += e;
            }
        l_run = l_run * alpha + (psum2.x + psum2.y);
        if (kt > 0 && !__all(alpha == 1.0f)) {
#pragma unroll
            for (int d = 0; d < DT; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc_o[d][r] *= alpha;
        }

        // ---- O^T += V^T . P^T, three products
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            u32x4 vh[2][DT], vl[2][DT];
            const uint32_t va = voff + bufo;
            if constexpr (DT == 2) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    if (j == 0 && s2 == 0) {
                        vh[0][0] = ldsx_read128<0>(va); vh[0][1] = ldsx_read128<512>(va);
                        vl[0][0] = ldsx_read128<kXPlane>(va); vl[0][1] = ldsx_read128<kXPlane + 512>(va);
                    } else if (j == 0) {
                        vh[1][0] = ldsx_read128<2 * DH * 16>(va); vh[1][1] = ldsx_read128<2 * DH * 16 + 512>(va);
                        vl[1][0] = ldsx_read128<kXPlane + 2 * DH * 16>(va); vl[1][1] = ldsx_read128<kXPlane + 2 * DH * 16 + 512>(va);
                    } else if (s2 == 0) {
                        vh[0][0] = ldsx_read128<4 * DH * 16>(va); vh[0][1] = ldsx_read128<4 * DH * 16 + 512>(va);
                        vl[0][0] = ldsx_read128<kXPlane + 4 * DH * 16>(va); vl[0][1] = ldsx_read128<kXPlane + 4 * DH * 16 + 512>(va);
                    } else {
                        vh[1][0] = ldsx_read128<6 * DH * 16>(va); vh[1][1] = ldsx_read128<6 * DH * 16 + 512>(va);
                        vl[1][0] = ldsx_read128<kXPlane + 6 * DH * 16>(va); vl[1][1] = ldsx_read128<kXPlane + 6 * DH * 16 + 512>(va);
                    }
                }
                ldsx_wait8<0>(vh[0][0], vh[0][1], vh[1][0], vh[1][1], vl[0][0], vl[0][1], vl[1][0], vl[1][1]);
            } else {          // one 32-feature tile: the fragment of token group 4 j + 2 s2 + hh
                if (j == 0) {
                    vh[0][0] = ldsx_read128<0>(va); vl[0][0] = ldsx_read128<kXPlane>(va);
                    vh[1][0] = ldsx_read128<2 * DH * 16>(va); vl[1][0] = ldsx_read128<kXPlane + 2 * DH * 16>(va);
                } else {
                    vh[0][0] = ldsx_read128<4 * DH * 16>(va); vl[0][0] = ldsx_read128<kXPlane + 4 * DH * 16>(va);
                    vh[1][0] = ldsx_read128<6 * DH * 16>(va); vl[1][0] = ldsx_read128<kXPlane + 6 * DH * 16>(va);
                }
                ldsx_wait4<0>(vh[0][0], vh[1][0], vl[0][0], vl[1][0]);
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                uint4 ph, pl;
                const float* e = reinterpret_cast<const float*>(&acc_s[j]) + 8 * s2;
                ph.x = pack_e2(e[0], e[1]); ph.y = pack_e2(e[2], e[3]);
                ph.z = pack_e2(e[4], e[5]); ph.w = pack_e2(e[6], e[7]);
                pl.x = pack_e2(e[0] - elo(ph.x), e[1] - ehi(ph.x));
                pl.y = pack_e2(e[2] - elo(ph.y), e[3] - ehi(ph.y));
                pl.z = pack_e2(e[4] - elo(ph.z), e[5] - ehi(ph.z));
                pl.w = pack_e2(e[6] - elo(ph.w), e[7] - ehi(ph.w));
                if (p.round_flags & 1) pl = uint4{0u, 0u, 0u, 0u};
                const ex8 pfh = __builtin_bit_cast(ex8, ph), pfl = __builtin_bit_cast(ex8, pl);
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    acc_o[d] = TT_MFMA_32x32x16Z(__builtin_bit_cast(ex8, vl[s2][d]), pfh, acc_o[d], 0, 0, 0);
                    acc_o[d] = TT_MFMA_32x32x16Z(__builtin_bit_cast(ex8, vh[s2][d]), pfl, acc_o[d], 0, 0, 0);
                    acc_o[d] = TT_MFMA_32x32x16Z(__builtin_bit_cast(ex8, vh[s2][d]), pfh, acc_o[d], 0, 0, 0);
                }
            }
        }
    }

    // ---- normalise and store both planes: lane = query row, registers = 4 consecutive d
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q_row < len) {
        uint16_t* op = p.out + (size_t)(t0 + q_row) * p.ld_out + head * DH;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float y0 = acc_o[d][4 * g + 0] * inv, y1 = acc_o[d][4 * g + 1] * inv, y2 = acc_o[d][4 * g + 2] * inv,
                      y3 = acc_o[d][4 * g + 3] * inv;
                asm("" : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3));   // opaque: no fusing of "* inv" into the "y - hi" below (see epilogue_x3)
                uint2 hi, lo;
                hi.x = pack_e2(y0, y1);
                hi.y = pack_e2(y2, y3);
                lo.x = pack_e2(y0 - elo(hi.x), y1 - ehi(hi.x));
                lo.y = pack_e2(y2 - elo(hi.y), y3 - ehi(hi.y));
                if (p.round_flags & 2) lo = uint2{0u, 0u};
                *reinterpret_cast<uint2*>(op + 32 * d + 8 * g + 4 * hh) = hi;
                *reinterpret_cast<uint2*>(op + p.out_lo_off + 32 * d + 8 * g + 4 * hh) = lo;
            }
    }
}

int attention_x3_launch(const AttnX3Params& p, hipStream_t st) {
    if (p.n_seq <= 0 || p.max_len <= 0) return TT_OK;
    if ((p.ld_qk % 8) || (p.q_col0 % 8) || (p.k_col0 % 8) || (p.lo_off % 8) || (p.ldvt % 8) || (p.ld_out % 4) || (p.out_lo_off % 4)) {
        tt_set_error("attention x3: leading dimensions / column offsets must keep 16-byte alignment");
        return TT_E_INVALID;
    }
    const int n_qt = (p.max_len + 32 * kXWaves - 1) / (32 * kXWaves);
    const long long pairs = (long long)p.heads * p.n_seq;
    if (pairs * n_qt > 0x7FFFFFFFLL) {
        tt_set_error("attention x3: %lld workgroups exceed the grid limit", pairs * n_qt);
        return TT_E_UNSUPPORTED;
    }
    AttnX3Params q = p;
    q.lazy = 8.0f;
    q.n_qt = n_qt;
    const int dh = p.head_dim ? p.head_dim : 64;
    if (dh != 64 && dh != 32) {
        tt_set_error("attention x3: head_dim %d (64 or 32)", dh);
        return TT_E_UNSUPPORTED;
    }
    TtProfScope prof(TT_K_ATTENTION, st);
    if (dh == 64) {
        constexpr int kLds = 2 * 4 * kXKTile * 64 * 2;       // 64 KiB: two buffers of K hi | K lo | V hi | V lo
        TT_SET_MAX_LDS(attention_x3_kernel<64>, kLds);
        hipLaunchKernelGGL(attention_x3_kernel<64>, dim3((unsigned)(pairs * n_qt)), dim3(64 * kXWaves), kLds, st, q);
    } else {
        constexpr int kLds = 2 * 4 * kXKTile * 32 * 2;       // 32 KiB
        TT_SET_MAX_LDS(attention_x3_kernel<32>, kLds);
        hipLaunchKernelGGL(attention_x3_kernel<32>, dim3((unsigned)(pairs * n_qt)), dim3(64 * kXWaves), kLds, st, q);
    }
    TT_CHECK_LAUNCH();
    return TT_OK;
}

// ---- CLS-only tail of the last layer (tt_encoder_forward_x3_cls): one wave per (sequence, head), the single query row --------
// The bf16 path's attention_cls_kernel with every operand as hi + lo planes: q, k and v are rebuilt in fp32 (hi + lo carries 16
// mantissa bits), the dot products, the softmax and the value sum are fp32, the context goes out as planes again.
template <int DH>
__global__ __launch_bounds__(64) void attention_cls_x3_kernel(AttnX3Params p) {
    extern __shared__ __attribute__((aligned(16))) float probs_x[];   // [max_len rounded up to 8]
    const int seq = blockIdx.x, head = blockIdx.y, lane = threadIdx.x;
    const int len = p.seq_len[seq], t0 = p.seq_start[seq];
    const int t0a = t0 & ~7, off = t0 - t0a, alen = off + len;
    auto add8 = [](const uint4& h, const uint4& l, float (&f)[8]) {
        f[0] = elo(h.x) + elo(l.x); f[1] = ehi(h.x) + ehi(l.x);
        f[2] = elo(h.y) + elo(l.y); f[3] = ehi(h.y) + ehi(l.y);
        f[4] = elo(h.z) + elo(l.z); f[5] = ehi(h.z) + ehi(l.z);
        f[6] = elo(h.w) + elo(l.w); f[7] = ehi(h.w) + ehi(l.w);
    };
    float q[DH];
    {
        const uint16_t* qp = p.qk + (size_t)t0 * p.ld_qk + p.q_col0 + head * DH;
#pragma unroll
        for (int c = 0; c < DH / 8; ++c) {
            float f[8];
            add8(*reinterpret_cast<const uint4*>(qp + c * 8), *reinterpret_cast<const uint4*>(qp + p.lo_off + c * 8), f);
#pragma unroll
            for (int i = 0; i < 8; ++i) q[c * 8 + i] = f[i];
        }
    }
    const float sc = p.scale * 1.4426950408889634f;
    float mx = -__builtin_inff();
    for (int j = off + lane; j < alen; j += 64) {
        const uint16_t* kp = p.qk + (size_t)(t0a + j) * p.ld_qk + p.k_col0 + head * DH;
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < DH / 8; ++c) {
            float f[8];
            add8(*reinterpret_cast<const uint4*>(kp + c * 8), *reinterpret_cast<const uint4*>(kp + p.lo_off + c * 8), f);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc = fmaf(q[c * 8 + i], f[i], acc);
        }
        acc *= sc;
        probs_x[j] = acc;
        mx = fmaxf(mx, acc);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
    const int len8 = (alen + 7) & ~7;
    for (int j = lane; j < len8; j += 64) {
        float e = 0.f;
        if (j >= off && j < alen) e = __builtin_amdgcn_exp2f(probs_x[j] - mx);
        probs_x[j] = e;
        sum += e;
    }
    sum = wave_sum_x(sum);
    __syncthreads();
    const int d = lane;        // one feature per lane (DH = 32: the upper half of the wave has none)
    if (d >= DH) return;
    float o = 0.f;
    for (int g8 = 0; g8 * 8 < alen; ++g8) {
        const size_t at = (size_t)(t0a / 8 + g8) * p.ldvt + (size_t)(head * DH + d) * 8;
        float v[8];
        add8(*reinterpret_cast<const uint4*>(p.vt + at), *reinterpret_cast<const uint4*>(p.vt_lo + at), v);
        const float4 pa = *reinterpret_cast<const float4*>(probs_x + g8 * 8);
        const float4 pb = *reinterpret_cast<const float4*>(probs_x + g8 * 8 + 4);
        o = fmaf(pa.x, v[0], o); o = fmaf(pa.y, v[1], o); o = fmaf(pa.z, v[2], o); o = fmaf(pa.w, v[3], o);
        o = fmaf(pb.x, v[4], o); o = fmaf(pb.y, v[5], o); o = fmaf(pb.z, v[6], o); o = fmaf(pb.w, v[7], o);
    }
    o /= sum;
    const float hi = rbf(o);
    uint16_t* op = p.out + (size_t)seq * p.ld_out + head * DH + d;
    op[0] = (uint16_t)(pack_e2(hi, 0.f) & 0xFFFFu);
    op[p.out_lo_off] = (uint16_t)(pack_e2(o - hi, 0.f) & 0xFFFFu);
}

// rows seq_start[b] of an fp32 [T][H] matrix -> dst [n_pad][H] (rows beyond n: zeros)
__global__ __launch_bounds__(256) void gather_rows_f32_kernel(const float* src, const int32_t* rows, int n, int n_pad, int H, float* dst) {
    const int b = blockIdx.x;
    if (b >= n_pad) return;
    for (int c = threadIdx.x * 4; c < H; c += 256 * 4) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (b < n) v = *reinterpret_cast<const float4*>(src + (size_t)rows[b] * H + c);
        *reinterpret_cast<float4*>(dst + (size_t)b * H + c) = v;
    }
}

int attention_cls_x3_launch(const AttnX3Params& p, hipStream_t st) {
    if (p.n_seq <= 0) return TT_OK;
    const size_t lds = (size_t)((p.max_len + 14) / 8 * 8) * sizeof(float);
    if (lds > 160 * 1024) {
        tt_set_error("attention_cls_x3: max_len %d exceeds the LDS score buffer", p.max_len);
        return TT_E_UNSUPPORTED;
    }
    const int dh = p.head_dim ? p.head_dim : 64;
    if (dh != 64 && dh != 32) {
        tt_set_error("attention_cls_x3: head_dim %d (64 or 32)", dh);
        return TT_E_UNSUPPORTED;
    }
    TtProfScope prof(TT_K_ATTENTION, st);
    if (dh == 64) {
        TT_SET_MAX_LDS(attention_cls_x3_kernel<64>, 160 * 1024);
        hipLaunchKernelGGL(attention_cls_x3_kernel<64>, dim3(p.n_seq, p.heads), dim3(64), lds, st, p);
    } else {
        TT_SET_MAX_LDS(attention_cls_x3_kernel<32>, 160 * 1024);
        hipLaunchKernelGGL(attention_cls_x3_kernel<32>, dim3(p.n_seq, p.heads), dim3(64), lds, st, p);
    }
    TT_CHECK_LAUNCH();
    return TT_OK;
}

inline int x3_cls_pad(int n_seq) { return n_seq <= 256 ? (n_seq + 63) / 64 * 64 : (n_seq + 255) / 256 * 256; }

// ---- forward ---------------------------------------------------------------------------------------------------------
struct X3Ws {
    size_t off_xa, off_xb, off_y, off_xpl, off_qk, off_vt, off_vtlo, off_ctx, off_ffn, total;
    size_t off_cctx, off_cx, off_cy, off_cx1, off_cxpl, off_cffn;     // CLS tail (n_cls > 0)
};

X3Ws x3_plan(const tt_encoder_weights_x3* w, int n_rows, int n_cls = 0) {
    X3Ws e{};
    const size_t H = (size_t)w->hidden, F = (size_t)w->ffn, T = ((size_t)n_rows + 255) / 256 * 256;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += tt_align_up(bytes, 256); return o; };
    e.off_xa = take(T * H * 4);
    e.off_xb = take(T * H * 4);
    e.off_y = take(T * H * 4);
    e.off_xpl = take(T * 2 * H * 2);
    e.off_qk = take(T * 4 * H * 2);
    e.off_vt = take(T * H * 2);
    e.off_vtlo = take(T * H * 2);
    e.off_ctx = take(T * 2 * H * 2);
    e.off_ffn = take(T * 2 * F * 2);
    if (n_cls > 0) {
        const size_t B = (size_t)x3_cls_pad(n_cls);
        e.off_cctx = take(B * 2 * H * 2);
        e.off_cx = take(B * H * 4);
        e.off_cy = take(B * H * 4);
        e.off_cx1 = take(B * H * 4);
        e.off_cxpl = take(B * 2 * H * 2);
        e.off_cffn = take(B * 2 * F * 2);
    }
    e.total = off;
    return e;
}

int check_weights_x3(const tt_encoder_weights_x3* w) {
    TT_CHECK_ARG(w != nullptr, "null weights");
    // round 6: multiples of 128 (the GEMMs' last column tile may be partial: gemm.hip; the row kernels' last 256-element chunk may be
    // half), 64- or 32-wide heads -- bge-small-en-v1.5 and ms-marco-MiniLM-L-6-v2 (384 = 12 x 32, ffn 1536) run in the reference's precision
    TT_CHECK_ARG(w->hidden > 0 && w->hidden % 128 == 0 && w->hidden <= 1024, "hidden=%d: the split-plane path takes multiples of 128 up to 1024", w->hidden);
    TT_CHECK_ARG(w->heads > 0 && (w->hidden == w->heads * 64 || w->hidden == w->heads * 32),
                 "heads=%d at hidden %d: the split-plane attention takes 64- or 32-wide heads", w->heads, w->hidden);
    TT_CHECK_ARG(w->ffn > 0 && w->ffn % 64 == 0, "ffn=%d must be a multiple of 64", w->ffn);
    TT_CHECK_ARG(w->layers >= 0 && (w->layers == 0 || w->layer != nullptr), "layer array missing");
    TT_CHECK_ARG(w->word_emb && w->pos_emb && w->type_emb && w->emb_ln_g && w->emb_ln_b, "embedding tables missing");
    return TT_OK;
}

inline dim3 row_grid_x(int rows) { return dim3((unsigned)((rows + 3) / 4)); }

}  // namespace

extern "C" {

size_t tt_encoder_x3_workspace_bytes(const tt_encoder_weights_x3* w, int n_rows) {
    if (!w || n_rows <= 0) return 0;
    return x3_plan(w, n_rows).total;
}

}  // extern "C"

namespace {
// hidden_out: the last hidden state [n_rows][H]; cls_out (instead): the last hidden state of every sequence's FIRST row only,
// [x3_cls_pad(n_seq)][H] -- the last layer then runs its attention, output projection, LayerNorms and FFN for those rows only
int forward_x3_impl(const tt_encoder_weights_x3* w, const int32_t* ids, const int32_t* pos, const int32_t* type_ids,
                    const int32_t* seq_start, const int32_t* seq_len, int n_seq, int n_rows, int max_len,
                    float* hidden_out, float* cls_out, void* workspace, size_t workspace_bytes, void* stream) {
    if (int rc = check_weights_x3(w)) return rc;
    TT_CHECK_ARG(n_rows > 0 && (n_rows % 256 == 0 || (n_rows < 256 && n_rows % 64 == 0)),
                 "n_rows=%d must be a positive multiple of 256 (or 64 / 128 / 192: skinny GEMMs)", n_rows);
    TT_CHECK_ARG(n_seq > 0 && max_len > 0, "n_seq=%d max_len=%d", n_seq, max_len);
    TT_CHECK_ARG(ids && pos && seq_start && seq_len && (hidden_out || cls_out), "null pointer");
    const bool cls_tail = cls_out != nullptr && w->layers > 0;
    const X3Ws e = x3_plan(w, n_rows, cls_tail ? n_seq : 0);
    if (!workspace || workspace_bytes < e.total) {
        tt_set_error("tt_encoder_forward_x3: workspace %zu < required %zu bytes", workspace_bytes, e.total);
        return TT_E_WORKSPACE;
    }
    TT_CHECK_ARG(((uintptr_t)workspace % 256) == 0, "workspace must be 256-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    const int H = w->hidden, F = w->ffn, T = n_rows;
    float* xa = (float*)(ws + e.off_xa);
    float* xb = (float*)(ws + e.off_xb);
    float* y = (float*)(ws + e.off_y);
    uint16_t* xpl = (uint16_t*)(ws + e.off_xpl);
    uint16_t* qk = (uint16_t*)(ws + e.off_qk);
    uint16_t* vt = (uint16_t*)(ws + e.off_vt);
    uint16_t* vtlo = (uint16_t*)(ws + e.off_vtlo);
    uint16_t* ctx = (uint16_t*)(ws + e.off_ctx);
    uint16_t* ffn = (uint16_t*)(ws + e.off_ffn);
    TT_CHECK_HIP(hipMemsetAsync(ctx, 0, (size_t)T * 2 * H * 2, st));    // rows of no sequence are never written by attention

    // TT_X3_ROUND_MASK (DIAGNOSTIC LIBRARY ONLY, `make DIAG=1`; default 0, read per forward there): re-introduce ONE of the bf16 path's rounding points at a time
    // into this fp32-grade forward -- the error budget of the bf16 mode (tools/probes/bf16_error_budget.py).  bit 1: Q / K / V
    // projections' outputs, 2: softmax probabilities, 3: attention context, 4: pre-LayerNorm sums (attention output + residual,
    // FFN output + residual), 5: LayerNorm outputs (incl. the embedding LayerNorm; the residual branch reads the rounded copy, as
    // in the bf16 path), 6: FFN intermediate (GELU output).  (Bit 0, bf16 WEIGHTS, is applied where the planes are made:
    // encoder_x3.EncoderWeightsX3(round_weights=True).)
    const int rmask = TT_DIAG_ENV_INT("TT_X3_ROUND_MASK", 0);
    const int ln_out = (rmask & 32) ? 2 : 0, ln_in = (rmask & 16) ? 1 : 0;
    // fp16 planes (22 significand bits; the default reference implementation): the residual branch is REBUILT from the planes of the
    // LayerNorm output, hi + lo (abs. error <= max(2^-22 |x|, 2^-25)), instead of carried as a separate fp32 copy -- a LayerNorm
    // pass then writes 8 bytes per element (two planes) instead of 12 and the residual epilogue reads the same 4.  bf16 planes
    // (16 bits) keep the fp32 copy.  TT_X3_RES_PLANES=0: the fp32 copy, the A/B switch; the rounding diagnostics (rmask) keep it too.
    static const bool res_planes_env = TT_DIAG_ENV_INT("TT_X3_RES_PLANES", 1) != 0;
    const bool res_planes = kF16 && res_planes_env && rmask == 0;
    float* x = (w->layers == 0 && hidden_out) ? hidden_out : xa;
    {
        TtProfScope prof(TT_K_ROWOPS, st);
        hipLaunchKernelGGL(embed_ln_x3_kernel, row_grid_x(T), dim3(kRowThreadsX), 0, st, ids, pos, type_ids, w->word_emb, w->pos_emb,
                           w->type_emb, w->emb_ln_g, w->emb_ln_b, x, xpl, T, H, w->vocab, w->max_pos, w->type_vocab, w->ln_eps, ln_out);
        TT_CHECK_LAUNCH();
    }
    for (int l = 0; l < w->layers; ++l) {
        const tt_layer_weights_x3& lw = w->layer[l];
        TT_CHECK_ARG(lw.qkv_w && lw.qkv_b && lw.o_w && lw.o_b && lw.ln1_g && lw.ln1_b && lw.ffn1_w && lw.ffn1_b && lw.ffn2_w &&
                         lw.ffn2_b && lw.ln2_g && lw.ln2_b, "layer %d has a null weight pointer", l);
        // Q, K columns -> planes [T][4H] (hi at [0, 2H), lo at [2H, 4H)); V columns -> V8 hi / lo
        GemmParams g{};
        g.x3 = 1;
        g.A = xpl; g.lda = 2 * H; g.W = (const uint16_t*)lw.qkv_w; g.ldw = 2 * H; g.bias = lw.qkv_b;
        g.C = qk; g.ldc = 4 * H; g.c_lo_off = 2 * H; g.M = T; g.N = 2 * H; g.K = H;
        g.x3_zero_lo = (rmask & 2) ? 1 : 0;
        if (int rc = tt_gemm_launch(g, TT_EPI_BIAS, st)) return rc;
        GemmParams gv = g;
        gv.W = (const uint16_t*)lw.qkv_w + (size_t)2 * H * 2 * H;
        gv.bias = lw.qkv_b + 2 * H;
        gv.N = H; gv.vt = vt; gv.vt_lo = vtlo; gv.ldvt = 8 * H; gv.vt_col0 = 0;
        if (int rc = tt_gemm_launch(gv, TT_EPI_VT, st)) return rc;
        if (cls_tail && l == w->layers - 1) {
            // ---- last layer, first rows only: one-query attention per (sequence, head), then the output projection, the
            //      LayerNorms and the FFN on n_seq (padded) rows instead of n_rows
            const int Bp = x3_cls_pad(n_seq);
            uint16_t* cctx = (uint16_t*)(ws + e.off_cctx);
            float* cx = (float*)(ws + e.off_cx);
            float* cy = (float*)(ws + e.off_cy);
            float* cx1 = (float*)(ws + e.off_cx1);
            uint16_t* cxpl = (uint16_t*)(ws + e.off_cxpl);
            uint16_t* cffn = (uint16_t*)(ws + e.off_cffn);
            TT_CHECK_HIP(hipMemsetAsync(cctx, 0, (size_t)Bp * 2 * H * 2, st));
            AttnX3Params ac{};
            ac.qk = qk; ac.ld_qk = 4 * H; ac.q_col0 = 0; ac.k_col0 = H; ac.lo_off = 2 * H; ac.vt = vt; ac.vt_lo = vtlo; ac.ldvt = 8 * H;
            ac.out = cctx; ac.ld_out = 2 * H; ac.out_lo_off = H; ac.seq_start = seq_start; ac.seq_len = seq_len;
            ac.n_seq = n_seq; ac.heads = w->heads; ac.max_len = max_len; ac.head_dim = H / w->heads;
            ac.scale = 1.0f / sqrtf((float)ac.head_dim);
            if (int rc = attention_cls_x3_launch(ac, st)) return rc;
            {
                TtProfScope prof(TT_K_ROWOPS, st);
                hipLaunchKernelGGL(gather_rows_f32_kernel, dim3(Bp), dim3(256), 0, st, x, seq_start, n_seq, Bp, H, cx);
                TT_CHECK_LAUNCH();
            }
            GemmParams go{};
            go.x3 = 1;
            go.A = cctx; go.lda = 2 * H; go.W = (const uint16_t*)lw.o_w; go.ldw = 2 * H; go.bias = lw.o_b;
            go.res32 = cx; go.ldr = H; go.C32 = cy; go.ldc = H; go.M = Bp; go.N = H; go.K = H;
            if (int rc = tt_gemm_launch(go, TT_EPI_RESIDUAL, st)) return rc;
            {
                TtProfScope prof(TT_K_ROWOPS, st);
                hipLaunchKernelGGL(layernorm_x3_kernel, row_grid_x(Bp), dim3(kRowThreadsX), 0, st, cy, cx1, cxpl, lw.ln1_g, lw.ln1_b, Bp, H,
                                   w->ln_eps, ln_in | ln_out);
                TT_CHECK_LAUNCH();
            }
            GemmParams g1{};
            g1.x3 = 1;
            g1.A = cxpl; g1.lda = 2 * H; g1.W = (const uint16_t*)lw.ffn1_w; g1.ldw = 2 * H; g1.bias = lw.ffn1_b;
            g1.C = cffn; g1.ldc = 2 * F; g1.c_lo_off = F; g1.M = Bp; g1.N = F; g1.K = H;
            g1.x3_zero_lo = (rmask & 64) ? 1 : 0;
            if (int rc = tt_gemm_launch(g1, TT_EPI_GELU, st)) return rc;
            GemmParams g2{};
            g2.x3 = 1;
            g2.A = cffn; g2.lda = 2 * F; g2.W = (const uint16_t*)lw.ffn2_w; g2.ldw = 2 * F; g2.bias = lw.ffn2_b;
            g2.res32 = cx1; g2.ldr = H; g2.C32 = cy; g2.ldc = H; g2.M = Bp; g2.N = H; g2.K = F;
            if (int rc = tt_gemm_launch(g2, TT_EPI_RESIDUAL, st)) return rc;
            TtProfScope prof(TT_K_ROWOPS, st);
            hipLaunchKernelGGL(layernorm_x3_kernel, row_grid_x(Bp), dim3(kRowThreadsX), 0, st, cy, cls_out, (uint16_t*)nullptr, lw.ln2_g,
                               lw.ln2_b, Bp, H, w->ln_eps, ln_in | ln_out);
            TT_CHECK_LAUNCH();
            return TT_OK;
        }
        AttnX3Params a{};
        a.qk = qk; a.ld_qk = 4 * H; a.q_col0 = 0; a.k_col0 = H; a.lo_off = 2 * H; a.vt = vt; a.vt_lo = vtlo; a.ldvt = 8 * H;
        a.out = ctx; a.ld_out = 2 * H; a.out_lo_off = H; a.seq_start = seq_start; a.seq_len = seq_len;
        a.n_seq = n_seq; a.heads = w->heads; a.max_len = max_len; a.head_dim = H / w->heads;
        a.scale = 1.0f / sqrtf((float)a.head_dim);
        a.round_flags = ((rmask & 4) ? 1 : 0) | ((rmask & 8) ? 2 : 0);
        if (int rc = attention_x3_launch(a, st)) return rc;
        GemmParams go{};
        go.x3 = 1;
        go.A = ctx; go.lda = 2 * H; go.W = (const uint16_t*)lw.o_w; go.ldw = 2 * H; go.bias = lw.o_b;
        go.res32 = x; go.ldr = H; go.C32 = y; go.ldc = H; go.M = T; go.N = H; go.K = H;
        if (res_planes) { go.res32 = nullptr; go.res_planes = xpl; go.res_lo_off = H; go.ldr = 2 * H; }
        if (int rc = tt_gemm_launch(go, TT_EPI_RESIDUAL, st)) return rc;
        float* x1 = (x == xa) ? xb : xa;
        {
            TtProfScope prof(TT_K_ROWOPS, st);
            hipLaunchKernelGGL(layernorm_x3_kernel, row_grid_x(T), dim3(kRowThreadsX), 0, st, y, res_planes ? (float*)nullptr : x1, xpl, lw.ln1_g,
                               lw.ln1_b, T, H, w->ln_eps, ln_in | ln_out);
            TT_CHECK_LAUNCH();
        }
        GemmParams g1{};
        g1.x3 = 1;
        g1.A = xpl; g1.lda = 2 * H; g1.W = (const uint16_t*)lw.ffn1_w; g1.ldw = 2 * H; g1.bias = lw.ffn1_b;
        g1.C = ffn; g1.ldc = 2 * F; g1.c_lo_off = F; g1.M = T; g1.N = F; g1.K = H;
        g1.x3_zero_lo = (rmask & 64) ? 1 : 0;
        if (int rc = tt_gemm_launch(g1, TT_EPI_GELU, st)) return rc;
        GemmParams g2{};
        g2.x3 = 1;
        g2.A = ffn; g2.lda = 2 * F; g2.W = (const uint16_t*)lw.ffn2_w; g2.ldw = 2 * F; g2.bias = lw.ffn2_b;
        g2.res32 = x1; g2.ldr = H; g2.C32 = y; g2.ldc = H; g2.M = T; g2.N = H; g2.K = F;
        if (res_planes) { g2.res32 = nullptr; g2.res_planes = xpl; g2.res_lo_off = H; g2.ldr = 2 * H; }
        if (int rc = tt_gemm_launch(g2, TT_EPI_RESIDUAL, st)) return rc;
        const bool last = l == w->layers - 1;
        float* dst = last ? hidden_out : x;
        // (the fp32 copy is still written where something reads it: the forward's output, and the input of a first-rows-only last layer)
        const bool want32 = !res_planes || last || (cls_tail && l == w->layers - 2);
        {
            TtProfScope prof(TT_K_ROWOPS, st);
            hipLaunchKernelGGL(layernorm_x3_kernel, row_grid_x(T), dim3(kRowThreadsX), 0, st, y, want32 ? dst : (float*)nullptr,
                               last ? (uint16_t*)nullptr : xpl, lw.ln2_g, lw.ln2_b, T, H, w->ln_eps, ln_in | ln_out);
            TT_CHECK_LAUNCH();
        }
        x = dst;
    }
    if (cls_out) {     // no layers: the "last hidden state" is the embedding LayerNorm's output -- gather the first rows
        TtProfScope prof(TT_K_ROWOPS, st);
        const int Bp = x3_cls_pad(n_seq);
        hipLaunchKernelGGL(gather_rows_f32_kernel, dim3(Bp), dim3(256), 0, st, x, seq_start, n_seq, Bp, H, cls_out);
        TT_CHECK_LAUNCH();
    }
    return TT_OK;
}
}  // namespace

extern "C" {

int tt_encoder_forward_x3(const tt_encoder_weights_x3* w, const int32_t* ids, const int32_t* pos, const int32_t* type_ids,
                          const int32_t* seq_start, const int32_t* seq_len, int n_seq, int n_rows, int max_len,
                          float* hidden_out, void* workspace, size_t workspace_bytes, void* stream) {
    TT_CHECK_ARG(hidden_out != nullptr, "null pointer");
    return forward_x3_impl(w, ids, pos, type_ids, seq_start, seq_len, n_seq, n_rows, max_len, hidden_out, nullptr, workspace,
                           workspace_bytes, stream);
}

size_t tt_encoder_x3_cls_workspace_bytes(const tt_encoder_weights_x3* w, int n_rows, int n_seq) {
    if (!w || n_rows <= 0 || n_seq <= 0) return 0;
    return x3_plan(w, n_rows, n_seq).total;
}

int tt_encoder_forward_x3_cls(const tt_encoder_weights_x3* w, const int32_t* ids, const int32_t* pos, const int32_t* type_ids,
                              const int32_t* seq_start, const int32_t* seq_len, int n_seq, int n_rows, int max_len,
                              float* cls_out, void* workspace, size_t workspace_bytes, void* stream) {
    TT_CHECK_ARG(cls_out != nullptr, "null pointer");
    return forward_x3_impl(w, ids, pos, type_ids, seq_start, seq_len, n_seq, n_rows, max_len, nullptr, cls_out, workspace,
                           workspace_bytes, stream);
}

int tt_rerank_head_x3(const tt_encoder_weights_x3* w, const float* hidden_f32, const int32_t* rows, int n_seq, float* scores,
                      float* logits, void* workspace, size_t workspace_bytes, void* stream) {
    TT_CHECK_ARG(w != nullptr, "null weights");
    tt_encoder_weights_f32 h{};     // the head is a [n_seq x H x H] product: the fp32 kernels (f32_path.hip) on fp32 head weights
    h.hidden = w->hidden; h.layers = 0; h.heads = w->heads; h.ffn = w->ffn; h.vocab = w->vocab; h.max_pos = w->max_pos;
    h.type_vocab = w->type_vocab; h.ln_eps = w->ln_eps;
    h.word_emb = w->word_emb; h.pos_emb = w->pos_emb; h.type_emb = w->type_emb; h.emb_ln_g = w->emb_ln_g; h.emb_ln_b = w->emb_ln_b;
    h.cls_dense_w = w->cls_dense_w; h.cls_dense_b = w->cls_dense_b; h.cls_out_w = w->cls_out_w; h.cls_out_b = w->cls_out_b;
    return tt_rerank_head_f32(&h, hidden_f32, rows, n_seq, scores, logits, workspace, workspace_bytes, stream);
}

/* building blocks for the parity tests */
int tt_split_planes(const float* in_f32, int64_t rows, int cols, void* out_planes, void* stream) {
    TT_CHECK_ARG(in_f32 && out_planes && rows >= 0 && cols > 0 && cols % 4 == 0, "bad argument");
    if (rows == 0) return TT_OK;
    const int64_t n4 = rows * (cols / 4);
    hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in_f32,
                       (uint16_t*)out_planes, rows, cols);
    TT_CHECK_LAUNCH();
    return TT_OK;
}

int tt_gemm_x3(const void* a_planes, const void* w_planes, const float* bias, const float* residual_f32, void* c_planes,
               float* c_f32, int m, int n, int k, int epilogue, void* stream) {
    TT_CHECK_ARG(a_planes && w_planes && bias, "null pointer");
    TT_CHECK_ARG(epilogue == TT_EPI_BIAS || epilogue == TT_EPI_GELU || epilogue == TT_EPI_RESIDUAL, "epilogue %d", epilogue);
    GemmParams g{};
    g.x3 = 1;
    g.A = (const uint16_t*)a_planes; g.lda = 2 * k; g.W = (const uint16_t*)w_planes; g.ldw = 2 * k; g.bias = bias;
    g.M = m; g.N = n; g.K = k;
    if (epilogue == TT_EPI_RESIDUAL) {
        TT_CHECK_ARG(residual_f32 && c_f32, "residual epilogue: fp32 residual and fp32 output");
        g.res32 = residual_f32; g.ldr = n; g.C32 = c_f32; g.ldc = n;
    } else {
        TT_CHECK_ARG(c_planes, "planes output missing");
        g.C = (uint16_t*)c_planes; g.ldc = 2 * n; g.c_lo_off = n;
    }
    return tt_gemm_launch(g, epilogue, (hipStream_t)stream);
}

int tt_attention_x3(const void* qk_planes, int ld_qk, int q_col0, int k_col0, int lo_off, const void* vt_hi, const void* vt_lo,
                    int ldvt, void* out_planes, int ld_out, int out_lo_off, const int32_t* seq_start, const int32_t* seq_len,
                    int n_seq, int heads, int max_len, void* stream) {
    TT_CHECK_ARG(qk_planes && vt_hi && vt_lo && out_planes && seq_start && seq_len, "null pointer");
    AttnX3Params a{};
    a.qk = (const uint16_t*)qk_planes; a.ld_qk = ld_qk; a.q_col0 = q_col0; a.k_col0 = k_col0; a.lo_off = lo_off;
    a.vt = (const uint16_t*)vt_hi; a.vt_lo = (const uint16_t*)vt_lo; a.ldvt = ldvt;
    a.out = (uint16_t*)out_planes; a.ld_out = ld_out; a.out_lo_off = out_lo_off;
    a.seq_start = seq_start; a.seq_len = seq_len; a.n_seq = n_seq; a.heads = heads; a.max_len = max_len; a.scale = 0.125f;
    a.head_dim = 64;           // (the building block of the parity tests: 64-wide heads; 32-wide ones are tested through the forward)
    return attention_x3_launch(a, (hipStream_t)stream);
}

}  // extern "C"
