// Variable-length bidirectional self-attention (flash-style, online softmax), gfx950 MFMA.
//
//   ctx[t][h*dh + d] = sum_k softmax_k(Q[t].K[k] / sqrt(dh)) V[k][d]   over the keys of t's sequence
//
// Replaces the scaled-dot-product attention inside the reference's XLM-R / BERT encoder
// layers (additive -inf key-padding mask == attending only to the sequence's own tokens
// in the packed varlen layout).
//
// One workgroup = 4 waves = 128 query rows of one (sequence, head); keys stream through LDS
// in tiles of 64.  Both products run "swapped" on v_mfma_f32_32x32x16_bf16 so that the
// QUERY sits on the lane for the whole kernel:
//   S^T[key][q] = K . Q^T      A = K fragment (ds_read_b128, XOR-swizzled), B = Q fragment (registers)
//   O^T[d][q]  += V^T . P^T    A = V^T fragment (2 x ds_read_b64 from a padded [d][key] image),
//                              B = the S^T accumulator itself, converted to bf16 in place
// so the softmax row statistics are lane-local (one cross-half exchange per tile), P never
// goes through LDS, and V is consumed from the token-blocked transposed copy the QKV GEMM
// epilogue wrote (V8: [token/8][feature][8]: a 64-key x dh tile is 8 contiguous 1-KiB runs).
#include "common.h"
#include "encoder.h"

namespace {

constexpr int kAttThreads = 256;
constexpr int kQTile = 128;   // query rows per workgroup (32 per wave)
constexpr int kKTile = 64;    // keys per LDS tile
constexpr int kVtStride = kKTile * 2 + 8;  // bytes per V^T row in LDS (padded: conflict-free ds_read_b64)

template <int DH>
__global__ __launch_bounds__(kAttThreads) void attention_kernel(AttnParams p) {
    constexpr int RB = DH * 2;              // bytes per K row
    constexpr int CH = RB / 16;             // 16-B chunks per K row
    constexpr int RPB = 256 / RB;           // K rows per 256-B bank row
    constexpr int KS = DH / 16;             // k-steps of Q.K
    constexpr int DT = DH / 32;             // 32-row d tiles of O^T
    __shared__ __attribute__((aligned(16))) char k_lds[kKTile * RB];
    __shared__ __attribute__((aligned(16))) char vt_lds[DH * kVtStride];

    const int seq = blockIdx.z, head = blockIdx.y, qt = blockIdx.x;
    const int len = p.seq_len[seq];
    if (qt * kQTile >= len) return;
    const int t0 = p.seq_start[seq];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & 31, hh = lane >> 5;

    // ---- Q fragments (B operand), straight from global ---------------------------------
    const int q_row = qt * kQTile + wave * 32 + ql;        // row inside the sequence
    const int q_row_c = q_row < len ? q_row : len - 1;     // clamp: result discarded
    const uint16_t* qp = p.qk + (size_t)(t0 + q_row_c) * p.ld_qk + p.q_col0 + head * DH + hh * 8;
    bf16x8 qf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qp + s * 16);

    f32x16 acc_o[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[d][r] = 0.f;
    float m_run = -__builtin_inff();
    float l_run = 0.f;
    const float sc = p.scale * 1.4426950408889634f;  // fold log2(e): softmax via exp2

    const int n_kt = (len + kKTile - 1) / kKTile;
    // waves whose 32 query rows all lie beyond the sequence only help staging the tiles
    const bool wave_active = qt * kQTile + wave * 32 < len;   // wave-uniform

    // ---- staging: each thread moves KPT 16-B pieces of the K tile and VPT of the V^T tile;
    // the NEXT tile's pieces are loaded into registers before the current tile is computed and
    // written to LDS after it (global latency hidden behind the MFMA / softmax work).
    constexpr int KPT = kKTile * CH / kAttThreads;          // 2 (dh 64) or 1 (dh 32)
    constexpr int VPT = DH * (kKTile / 8) / kAttThreads;    // 2 (dh 64) or 1 (dh 32)
    uint4 kreg[KPT], vreg[VPT];
    auto load_tile = [&](int kt) {
        const int k0 = kt * kKTile;
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const int piece = tid + i * kAttThreads;
            const int r = piece / CH, c = piece % CH;
            kreg[i] = make_uint4(0, 0, 0, 0);
            if (k0 + r < len)
                kreg[i] = *reinterpret_cast<const uint4*>(p.qk + (size_t)(t0 + k0 + r) * p.ld_qk + p.k_col0 + head * DH + c * 8);
        }
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int piece = tid + i * kAttThreads;
            const int d = piece % DH, c = piece / DH;       // consecutive threads -> consecutive features (16 B apart)
            vreg[i] = make_uint4(0, 0, 0, 0);
            if (k0 + c * 8 < len)
                vreg[i] = *reinterpret_cast<const uint4*>(p.vt + (size_t)((t0 + k0) / 8 + c) * p.ldvt + (size_t)(head * DH + d) * 8);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const int piece = tid + i * kAttThreads;
            const int r = piece / CH, c = piece % CH;
            *reinterpret_cast<uint4*>(k_lds + r * RB + ((c ^ ((r / RPB) & (CH - 1))) << 4)) = kreg[i];
        }
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int piece = tid + i * kAttThreads;
            const int d = piece % DH, c = piece / DH;
            uint2* dst = reinterpret_cast<uint2*>(vt_lds + d * kVtStride + c * 16);
            dst[0] = make_uint2(vreg[i].x, vreg[i].y);
            dst[1] = make_uint2(vreg[i].z, vreg[i].w);
        }
    };

    load_tile(0);
    for (int kt = 0; kt < n_kt; ++kt) {
        const int k0 = kt * kKTile;
        if (kt > 0) __syncthreads();          // everyone is done reading the previous tile
        store_tile();
        if (kt + 1 < n_kt) load_tile(kt + 1);  // in flight while this tile is computed
        __syncthreads();

        if (!wave_active) continue;
        // ---- S^T = K . Q^T for two 32-key tiles -------------------------------------------------
        f32x16 acc_s[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_s[j][r] = 0.f;
            const int key = 32 * j + ql;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int chunk = 2 * s + hh;
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(k_lds + key * RB + ((chunk ^ ((key / RPB) & (CH - 1))) << 4));
                acc_s[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], acc_s[j], 0, 0, 0);
            }
        }
        // ---- mask the tail, running max, exponentials --------------------------------------------
        // The softmax scale (and log2 e) is folded into one FMA per score: p = 2^(s*sc - m), with m
        // tracked in the scaled domain; v_exp_f32 is used raw (arguments are <= 0, a result that
        // underflows is 0 either way), the libm exp2f wraps it in 5 more instructions per value.
        const bool tail = k0 + kKTile > len;  // wave-uniform
        float mx = -__builtin_inff();
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (tail) {
                    const int key = k0 + 32 * j + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    if (key >= len) acc_s[j][r] = -__builtin_inff();
                }
                mx = fmaxf(mx, acc_s[j][r]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx * sc);       // finite: every tile holds >= 1 valid key
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // first tile: 2^-inf = 0
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
        if (!__all(alpha == 1.0f)) {          // the running max rarely moves after the first tiles
#pragma unroll
            for (int d = 0; d < DT; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc_o[d][r] *= alpha;
        }

        // ---- O^T += V^T . P^T ----------------------------------------------------------------------
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                uint4 pb;
                pb.x = pack_bf16x2(acc_s[j][8 * s2 + 0], acc_s[j][8 * s2 + 1]);
                pb.y = pack_bf16x2(acc_s[j][8 * s2 + 2], acc_s[j][8 * s2 + 3]);
                pb.z = pack_bf16x2(acc_s[j][8 * s2 + 4], acc_s[j][8 * s2 + 5]);
                pb.w = pack_bf16x2(acc_s[j][8 * s2 + 6], acc_s[j][8 * s2 + 7]);
                const bf16x8 pf = __builtin_bit_cast(bf16x8, pb);
                const int kb = 32 * j + 16 * s2 + 4 * hh;   // keys kb..kb+3 and kb+8..kb+11
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    const char* vrow = vt_lds + (32 * d + ql) * kVtStride + kb * 2;
                    const uint2 lo = *reinterpret_cast<const uint2*>(vrow);
                    const uint2 hi = *reinterpret_cast<const uint2*>(vrow + 16);
                    const uint4 vv = make_uint4(lo.x, lo.y, hi.x, hi.y);
                    acc_o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vv), pf, acc_o[d], 0, 0, 0);
                }
            }
    }

    // ---- normalise and store: lane = query row, registers = 4 consecutive d ------------------
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q_row < len) {
        uint16_t* op = p.out + (size_t)(t0 + q_row) * p.ld_out + head * DH;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 o;
                o.x = pack_bf16x2(acc_o[d][4 * g + 0] * inv, acc_o[d][4 * g + 1] * inv);
                o.y = pack_bf16x2(acc_o[d][4 * g + 2] * inv, acc_o[d][4 * g + 3] * inv);
                *reinterpret_cast<uint2*>(op + 32 * d + 8 * g + 4 * hh) = o;
            }
    }
}


// ---- CLS-only attention (last layer): one wave per (sequence, head), the single query row t0 -------------
// Scores with the key on the lane (fp32 dot products, K rows read as 16-B pieces), softmax across the wave,
// then the value sum with the FEATURE on the lane (V8 layout: one 16-B load brings 8 keys of a feature).
// Only the CLS token of the last layer is ever read by the pooling / classification head, so the other
// rows' attention (and their output projection / FFN) is skipped: 1/24 of the encoder's flops.
template <int DH>
__global__ __launch_bounds__(64) void attention_cls_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) float probs[];   // [max_len rounded up to 8]
    const int seq = blockIdx.x, head = blockIdx.y, lane = threadIdx.x;
    const int len = p.seq_len[seq], t0 = p.seq_start[seq];
    // query row -> registers (every lane holds the whole q: uniform address)
    float q[DH];
    {
        const uint16_t* qp = p.qk + (size_t)t0 * p.ld_qk + p.q_col0 + head * DH;
#pragma unroll
        for (int c = 0; c < DH / 8; ++c) {
            const uint4 u = *reinterpret_cast<const uint4*>(qp + c * 8);
            q[c * 8 + 0] = __uint_as_float(u.x << 16); q[c * 8 + 1] = __uint_as_float(u.x & 0xFFFF0000u);
            q[c * 8 + 2] = __uint_as_float(u.y << 16); q[c * 8 + 3] = __uint_as_float(u.y & 0xFFFF0000u);
            q[c * 8 + 4] = __uint_as_float(u.z << 16); q[c * 8 + 5] = __uint_as_float(u.z & 0xFFFF0000u);
            q[c * 8 + 6] = __uint_as_float(u.w << 16); q[c * 8 + 7] = __uint_as_float(u.w & 0xFFFF0000u);
        }
    }
    const float sc = p.scale * 1.4426950408889634f;
    float mx = -__builtin_inff();
    for (int j = lane; j < len; j += 64) {
        const uint16_t* kp = p.qk + (size_t)(t0 + j) * p.ld_qk + p.k_col0 + head * DH;
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < DH / 8; ++c) {
            const uint4 u = *reinterpret_cast<const uint4*>(kp + c * 8);
            acc = fmaf(q[c * 8 + 0], __uint_as_float(u.x << 16), acc);
            acc = fmaf(q[c * 8 + 1], __uint_as_float(u.x & 0xFFFF0000u), acc);
            acc = fmaf(q[c * 8 + 2], __uint_as_float(u.y << 16), acc);
            acc = fmaf(q[c * 8 + 3], __uint_as_float(u.y & 0xFFFF0000u), acc);
            acc = fmaf(q[c * 8 + 4], __uint_as_float(u.z << 16), acc);
            acc = fmaf(q[c * 8 + 5], __uint_as_float(u.z & 0xFFFF0000u), acc);
            acc = fmaf(q[c * 8 + 6], __uint_as_float(u.w << 16), acc);
            acc = fmaf(q[c * 8 + 7], __uint_as_float(u.w & 0xFFFF0000u), acc);
        }
        acc *= sc;
        probs[j] = acc;
        mx = fmaxf(mx, acc);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    float sum = 0.f;
    const int len8 = (len + 7) & ~7;
    for (int j = lane; j < len8; j += 64) {
        float e = 0.f;
        if (j < len) e = __builtin_amdgcn_exp2f(probs[j] - mx);
        probs[j] = e;      // keys beyond len (same 8-group) get probability 0
        sum += e;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    __syncthreads();
    // value sum: lane = feature (DH=64) or (feature, key-group parity) (DH=32)
    constexpr int PARTS = 64 / DH;
    const int d = lane % DH, part = lane / DH;
    float o = 0.f;
    for (int g8 = part; g8 * 8 < len; g8 += PARTS) {
        const uint4 u = *reinterpret_cast<const uint4*>(p.vt + (size_t)(t0 / 8 + g8) * p.ldvt + (size_t)(head * DH + d) * 8);
        const float4 pa = *reinterpret_cast<const float4*>(probs + g8 * 8);
        const float4 pb = *reinterpret_cast<const float4*>(probs + g8 * 8 + 4);
        o = fmaf(pa.x, __uint_as_float(u.x << 16), o); o = fmaf(pa.y, __uint_as_float(u.x & 0xFFFF0000u), o);
        o = fmaf(pa.z, __uint_as_float(u.y << 16), o); o = fmaf(pa.w, __uint_as_float(u.y & 0xFFFF0000u), o);
        o = fmaf(pb.x, __uint_as_float(u.z << 16), o); o = fmaf(pb.y, __uint_as_float(u.z & 0xFFFF0000u), o);
        o = fmaf(pb.z, __uint_as_float(u.w << 16), o); o = fmaf(pb.w, __uint_as_float(u.w & 0xFFFF0000u), o);
    }
    if constexpr (PARTS == 2) o += __shfl_xor(o, 32, 64);
    if (part == 0) p.out[(size_t)seq * p.ld_out + head * DH + d] = f32_to_bf16_bits(o / sum);
}

}  // namespace

int tt_attention_launch(const AttnParams& p, hipStream_t st) {
    if (p.n_seq <= 0 || p.max_len <= 0) return TT_OK;
    if ((p.ld_qk % 8) || (p.q_col0 % 8) || (p.k_col0 % 8) || (p.ldvt % 8) || (p.ld_out % 4)) {
        tt_set_error("attention: leading dimensions / column offsets must keep 16-byte alignment");
        return TT_E_INVALID;
    }
    const dim3 grid((p.max_len + kQTile - 1) / kQTile, p.heads, p.n_seq);
    TtProfScope prof(TT_K_ATTENTION, st);
    if (p.head_dim == 64) {
        hipLaunchKernelGGL(attention_kernel<64>, grid, dim3(kAttThreads), 0, st, p);
    } else if (p.head_dim == 32) {
        hipLaunchKernelGGL(attention_kernel<32>, grid, dim3(kAttThreads), 0, st, p);
    } else {
        tt_set_error("attention: head_dim %d not in {32, 64}", p.head_dim);
        return TT_E_UNSUPPORTED;
    }
    TT_CHECK_LAUNCH();
    return TT_OK;
}

// out: [n_seq][ld_out] (row = sequence index, NOT token row)
int tt_attention_cls_launch(const AttnParams& p, hipStream_t st) {
    if (p.n_seq <= 0) return TT_OK;
    const size_t lds = (size_t)((p.max_len + 7) / 8 * 8) * sizeof(float);
    if (lds > 160 * 1024) {
        tt_set_error("attention_cls: max_len %d exceeds the LDS score buffer", p.max_len);
        return TT_E_UNSUPPORTED;
    }
    TtProfScope prof(TT_K_ATTENTION, st);
    const dim3 grid(p.n_seq, p.heads);
    if (p.head_dim == 64) {
        static thread_local bool a64 = false;
        if (!a64) { TT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attention_cls_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); a64 = true; }
        hipLaunchKernelGGL(attention_cls_kernel<64>, grid, dim3(64), lds, st, p);
    } else if (p.head_dim == 32) {
        static thread_local bool a32 = false;
        if (!a32) { TT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attention_cls_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); a32 = true; }
        hipLaunchKernelGGL(attention_cls_kernel<32>, grid, dim3(64), lds, st, p);
    } else {
        tt_set_error("attention: head_dim %d not in {32, 64}", p.head_dim);
        return TT_E_UNSUPPORTED;
    }
    TT_CHECK_LAUNCH();
    return TT_OK;
}
