// bf16 GEMM with fused epilogues for the encoder layers, gfx950 MFMA.
//
//   C[M,N] = epi(A[M,K] . W[N,K]^T + bias[N])      A, W, C bf16 row-major, fp32 accumulate
//
// W is the HF nn.Linear weight as stored ([out, in]), so both operands are K-contiguous and
// every MFMA fragment is one 16-byte read.  These are the dense contractions of the
// reference's bi-encoder / cross-encoder forward (transformers XLMRobertaLayer / BertLayer:
// QKV, attention output, FFN up, FFN down).
//
// Two kernels: gemm_kernel_v3 (256x256x64 tile, 8 waves, ping-pong LDS/MFMA slots, see "v3" below) for shapes
// divisible by 256 -- every GEMM of the 1024-wide models -- and gemm_kernel (v1: 128x128x64 tile, 4 waves x
// 64x64, 2 blocks per CU, one barrier per K-step) for the rest (bge-small's 384 / 1152 / 1536 columns).
// Both stage operands with global_load_lds_dwordx4 into [row][64] bf16 tiles whose 16-B slots are
// XOR-swizzled on the SOURCE address, issue the MFMA "swapped" (a = W fragment, b = A fragment) so a lane
// ends up with consecutive N-columns of one row, and order blocks XCD-contiguously in super-tiles.
// A 256x128 3-stage counted-vmcnt variant of v1 (the former v2) was measured no faster and was dropped.
//
// Roofline: MFMA-bound; 2*M*N*K flops per launch.
#include "common.h"
#include <type_traits>

#include "encoder.h"
#include "f16c.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int kGemmThreads = 256;
constexpr int kTileBytes = BM * BK * 2;          // 16 KiB per operand tile
constexpr int kStageBytes = 2 * kTileBytes;      // A + W
constexpr int kGemmLds = 2 * kStageBytes;        // 64 KiB

// GELU(x) = 0.5 x (1 + erf(x / sqrt 2)), exact-erf form (HF "gelu"), written as
//   gelu(x) = max(x, 0) - |x| Phi(-|x|),   Phi(-u) = 2^Q(u),
// Q = degree-7 minimax fit of log2 Phi(-u) on [0, 9] (leading coefficient negative, so the term underflows to 0
// beyond the fitted range).  Relative error of the correction term <= 7.5e-5 everywhere (abs error of the result
// <= 1e-5; the bf16 output resolves 2e-3 relative), and it costs 7 FMAs + one v_exp_f32 per value: the previous
// Abramowitz-Stegun erf (rcp + exp + 14 plain ops) made the FFN-up epilogue 10 % of that GEMM; libm erff 25 %.
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
    const f32x2 u = f32x2{fabsf(x.x), fabsf(x.y)};
    f32x2 q = u * -9.697845371e-07f + 4.016999810e-05f;
    q = q * u + -7.211678312e-04f;
    q = q * u + 7.490924560e-03f;
    q = q * u + -5.142170191e-02f;
    q = q * u + -4.614778757e-01f;
    q = q * u + -1.149914980e+00f;
    q = q * u + -1.000091195e+00f;
    const f32x2 e = f32x2{__builtin_amdgcn_exp2f(q.x), __builtin_amdgcn_exp2f(q.y)};
    const f32x2 r = f32x2{fmaxf(x.x, 0.f), fmaxf(x.y, 0.f)};
    return r - u * e;
}
__device__ __forceinline__ float gelu_erf(float x) {
    const f32x2 y = gelu_erf2(f32x2{x, x});
    return y.x;
}

// ---- epilogue of one wave's 64x64 tile: lane holds C[m][n..n+3], m = tile row (l&15), n = 4*(l>>4)
template <int EPI, int NT, int MT>
__device__ __forceinline__ void gemm_epilogue_tile(const GemmParams& p, f32x4 (&acc)[NT][MT], int mw, int nw, int lane) {
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int n = nw + i * 16 + (lane >> 4) * 4;
        const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n);
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int m = mw + j * 16 + (lane & 15);
            float v0 = acc[i][j][0] + b4.x, v1 = acc[i][j][1] + b4.y, v2 = acc[i][j][2] + b4.z, v3 = acc[i][j][3] + b4.w;
            if constexpr (EPI == TT_EPI_GELU) {
                v0 = gelu_erf(v0); v1 = gelu_erf(v1); v2 = gelu_erf(v2); v3 = gelu_erf(v3);
            } else if constexpr (EPI == TT_EPI_TANH) {
                v0 = tanhf(v0); v1 = tanhf(v1); v2 = tanhf(v2); v3 = tanhf(v3);
            } else if constexpr (EPI == TT_EPI_RESIDUAL) {
                const uint2 r = *reinterpret_cast<const uint2*>(p.residual + (size_t)m * p.ldr + n);
                v0 += elo(r.x);
                v1 += ehi(r.x);
                v2 += elo(r.y);
                v3 += ehi(r.y);
            }
            if constexpr (EPI == TT_EPI_QKV) {
                if (n >= p.vt_col0) {
                    // V third: store transposed, VT[n - vt_col0][m]
                    // V8 layout: vt[(m / 8) * ldvt + feature * 8 + m % 8]
                    uint16_t* vt = p.vt + (size_t)(m >> 3) * p.ldvt + (size_t)(n - p.vt_col0) * 8 + (m & 7);
                    vt[0] = f32_to_ebits(v0);
                    vt[8] = f32_to_ebits(v1);
                    vt[16] = f32_to_ebits(v2);
                    vt[24] = f32_to_ebits(v3);
                    continue;
                }
            }
            uint2 o;
            o.x = pack_e2(v0, v1);
            o.y = pack_e2(v2, v3);
            *reinterpret_cast<uint2*>(p.C + (size_t)m * p.ldc + n) = o;
        }
    }
}

// (split planes carry the element type of the build: two bf16 planes in the bf16 instantiation -- "bf16x3" --, two fp16 planes
// in the fp16 one -- "f16x3")
template <bool X3>
__device__ __forceinline__ uint32_t pack_sel(float lo, float hi) {
    return pack_e2(lo, hi);
}

template <int EPI>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x4 (&acc)[4][4], int mw, int nw, int lane) {
    gemm_epilogue_tile<EPI, 4, 4>(p, acc, mw, nw, lane);
}

// ---- wide epilogue: 16-byte stores -----------------------------------------------------------
// In the swapped accumulator layout a lane holds 4 consecutive columns of one row (8-B stores, and a
// row-per-lane store tail is store-ISSUE bound: 32 dwordx2 per lane cost ~10 us per 256x256 tile).
// v_permlane16_swap between the register of m-tile 2j (vdst) and of m-tile 2j+1 (src) hands the
// lanes of even 16-lane rows their right-hand neighbour's 4 columns of m-tile 2j and the lanes of
// odd rows their left-hand neighbour's 4 columns of m-tile 2j+1: every lane then owns 8 consecutive
// columns of ONE row -> one 16-B store (and 16-B residual load) per pair of tiles.
// CM (experiment, TT_GEMM_ABLATE=7): store feature-chunk-major, C[(n / 8) * M + m][8] -- every lane's 16-byte chunk goes out
// directly (256-byte runs per 16-lane group), no LDS transposition; times what a chunk-major activation layout would cost.
template <int EPI, int NT, int MT, bool CM = false>
__device__ __forceinline__ void gemm_epilogue_wide(const GemmParams& p, f32x4 (&acc)[NT][MT], int mw, int nw, int lane) {
    static_assert(MT % 2 == 0, "m-tiles are processed in pairs");
    const int g = lane >> 4;           // 16-lane row: columns 4g..4g+3 of the tile before the swap
    const bool odd = (g & 1) != 0;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        if constexpr (EPI == TT_EPI_QKV) {
            if (nw + i * 16 >= p.vt_col0) {   // V third (block-uniform): transposed 2-byte stores, narrow path
#pragma unroll
                for (int j = 0; j < MT; ++j) {
                    const int n = nw + i * 16 + g * 4;
                    const int m = mw + j * 16 + (lane & 15);
                    const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n);
                    uint16_t* vt = p.vt + (size_t)(m >> 3) * p.ldvt + (size_t)(n - p.vt_col0) * 8 + (m & 7);
                    vt[0] = f32_to_ebits(acc[i][j][0] + b4.x);
                    vt[8] = f32_to_ebits(acc[i][j][1] + b4.y);
                    vt[16] = f32_to_ebits(acc[i][j][2] + b4.z);
                    vt[24] = f32_to_ebits(acc[i][j][3] + b4.w);
                }
                continue;
            }
        }
        const int n = nw + i * 16 + (g & ~1) * 4;             // first of this lane's 8 columns
        const float4 ba = *reinterpret_cast<const float4*>(p.bias + n);
        const float4 bb = *reinterpret_cast<const float4*>(p.bias + n + 4);
        const float bias8[8] = {ba.x, ba.y, ba.z, ba.w, bb.x, bb.y, bb.z, bb.w};
#pragma unroll
        for (int j = 0; j < MT; j += 2) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[i][j][k]),
                                                                __float_as_uint(acc[i][j + 1][k]), false, false);
                v[k] = __uint_as_float(r[0]);
                v[4 + k] = __uint_as_float(r[1]);
            }
            const int m = mw + (odd ? j + 1 : j) * 16 + (lane & 15);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += bias8[k];
            if constexpr (EPI == TT_EPI_GELU) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = gelu_erf(v[k]);
            } else if constexpr (EPI == TT_EPI_TANH) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = tanhf(v[k]);
            } else if constexpr (EPI == TT_EPI_RESIDUAL) {
                const uint4 r = *reinterpret_cast<const uint4*>(p.residual + (size_t)m * p.ldr + n);
                v[0] += elo(r.x); v[1] += ehi(r.x);
                v[2] += elo(r.y); v[3] += ehi(r.y);
                v[4] += elo(r.z); v[5] += ehi(r.z);
                v[6] += elo(r.w); v[7] += ehi(r.w);
            }
            uint4 o;
            o.x = pack_e2(v[0], v[1]); o.y = pack_e2(v[2], v[3]);
            o.z = pack_e2(v[4], v[5]); o.w = pack_e2(v[6], v[7]);
            if constexpr (CM) *reinterpret_cast<uint4*>(p.C + ((size_t)(n >> 3) * p.M + m) * 8) = o;
            else *reinterpret_cast<uint4*>(p.C + (size_t)m * p.ldc + n) = o;
        }
    }
}

// ---- V^T epilogue (un-swapped accumulators: lane = feature column l&15, registers = 4 consecutive tokens)
// permlane16_swap between the registers of n-tile 2i and 2i+1 gives every lane 8 consecutive tokens of one
// feature: one 16-B store into the V8 buffer ([token/8][feature][8 tokens]) instead of eight 2-byte stores.
template <int NT, int MT, bool FP8 = false, bool X3 = false>
__device__ __forceinline__ void gemm_epilogue_vt(const GemmParams& p, f32x4 (&acc)[NT][MT], int mw, int nw, int lane) {
    static_assert(NT % 2 == 0, "n-tiles are processed in pairs");
    const int g = lane >> 4;
    const bool odd = (g & 1) != 0;
#pragma unroll
    for (int i = 0; i < NT; i += 2) {
        const int n = nw + (odd ? i + 1 : i) * 16 + (lane & 15);
        const float b = p.bias[n];
        float sw = 1.f;
        if constexpr (FP8) sw = p.w_scale[n];
        uint16_t* col = p.vt + (size_t)(n - p.vt_col0) * 8;   // V8 layout: + (m / 8) * ldvt
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            float v[8];
            const int m = mw + j * 16 + (g & ~1) * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[i][j][k]),
                                                                __float_as_uint(acc[i + 1][j][k]), false, false);
                v[k] = __uint_as_float(r[0]);
                v[4 + k] = __uint_as_float(r[1]);
            }
            if constexpr (FP8) {
                const float4 a0 = *reinterpret_cast<const float4*>(p.a_scale + m);
                const float4 a1 = *reinterpret_cast<const float4*>(p.a_scale + m + 4);
                v[0] *= a0.x * sw; v[1] *= a0.y * sw; v[2] *= a0.z * sw; v[3] *= a0.w * sw;
                v[4] *= a1.x * sw; v[5] *= a1.y * sw; v[6] *= a1.z * sw; v[7] *= a1.w * sw;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += b;
            uint4 o;
            o.x = pack_sel<X3>(v[0], v[1]); o.y = pack_sel<X3>(v[2], v[3]);
            o.z = pack_sel<X3>(v[4], v[5]); o.w = pack_sel<X3>(v[6], v[7]);
            *reinterpret_cast<uint4*>(col + (size_t)(m >> 3) * p.ldvt) = o;
            if constexpr (X3) {   // lo plane: what the bf16 rounding of the hi plane left behind
                uint4 l;
                l.x = pack_e2(v[0] - elo(o.x), v[1] - ehi(o.x));
                l.y = pack_e2(v[2] - elo(o.y), v[3] - ehi(o.y));
                l.z = pack_e2(v[4] - elo(o.z), v[5] - ehi(o.z));
                l.w = pack_e2(v[6] - elo(o.w), v[7] - ehi(o.w));
                if (p.x3_zero_lo) l = uint4{0u, 0u, 0u, 0u};
                *reinterpret_cast<uint4*>(p.vt_lo + (size_t)(n - p.vt_col0) * 8 + (size_t)(m >> 3) * p.ldvt) = l;
            }
        }
    }
}

// issue the global->LDS copies of one K-step (both operand tiles) for this wave
__device__ __forceinline__ void stage_tile(const GemmParams& p, char* stage, int wave, int lane, int m0, int n0, int k0) {
    const int lrow = lane >> 3;   // row inside the 8-row piece
    const int slot = lane & 7;    // 16-B slot inside the 128-B row
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = 32 * wave + 8 * j + lrow;
        const int chunk = slot ^ ((row >> 1) & 7);
        const uint16_t* src = p.A + (size_t)(m0 + row) * p.lda + k0 + chunk * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(stage + (32 * wave + 8 * j) * 128),
                                         16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = 32 * wave + 8 * j + lrow;
        const int chunk = slot ^ ((row >> 1) & 7);
        const uint16_t* src = p.W + (size_t)(n0 + row) * p.K + k0 + chunk * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(stage + kTileBytes + (32 * wave + 8 * j) * 128),
                                         16, 0, 0);
    }
}

template <int EPI>
__global__ __launch_bounds__(kGemmThreads, 2) void gemm_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // ---- block -> tile: XCD-contiguous ranges, then SMxSN super-tiles -----------------
    const int mt_n = p.M / BM, nt_n = p.N / BN;
    int L = blockIdx.x;
    {
        const int nwg = gridDim.x;
        if ((nwg & 7) == 0) L = (L & 7) * (nwg >> 3) + (L >> 3);
    }
    const int SN = nt_n < 8 ? nt_n : 8;
    const int SM = 8;
    const int per_super = SM * SN;
    const int supers_n = (nt_n + SN - 1) / SN;
    const int s = L / per_super, w = L % per_super;
    const int tm = (s / supers_n) * SM + w / SN;
    const int tn = (s % supers_n) * SN + w % SN;
    if (tm >= mt_n || tn >= nt_n) return;
    const int m0 = tm * BM, n0 = tn * BN;

    f32x4 acc[4][4];  // [nt][mt]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;
    stage_tile(p, smem, wave, lane, m0, n0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    __syncthreads();

    // per-lane fragment read offsets: row (l&15) of a 16-row tile, 16-B chunk (l>>4) (+4 for s=1)
    const int frow = lane & 15;
    const int fchk = lane >> 4;

    for (int kt = 0; kt < nk; ++kt) {
        char* cur = smem + (kt & 1) * kStageBytes;
        if (kt + 1 < nk) stage_tile(p, smem + ((kt + 1) & 1) * kStageBytes, wave, lane, m0, n0, (kt + 1) * BK);
        const char* tA = cur;
        const char* tW = cur + kTileBytes;
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
            ex8 wf[4], xf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rw = wn * 64 + i * 16 + frow;
                wf[i] = *reinterpret_cast<const ex8*>(tW + rw * 128 + (((4 * ss + fchk) ^ ((rw >> 1) & 7)) << 4));
                const int ra = wm * 64 + i * 16 + frow;
                xf[i] = *reinterpret_cast<const ex8*>(tA + ra * 128 + (((4 * ss + fchk) ^ ((ra >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = TT_MFMA_16x16x32(wf[i], xf[j], acc[i][j]);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): next tile landed
        __syncthreads();
    }

    gemm_epilogue<EPI>(p, acc, m0 + wm * 64, n0 + wn * 64, lane);
}

// ---------------------------------------------------------------------------------------------
// v3: 256x256x64 block tile, 8 waves of 128x64, "ping-pong" schedule.
//
// LDS image: 8 half-tile slots of 16 KiB = {A,W} x {lo,hi 128 rows} x {even,odd K-tile}.  Wave w
// (wm = w>>2, wn = w&3) owns rows wm*64+[0,64) of BOTH A halves and columns wn*32+[0,32) of BOTH W
// halves, so quadrant (qm,qn) of its 128x64 output needs exactly half-tiles A[qm], W[qn].
// Per K-tile every wave runs  L1 C1 L2 C2 L3 C3 L4 C4  with a raw s_barrier after each slot:
//   L = LDS fragment reads for the next C + 2 global->LDS copies (one half-tile per L, all waves)
//   C = 16 MFMAs (one quadrant x K=64)
// Waves 4-7 run ONE SLOT BEHIND waves 0-3 (one extra barrier up front), and each SIMD hosts one
// wave of either group: while one group's waves are in a C slot (matrix pipe) the other group's
// are in an L slot (LDS + DMA issue), so LDS traffic overlaps the MFMAs instead of adding to them
// (v1/v2 spend as many CU cycles on LDS as on MFMA and do not overlap them: 36 % MFMA busy).
// Quadrant order (0,0) (0,1) (1,1) (1,0); fragments: A-sub 8 reads in L1/L3, W-sub1 4 reads in L2,
// W-sub0 4 reads in L1 and kept in registers for C4.  A half-tile slot is re-staged as soon as
// both groups are past its last read (two K-tiles ahead):
//   L1(t): A-hi(t+1)   L2(t): A-lo(t+2)   L3(t): W-lo(t+2)   L4(t): W-hi(t+2)
// so >= 11 slots pass between a copy's issue and its first read; every L ends with a counted
// s_waitcnt vmcnt(10) (the five youngest half-tile copies may still be in flight) + lgkmcnt(0).
namespace v3 {
constexpr int BM3 = 256, BN3 = 256;
constexpr int kThreads3 = 512;
constexpr int kHalf = 128 * BK * 2;   // 16 KiB half-tile
constexpr int kBiasOff = 8 * kHalf;   // 1 KiB: the tile's 256 bias values, staged by wave 0 in the prologue
constexpr int kScaleOff = kBiasOff + 2048;   // fp8: {row scales, column scales} x 2 buffers, 1 KiB each
constexpr int kLds3 = 8 * kHalf + 2048 + 4096;   // 134 KiB
// slot offsets: [operand A=0/W=1][half][buf]
__device__ __forceinline__ constexpr int slot_off(int operand, int half, int buf) { return ((operand * 2 + half) * 2 + buf) * kHalf; }

// Row r (0..127) of W half-tile h holds output column n0 + w_row_of(r) + 32 h: the halves interleave in runs of 32
// so that the 64 columns a wave owns (32 of either half) are CONTIGUOUS in the output, n0 + 64 wn + [0, 64): one
// full 128-byte line per output row, which the epilogue stores whole.
__device__ __forceinline__ constexpr int w_row_of(int r) { return (r >> 5) * 64 + (r & 31); }

// One 1-KiB LDS-DMA copy: lane l's 16 bytes from (uniform base + per-lane 32-bit offset) to LDS byte address
// lds_addr + 16 l.  Written as asm for the SGPR-base addressing form: through the builtin hipcc adds base and
// offset into a 64-bit VGPR pair per copy (one VALU op, two temporaries and 64-bit copies of every per-lane
// offset: 10+ VGPRs in a kernel that has none to spare).  M0 is not used by anything else in these kernels.
__device__ __forceinline__ void glds16(const void* base, uint32_t voff, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(voff), "s"(base), "s"(lds_addr) : "memory", "m0");
}

// One half-tile = 2 global->LDS copies per wave.  The source is addressed as a wave-uniform base
// (SGPR pair: operand + first row of the half + K offset) plus a per-lane 32-bit byte offset that is
// constant for the whole kernel (voff[j]), so a copy costs no vector address arithmetic.
template <int OPERAND, int ES = 2>
__device__ __forceinline__ void stage_half(const GemmParams& p, char* smem, int half, int buf, int tile, int wave,
                                           const uint32_t (&voff)[2], int m0, int n0, int tile_bytes = 128) {
    char* dst = smem + slot_off(OPERAND, half, buf);
    // ES = bytes per element (2 bf16, 1 fp8); a K-tile is 128 bytes of every row either way (split planes: 64 bytes of the
    // hi plane + 64 of the lo plane, tile_bytes = 64; the plane offset is part of the per-lane voff)
    const char* base;
    // (p.xp bit 17, TT_GEMM_DEBUG_A0: every tile reads the A rows of row-block 0 -- wrong results, all A reads L2 hits: what the A misses cost)
    if constexpr (OPERAND == 0) base = reinterpret_cast<const char*>(p.A) + ((size_t)(((p.xp & 0x20000) ? 0 : m0) + half * 128) * p.lda) * ES + tile * tile_bytes;
    else base = reinterpret_cast<const char*>(p.W) + ((size_t)(n0 + half * 32) * (p.ldw ? p.ldw : p.K)) * ES + tile * tile_bytes;   // see w_row_of()
    // keep the base in SGPRs (otherwise hipcc folds it into per-lane 64-bit VGPR addresses and
    // pays two 64-bit vector adds per copy)
    const unsigned long long b64 = reinterpret_cast<unsigned long long>(base);
    const unsigned int blo = __builtin_amdgcn_readfirstlane((unsigned int)b64);
    const unsigned int bhi = __builtin_amdgcn_readfirstlane((unsigned int)(b64 >> 32));
    base = reinterpret_cast<const char*>(((unsigned long long)bhi << 32) | blo);
    const uint32_t lds_dst = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)dst + (16 * wave) * 128;
#pragma unroll
    for (int j = 0; j < 2; ++j) glds16(base, voff[j], lds_dst + 8 * j * 128);
}

// Whole-wave epilogue of the 256x256 kernel (acc[qm][qn][nt][mt], wide 16-B layout after permlane16_swap).
// No ordinary global load is left in it: the tile's bias strip is staged to LDS in the prologue and the residual
// tile rides the operand pipeline as two pseudo K-tiles (see stage_res), so the only vector-memory waits are the
// counted ones written here.  History: with the bias / residual loads inside the per-tile loop every iteration
// paid a global-load latency (11 k cycles per tile); hoisted "up front" hipcc still split the 16 residual loads
// into four waited batches (6 us per tile, +19 % on the attention-output GEMM).
//
// Residual image in LDS (wave-private, so no barrier is needed before reading it): part P = qm*2 + pr holds the
// wave's rows (pr*2 + odd)*16 + [0,16) of quadrant row qm as two 2-KiB pieces [16 rows][8 chunks of 16 B]
// (chunk c = qn*4 + nt*2 + (g>>1), XOR-swizzled with (row>>1)&7 like the operand tiles), the odd = 0 piece in
// half-tile slot kResSlot[P], the odd = 1 piece four slots further, both at byte 2048*wave.
__device__ __forceinline__ constexpr int res_slot(int part) { return part == 0 ? 0 : part == 1 ? 2 : part == 2 ? 1 : 3; }

// LDS reads the compiler must not see as LDS reads: hipcc puts s_waitcnt vmcnt(0) in front of any ds_read that may
// alias an LDS-DMA still in flight, which would serialise the residual parts.  The caller waits (lds_wait) before
// the first use.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int OFF>
__device__ __forceinline__ u32x4 lds_read128_async(uint32_t addr) {
    u32x4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
__device__ __forceinline__ void lds_wait(u32x4& a, u32x4& b, u32x4& c, u32x4& d) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}

struct NoNext { __device__ __forceinline__ void operator()(int) const {} };

// has_next / issue_next (persistent kernel): after the wave has read residual part P out of its private pieces it
// refills exactly those pieces with the NEXT tile's operand rows (issue_next(P): 4 copies), so the counted waits
// grow by 4 per finished block.
__device__ __forceinline__ void lds_write128_async(uint32_t addr, uint4 v) {
    const u32x4 d = u32x4{v.x, v.y, v.z, v.w};
    asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(d) : "memory");
}

// STAGE: transpose every (qm, pr) block through the wave's private LDS pieces (the layout of the residual parts,
// which the staged output overwrites in place) so that a store instruction writes whole 128-byte lines.
// Needs the operand slots to be free: the one-tile-per-block kernel.
__device__ __forceinline__ float lds_read32_sync(uint32_t addr) {
    float r;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr));
    return r;
}

// FP8: acc holds sums of e4m3 products; the tile's row scales (a_scale[m0..]) and column scales (w_scale[n0..]) sit in
// two more 1-KiB LDS strips behind the bias strip pair: sa at scale_off, sw at scale_off + 1024.
template <int EPI, bool STAGE, bool FP8, class Next>
__device__ __forceinline__ void epilogue_all(const GemmParams& p, f32x4 (&acc)[2][2][2][4], const char* smem, int bias_off,
                                             int scale_off, int m0, int n0, int wm, int wn, int wave, int lane, bool has_next,
                                             Next issue_next, int lnf_col_off = 0, int lnf_row_off = 0) {
    const int g = lane >> 4;
    const bool odd = (g & 1) != 0;
    const int ncol = wn * 64 + (g & ~1) * 4;           // + qn*32 + nt*16: first of this lane's 8 columns
    const int mrow = wm * 64 + (lane & 15);            // + qm*128 + (pair*2 + odd)*16
    const int l15 = lane & 15;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)smem;
    const uint32_t rbase = lds0 + (odd ? 4 * kHalf : 0) + 2048 * wave + l15 * 128;
    const uint32_t rx = ((g >> 1) ^ ((l15 >> 1) & 7)) << 4;
    const uint32_t raddr[2][2] = {{rbase + (rx ^ 0u), rbase + (rx ^ 32u)}, {rbase + (rx ^ 64u), rbase + (rx ^ 96u)}};  // [qn][nt]
    constexpr uint32_t kOffB[4] = {0, 2 * kHalf, 1 * kHalf, 3 * kHalf};   // res_slot(part) * kHalf
    // row-major read-back: rows (lane>>3) and 8 + (lane>>3) of a 16-row piece, chunk lane & 7
    const uint32_t srow0 = lane >> 3, srow1 = 8 + (lane >> 3);
    const uint32_t saddr[2] = {lds0 + 2048 * wave + srow0 * 128 + (((lane & 7) ^ ((srow0 >> 1) & 7)) << 4),
                               lds0 + 2048 * wave + srow1 * 128 + (((lane & 7) ^ ((srow1 >> 1) & 7)) << 4)};

    float4 bias[2][2][2];
    {
        const uint32_t baddr = lds0 + bias_off + ncol * 4;
        u32x4 b[8];
        b[0] = lds_read128_async<0>(baddr);        b[1] = lds_read128_async<16>(baddr);
        b[2] = lds_read128_async<64>(baddr);       b[3] = lds_read128_async<80>(baddr);
        b[4] = lds_read128_async<128>(baddr);      b[5] = lds_read128_async<144>(baddr);
        b[6] = lds_read128_async<192>(baddr);      b[7] = lds_read128_async<208>(baddr);
        lds_wait(b[0], b[1], b[2], b[3]);
        lds_wait(b[4], b[5], b[6], b[7]);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            bias[i >> 2][(i >> 1) & 1][i & 1] = float4{__uint_as_float(b[i].x), __uint_as_float(b[i].y), __uint_as_float(b[i].z),
                                                       __uint_as_float(b[i].w)};
    }
#if TT_DIAG
    // LayerNorm-folding experiment (GemmParams.lnf): the column strip(s) and the tile's row statistics sit in LDS behind the bias
    // strips (staged in the prologue like them); consumer form: the column sums of the folded weight, kept in registers like the bias
    const int lnf = (FP8 || EPI == TT_EPI_TANH) ? 0 : p.lnf;
    float4 lcs[2][2][2];
    if (lnf == 1) {
        const uint32_t caddr = lds0 + lnf_col_off + ncol * 4;
        u32x4 b[8];
        b[0] = lds_read128_async<0>(caddr);        b[1] = lds_read128_async<16>(caddr);
        b[2] = lds_read128_async<64>(caddr);       b[3] = lds_read128_async<80>(caddr);
        b[4] = lds_read128_async<128>(caddr);      b[5] = lds_read128_async<144>(caddr);
        b[6] = lds_read128_async<192>(caddr);      b[7] = lds_read128_async<208>(caddr);
        lds_wait(b[0], b[1], b[2], b[3]);
        lds_wait(b[4], b[5], b[6], b[7]);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            lcs[i >> 2][(i >> 1) & 1][i & 1] = float4{__uint_as_float(b[i].x), __uint_as_float(b[i].y), __uint_as_float(b[i].z),
                                                      __uint_as_float(b[i].w)};
    }
#endif
    float4 wsc[2][2][2];
    float asc[2][2];   // [qm][pr]: this lane's output row of the block
    if constexpr (FP8) {
        const uint32_t saddr_w = lds0 + scale_off + 1024 + ncol * 4;
        u32x4 b[8];
        b[0] = lds_read128_async<0>(saddr_w);        b[1] = lds_read128_async<16>(saddr_w);
        b[2] = lds_read128_async<64>(saddr_w);       b[3] = lds_read128_async<80>(saddr_w);
        b[4] = lds_read128_async<128>(saddr_w);      b[5] = lds_read128_async<144>(saddr_w);
        b[6] = lds_read128_async<192>(saddr_w);      b[7] = lds_read128_async<208>(saddr_w);
        lds_wait(b[0], b[1], b[2], b[3]);
        lds_wait(b[4], b[5], b[6], b[7]);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            wsc[i >> 2][(i >> 1) & 1][i & 1] = float4{__uint_as_float(b[i].x), __uint_as_float(b[i].y), __uint_as_float(b[i].z),
                                                      __uint_as_float(b[i].w)};
#pragma unroll
        for (int qm = 0; qm < 2; ++qm)
#pragma unroll
            for (int pr = 0; pr < 2; ++pr)
                asc[qm][pr] = lds_read32_sync(lds0 + scale_off + (qm * 128 + mrow + (pr * 2 + (odd ? 1 : 0)) * 16) * 4);
    }
#pragma unroll
    for (int qm = 0; qm < 2; ++qm)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            u32x4 res[2][2];
            if constexpr (EPI == TT_EPI_RESIDUAL) {
                // all but the N youngest vector-memory operations of this wave are done: the residual parts
                // after this one (4 copies each) plus the 4 stores (and, in the persistent kernel, the 4
                // next-tile copies) of every finished (qm, pr) block: N = 12, or 12 + 4 * block
                const int blk = qm * 2 + pr;
                if (!has_next || blk == 0) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else if (blk == 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else if (blk == 2) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
                constexpr int kOff[4] = {0, 2 * kHalf, 1 * kHalf, 3 * kHalf};   // res_slot(part) * kHalf
                if (qm == 0 && pr == 0) {
                    res[0][0] = lds_read128_async<kOff[0]>(raddr[0][0]); res[0][1] = lds_read128_async<kOff[0]>(raddr[0][1]);
                    res[1][0] = lds_read128_async<kOff[0]>(raddr[1][0]); res[1][1] = lds_read128_async<kOff[0]>(raddr[1][1]);
                } else if (qm == 0 && pr == 1) {
                    res[0][0] = lds_read128_async<kOff[1]>(raddr[0][0]); res[0][1] = lds_read128_async<kOff[1]>(raddr[0][1]);
                    res[1][0] = lds_read128_async<kOff[1]>(raddr[1][0]); res[1][1] = lds_read128_async<kOff[1]>(raddr[1][1]);
                } else if (qm == 1 && pr == 0) {
                    res[0][0] = lds_read128_async<kOff[2]>(raddr[0][0]); res[0][1] = lds_read128_async<kOff[2]>(raddr[0][1]);
                    res[1][0] = lds_read128_async<kOff[2]>(raddr[1][0]); res[1][1] = lds_read128_async<kOff[2]>(raddr[1][1]);
                } else {
                    res[0][0] = lds_read128_async<kOff[3]>(raddr[0][0]); res[0][1] = lds_read128_async<kOff[3]>(raddr[0][1]);
                    res[1][0] = lds_read128_async<kOff[3]>(raddr[1][0]); res[1][1] = lds_read128_async<kOff[3]>(raddr[1][1]);
                }
                lds_wait(res[0][0], res[0][1], res[1][0], res[1][1]);
                if (has_next) issue_next(qm * 2 + pr);
            }
#if TT_DIAG
            float lrs = 1.f, lnm = 0.f, lsum = 0.f, lsq = 0.f;       // this lane's row of the block: rstd, -mu rstd; output partials
            if (lnf) {
                const uint32_t ra = lds0 + lnf_row_off + (uint32_t)(qm * 128 + mrow + (pr * 2 + (odd ? 1 : 0)) * 16) * 8u;
                // (two 32-bit reads into registers of their own: read as one 64-bit pair, broadcasting the pair's HIGH dword into a packed-f32
                // FMA is the op_sel form csrc/check_isa.py bans -- profiles/r03_pk_mfma_hazard.log)
                asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:4\n\ts_waitcnt lgkmcnt(0)" : "=&v"(lrs), "=&v"(lnm) : "v"(ra));
            }
#endif
#pragma unroll
            for (int qn = 0; qn < 2; ++qn)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    float v[8];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[qm][qn][nt][2 * pr][k]),
                                                                        __float_as_uint(acc[qm][qn][nt][2 * pr + 1][k]),
                                                                        false, false);
                        v[k] = __uint_as_float(r[0]);
                        v[4 + k] = __uint_as_float(r[1]);
                    }
                    const float4 b0 = bias[qn][nt][0], b1 = bias[qn][nt][1];
#if TT_DIAG
                    if (lnf == 1) {          // consumer: rstd acc + (nmr cs + bias)
                        const float4 c0 = lcs[qn][nt][0], c1 = lcs[qn][nt][1];
                        v[0] = fmaf(v[0], lrs, fmaf(lnm, c0.x, b0.x)); v[1] = fmaf(v[1], lrs, fmaf(lnm, c0.y, b0.y));
                        v[2] = fmaf(v[2], lrs, fmaf(lnm, c0.z, b0.z)); v[3] = fmaf(v[3], lrs, fmaf(lnm, c0.w, b0.w));
                        v[4] = fmaf(v[4], lrs, fmaf(lnm, c1.x, b1.x)); v[5] = fmaf(v[5], lrs, fmaf(lnm, c1.y, b1.y));
                        v[6] = fmaf(v[6], lrs, fmaf(lnm, c1.z, b1.z)); v[7] = fmaf(v[7], lrs, fmaf(lnm, c1.w, b1.w));
                    } else
#endif
                    if constexpr (FP8) {
                        const float sa = asc[qm][pr];
                        const float4 s0 = wsc[qn][nt][0], s1 = wsc[qn][nt][1];
                        v[0] = fmaf(v[0] * sa, s0.x, b0.x); v[1] = fmaf(v[1] * sa, s0.y, b0.y);
                        v[2] = fmaf(v[2] * sa, s0.z, b0.z); v[3] = fmaf(v[3] * sa, s0.w, b0.w);
                        v[4] = fmaf(v[4] * sa, s1.x, b1.x); v[5] = fmaf(v[5] * sa, s1.y, b1.y);
                        v[6] = fmaf(v[6] * sa, s1.z, b1.z); v[7] = fmaf(v[7] * sa, s1.w, b1.w);
                    } else {
                        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
                        v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
                    }
                    if constexpr (EPI == TT_EPI_GELU) {
#pragma unroll
                        for (int k = 0; k < 8; k += 2) {
                            const f32x2 y = gelu_erf2(f32x2{v[k], v[k + 1]});
                            v[k] = y.x; v[k + 1] = y.y;
                        }
                    } else if constexpr (EPI == TT_EPI_TANH) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] = tanhf(v[k]);
                    } else if constexpr (EPI == TT_EPI_RESIDUAL) {
                        const u32x4 r = res[qn][nt];
#if TT_DIAG
                        if (lnf == 2) {      // producer: the residual tile is the raw pre-LayerNorm sum; LayerNorm rebuilt in fp32
                            const uint32_t ga = lds0 + lnf_col_off + (uint32_t)(ncol + qn * 32 + nt * 16) * 4u;
                            u32x4 g0 = lds_read128_async<0>(ga), g1 = lds_read128_async<16>(ga);
                            u32x4 e0 = lds_read128_async<1024>(ga), e1 = lds_read128_async<1040>(ga);
                            lds_wait(g0, g1, e0, e1);
                            const float gm[8] = {__uint_as_float(g0.x), __uint_as_float(g0.y), __uint_as_float(g0.z), __uint_as_float(g0.w),
                                                 __uint_as_float(g1.x), __uint_as_float(g1.y), __uint_as_float(g1.z), __uint_as_float(g1.w)};
                            const float bt[8] = {__uint_as_float(e0.x), __uint_as_float(e0.y), __uint_as_float(e0.z), __uint_as_float(e0.w),
                                                 __uint_as_float(e1.x), __uint_as_float(e1.y), __uint_as_float(e1.z), __uint_as_float(e1.w)};
                            const float y[8] = {elo(r.x), ehi(r.x), elo(r.y), ehi(r.y), elo(r.z), ehi(r.z), elo(r.w), ehi(r.w)};
#pragma unroll
                            for (int k = 0; k < 8; ++k) v[k] += fmaf(y[k], lrs * gm[k], fmaf(lnm, gm[k], bt[k]));
                        } else
#endif
                        {
                        v[0] += elo(r.x); v[1] += ehi(r.x);
                        v[2] += elo(r.y); v[3] += ehi(r.y);
                        v[4] += elo(r.z); v[5] += ehi(r.z);
                        v[6] += elo(r.w); v[7] += ehi(r.w);
                        }
                    }
                    const int m = m0 + qm * 128 + mrow + (pr * 2 + (odd ? 1 : 0)) * 16;
                    const int n = n0 + qn * 32 + nt * 16 + ncol;
                    uint4 o;
                    o.x = pack_e2(v[0], v[1]); o.y = pack_e2(v[2], v[3]);
                    o.z = pack_e2(v[4], v[5]); o.w = pack_e2(v[6], v[7]);
#if TT_DIAG
                    if (lnf == 2) {          // statistics of the values the consumer will read: the bf16-rounded outputs
                        const float w8[8] = {elo(o.x), ehi(o.x), elo(o.y), ehi(o.y), elo(o.z), ehi(o.z), elo(o.w), ehi(o.w)};
#pragma unroll
                        for (int k = 0; k < 8; ++k) { lsum += w8[k]; lsq = fmaf(w8[k], w8[k], lsq); }
                    }
#endif
                    if constexpr (STAGE) {
                        lds_write128_async(raddr[qn][nt] + kOffB[qm * 2 + pr], o);
                    } else {
                        *reinterpret_cast<uint4*>(p.C + (size_t)m * p.ldc + n) = o;
                    }
                }
#if TT_DIAG
            if (lnf == 2) {
                // the row's 64 columns of this wave sit in two lanes (g and g ^ 2: lanes 32 apart)
                lsum += __shfl_xor(lsum, 32, 64);
                lsq += __shfl_xor(lsq, 32, 64);
                if ((g >> 1) == 0) {
                    const int m = m0 + qm * 128 + mrow + (pr * 2 + (odd ? 1 : 0)) * 16;
                    *reinterpret_cast<float2*>(p.lnf_part + ((size_t)m * (p.N >> 6) + (size_t)((n0 >> 6) + wn)) * 2) = float2{lsum, lsq};
                }
            }
#endif
            if constexpr (STAGE) {
                // the block's 32 rows x 128 B now sit row-major in this wave's two private pieces: read them back a
                // row per 8 lanes and store whole 128-byte lines instead of 32-byte runs
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                u32x4 t[4];
                t[0] = lds_read128_async<0>(saddr[0] + kOffB[qm * 2 + pr]);
                t[1] = lds_read128_async<0>(saddr[1] + kOffB[qm * 2 + pr]);
                t[2] = lds_read128_async<0>(saddr[0] + kOffB[qm * 2 + pr] + 4 * kHalf);
                t[3] = lds_read128_async<0>(saddr[1] + kOffB[qm * 2 + pr] + 4 * kHalf);
                lds_wait(t[0], t[1], t[2], t[3]);
                size_t crow = (size_t)(m0 + qm * 128 + wm * 64 + pr * 32 + (lane >> 3)) * p.ldc + n0 + wn * 64 + (lane & 7) * 8;
#if TT_DIAG
                // experiment (TT_GEMM_HEAD_MAJOR=1, diagnostic library): the output head-major, C[n / 64][M][64] -- a wave's 32 rows
                // x 128 bytes become ONE 4-KiB run instead of 32 lines 2 N bytes apart (profiles/r04_gemm_writeback_ab.log)
                if (p.xp & 0x40000) crow = ((size_t)((n0 + wn * 64) >> 6) * p.M + (m0 + qm * 128 + wm * 64 + pr * 32 + (lane >> 3))) * 64 + (lane & 7) * 8;
                const size_t rstride = (p.xp & 0x40000) ? 64 : (size_t)p.ldc;
#else
                const size_t rstride = (size_t)p.ldc;
#endif
                if (p.C8) {
                    // e4m3 output: quantise the bf16-rounded values with the tensor's static scale, 8 bytes per lane
                    // (64 contiguous bytes per row and instruction)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float s = p.c8_inv_scale;
                        const unsigned w[4] = {t[i].x, t[i].y, t[i].z, t[i].w};
                        float f[8];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {   // saturate: the conversion itself returns NaN beyond the e4m3 range
                            f[2 * k] = __builtin_amdgcn_fmed3f(__uint_as_float(w[k] << 16) * s, -448.0f, 448.0f);
                            f[2 * k + 1] = __builtin_amdgcn_fmed3f(__uint_as_float(w[k] & 0xFFFF0000u) * s, -448.0f, 448.0f);
                        }
                        int lo = 0, hi = 0;
                        lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
                        lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
                        hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
                        hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
                        *reinterpret_cast<uint2*>(p.C8 + crow + (size_t)(8 * i) * p.ldc) = make_uint2((unsigned)lo, (unsigned)hi);
                    }
                } else {
                    uint16_t* cp = p.C + crow;
                    if (p.nt_store) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) __builtin_nontemporal_store(t[i], reinterpret_cast<u32x4*>(cp + (size_t)(8 * i) * rstride));
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            *reinterpret_cast<uint4*>(cp + (size_t)(8 * i) * rstride) = uint4{t[i].x, t[i].y, t[i].z, t[i].w};
                    }
                }
            }
            if constexpr (EPI == TT_EPI_RESIDUAL) __builtin_amdgcn_sched_barrier(0);   // keep the 4 stores in their block
        }
}

// ---- split-bf16 (bf16x3) epilogue: reference precision out of the bf16 main loop (GemmParams.x3) -------------------------
// Same accumulator layout and column mapping as epilogue_all, no LDS staging (the main loop is three times as long as the
// bf16 one, the epilogue's share a third): after the permlane16 swap a lane owns 8 consecutive columns of one row.
//   BIAS / GELU (exact erf: the degree-7 fit above is a bf16-grade approximation): two bf16 planes, hi = bf16(v),
//   lo = bf16(v - hi), 16 bytes each per lane;  RESIDUAL: fp32 out = acc + bias + fp32 residual, 32 bytes per lane.
__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

template <int EPI, bool RP = false>   // RP: the residual comes as the two planes of the activation (GemmParams.res_planes), not as fp32
__device__ __forceinline__ void epilogue_x3(const GemmParams& p, f32x4 (&acc)[2][2][2][4], const char* smem, int bias_off, int m0,
                                            int n0, int wm, int wn, int lane) {
    const int g = lane >> 4;
    const bool odd = (g & 1) != 0;
    const int ncol = wn * 64 + (g & ~1) * 4;
    const int mrow = wm * 64 + (lane & 15);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)smem;
    float4 bias[2][2][2];
    {
        const uint32_t baddr = lds0 + bias_off + ncol * 4;
        u32x4 b[8];
        b[0] = lds_read128_async<0>(baddr);        b[1] = lds_read128_async<16>(baddr);
        b[2] = lds_read128_async<64>(baddr);       b[3] = lds_read128_async<80>(baddr);
        b[4] = lds_read128_async<128>(baddr);      b[5] = lds_read128_async<144>(baddr);
        b[6] = lds_read128_async<192>(baddr);      b[7] = lds_read128_async<208>(baddr);
        lds_wait(b[0], b[1], b[2], b[3]);
        lds_wait(b[4], b[5], b[6], b[7]);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            bias[i >> 2][(i >> 1) & 1][i & 1] = float4{__uint_as_float(b[i].x), __uint_as_float(b[i].y), __uint_as_float(b[i].z),
                                                       __uint_as_float(b[i].w)};
    }
#pragma unroll
    for (int qm = 0; qm < 2; ++qm)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const int m = m0 + qm * 128 + mrow + (pr * 2 + (odd ? 1 : 0)) * 16;
            float4 r0[2][2], r1[2][2];
            if constexpr (EPI == TT_EPI_RESIDUAL) {   // the block's 8 residual loads in flight together
#pragma unroll
                for (int qn = 0; qn < 2; ++qn)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        if constexpr (RP) {      // 8 columns of both planes: 16 + 16 bytes instead of 32 of fp32
                            const uint16_t* rp = p.res_planes + (size_t)m * p.ldr + n0 + qn * 32 + nt * 16 + ncol;
                            const uint4 h = *reinterpret_cast<const uint4*>(rp), l = *reinterpret_cast<const uint4*>(rp + p.res_lo_off);
                            r0[qn][nt] = float4{__uint_as_float(h.x), __uint_as_float(h.y), __uint_as_float(h.z), __uint_as_float(h.w)};
                            r1[qn][nt] = float4{__uint_as_float(l.x), __uint_as_float(l.y), __uint_as_float(l.z), __uint_as_float(l.w)};
                        } else {
                            const float* rp = p.res32 + (size_t)m * p.ldr + n0 + qn * 32 + nt * 16 + ncol;
                            r0[qn][nt] = *reinterpret_cast<const float4*>(rp);
                            r1[qn][nt] = *reinterpret_cast<const float4*>(rp + 4);
                        }
                    }
            }
#pragma unroll
            for (int qn = 0; qn < 2; ++qn)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    float v[8];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[qm][qn][nt][2 * pr][k]),
                                                                        __float_as_uint(acc[qm][qn][nt][2 * pr + 1][k]),
                                                                        false, false);
                        v[k] = __uint_as_float(r[0]);
                        v[4 + k] = __uint_as_float(r[1]);
                    }
                    const float4 b0 = bias[qn][nt][0], b1 = bias[qn][nt][1];
                    v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
                    v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
                    const int n = n0 + qn * 32 + nt * 16 + ncol;
                    if constexpr (EPI == TT_EPI_RESIDUAL) {
                        float4 a = r0[qn][nt], b = r1[qn][nt];
                        if constexpr (RP) {      // a = the hi plane's 8 elements, b = the lo plane's: residual = hi + lo
                            const uint32_t h0 = __float_as_uint(a.x), h1 = __float_as_uint(a.y), h2 = __float_as_uint(a.z), h3 = __float_as_uint(a.w);
                            const uint32_t l0 = __float_as_uint(b.x), l1 = __float_as_uint(b.y), l2 = __float_as_uint(b.z), l3 = __float_as_uint(b.w);
                            a = float4{elo(h0) + elo(l0), ehi(h0) + ehi(l0), elo(h1) + elo(l1), ehi(h1) + ehi(l1)};
                            b = float4{elo(h2) + elo(l2), ehi(h2) + ehi(l2), elo(h3) + elo(l3), ehi(h3) + ehi(l3)};
                        }
                        float* cp = p.C32 + (size_t)m * p.ldc + n;
                        *reinterpret_cast<float4*>(cp) = float4{v[0] + a.x, v[1] + a.y, v[2] + a.z, v[3] + a.w};
                        *reinterpret_cast<float4*>(cp + 4) = float4{v[4] + b.x, v[5] + b.y, v[6] + b.z, v[7] + b.w};
                    } else {
                        if constexpr (EPI == TT_EPI_GELU) {
#pragma unroll
                            for (int k = 0; k < 8; ++k) {
                                v[k] = gelu_exact(v[k]);
                                // opaque: under -ffp-contract=fast the final multiply of the GELU fuses with the "v - hi" below
                                // in SOME unrolled copies of this block only -- the lo plane then depends on the row's position
                                // in the tile by an ulp (found by the row-permutation test)
                                asm("" : "+v"(v[k]));
                            }
                        }
                        uint4 o, l;
                        o.x = pack_e2(v[0], v[1]); o.y = pack_e2(v[2], v[3]);
                        o.z = pack_e2(v[4], v[5]); o.w = pack_e2(v[6], v[7]);
                        l.x = pack_e2(v[0] - elo(o.x), v[1] - ehi(o.x));
                        l.y = pack_e2(v[2] - elo(o.y), v[3] - ehi(o.y));
                        l.z = pack_e2(v[4] - elo(o.z), v[5] - ehi(o.z));
                        l.w = pack_e2(v[6] - elo(o.w), v[7] - ehi(o.w));
                        if (p.x3_zero_lo) l = uint4{0u, 0u, 0u, 0u};
                        uint16_t* cp = p.C + (size_t)m * p.ldc + n;
                        *reinterpret_cast<uint4*>(cp) = o;
                        *reinterpret_cast<uint4*>(cp + p.c_lo_off) = l;
                    }
                }
        }
}

// ---- f16c epilogue (GemmParams.xc, GELU): the output as c-planes, i.e. as the NEXT GEMM's A operand ---------------------------
// Same accumulator layout as epilogue_x3: after the permlane16 swap a lane owns 8 consecutive columns of one row; the 32
// columns of a scale block (n0 + wn 64 + qn 32 + [0, 32)) are this lane's two chunks (nt = 0, 1) and those of the lane 32
// further on (the other 8-column half): the block's absmax is one cross-lane exchange.  Per block: E = exponent of the absmax,
// scale byte s = E - 7 (biased); hi = fp16(v); x8 = e4m3(v 2^(7 - E)); lo8 = e4m3((v - hi) 2^(18 - E)) -- |v - hi| <= 2^(E - 11).
// One byte per (row, block) goes to the tiled scale array of a consumer with K = p.N.
// f16c, Q / K projection: the output as TWO fp16 planes, hi = fp16(v) at C[m][n], lo = fp16(v - hi) at C[m][c_lo_off + n] (22
// significand bits for the attention's score product: f16c_path.hip attention_qk2_kernel).  epilogue_x3's structure.
__device__ __forceinline__ void epilogue_h2(const GemmParams& p, f32x4 (&acc)[2][2][2][4], const char* smem, int bias_off, int m0, int n0,
                                            int wm, int wn, int lane) {
    const int g = lane >> 4;
    const bool odd = (g & 1) != 0;
    const int ncol = wn * 64 + (g & ~1) * 4;
    const int mrow = wm * 64 + (lane & 15);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)smem;
    float4 bias[2][2][2];
    {
        const uint32_t baddr = lds0 + bias_off + ncol * 4;
        u32x4 b[8];
        b[0] = lds_read128_async<0>(baddr);        b[1] = lds_read128_async<16>(baddr);
        b[2] = lds_read128_async<64>(baddr);       b[3] = lds_read128_async<80>(baddr);
        b[4] = lds_read128_async<128>(baddr);      b[5] = lds_read128_async<144>(baddr);
        b[6] = lds_read128_async<192>(baddr);      b[7] = lds_read128_async<208>(baddr);
        lds_wait(b[0], b[1], b[2], b[3]);
        lds_wait(b[4], b[5], b[6], b[7]);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            bias[i >> 2][(i >> 1) & 1][i & 1] = float4{__uint_as_float(b[i].x), __uint_as_float(b[i].y), __uint_as_float(b[i].z),
                                                       __uint_as_float(b[i].w)};
    }
#pragma unroll
    for (int qm = 0; qm < 2; ++qm)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const int m = m0 + qm * 128 + mrow + (pr * 2 + (odd ? 1 : 0)) * 16;
#pragma unroll
            for (int qn = 0; qn < 2; ++qn)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    float v[8];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[qm][qn][nt][2 * pr][k]),
                                                                        __float_as_uint(acc[qm][qn][nt][2 * pr + 1][k]),
                                                                        false, false);
                        v[k] = __uint_as_float(r[0]);
                        v[4 + k] = __uint_as_float(r[1]);
                    }
                    const float4 b0 = bias[qn][nt][0], b1 = bias[qn][nt][1];
                    v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
                    v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
#pragma unroll
                    for (int k = 0; k < 8; ++k) asm("" : "+v"(v[k]));      // opaque before the split (see epilogue_x3)
                    uint4 o, l;
                    o.x = pack_e2(v[0], v[1]); o.y = pack_e2(v[2], v[3]);
                    o.z = pack_e2(v[4], v[5]); o.w = pack_e2(v[6], v[7]);
                    l.x = pack_e2(v[0] - elo(o.x), v[1] - ehi(o.x)); l.y = pack_e2(v[2] - elo(o.y), v[3] - ehi(o.y));
                    l.z = pack_e2(v[4] - elo(o.z), v[5] - ehi(o.z)); l.w = pack_e2(v[6] - elo(o.w), v[7] - ehi(o.w));
                    uint16_t* cp = p.C + (size_t)m * p.ldc + n0 + qn * 32 + nt * 16 + ncol;
                    *reinterpret_cast<uint4*>(cp) = o;
                    *reinterpret_cast<uint4*>(cp + p.c_lo_off) = l;
                }
        }
}

// exact-erf GELU to fp32 grade without libm: gelu(x) = max(x, 0) - |x| / 2 * erfc(|x| / sqrt 2), erfc by the Chebyshev fit
// t exp(-z^2 + P(t)), t = 1 / (1 + z / 2) (fractional error < 1.2e-7 for every z >= 0: the error of the GELU is below
// 1.2e-7 of the CORRECTION term, i.e. relatively accurate on both tails).  One v_rcp_f32, ten FMAs, one v_exp_f32: the libm
// erff of epilogue_x3 made this GEMM's epilogue a quarter of its time (2.48 matrix-time units against the 1.95 of the same K
// with a bias epilogue, profiles/r04_f16c_gemm_shapes.log).
__device__ __forceinline__ float gelu_erfc(float x) {
    const float ax = fabsf(x);
    const float z = ax * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.5f, z, 1.0f));
    float p = 0.17087277f;
    p = fmaf(p, t, -0.82215223f);
    p = fmaf(p, t, 1.48851587f);
    p = fmaf(p, t, -1.13520398f);
    p = fmaf(p, t, 0.27886807f);
    p = fmaf(p, t, -0.18628806f);
    p = fmaf(p, t, 0.09678418f);
    p = fmaf(p, t, 0.37409196f);
    p = fmaf(p, t, 1.00002368f);
    p = fmaf(p, t, -1.26551223f);
    const float e = t * __builtin_amdgcn_exp2f((p - z * z) * 1.4426950408889634f);      // erfc(z)
    return fmaxf(x, 0.f) - 0.5f * ax * e;
}

template <int EPI>
__device__ __forceinline__ void epilogue_xc(const GemmParams& p, f32x4 (&acc)[2][2][2][4], const char* smem, int bias_off, int m0,
                                            int n0, int wm, int wn, int lane) {
    static_assert(EPI == TT_EPI_GELU || EPI == TT_EPI_BIAS, "c-planes come out of the bias / GELU epilogues");
    const int g = lane >> 4;
    const bool odd = (g & 1) != 0;
    const int ncol = wn * 64 + (g & ~1) * 4;
    const int mrow = wm * 64 + (lane & 15);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)smem;
    float4 bias[2][2][2];
    {
        const uint32_t baddr = lds0 + bias_off + ncol * 4;
        u32x4 b[8];
        b[0] = lds_read128_async<0>(baddr);        b[1] = lds_read128_async<16>(baddr);
        b[2] = lds_read128_async<64>(baddr);       b[3] = lds_read128_async<80>(baddr);
        b[4] = lds_read128_async<128>(baddr);      b[5] = lds_read128_async<144>(baddr);
        b[6] = lds_read128_async<192>(baddr);      b[7] = lds_read128_async<208>(baddr);
        lds_wait(b[0], b[1], b[2], b[3]);
        lds_wait(b[4], b[5], b[6], b[7]);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            bias[i >> 2][(i >> 1) & 1][i & 1] = float4{__uint_as_float(b[i].x), __uint_as_float(b[i].y), __uint_as_float(b[i].z),
                                                       __uint_as_float(b[i].w)};
    }
    char* cb = reinterpret_cast<char*>(p.C);
    const size_t row_bytes = (size_t)p.ldc * 2;          // = 4 N
    const int nks_out = p.N >> 7;                        // 128-element K-tiles of the consumer
#pragma unroll
    for (int qm = 0; qm < 2; ++qm)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const int m = m0 + qm * 128 + mrow + (pr * 2 + (odd ? 1 : 0)) * 16;
            char* crow = cb + (size_t)m * row_bytes;
#pragma unroll
            for (int qn = 0; qn < 2; ++qn) {
                float v[2][8];
                float amax = 0.f;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[qm][qn][nt][2 * pr][k]),
                                                                        __float_as_uint(acc[qm][qn][nt][2 * pr + 1][k]),
                                                                        false, false);
                        v[nt][k] = __uint_as_float(r[0]);
                        v[nt][4 + k] = __uint_as_float(r[1]);
                    }
                    const float4 b0 = bias[qn][nt][0], b1 = bias[qn][nt][1];
                    v[nt][0] += b0.x; v[nt][1] += b0.y; v[nt][2] += b0.z; v[nt][3] += b0.w;
                    v[nt][4] += b1.x; v[nt][5] += b1.y; v[nt][6] += b1.z; v[nt][7] += b1.w;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        if constexpr (EPI == TT_EPI_GELU) v[nt][k] = gelu_erfc(v[nt][k]);
                        asm("" : "+v"(v[nt][k]));        // opaque before it is split (see epilogue_x3)
                        amax = fmaxf(amax, fabsf(v[nt][k]));
                    }
                }
                amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
                int sbyte, sh;
                xc_block_scale(amax, sbyte, sh);
                const int n_blk = n0 + wn * 64 + qn * 32;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const int n = n_blk + nt * 16 + (g & ~1) * 4;
                    uint2 h0, h1;
                    uint32_t xa, xb, la, lb;
                    xc_split4(v[nt][0], v[nt][1], v[nt][2], v[nt][3], sh, sh + 11, h0, xa, la);
                    xc_split4(v[nt][4], v[nt][5], v[nt][6], v[nt][7], sh, sh + 11, h1, xb, lb);
                    *reinterpret_cast<uint4*>(crow + (size_t)n * 2) = uint4{h0.x, h0.y, h1.x, h1.y};
                    *reinterpret_cast<uint2*>(crow + (size_t)2 * p.N + n) = make_uint2(xa, xb);
                    *reinterpret_cast<uint2*>(crow + (size_t)3 * p.N + n) = make_uint2(la, lb);
                }
                if ((lane & 32) == 0) p.c_scales[xc_a_scale_at(m, n_blk >> 5, nks_out)] = (uint8_t)sbyte;
            }
        }
}

// TT_EPI_SCAN: nothing is stored.  In the swapped accumulator layout a lane holds ONE corpus row (m-tile row l & 15)
// and four consecutive queries per 16x16 tile, so the per-query threshold filter is four compares against a float4
// of the tile's threshold strip (LDS, staged like a bias strip).  Survivors are rare (about k * rows / sample rows
// per query over the whole pass: tens per tile) but each one needs a RETURNING global atomic on its query's list
// counter; issued where they are found -- one divergent region after the other -- their round trips add up to ~20 %
// of a tile (measured: 5.2 -> 4.8 ms per pass when a 4x larger sample quarters the survivors).  So the waves only
// RECORD survivors, in an LDS list behind the operand slots (LDS atomic for the slot), and after one barrier the
// first n threads append one survivor each: all global atomics of a tile are in flight together, one round trip.
// NaN scores (tombstoned rows) fail the compare.  STAGED = false (persistent form, no workgroup barrier available
// between its wave groups here): the direct per-survivor path.
constexpr int kScanHitOff = kLds3;            // [0]: count, [16...): kScanHitCap x {score bits, (query << 8) | row in tile}
constexpr int kScanHitCap = 1024;
constexpr int kLdsScan = kLds3 + 16 + kScanHitCap * 8;
// MODE 2: wave-private lists (round 3): every wave records ITS survivors in its own LDS list (count + kScanWaveCap entries)
// and appends them itself once its compares are done -- the same "all atomics of a tile in flight together", but with no
// workgroup barrier and no shared counter, so the two wave groups of the PERSISTENT kernel (one slot apart through the tile
// boundary, no common barrier there) can use it: the persistent form then saves the per-tile prologue and dispatch gap.
constexpr int kScanWaveCap = 120;             // survivors per wave and tile before falling back to direct appends (expected: ~3)
constexpr int kScanWaveBytes = 16 + kScanWaveCap * 8;      // 976 B
constexpr int kLdsScanW = kLds3 + 8 * 1024;   // 8 waves x 1 KiB

__device__ __forceinline__ void scan_append(const GemmParams& p, int q, int32_t row, float v) {
    const int pos = atomicAdd(p.scan_cnt + q, 1);
    if (pos < p.scan_cap) {
        p.scan_scores[(size_t)q * p.scan_cap + pos] = v;
        p.scan_idx[(size_t)q * p.scan_cap + pos] = p.scan_idx_base + row;
    }
}

// MODE 0: direct per-survivor appends; 1: workgroup-shared LDS list + one barrier; 2: wave-private LDS lists, no barrier
template <int MODE>
__device__ __forceinline__ void scan_filter_epilogue(const GemmParams& p, f32x4 (&acc)[2][2][2][4], char* smem, int thr_off,
                                                     int m0, int wm, int wn, int lane) {
    // first row this tile reports: m0, except for the shifted last tile of a shard whose rows are not a multiple of 256
    const int row_min = (p.scan_rows && (p.scan_rows & 255) && m0 == p.scan_rows - 256) ? (p.scan_rows & ~255) : m0;
    const int g = lane >> 4, l15 = lane & 15;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)smem;
    const int wave_id = wm * 4 + wn;
    char* hbase = MODE == 2 ? smem + kScanHitOff + wave_id * 1024 : smem + kScanHitOff;
    int* hit_cnt = reinterpret_cast<int*>(hbase);
    uint2* hits = reinterpret_cast<uint2*>(hbase + 16);
    constexpr int kCap = MODE == 2 ? kScanWaveCap : kScanHitCap;
#pragma unroll
    for (int qn = 0; qn < 2; ++qn)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int q0 = wn * 64 + qn * 32 + nt * 16 + g * 4;
            u32x4 tb = lds_read128_async<0>(lds0 + thr_off + q0 * 4);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tb));
            const float t[4] = {__uint_as_float(tb.x), __uint_as_float(tb.y), __uint_as_float(tb.z), __uint_as_float(tb.w)};
#pragma unroll
            for (int qm = 0; qm < 2; ++qm)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const f32x4 v = acc[qm][qn][nt][mt];
                    const int rt = qm * 128 + wm * 64 + mt * 16 + l15;          // row inside the tile
                    // (the shard's last tile when its rows are not a multiple of 256: it starts 256 rows before the end, the rows
                    // it shares with its predecessor were reported by that one)
                    const bool hit = ((v[0] >= t[0]) | (v[1] >= t[1]) | (v[2] >= t[2]) | (v[3] >= t[3])) & (m0 + rt >= row_min);
                    if (hit) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (v[r] >= t[r]) {
                                const int q = q0 + r;
                                if constexpr (MODE != 0) {
                                    const int pos = atomicAdd(hit_cnt, 1);
                                    if (pos < kCap) hits[pos] = make_uint2(__float_as_uint(v[r]), (uint32_t)((q << 8) | rt));
                                    else scan_append(p, q, m0 + rt, v[r]);       // more survivors than the list holds
                                } else {
                                    scan_append(p, q, m0 + rt, v[r]);
                                }
                            }
                    }
                }
        }
    if constexpr (MODE == 1) {
        __syncthreads();
        const int n = *hit_cnt < kScanHitCap ? *hit_cnt : kScanHitCap;
        for (int i = threadIdx.x; i < n; i += kThreads3) {
            const uint2 e = hits[i];
            scan_append(p, (int)(e.y >> 8), m0 + (int)(e.y & 255u), __uint_as_float(e.x));
        }
    } else if constexpr (MODE == 2) {
        // wave-synchronous: this wave's LDS atomics and list writes are complete once its lgkm counter is drained
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        int n = *reinterpret_cast<volatile int*>(hit_cnt);
        n = n < kScanWaveCap ? n : kScanWaveCap;
        for (int i = lane; i < n; i += 64) {
            const uint2 e = hits[i];
            scan_append(p, (int)(e.y >> 8), m0 + (int)(e.y & 255u), __uint_as_float(e.x));
        }
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) *reinterpret_cast<volatile int*>(hit_cnt) = 0;     // ready for this wave's next tile
    }
}

// Sample form of the scan contraction (GemmParams.scan_dense): per query the maximum of every 32-row group of the tile.  A lane
// holds 4 queries x 1 row of each 16 x 16 MFMA tile (row = lane & 15): the two MFMA tiles of a group are combined in registers,
// the 16 rows of a tile across lanes with four DPP steps (quad_perm xor 1, xor 2, row_half_mirror, row_mirror: max is symmetric,
// so every lane ends with the group's maximum); v_max_f32 drops NaN operands, so tombstoned rows do not count.
__device__ __forceinline__ float dpp_max16(float v) {
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true)));
    return v;
}
__device__ __forceinline__ void scan_sample_epilogue(const GemmParams& p, f32x4 (&acc)[2][2][2][4], int tile, int wm, int wn, int lane) {
    const int g = lane >> 4, l15 = lane & 15;
#pragma unroll
    for (int qn = 0; qn < 2; ++qn)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int q0 = wn * 64 + qn * 32 + nt * 16 + g * 4;
#pragma unroll
            for (int qm = 0; qm < 2; ++qm)
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const f32x4 a = acc[qm][qn][nt][2 * half], b = acc[qm][qn][nt][2 * half + 1];
                    float m[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) m[r] = dpp_max16(fmaxf(a[r], b[r]));
                    if (l15 == 0) {
                        const int grp = tile * 8 + qm * 4 + wm * 2 + half;
#pragma unroll
                        for (int r = 0; r < 4; ++r) p.scan_dense[(size_t)(q0 + r) * p.scan_dense_stride + grp] = m[r];
                    }
                }
        }
}

#define TT_SLOT_END()                                         \
    do {                                                      \
        __builtin_amdgcn_sched_barrier(0);                    \
        __builtin_amdgcn_s_barrier();                         \
        __builtin_amdgcn_sched_barrier(0);                    \
    } while (0)

typedef int v8i __attribute__((ext_vector_type(8)));
struct Frag2 { ex8 lo, hi; };
// one K = 128 step of e4m3 products (v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales: twice the cycles of
// the bf16 16x16x32 form at four times the K).  Lane (row, g) supplies bytes [16 g, 16 g + 16) and
// [64 + 16 g, 64 + 16 g + 16) of its row's 128-byte K-tile for BOTH operands -- the two fragments the bf16 path
// reads -- which is a permutation of k the instruction applies to both operands alike.
__device__ __forceinline__ f32x4 mfma_fp8(const ex8& a0, const ex8& a1, const ex8& b0, const ex8& b1, f32x4 c) {
    const v8i a = __builtin_bit_cast(v8i, Frag2{a0, a1});
    const v8i b = __builtin_bit_cast(v8i, Frag2{b0, b1});
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
}

// ---- f16c operands (GemmParams.xc): fp16 tiles, then block-scaled e4m3 tiles, in one K stream ------------------------------
// The e4m3 tiles' E8M0 block scales ride the operand pipeline in GROUPS OF FOUR K-tiles: one 1-KiB LDS-DMA per wave and group
// (wave w: the A strip (w even) or W strip (w odd) of the group's tile w / 2) into one of two 8-KiB group buffers behind the
// operand slots, issued two tiles ahead of the group's first tile.  (A copy per wave and TILE -- 256 bytes each -- made an
// e4m3 tile 1.18x as long as an fp16 tile: an LDS-DMA issue costs ~100 cycles whatever its size.)
constexpr int kXcScaleOff = kLds3;            // 2 groups x 4 tiles x {A strip 1 KiB, W strip 1 KiB}
constexpr int kLdsXc = kLds3 + 2 * 8192;      // 150 KiB
// a strip is stored in the order its readers want it (one ds_read_b32 = the four scales of a lane's MFMAs):
//   A: byte ((qm 2 + wm) 16 + frow) 16 + g 4 + mt  = scale of tile row qm 128 + wm 64 + mt 16 + frow, block g
//   W: byte ((wn 16 + frow) 4 + g) 4 + qn 2 + nt    = scale of tile column wn 64 + qn 32 + nt 16 + frow, block g
__device__ __forceinline__ void glds4(const void* base, uint32_t voff, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1"
                 :: "v"(voff), "s"(base), "s"(lds_addr) : "memory", "m0");
}
// one K = 128 step of e4m3 products with block scales: byte OA of lane (row, g)'s scale_a is the E8M0 scale of block g (k in
// [32 g, 32 g + 32)) of that row of operand a, byte OB of scale_b likewise for operand b (tools/probes/mfma_scale_probe.cpp)
template <int OA, int OB>
__device__ __forceinline__ f32x4 mfma_fp8_scaled(const ex8& a0, const ex8& a1, const ex8& b0, const ex8& b1, f32x4 c, uint32_t scale_a,
                                                 uint32_t scale_b) {
    const v8i a = __builtin_bit_cast(v8i, Frag2{a0, a1});
    const v8i b = __builtin_bit_cast(v8i, Frag2{b0, b1});
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, OA, (int)scale_a, OB, (int)scale_b);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

template <int EPI, int SLOTS, bool FP8 = false, bool X3 = false, bool XC = false, bool RP = false>
__global__ __launch_bounds__(kThreads3, 2) void gemm_kernel_v3(GemmParams p) {
    static_assert(!(FP8 && X3), "split-bf16 operands are bf16");
    static_assert(!XC || (!FP8 && !X3 && kF16), "f16c operands: the fp16 instantiation, no other operand mode");
    constexpr int ES = FP8 ? 1 : 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    // (split planes: N may end inside the last column tile -- a multiple of 64, so whole waves' 64-column strips are in or out)
    const int mt_n = p.M / BM3, nt_n = (p.N + BN3 - 1) / BN3;
    int L = blockIdx.x;
    {
        const int nwg = gridDim.x;
        if ((nwg & 7) == 0) L = (L & 7) * (nwg >> 3) + (L >> 3);
    }
    const int SN = p.sn > 0 ? p.sn : (nt_n < 4 ? nt_n : 4);
    const int SM = 32 / SN > 0 ? 32 / SN : 1;
    const int per_super = SM * SN;
    const int supers_n = (nt_n + SN - 1) / SN;
    const int sidx = L / per_super, widx = L % per_super;
    const int tm = (sidx / supers_n) * SM + widx / SN;
    const int tn = (sidx % supers_n) * SN + widx % SN;
    if (tm >= mt_n || tn >= nt_n) return;
    int m0_nominal = tm * BM3;
    if constexpr (EPI == TT_EPI_SCAN) {
        if (p.scan_tile_stride > 1) m0_nominal *= p.scan_tile_stride;                           // sample form: every S-th row tile
        if (p.scan_rows && m0_nominal + BM3 > p.scan_rows) m0_nominal = p.scan_rows - BM3;      // the shard's last 256 rows
    }
    const int m0 = m0_nominal, n0 = tn * BN3;
    // split planes (X3; round 4): a K-tile is 32 elements of BOTH planes -- a row of the tile in LDS is [hi: 64 bytes | lo: 64
    // bytes] -- and its three products hi.hi, hi.lo, lo.hi are three MFMAs on the SAME four fragments: a third less LDS-DMA,
    // a third fewer LDS reads and barriers per MFMA than round 3's three passes over K (one virtual stream of 3 K / 64 tiles)
    // f16c (XC): K / 64 fp16 tiles, then K / 64 e4m3 tiles (x8.w_lo8: K / 128, lo8.w_x8: K / 128) -- the rows ARE that stream
    const int nk1 = p.K * ES / 128;
    constexpr int TB = X3 ? 64 : 128;                 // bytes a K-tile advances in a plane
    const int nk = X3 ? 2 * nk1 : (XC ? 2 * nk1 : nk1);
    auto tile_a = [&](int t) { return t; };
    auto tile_w = [&](int t) { return t; };

    f32x4 acc[2][2][2][4];  // [qm][qn][n-tile][m-tile]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int d = 0; d < 4; ++d) acc[a][b][c][d] = f32x4{0.f, 0.f, 0.f, 0.f};

    // per-lane fragment offsets inside a half-tile (row frow of a 16-row tile, k-substep 0 / 1)
    const int frow = lane & 15, fchk = lane >> 4, sw = (frow >> 1) & 7;
    const int off0 = frow * 128 + ((fchk ^ sw) << 4);
    const int off1 = frow * 128 + (((4 + fchk) ^ sw) << 4);
    const int a_row0 = wm * 64 * 128;   // byte offset of this wave's first A row inside a half-tile
    const int w_row0 = wn * 32 * 128;

    // per-lane source byte offsets of this wave's two copies inside a half-tile (rows 16w+8j+lane/8,
    // 16-B chunk (lane&7) ^ swizzle(row))
    uint32_t voffA[2], voffW[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = 16 * wave + 8 * j + (lane >> 3);
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        // (split planes: source chunks 0-3 of a tile row are the hi plane's 64 bytes, 4-7 the lo plane's, K elements further)
        const uint32_t coff = X3 ? (chunk < 4 ? (uint32_t)chunk * 16u : (uint32_t)p.K * 2u + (uint32_t)(chunk - 4) * 16u) : (uint32_t)chunk * 16u;
        voffA[j] = (uint32_t)r * (uint32_t)p.lda * (uint32_t)ES + coff;
        // last column tile of an N that is not a multiple of 256 (split planes, N % 64 == 0: bge-small's 384 columns): the W rows of
        // the 64-column strips beyond N are read from the tile's first row instead (finite garbage products, never stored)
        const int wr = (X3 && n0 + (r >> 5) * 64 + 64 > p.N) ? 0 : w_row_of(r);
        voffW[j] = (uint32_t)wr * (uint32_t)(p.ldw ? p.ldw : p.K) * (uint32_t)ES + coff;
    }

    unsigned long long* dbg0 = reinterpret_cast<unsigned long long*>(p.vt);
    auto cstamp = [&](int slot) {
        if constexpr (SLOTS == 46) {
            if ((int)blockIdx.x == (p.xp >> 20) && tid == 0 && dbg0) dbg0[20 + slot] = __builtin_amdgcn_s_memtime();   // stamped workgroup: TT_GEMM_STAMP_BLOCK
            // every workgroup: entry / exit time and where it ran (tools/gemm_stamps: dispatch gaps per CU)
            if ((slot == 0 || slot == 3) && tid == 0 && dbg0) {
                unsigned long long* rec = dbg0 + 256 + (size_t)blockIdx.x * 4;
                rec[slot == 0 ? 0 : 1] = __builtin_amdgcn_s_memtime();
                if (slot == 0)
                    rec[2] = ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32) |   // HW_REG_XCC_ID
                             (unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11));              // HW_REG_HW_ID
            }
        }
    };
    // residual tile -> LDS, four parts of (2 pieces x 2 copies) per wave; layout: see epilogue_all
    constexpr bool kResLds = (EPI == TT_EPI_RESIDUAL) && !X3 && !XC;   // (split-bf16 / f16c: the residual is fp32, read by the epilogue)
    uint32_t voffR[2] = {0u, 0u};
    if constexpr (kResLds) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r16 = 8 * j + (lane >> 3);
            const int c = (lane & 7) ^ ((r16 >> 1) & 7);
            voffR[j] = (uint32_t)r16 * (uint32_t)p.ldr * 2u + (uint32_t)(c * 16);
        }
    }
    auto stage_res = [&](int part) {
        if constexpr (kResLds) {
#pragma unroll
            for (int od = 0; od < 2; ++od) {
                const char* base = reinterpret_cast<const char*>(
                    p.residual + (size_t)(m0 + (part >> 1) * 128 + wm * 64 + ((part & 1) * 2 + od) * 16) * p.ldr + n0 + wn * 64);
                const unsigned long long b64 = reinterpret_cast<unsigned long long>(base);
                const unsigned int blo = __builtin_amdgcn_readfirstlane((unsigned int)b64);
                const unsigned int bhi = __builtin_amdgcn_readfirstlane((unsigned int)(b64 >> 32));
                base = reinterpret_cast<const char*>(((unsigned long long)bhi << 32) | blo);
                char* dst = smem + (res_slot(part) + 4 * od) * kHalf + 2048 * wave;
                const uint32_t lds_dst = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)dst;
#pragma unroll
                for (int j = 0; j < 2; ++j) glds16(base, voffR[j], lds_dst + 1024 * j);
            }
        }
    };

    cstamp(0);
    if constexpr (EPI == TT_EPI_SCAN) {
        // survivor counts of this tile: the shared list's, or (wave-private lists, p.xp bit 16) one per wave
        if (tid == 0) *reinterpret_cast<int*>(smem + kScanHitOff) = 0;
        if ((p.xp & 0x10000) && lane == 0) *reinterpret_cast<int*>(smem + kScanHitOff + wave * 1024) = 0;
    }
    // bias strip of this tile (256 floats = one 1-KiB copy), oldest operation of wave 0's queue
    if (wave == 0) {
        uint32_t boff = lane * 16;
        if constexpr (X3) {          // (a partial last column tile: stay inside the bias vector)
            const uint32_t last = (uint32_t)((p.N - n0 < BN3 ? p.N - n0 : BN3) * 4 - 16);
            boff = boff < last ? boff : last;
        }
        glds16(p.bias + n0, boff, (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem + kBiasOff);
    }
    if constexpr (FP8 && EPI != TT_EPI_VT) {   // (the V^T epilogue reads its scales from global memory)
        if (wave == 1) glds16(p.a_scale + m0, lane * 16, (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem + kScaleOff);
        if (wave == 2) glds16(p.w_scale + n0, lane * 16, (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem + kScaleOff + 1024);
    }
#if TT_DIAG
    if constexpr (!FP8 && !X3 && !XC && (EPI == TT_EPI_BIAS || EPI == TT_EPI_GELU || EPI == TT_EPI_RESIDUAL)) {
        if (p.lnf) {      // LayerNorm-folding experiment: column strip(s) + the tile's 256 row statistics, oldest operations of their waves
            const uint32_t l0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
            if (wave == 1) glds16(p.lnf_c0 + n0, lane * 16, l0 + kScaleOff);
            if (wave == 2 && p.lnf == 2) glds16(p.lnf_c1 + n0, lane * 16, l0 + kScaleOff + 1024);
            if (wave == 3) glds16(p.lnf_rows + (size_t)m0 * 2, lane * 16, l0 + kScaleOff + 2048);
            if (wave == 4) glds16(p.lnf_rows + (size_t)(m0 + 128) * 2, lane * 16, l0 + kScaleOff + 3072);
        }
    }
#endif
    // ---- prologue: tile 0 complete, tile 1 without its A-hi (issued in L1(0)); same order as steady state
    stage_half<0, ES>(p, smem, 0, 0, 0, wave, voffA, m0, n0, TB);
    stage_half<1, ES>(p, smem, 0, 0, 0, wave, voffW, m0, n0, TB);
    stage_half<1, ES>(p, smem, 1, 0, 0, wave, voffW, m0, n0, TB);
    stage_half<0, ES>(p, smem, 1, 0, 0, wave, voffA, m0, n0, TB);
    if (nk > 1) {
        stage_half<0, ES>(p, smem, 0, 1, tile_a(1), wave, voffA, m0, n0, TB);
        stage_half<1, ES>(p, smem, 0, 1, tile_w(1), wave, voffW, m0, n0, TB);
    }
    if (nk > 1) {
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // A-hi(0), A-lo(1), W-lo(1) may be in flight
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    TT_SLOT_END();
    cstamp(1);
    const bool late = wave >= 4;   // (grouping by SIMD parity instead, both waves of a SIMD in the same slot: -36 %)
    if (late) TT_SLOT_END();   // waves 4-7 run one slot behind

    ex8 xf[4][2], wf0[2][2], wf1[2][2];

    // f16c: an e4m3 tile is read exactly like an fp16 tile -- lane (row, g) takes bytes [16 g, 16 g + 16) and [64 + 16 g,
    // 64 + 16 g + 16) of its row.  That IS the instruction's K order (registers 0-3 of lane group g: k = 16 g ..., registers
    // 4-7: k = 64 + 16 g ...), and the E8M0 scale of the tile's 32-element block b = k / 32 is taken from lane group b's scale
    // register, whatever lanes hold the block's data (measured: tools/probes/mfma_scale_map2.cpp) -- so lane (row, g) supplies
    // the scale of block g of its row, which is how the scale strips are laid out.
    auto read_a = [&](const char* base) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            xf[mt][0] = *reinterpret_cast<const ex8*>(base + a_row0 + mt * 2048 + off0);
            xf[mt][1] = *reinterpret_cast<const ex8*>(base + a_row0 + mt * 2048 + off1);
        }
    };
    auto read_w = [&](ex8(&wf)[2][2], const char* base) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            wf[nt][0] = *reinterpret_cast<const ex8*>(base + w_row0 + nt * 2048 + off0);
            wf[nt][1] = *reinterpret_cast<const ex8*>(base + w_row0 + nt * 2048 + off1);
        }
    };
    // f16c: scale strips of the four e4m3 K-tiles t .. t + 3 (t - nk1 a multiple of 4; nk1 is one: K % 256 == 0) -> group
    // buffer ((t - nk1) >> 2) & 1: wave w copies tile t + w / 2's A strip (w even) or W strip (w odd), 1 KiB
    uint32_t sa_reg[2] = {0u, 0u}, sw_reg = 0u;
    const int sa_off = (wm * 16 + frow) * 16 + fchk * 4;                 // + qm * 512
    const int sw_off = 1024 + (wn * 16 + frow) * 16 + fchk * 4;
    auto stage_scales = [&](int t) {
        if constexpr (XC) {
            const int u = t - nk1 + (wave >> 1), nks = nk1 >> 1;          // nks = K / 128 e4m3 tiles per part
            const uint8_t* src = (wave & 1) ? p.w_scales + ((size_t)(n0 >> 8) * nk1 + u) * 1024
                                            : p.a_scales + ((size_t)(m0 >> 8) * nks + (u >= nks ? u - nks : u)) * 1024;
            const unsigned long long b64 = reinterpret_cast<unsigned long long>(src);
            const unsigned int blo = __builtin_amdgcn_readfirstlane((unsigned int)b64);
            const unsigned int bhi = __builtin_amdgcn_readfirstlane((unsigned int)(b64 >> 32));
            const uint32_t dst = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem + kXcScaleOff +
                                 (((t - nk1) >> 2) & 1) * 8192 + wave * 1024;
            glds16(reinterpret_cast<const void*>(((unsigned long long)bhi << 32) | blo), (uint32_t)lane * 16u, dst);
        }
    };
    // Tiles of the V third of a QKV projection are produced un-swapped (a = X, b = W): a lane then holds 4
    // consecutive TOKENS of one feature, which is what the transposed V^T store wants.
    constexpr bool vblk = (EPI == TT_EPI_VT);
    auto mma = [&](f32x4(&c)[2][4], const ex8(&wf)[2][2]) {
        __builtin_amdgcn_s_setprio(1);
        if constexpr (FP8) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    if constexpr (vblk) c[nt][mt] = mfma_fp8(xf[mt][0], xf[mt][1], wf[nt][0], wf[nt][1], c[nt][mt]);
                    else c[nt][mt] = mfma_fp8(wf[nt][0], wf[nt][1], xf[mt][0], xf[mt][1], c[nt][mt]);
                }
        } else if constexpr (X3) {
            // fragment [.][0] = the tile's 32 hi-plane elements, [.][1] = its 32 lo-plane elements
#pragma unroll
            for (int term = 0; term < 3; ++term) {
                const int sa = term == 2 ? 1 : 0, sw = term == 1 ? 1 : 0;       // hi.hi, x_hi.w_lo, x_lo.w_hi
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        if constexpr (vblk) c[nt][mt] = TT_MFMA_16x16x32(xf[mt][sa], wf[nt][sw], c[nt][mt]);
                        else c[nt][mt] = TT_MFMA_16x16x32(wf[nt][sw], xf[mt][sa], c[nt][mt]);
                    }
            }
        } else if constexpr (vblk) {
#pragma unroll
            for (int ss = 0; ss < 2; ++ss)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        c[nt][mt] = TT_MFMA_16x16x32(xf[mt][ss], wf[nt][ss], c[nt][mt]);
        } else {
#pragma unroll
            for (int ss = 0; ss < 2; ++ss)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        c[nt][mt] = TT_MFMA_16x16x32(wf[nt][ss], xf[mt][ss], c[nt][mt]);
        }
        __builtin_amdgcn_s_setprio(0);
    };
    // f16c, e4m3 tiles: one scaled MFMA per 16 x 16 output tile and K-tile; the A rows' scales are the four bytes of sa (byte
    // mt), the W columns' of sw (byte qn 2 + nt)
    auto mma8 = [&](f32x4(&c)[2][4], const ex8(&wf)[2][2], uint32_t sa, uint32_t sw, auto qnc) {
        if constexpr (XC) {
            constexpr int QN = decltype(qnc)::value;
            __builtin_amdgcn_s_setprio(1);
            static_for<2>([&](auto ntc) {
                static_for<4>([&](auto mtc) {
                    constexpr int nt = decltype(ntc)::value, mt = decltype(mtc)::value;
                    if constexpr (vblk) c[nt][mt] = mfma_fp8_scaled<mt, QN * 2 + nt>(xf[mt][0], xf[mt][1], wf[nt][0], wf[nt][1], c[nt][mt], sa, sw);
                    else c[nt][mt] = mfma_fp8_scaled<QN * 2 + nt, mt>(wf[nt][0], wf[nt][1], xf[mt][0], xf[mt][1], c[nt][mt], sw, sa);
                });
            });
            __builtin_amdgcn_s_setprio(0);
        }
    };
    // 4-slot variant: La (A-lo, W-lo, W-hi fragments) | Ca (quadrants 00, 01) | Lb (A-hi) | Cb (11, 10):
    // half as many barriers per MFMA.
    unsigned long long* dbg = reinterpret_cast<unsigned long long*>(p.vt);
    int dbg_i = 0;
    auto stamp = [&](int t) {
        if constexpr (SLOTS == 46) {
            if ((int)blockIdx.x == (p.xp >> 20) && t == 6 && (lane == 0) && dbg) {
                __builtin_amdgcn_sched_barrier(0);
                dbg[wave * 32 + dbg_i] = __builtin_amdgcn_s_memtime();
                ++dbg_i;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    auto wait_n = [&](int n) {
        if (n == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        else if (n == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        else if (n == 9) asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)" ::: "memory");
        else if (n == 7) asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        TT_SLOT_END();
    };
    // Copy schedule:  La(t): W-hi(t+1), A-hi(t+1)   Lb(t): A-lo(t+2), W-lo(t+2) [f16c: + the scale strips of e4m3 tile t+2].
    // (Moving half of the copies into the C slots was measured neutral-to-worse: an LDS-DMA issue
    // costs ~100 cycles wherever it sits, and the C slot is then as long as the L slot.)
    // (f8c: the tile's flavour in the f16c stream -- false: fp16 MFMAs, true: scaled e4m3 MFMAs; always false otherwise)
    // Energy ablations (diagnostic library only, TT_GEMM_ENERGY; wrong results): from the third K-tile on, 1 = no operand copies
    // (the MFMAs run on stale LDS tiles: no LDS-DMA, L2 or HBM traffic), 2 = no fragment reads (stale registers: no LDS reads),
    // 3 = both.  Run under the power cap with random operands, the rate each gains is that part's share of the energy per flop.
#if TT_DIAG
    const int eabl = (p.xp & 0x80000) ? ((p.xp >> 4) & 15) : 0;
#else
    constexpr int eabl = 0;
#endif
    auto tile4 = [&](int t, auto bufc, auto f8c) {
        constexpr int B = decltype(bufc)::value;
        constexpr bool F8T = XC && decltype(f8c)::value;
        const bool dma_on = !((eabl & 1) && t >= 2), rd_on = !((eabl & 2) && t >= 2);
        const bool more1 = t + 1 < nk && dma_on, more2 = t + 2 < nk && dma_on;
        // f16c: vector-memory operations that may still be in flight at the two counted waits -- the copies of Lb(t - 1)
        // resp. Lb(t) include one scale strip when the tile they prefetch (t + 1 resp. t + 2) is an e4m3 tile
        // (a group of four e4m3 tiles, when the tile prefetched there opens one)
        const int sc_la = (XC && t + 1 >= nk1 && t + 1 < nk && ((t + 1 - nk1) & 3) == 0) ? 1 : 0;
        const int sc_lb = (XC && t + 2 >= nk1 && t + 2 < nk && ((t + 2 - nk1) & 3) == 0) ? 1 : 0;
        const char* sbuf = smem + kXcScaleOff + (((t - nk1) >> 2) & 1) * 8192 + ((t - nk1) & 3) * 2048;
        // La
        stamp(t);                                  // 0: La start
        if (more1) {
            stage_half<1, ES>(p, smem, 1, B ^ 1, tile_w(t + 1), wave, voffW, m0, n0, TB);
            stage_half<0, ES>(p, smem, 1, B ^ 1, tile_a(t + 1), wave, voffA, m0, n0, TB);
        } else if (kResLds) {
            stage_res(1);                          // last tile (B = 1): hi slots of buffer 0
        }
        stamp(t);                                  // 1: copies issued
        if (rd_on) {
            read_a(smem + slot_off(0, 0, B));
            read_w(wf0, smem + slot_off(1, 0, B));
            read_w(wf1, smem + slot_off(1, 1, B));
        }
        if constexpr (F8T) {
            sa_reg[0] = *reinterpret_cast<const uint32_t*>(sbuf + sa_off);
            sw_reg = *reinterpret_cast<const uint32_t*>(sbuf + sw_off);
        }
        stamp(t);                                  // 2: reads issued
        if (SLOTS == 46) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        stamp(t);                                  // 3: reads returned
        wait_n((t + 1 < nk || kResLds) ? 8 + sc_la : 0);
        stamp(t);                                  // 4: past barrier (Ca start)
        if constexpr (F8T) {
            mma8(acc[0][0], wf0, sa_reg[0], sw_reg, std::integral_constant<int, 0>{});
            mma8(acc[0][1], wf1, sa_reg[0], sw_reg, std::integral_constant<int, 1>{});
        } else {
            mma(acc[0][0], wf0);                   // Ca
            mma(acc[0][1], wf1);
        }
        stamp(t);                                  // 5: MFMAs issued
        TT_SLOT_END();
        // Lb
        stamp(t);                                  // 6: Lb start
        if (more2) {
            stage_half<0, ES>(p, smem, 0, B, tile_a(t + 2), wave, voffA, m0, n0, TB);
            stage_half<1, ES>(p, smem, 0, B, tile_w(t + 2), wave, voffW, m0, n0, TB);
            if (sc_lb) stage_scales(t + 2);
        } else if (kResLds) {
            stage_res(B == 0 ? 0 : 2);             // tile nk-2 (B = 0): lo slots of buffer 0; tile nk-1: of buffer 1
        }
        stamp(t);                                  // 7
        if (rd_on) read_a(smem + slot_off(0, 1, B));
        if constexpr (F8T) sa_reg[1] = *reinterpret_cast<const uint32_t*>(sbuf + 512 + sa_off);
        stamp(t);                                  // 8
        if (SLOTS == 46) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        stamp(t);                                  // 9
        wait_n((t + 2 < nk || kResLds) ? 6 + sc_lb : 0);
        stamp(t);                                  // 10: Cb start
        if constexpr (F8T) {
            mma8(acc[1][1], wf1, sa_reg[1], sw_reg, std::integral_constant<int, 1>{});
            mma8(acc[1][0], wf0, sa_reg[1], sw_reg, std::integral_constant<int, 0>{});
        } else {
            mma(acc[1][1], wf1);                   // Cb
            mma(acc[1][0], wf0);
        }
        stamp(t);                                  // 11
        TT_SLOT_END();
        stamp(t);                                  // 12
    };

    if constexpr (XC) {
        for (int t = 0; t < nk1; t += 2) {          // nk1 = K / 64 is even (K a multiple of 128)
            tile4(t, std::integral_constant<int, 0>{}, std::false_type{});
            tile4(t + 1, std::integral_constant<int, 1>{}, std::false_type{});
        }
        for (int t = nk1; t < nk; t += 2) {
            tile4(t, std::integral_constant<int, 0>{}, std::true_type{});
            tile4(t + 1, std::integral_constant<int, 1>{}, std::true_type{});
        }
    } else {
        for (int t = 0; t < nk; t += 2) {
            tile4(t, std::integral_constant<int, 0>{}, std::false_type{});
            if (t + 1 < nk) tile4(t + 1, std::integral_constant<int, 1>{}, std::false_type{});
        }
    }
    cstamp(2);
    if (!late) TT_SLOT_END();   // match the extra barrier the late group took up front
    stage_res(3);               // every operand read is done: hi slots of buffer 1

    // ---- epilogue ---------------------------------------------------------------------------------
    if constexpr (X3) {
        if (n0 + wn * 64 >= p.N) return;       // this wave's 64-column strip lies beyond N (partial last column tile): nothing to store
    }
    if constexpr (vblk) {
#pragma unroll
        for (int qm = 0; qm < 2; ++qm)
#pragma unroll
            for (int qn = 0; qn < 2; ++qn)
                gemm_epilogue_vt<2, 4, FP8, X3>(p, acc[qm][qn], m0 + qm * 128 + wm * 64, n0 + wn * 64 + qn * 32, lane);
    } else if constexpr (X3) {
        epilogue_x3<EPI, RP>(p, acc, smem, kBiasOff, m0, n0, wm, wn, lane);
    } else if constexpr (XC && EPI == TT_EPI_RESIDUAL) {
        epilogue_x3<EPI>(p, acc, smem, kBiasOff, m0, n0, wm, wn, lane);          // fp32 out = acc + bias + fp32 residual
    } else if constexpr (XC && EPI == TT_EPI_GELU) {
        epilogue_xc<EPI>(p, acc, smem, kBiasOff, m0, n0, wm, wn, lane);          // c-planes out
    } else if constexpr (XC && EPI == TT_EPI_BIAS) {
        if (p.c_lo_off) epilogue_h2(p, acc, smem, kBiasOff, m0, n0, wm, wn, lane);                  // two fp16 planes (Q, K)
        else epilogue_all<EPI, true, false>(p, acc, smem, kBiasOff, kScaleOff, m0, n0, wm, wn, wave, lane, false, NoNext{});
    } else if constexpr (EPI == TT_EPI_QKV) {
#pragma unroll
        for (int qm = 0; qm < 2; ++qm)
#pragma unroll
            for (int qn = 0; qn < 2; ++qn)
                gemm_epilogue_wide<EPI, 2, 4>(p, acc[qm][qn], m0 + qm * 128 + wm * 64, n0 + wn * 64 + qn * 32, lane);
    } else if constexpr (EPI == TT_EPI_SCAN) {
        if (p.scan_dense) scan_sample_epilogue(p, acc, tm, wm, wn, lane);
        else if (p.xp & 0x10000) scan_filter_epilogue<2>(p, acc, smem, kBiasOff, m0, wm, wn, lane);
        else scan_filter_epilogue<1>(p, acc, smem, kBiasOff, m0, wm, wn, lane);
    } else if constexpr (SLOTS == 47) {
#pragma unroll
        for (int qm = 0; qm < 2; ++qm)
#pragma unroll
            for (int qn = 0; qn < 2; ++qn)
                gemm_epilogue_wide<EPI, 2, 4, true>(p, acc[qm][qn], m0 + qm * 128 + wm * 64, n0 + wn * 64 + qn * 32, lane);
    } else {
        epilogue_all<EPI, true, FP8>(p, acc, smem, kBiasOff, kScaleOff, m0, n0, wm, wn, wave, lane, false, NoNext{}, kScaleOff, kScaleOff + 2048);
    }
    if constexpr (SLOTS == 46) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        cstamp(3);
    }
}

// ---- persistent, continuous variant (bias / GELU / tanh epilogues) -----------------------------------------
// One workgroup per CU walks the tile list (virtual block id v = blockIdx.x + i * gridDim.x, same XCD-aware
// super-tile order as the one-tile-per-block kernel) and the K stream never drains: the copy schedule simply
// runs on into the NEXT output tile (K-tile index u >= nk means K-tile u - nk of the next tile), so a tile's
// main loop starts on landed data, and the two wave groups stay one slot apart THROUGH the epilogue -- while
// waves 0-3 convert and store their accumulators, waves 4-7 still run the last MFMA slot, and vice versa.
// A K=1024 tile of the one-tile-per-block kernel spends ~20 % of its time outside the main loop (dispatch,
// first-load latency, pipeline ramp, epilogue).
// Vector-memory bookkeeping: stores count in vmcnt and return in order with the copies.  The first copies of the
// next tile (La(0)) are issued BEFORE the 16 epilogue stores, so the first two waits of a non-first tile allow
// 8 + 16 and 6 + 16 operations; by the third the stores are four slots old.
// The residual epilogue needs all eight slots for the residual tile and stays on the one-tile-per-block kernel.
template <int EPI, bool FP8 = false, bool STAMP = false>
__global__ __launch_bounds__(kThreads3, 2) void gemm_kernel_p(GemmParams p, int nwg) {
    constexpr int ES = FP8 ? 1 : 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    const int mt_n = p.M / BM3, nt_n = p.N / BN3;
    const int SN = p.sn > 0 ? p.sn : (nt_n < 4 ? nt_n : 4);
    const int SM = 32 / SN > 0 ? 32 / SN : 1;
    const int per_super = SM * SN;
    const int supers_n = (nt_n + SN - 1) / SN;
    const int nk = p.K * ES / 128;
    auto decode = [&](int v, int& m0, int& n0) {
        const int L = (v & 7) * (nwg >> 3) + (v >> 3);
        const int sidx = L / per_super, widx = L % per_super;
        const int tm = (sidx / supers_n) * SM + widx / SN;
        const int tn = (sidx % supers_n) * SN + widx % SN;
        int mm = tm * BM3;
        if constexpr (EPI == TT_EPI_SCAN) {
            if (p.scan_rows && mm + BM3 > p.scan_rows) mm = p.scan_rows - BM3;                    // the shard's last 256 rows
        }
        m0 = __builtin_amdgcn_readfirstlane(mm);
        n0 = __builtin_amdgcn_readfirstlane(tn * BN3);
        return tm < mt_n && tn < nt_n;
    };
    auto next_valid = [&](int v, int& m0, int& n0) {
        for (v += gridDim.x; v < nwg; v += gridDim.x)
            if (decode(v, m0, n0)) return v;
        return nwg;
    };

    const int frow = lane & 15, fchk = lane >> 4, sw = (frow >> 1) & 7;
    const int off0 = frow * 128 + ((fchk ^ sw) << 4);
    const int off1 = frow * 128 + (((4 + fchk) ^ sw) << 4);
    const int a_row0 = wm * 64 * 128;
    const int w_row0 = wn * 32 * 128;
    uint32_t voffA[2], voffW[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = 16 * wave + 8 * j + (lane >> 3);
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        voffA[j] = (uint32_t)r * (uint32_t)p.lda * (uint32_t)ES + (uint32_t)chunk * 16u;
        voffW[j] = (uint32_t)w_row_of(r) * (uint32_t)(p.ldw ? p.ldw : p.K) * (uint32_t)ES + (uint32_t)chunk * 16u;
    }

    int m0 = 0, n0 = 0, m1 = 0, n1 = 0;
    int v = blockIdx.x;
    if (!decode(v, m0, n0)) v = next_valid(v, m0, n0);
    if (v >= nwg) return;
    int vn = next_valid(v, m1, n1);
    bool has_next = vn < nwg;

    // bias strip of a tile (wave 0) and, fp8, its row / column scale strips (waves 1, 2): one copy per wave
    auto stage_strips = [&](int mm0, int nn0, int par) {
        const uint32_t l0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
        if (wave == 0) glds16(p.bias + nn0, lane * 16, l0 + kBiasOff + par * 1024);
        if constexpr (FP8) {
            if (wave == 1) glds16(p.a_scale + mm0, lane * 16, l0 + kScaleOff + par * 2048);
            if (wave == 2) glds16(p.w_scale + nn0, lane * 16, l0 + kScaleOff + par * 2048 + 1024);
        }
#if TT_DIAG
        if constexpr (!FP8 && (EPI == TT_EPI_BIAS || EPI == TT_EPI_GELU)) {
            if (p.lnf == 1) {   // consumer form only: column sums at kScaleOff + par KiB, row statistics at kScaleOff + 2 KiB + par 2 KiB (launched with 2 KiB more LDS)
                if (wave == 1) glds16(p.lnf_c0 + nn0, lane * 16, l0 + kScaleOff + par * 1024);
                if (wave == 3) glds16(p.lnf_rows + (size_t)mm0 * 2, lane * 16, l0 + kScaleOff + 2048 + par * 2048);
                if (wave == 5) glds16(p.lnf_rows + (size_t)(mm0 + 128) * 2, lane * 16, l0 + kScaleOff + 2048 + par * 2048 + 1024);
            }
        }
#endif
    };
    // hi / lo halves of K-tile u of the current tile, or of K-tile u - nk of the next one (nk is even: same buffer)
    auto issue_hi = [&](int u) {
        if (u < nk) {
            stage_half<1, ES>(p, smem, 1, u & 1, u, wave, voffW, m0, n0);
            stage_half<0, ES>(p, smem, 1, u & 1, u, wave, voffA, m0, n0);
        } else if (has_next) {
            stage_half<1, ES>(p, smem, 1, u & 1, u - nk, wave, voffW, m1, n1);
            stage_half<0, ES>(p, smem, 1, u & 1, u - nk, wave, voffA, m1, n1);
        }
    };
    auto issue_lo = [&](int u) {
        if (u < nk) {
            stage_half<0, ES>(p, smem, 0, u & 1, u, wave, voffA, m0, n0);
            stage_half<1, ES>(p, smem, 0, u & 1, u, wave, voffW, m0, n0);
        } else if (has_next) {
            stage_half<0, ES>(p, smem, 0, u & 1, u - nk, wave, voffA, m1, n1);
            stage_half<1, ES>(p, smem, 0, u & 1, u - nk, wave, voffW, m1, n1);
        }
    };

    const bool late = wave >= 4;
    int bpar = 0;
    bool first = true;
    // diagnostic build (TT_GEMM_ABLATE=8, tools/gemm_stamps_p): s_memtime of workgroup 0's waves 0 and 4 at the slot boundaries of
    // the first K-tiles of its 2nd..7th output tile -> p.vt[(wave >= 4) * 1024 + tile_no * 128 + index]
    int tile_no = 0;
    unsigned long long* sdbg = reinterpret_cast<unsigned long long*>(p.vt);
    auto pstamp = [&](int idx) {
        if constexpr (STAMP) {
            if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 4) && tile_no >= 1 && tile_no < 8 && idx < 128 && sdbg) {
                __builtin_amdgcn_sched_barrier(0);
                sdbg[(wave >= 4 ? 1024 : 0) + tile_no * 128 + idx] = __builtin_amdgcn_s_memtime();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // De-phasing (p.xp >> 8 = phases, in 1024-cycle units of delay per phase step): every CU of an XCD finishes its tiles at
    // the same moment, so the XCD's 32 x 128 KiB of output arrive at its 4 MiB L2 in one burst, which has to be written
    // back before it is accepted -- and the in-order vector-memory queue holds the next tile's copies behind those
    // stores.  A one-time start delay by position inside the XCD (blockIdx.x / 8: workgroups b and b + 8 share an XCD)
    // spreads the bursts over the tile period for the whole kernel (static tile lists keep the phase).
    if (const int ph = p.xp >> 8) {
        const int steps = ((blockIdx.x >> 3) % 32) * ph;       // x 1024 cycles
        for (int i = 0; i < steps; ++i) __builtin_amdgcn_s_sleep(16);
    }
    if constexpr (EPI == TT_EPI_SCAN) {     // wave-private survivor lists (scan_filter_epilogue<2>): this wave's count
        if (lane == 0) *reinterpret_cast<int*>(smem + kScanHitOff + wave * 1024) = 0;
    }
    // ---- prologue of the first tile: K-tile 0 complete, K-tile 1 without its hi halves (La(0) brings them)
    stage_strips(m0, n0, bpar);
    issue_lo(0);
    issue_hi(0);
    issue_lo(1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // A-hi(0), A-lo(1), W-lo(1) may be in flight
    TT_SLOT_END();
    if (late) TT_SLOT_END();   // waves 4-7 run one slot behind, for the whole kernel

    while (true) {
        f32x4 acc[2][2][2][4];  // [qm][qn][n-tile][m-tile]
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int d = 0; d < 4; ++d) acc[a][b][c][d] = f32x4{0.f, 0.f, 0.f, 0.f};
        ex8 xf[4][2], wf0[2][2], wf1[2][2];
        auto read_a = [&](const char* base) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                xf[mt][0] = *reinterpret_cast<const ex8*>(base + a_row0 + mt * 2048 + off0);
                xf[mt][1] = *reinterpret_cast<const ex8*>(base + a_row0 + mt * 2048 + off1);
            }
        };
        auto read_w = [&](ex8(&wf)[2][2], const char* base) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                wf[nt][0] = *reinterpret_cast<const ex8*>(base + w_row0 + nt * 2048 + off0);
                wf[nt][1] = *reinterpret_cast<const ex8*>(base + w_row0 + nt * 2048 + off1);
            }
        };
        auto mma = [&](f32x4(&c)[2][4], const ex8(&wf)[2][2]) {
            __builtin_amdgcn_s_setprio(1);
            if constexpr (FP8) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) c[nt][mt] = mfma_fp8(wf[nt][0], wf[nt][1], xf[mt][0], xf[mt][1], c[nt][mt]);
            } else {
#pragma unroll
                for (int ss = 0; ss < 2; ++ss)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt)
                            c[nt][mt] = TT_MFMA_16x16x32(wf[nt][ss], xf[mt][ss], c[nt][mt]);
            }
            __builtin_amdgcn_s_setprio(0);
        };
        auto wait_n = [&](int n) {
            if (n == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            else if (n == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
            else if (n == 22) asm volatile("s_waitcnt vmcnt(22) lgkmcnt(0)" ::: "memory");
            else if (n == 24) asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            TT_SLOT_END();
        };
        auto tile4 = [&](int t, auto bufc) {
            constexpr int B = decltype(bufc)::value;
            // this wave's 16 epilogue stores are the youngest-but-copies (the scan epilogue stores nothing: its rare
            // atomics return a value, so the compiler has already waited for them -- the plain counts are right)
            const bool after_epi = t == 0 && !first && EPI != TT_EPI_SCAN;
            // La: hi halves of K-tile t+1 (already issued, ahead of the stores, when after_epi)
            pstamp(t * 4 + 0);                        // La start
            if (!(t == 0 && !first)) issue_hi(t + 1);
            read_a(smem + slot_off(0, 0, B));
            read_w(wf0, smem + slot_off(1, 0, B));
            read_w(wf1, smem + slot_off(1, 1, B));
            // (xp bit 1: one more store-tolerant wait -- at La(1) the stores are still only behind hi(2'), lo(2'))
            if (after_epi || (t == 1 && !first && EPI != TT_EPI_SCAN && (p.xp & 2))) wait_n(24);
            else wait_n((t + 1 < nk || has_next) ? 8 : 0);
            pstamp(t * 4 + 1);                        // Ca start (past La's wait + barrier)
            mma(acc[0][0], wf0);                       // Ca
            mma(acc[0][1], wf1);
            TT_SLOT_END();
            pstamp(t * 4 + 2);                        // Lb start
            // Lb: lo halves of K-tile t+2; the other group's epilogue of the previous tile is over: the bias
            // strip it read can be refilled for the next tile
            if (t == 0 && has_next) stage_strips(m1, n1, bpar ^ 1);
            issue_lo(t + 2);
            read_a(smem + slot_off(0, 1, B));
            if (after_epi) wait_n(22);
            else wait_n((t + 2 < nk || has_next) ? 6 : 0);
            pstamp(t * 4 + 3);                        // Cb start
            mma(acc[1][1], wf1);                       // Cb
            mma(acc[1][0], wf0);
            TT_SLOT_END();
        };
        for (int t = 0; t < nk; t += 2) {
            tile4(t, std::integral_constant<int, 0>{});
            tile4(t + 1, std::integral_constant<int, 1>{});
        }
        pstamp(100);                                  // main loop done
        if (has_next) issue_hi(nk + 1);   // La(0) of the next tile, ahead of this tile's stores
        // Tile boundary.  With the groups one slot apart THROUGH the boundary their epilogues serialise: group A converts
        // and stores (4.3 k cycles for the bias epilogue, one wave per SIMD at half the vector issue rate) while B can only
        // run its last 512-cycle MFMA slot and then waits at a barrier A reaches after its epilogue; then B's epilogue runs
        // while A waits at the end of Ca(0) (stamps, tools/gemm_stamps_p: 6.3 k cycles in that one slot) -- 11.7 k of a
        // 59.7 k-cycle K = 1024 tile.  Aligned instead: A waits ONE slot for B's last MFMAs, both groups run their
        // epilogues side by side, and B waits one slot after it to fall behind again, exactly as at kernel start.
        // Measured: GELU 3.45 -> 3.41 ms, bias 3.18 -> 3.16 ms (N = 4096) -- only 1-2 %, because the stamps of the aligned
        // form show the real bound: the two epilogues together still take 8.6 k cycles, i.e. the 128 store instructions of
        // a tile (1 KiB each) issue at ~15 B/clk/CU whoever issues them (store-ISSUE bound, MI355X_MICROARCH.md's
        // "epilogue store tail" row); hiding them needs independent MFMA work on the CU, which one 8-wave workgroup with
        // 128 accumulator registers per wave does not have.
        const bool align = (p.xp & 4) == 0;    // default on (bit 2 of TT_GEMM_XP turns it OFF: the A/B switch)
        if (align && !late) TT_SLOT_END();

        // the epilogue's per-lane addresses are recomputed per tile from an opaque copy of the lane id: hoisted out
        // of the tile loop they would stay live through the main loop
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        if constexpr (EPI == TT_EPI_SCAN)
            scan_filter_epilogue<2>(p, acc, smem, kBiasOff + bpar * 1024, m0, wm, wn, lane_e);
        else
            epilogue_all<EPI, false, FP8>(p, acc, smem, kBiasOff + bpar * 1024, kScaleOff + bpar * 2048, m0, n0, wm, wn, wave, lane_e,
                                          false, NoNext{}, kScaleOff + bpar * 1024, kScaleOff + 2048 + bpar * 2048);
        pstamp(101);                                  // epilogue done (stores issued)
        ++tile_no;
        if (!has_next) {
            if (!align && !late) TT_SLOT_END();   // match the extra barrier the late group took up front
            break;
        }
        if (align && late) TT_SLOT_END();         // fall one slot behind again
        v = vn; m0 = m1; n0 = n1; bpar ^= 1; first = false;
        vn = next_valid(v, m1, n1);
        has_next = vn < nwg;
    }
}
#undef TT_SLOT_END
}  // namespace v3

// Super-tile shape: the 32 tiles an XCD works on at a time are SM row-blocks x SN column-blocks (SM * SN = 32); the A
// row-blocks of a super-tile are shared through that XCD's L2 by its SN column tiles.
inline int super_sn(int nt_n) {
    static const int env = TT_DIAG_ENV_INT("TT_GEMM_SN", 0);
    int sn = env > 0 ? env : 4;
    if (sn > nt_n) sn = nt_n;
    while (32 % sn) --sn;
    return sn;
}

inline bool small_grid_v1() {
    static const bool on = TT_DIAG_ENV_INT("TT_GEMM_SMALL_V1", 1) != 0;
    return on;
}

// fp8 operands: the 256x256 kernels only (bias / GELU / V^T epilogues), K-tiles of 128 elements
template <int EPI>
int launch_fp8(const GemmParams& p, hipStream_t st) {
    if constexpr (EPI == TT_EPI_BIAS || EPI == TT_EPI_GELU || EPI == TT_EPI_VT || EPI == TT_EPI_RESIDUAL) {
        const int nk = p.K / 128;
        if (EPI == TT_EPI_RESIDUAL && (!p.residual || p.ldr % 8)) {
            tt_set_error("gemm fp8: residual epilogue without residual");
            return TT_E_INVALID;
        }
        if (p.M % v3::BM3 || p.N % v3::BN3 || p.K % 128 || nk < 2 || (nk & 1) || p.ldc % 8 || p.lda % 16 || !p.a_scale || !p.w_scale) {
            tt_set_error("gemm fp8: M=%d N=%d K=%d must be multiples of 256/256/256 with row / column scales", p.M, p.N, p.K);
            return TT_E_UNSUPPORTED;
        }
        const int mt_n = p.M / v3::BM3, nt_n = p.N / v3::BN3;
        const int SN = super_sn(nt_n), SM = 32 / SN;
        const int supers = ((mt_n + SM - 1) / SM) * ((nt_n + SN - 1) / SN);
        int blocks = supers * SM * SN;
        blocks = (blocks + 7) / 8 * 8;
        // (the persistent kernel's fp8 GELU form spills; every fp8 GEMM runs one tile per workgroup)
        TT_SET_MAX_LDS((v3::gemm_kernel_v3<EPI, 4, true>), v3::kLds3);
        {
            TtProfScope prof(TT_K_GEMM, st);
            GemmParams q = p;
            q.sn = SN;
            hipLaunchKernelGGL((v3::gemm_kernel_v3<EPI, 4, true>), dim3(blocks), dim3(v3::kThreads3), v3::kLds3, st, q);
        }
        TT_CHECK_LAUNCH();
        return TT_OK;
    } else {
        tt_set_error("gemm fp8: epilogue %d has no fp8 form", EPI);
        return TT_E_UNSUPPORTED;
    }
}

// ---- skinny: M <= 256 rows (one query's embedding, the CLS-row tail of the last layer, the rerank head) -----------
// With a few dozen token rows a GEMM is a stream of the weight matrix and nothing else: the tiled kernels put
// 16 (N = 1024) to 64 workgroups on the chip and walk K in 64-element steps behind a barrier each (8 us at K = 1024,
// 30 us at K = 4096, times 6 GEMMs x 24 layers = most of a query embedding's 3.4 ms).  Here ONE WAVE owns 16 output
// columns x 16 rows x all of K: no LDS, no barrier, operands straight from global memory into MFMA fragments (a
// fragment is 16 contiguous bytes of a K-contiguous row in both operands), 24 K-steps of loads in flight per wave,
// N/16 x M/16 waves per launch.  The activations (<= 256 x K) and, by the row tiles of a column block, the weights
// are re-read from L2.  (Rounds 1-2: 16 columns x 64 rows per wave, 8 steps in flight -- TT_GEMM_SKINNY_MT=4.)
// Same instruction, operand order and K order as the tiled kernels, same epilogue code: results are bit-identical
// to theirs, so an embedding does not depend on whether the text was embedded alone or in a large batch.

// split-bf16 epilogue of one wave's tile (skinny kernel): same values, same operations as epilogue_x3
template <int EPI, int NT, int MT>
__device__ __forceinline__ void gemm_epilogue_tile_x3(const GemmParams& p, f32x4 (&acc)[NT][MT], int mw, int nw, int lane) {
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int n = nw + i * 16 + (lane >> 4) * 4;
        const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n);
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int m = mw + j * 16 + (lane & 15);
            float v[4] = {acc[i][j][0] + b4.x, acc[i][j][1] + b4.y, acc[i][j][2] + b4.z, acc[i][j][3] + b4.w};
            if constexpr (EPI == TT_EPI_RESIDUAL) {
                float4 r;
                if (p.res_planes) {      // (same values, same operations as epilogue_x3)
                    const uint16_t* rp = p.res_planes + (size_t)m * p.ldr + n;
                    const uint2 h = *reinterpret_cast<const uint2*>(rp), l = *reinterpret_cast<const uint2*>(rp + p.res_lo_off);
                    r = float4{elo(h.x) + elo(l.x), ehi(h.x) + ehi(l.x), elo(h.y) + elo(l.y), ehi(h.y) + ehi(l.y)};
                } else {
                    r = *reinterpret_cast<const float4*>(p.res32 + (size_t)m * p.ldr + n);
                }
                *reinterpret_cast<float4*>(p.C32 + (size_t)m * p.ldc + n) = float4{v[0] + r.x, v[1] + r.y, v[2] + r.z, v[3] + r.w};
            } else {
                if constexpr (EPI == TT_EPI_GELU) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        v[k] = v3::gelu_exact(v[k]);
                        asm("" : "+v"(v[k]));          // (see epilogue_x3)
                    }
                }
                uint2 o, l;
                o.x = pack_e2(v[0], v[1]); o.y = pack_e2(v[2], v[3]);
                l.x = pack_e2(v[0] - elo(o.x), v[1] - ehi(o.x));
                l.y = pack_e2(v[2] - elo(o.y), v[3] - ehi(o.y));
                if (p.x3_zero_lo) l = uint2{0u, 0u};
                if constexpr (EPI == TT_EPI_VT) {      // every column is a V feature: V8 layout, hi and lo planes
                    const size_t at = (size_t)(m >> 3) * p.ldvt + (size_t)(n - p.vt_col0) * 8 + (m & 7);
                    p.vt[at] = (uint16_t)(o.x & 0xFFFFu);       p.vt[at + 8] = (uint16_t)(o.x >> 16);
                    p.vt[at + 16] = (uint16_t)(o.y & 0xFFFFu);  p.vt[at + 24] = (uint16_t)(o.y >> 16);
                    p.vt_lo[at] = (uint16_t)(l.x & 0xFFFFu);      p.vt_lo[at + 8] = (uint16_t)(l.x >> 16);
                    p.vt_lo[at + 16] = (uint16_t)(l.y & 0xFFFFu); p.vt_lo[at + 24] = (uint16_t)(l.y >> 16);
                } else {
                    uint16_t* cp = p.C + (size_t)m * p.ldc + n;
                    *reinterpret_cast<uint2*>(cp) = o;
                    *reinterpret_cast<uint2*>(cp + p.c_lo_off) = l;
                }
            }
        }
    }
}

// X3: split-plane operands (GemmParams.x3) -- the same loop over a virtual K stream of 3 K / 32 steps: per 32 K elements the
// three products hi.hi, x_hi.w_lo, x_lo.w_hi, in the tiled kernel's order, so a row's result does not depend on which kernel computed it
// MT row tiles of 16 per wave, PF K-steps of loads in flight.  Round 3: one row tile per wave and 24 steps in flight (a
// query's 64 padded rows on 4 x N/16 waves instead of N/16: the N = 1024 GEMMs had 64 waves on 256 CUs, each waiting for its
// own 8 loads, 0.44 TB/s of weights); the MFMA chain of a 16 x 16 output tile is the same either way, so the bits are too.
template <int EPI, bool X3 = false, int MT = 1, int PF = 24>
__global__ __launch_bounds__(64) void gemm_skinny_kernel(GemmParams p) {
    const int lane = threadIdx.x;
    const int n0 = blockIdx.x * 16, m0 = blockIdx.y * (16 * MT);
    const int frow = lane & 15, fchk = lane >> 4;
    const uint16_t* wp = p.W + (size_t)(n0 + frow) * (X3 ? p.ldw : p.K) + fchk * 8;
    const uint16_t* ap = p.A + (size_t)(m0 + frow) * p.lda + fchk * 8;
    const size_t a16 = (size_t)16 * p.lda;
    f32x4 acc[1][MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[0][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    ex8 wb[PF], ab[PF][MT];
    const int nks1 = p.K / 32;
    const int nks = X3 ? 3 * nks1 : nks1;
    // (virtual step s = 3 * (K step) + term; term 0: hi.hi, 1: x_hi.w_lo, 2: x_lo.w_hi -- the tiled kernel's MFMA order per 32 K elements)
    auto step_a = [&](int s) { if constexpr (X3) return s / 3 + (s % 3 == 2 ? nks1 : 0); else return s; };
    auto step_w = [&](int s) { if constexpr (X3) return s / 3 + (s % 3 == 1 ? nks1 : 0); else return s; };
#pragma unroll
    for (int s = 0; s < PF; ++s)
        if (s < nks) {
            wb[s] = *reinterpret_cast<const ex8*>(wp + step_w(s) * 32);
#pragma unroll
            for (int j = 0; j < MT; ++j) ab[s][j] = *reinterpret_cast<const ex8*>(ap + j * a16 + step_a(s) * 32);
        }
    for (int ks = 0; ks < nks; ks += PF) {
#pragma unroll
        for (int s = 0; s < PF; ++s) {
            if (ks + s < nks) {
#pragma unroll
                for (int j = 0; j < MT; ++j)
                    acc[0][j] = TT_MFMA_16x16x32(wb[s], ab[s][j], acc[0][j]);
                const int nx = ks + s + PF;
                if (nx < nks) {
                    wb[s] = *reinterpret_cast<const ex8*>(wp + step_w(nx) * 32);
#pragma unroll
                    for (int j = 0; j < MT; ++j) ab[s][j] = *reinterpret_cast<const ex8*>(ap + j * a16 + step_a(nx) * 32);
                }
            }
        }
    }
    if constexpr (X3) gemm_epilogue_tile_x3<EPI, 1, MT>(p, acc, m0, n0, lane);
    else gemm_epilogue_tile<EPI, 1, MT>(p, acc, m0, n0, lane);
}

// ---- staged 128x128 kernel (round 6): the one- and two-round regime of a lone caller's rerank -----------------------------------------
// The reference's own call (K = 10 per index, top_n = 5: 10-20 pairs, M = 1-8 k rows) gives the 256x256 split-plane kernel 16-120 tiles
// for the N = 1024 projections: a launch lasts one 256x256 tile's latency on a fraction of the chip (M = 3072: attention output 64 us,
// FFN-down 190 us -- the same as at M = 7424), and the 128x128 kernel above has no split-plane form.  This is that form: a 128x128 tile
// per workgroup of 8 waves (64 x 32 each, two per SIMD), a FOUR-stage ring of 32-KiB K-steps in 128 KiB of LDS (one workgroup per CU),
// the copies of three steps in flight behind the one being multiplied (counted vmcnt, one raw barrier per step), fragments read with asm
// ds_reads one step ahead of the MFMAs (hipcc would put vmcnt(0) in front of ordinary reads), the step's LDS-DMA issues BETWEEN its MFMAs.
//   split planes: step = 32 K elements: [A hi][A lo][W hi][W lo], 128 rows x 64 B each; 16-B slot c of row r holds chunk
//                 c ^ (3 if r & 8 else 0): ds_read_b128 serves the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... -- every
//                 group holds all 16 fragment rows, rows 4-11 with the neighbouring chunk -- and the four rows that share a 64-byte quarter
//                 of the 256-byte bank row (r, r + 4, r + 8, r + 12) then sit in four different slots (with c ^ ((r >> 2) & 3), the
//                 swizzle for CONTIGUOUS 16-lane groups, every read was a 2-way conflict: SQ_LDS_BANK_CONFLICT 0 now); per step and
//                 16x16 output tile the three products hi.hi, x_hi.w_lo, x_lo.w_hi in the tiled kernel's order
//   16-bit:       step = 64 K elements: [A 128 x 128 B][W 128 x 128 B], the 128x128 kernel's swizzle and fragment order
// Measured (profiles/r06_staged_ab.log, r06_staged_pmc.log, r06_fill_rate.log): split planes, M = 3072: attention output 63 -> 28 us,
// FFN-down 190 -> 84 us; M = 1024: all four projections 2.0-2.5 x; two rounds (M = 5-8 k) 1.1-1.2 x.  16-bit, one round: 17 -> 15 and
// 49 -> 42 us.  Where a step's 0.64 us go (an ablation build, EXPERIMENTS.md round 6; split-plane FFN-down, 128 steps): with L2-hot
// operands nothing changes -- it is not memory; MFMAs + barriers alone 0.5 us per step (768 matrix-pipe cycles per SIMD: the 0.6-0.65 of
// the matrix peak the 256x256 kernel also reaches), copies + barriers alone 0.55 us, the empty loop (waits + barrier) 0.15 us: the step
// is max(MFMAs, copies) + the barrier, within 1.3 x of its matrix-pipe time.  (A CU with nothing else to do takes in a 32-KiB step per
// 0.43 us when every CU loads -- fill_rate.cpp: 77 GB/s per CU over 192 CUs, 104 over 64; ONE step in flight 45 GB/s, the two-stage
// kernel's rate.)  First version, 4 waves, the 8 issues of a step in a block in front of its MFMAs: 0.78 us per step; between the MFMAs
// 0.66; 8 waves 0.64 wherever the issues sit.  The 256x256 kernel needs half the fill per flop: beyond two rounds it wins.
// One accumulator per output element, K ascending in steps of 32, the same MFMA and the same epilogue code as the other kernels:
// the same bits, whichever kernel the row count selects (tests/test_x3_gpu.py::test_gemm_x3_rows_do_not_depend_on_the_kernel...,
// tests/test_encoder_gpu.py::test_rows_do_not_depend_on_the_kernel_that_computed_them).
constexpr int kStagedStages = 4;
constexpr int kStagedStageBytes = 32768;
constexpr int kStagedLds = kStagedStages * kStagedStageBytes;   // 128 KiB
constexpr int kStagedThreads = 512;                             // 8 waves: 2 (rows of 64) x 4 (columns of 32), two per SIMD

__device__ __forceinline__ void lds_wait12(v3::u32x4 (&a)[12]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                                          "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]));
}
__device__ __forceinline__ ex8 as_ex8(v3::u32x4 v) { return __builtin_bit_cast(ex8, v); }

template <int EPI, bool X3>
__global__ __launch_bounds__(kStagedThreads, 1) void gemm_staged_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;          // the wave's 64 x 32 piece of the tile: rows wm * 64, columns wn * 32

    // block -> tile: the 128x128 kernel's map (XCD-contiguous ranges, then 8 x SN super-tiles)
    const int mt_n = p.M / BM, nt_n = p.N / BN;
    int L = blockIdx.x;
    {
        const int nwg = gridDim.x;
        if ((nwg & 7) == 0) L = (L & 7) * (nwg >> 3) + (L >> 3);
    }
    const int SN = nt_n < 8 ? nt_n : 8;
    const int SM = 8;
    const int per_super = SM * SN;
    const int supers_n = (nt_n + SN - 1) / SN;
    const int s = L / per_super, w = L % per_super;
    const int tm = (s / supers_n) * SM + w / SN;
    const int tn = (s % supers_n) * SN + w % SN;
    if (tm >= mt_n || tn >= nt_n) return;
    const int m0 = tm * BM, n0 = tn * BN;

    constexpr int KE = X3 ? 32 : 64;             // K elements per step
    const int nk = p.K / KE;
    const int ldw = X3 ? p.ldw : p.K;
    // this wave's 4 copies per step (1 KiB each): per-lane byte offsets from (operand + first tile row + step), constant for the whole
    // kernel.  16-bit: rows 16 w + [0, 16) of A and of W, 8 rows of 128 B per copy.  Split planes: rows 16 w + [0, 16) of each of the
    // four planes, one copy of 16 rows x 64 B per plane.  Copies 0, 1 read A, copies 2, 3 read W.
    uint32_t voff[4], ldst[4];
    if constexpr (!X3) {
        const int lrow = lane >> 3, slot = lane & 7;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = 16 * wave + 8 * j + lrow;
            const int chunk = slot ^ ((row >> 1) & 7);
            voff[j] = (uint32_t)row * (uint32_t)p.lda * 2u + chunk * 16;
            voff[2 + j] = (uint32_t)row * (uint32_t)ldw * 2u + chunk * 16;
            ldst[j] = (16 * wave + 8 * j) * 128;
            ldst[2 + j] = 16384 + ldst[j];
        }
    } else {
        const int row = 16 * wave + (lane >> 2), slot = lane & 3;
        const int chunk = slot ^ (((row >> 3) & 1) * 3);
        voff[0] = (uint32_t)row * (uint32_t)p.lda * 2u + chunk * 16;                    // A hi
        voff[1] = voff[0] + (uint32_t)p.K * 2u;                                          // A lo: K elements behind in the row
        voff[2] = (uint32_t)row * (uint32_t)ldw * 2u + chunk * 16;                      // W hi
        voff[3] = voff[2] + (uint32_t)p.K * 2u;                                          // W lo
#pragma unroll
        for (int pl = 0; pl < 4; ++pl) ldst[pl] = pl * 8192 + 16 * wave * 64;
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    auto sgpr_ptr = [](const char* ptr) {
        const unsigned long long b64 = reinterpret_cast<unsigned long long>(ptr);
        const unsigned int blo = __builtin_amdgcn_readfirstlane((unsigned int)b64);
        const unsigned int bhi = __builtin_amdgcn_readfirstlane((unsigned int)(b64 >> 32));
        return reinterpret_cast<const char*>(((unsigned long long)bhi << 32) | blo);
    };
    const char* rowA = reinterpret_cast<const char*>(p.A) + (size_t)m0 * p.lda * 2;
    const char* rowW = reinterpret_cast<const char*>(p.W) + (size_t)n0 * ldw * 2;
    auto issue = [&](int kt) {
        const uint32_t st = lds0 + (kt & (kStagedStages - 1)) * kStagedStageBytes;
        const char* a = sgpr_ptr(rowA + (size_t)kt * KE * 2);
        const char* wgt = sgpr_ptr(rowW + (size_t)kt * KE * 2);
        v3::glds16(a, voff[0], st + ldst[0]);
        v3::glds16(a, voff[1], st + ldst[1]);
        v3::glds16(wgt, voff[2], st + ldst[2]);
        v3::glds16(wgt, voff[3], st + ldst[3]);
    };

    f32x4 acc[2][4];  // [nt][mt]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // per-lane fragment addresses inside a stage: row (l & 15) of a 16-row tile (+ i * 16 rows as the immediate), chunk (l >> 4)
    const int frow = lane & 15, fchk = lane >> 4;
    uint32_t fa[2], fw[2];      // 16-bit: sub-step 0 / 1 of the A and W tiles; split planes: hi / lo plane of A and of W
    if constexpr (!X3) {
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
            const int sw = ((4 * ss + fchk) ^ ((frow >> 1) & 7)) << 4;
            fa[ss] = (wm * 64 + frow) * 128 + sw;
            fw[ss] = 16384 + (wn * 32 + frow) * 128 + sw;
        }
    } else {
        const int sw = (fchk ^ (((frow >> 3) & 1) * 3)) << 4;
        fa[0] = (wm * 64 + frow) * 64 + sw;
        fa[1] = 8192 + fa[0];
        fw[0] = 16384 + (wn * 32 + frow) * 64 + sw;
        fw[1] = 8192 + fw[0];
    }
    constexpr int RS = X3 ? 16 * 64 : 16 * 128;     // bytes from one 16-row tile to the next (the reads' immediate offset)
#define TT_STAGED_RD4(dst, at, addr)                                                                  \
    do {                                                                                              \
        const uint32_t a_ = (addr);                                                                   \
        dst[(at) + 0] = v3::lds_read128_async<0>(a_);      dst[(at) + 1] = v3::lds_read128_async<RS>(a_);     \
        dst[(at) + 2] = v3::lds_read128_async<2 * RS>(a_); dst[(at) + 3] = v3::lds_read128_async<3 * RS>(a_); \
    } while (0)
#define TT_STAGED_RD2(dst, at, addr)                                                                  \
    do {                                                                                              \
        const uint32_t a_ = (addr);                                                                   \
        dst[(at) + 0] = v3::lds_read128_async<0>(a_);      dst[(at) + 1] = v3::lds_read128_async<RS>(a_);     \
    } while (0)
    // fragments of a step, 12 registers: 16-bit {x[0..3], w[0..1]} of sub-step 0, then of sub-step 1;
    //                                    split planes {x_hi[0..3], w_hi[0..1], w_lo[0..1], x_lo[0..3]}
    auto read_frags = [&](v3::u32x4 (&f)[12], int kt) {
        const uint32_t st = lds0 + (kt & (kStagedStages - 1)) * kStagedStageBytes;
        if constexpr (!X3) {
            TT_STAGED_RD4(f, 0, st + fa[0]); TT_STAGED_RD2(f, 4, st + fw[0]);
            TT_STAGED_RD4(f, 6, st + fa[1]); TT_STAGED_RD2(f, 10, st + fw[1]);
        } else {
            TT_STAGED_RD4(f, 0, st + fa[0]); TT_STAGED_RD2(f, 4, st + fw[0]);
            TT_STAGED_RD2(f, 6, st + fw[1]); TT_STAGED_RD4(f, 8, st + fa[1]);
        }
    };
    // the step's MFMAs in their fixed order -- (split planes) hi.hi, x_hi.w_lo, x_lo.w_hi over the wave's 8 output tiles, (16-bit)
    // sub-step 0, sub-step 1 -- cut into 4 equal groups; between(g) runs after group g (the step's 4 LDS-DMA issues, one per gap)
    auto multiply = [&](const v3::u32x4 (&f)[12], auto&& between) {
        constexpr int NMM = X3 ? 24 : 16;
        v3::static_for<NMM>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            constexpr int ph = t / 8, i = (t % 8) / 4, j = t % 4;
            if constexpr (!X3) acc[i][j] = TT_MFMA_16x16x32(as_ex8(f[(ph ? 10 : 4) + i]), as_ex8(f[(ph ? 6 : 0) + j]), acc[i][j]);
            else if constexpr (ph == 0) acc[i][j] = TT_MFMA_16x16x32(as_ex8(f[4 + i]), as_ex8(f[j]), acc[i][j]);             // hi.hi
            else if constexpr (ph == 1) acc[i][j] = TT_MFMA_16x16x32(as_ex8(f[6 + i]), as_ex8(f[j]), acc[i][j]);             // x_hi.w_lo
            else acc[i][j] = TT_MFMA_16x16x32(as_ex8(f[4 + i]), as_ex8(f[8 + j]), acc[i][j]);                                // x_lo.w_hi
            // copy 0 of the step goes out right behind the barrier (step(), below), copies 1-3 after each third of the MFMAs -- the last
            // one at the very end: the texture addresser should not run dry while the waves pass the barrier and read their fragments
            constexpr int g = (3 * (t + 1)) / NMM;                       // thirds completed after this MFMA
            if constexpr (g >= 1 && (3 * t) / NMM < g) {
                __builtin_amdgcn_sched_barrier(0);
                between(std::integral_constant<int, g>{});
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    };
    auto wait_groups = [](int g) {                  // at most g of this wave's 4-copy groups still in flight
        if (g >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (g == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (g == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    // One step: (1) this wave's copies of step kt + 1 have landed, (2) its fragment reads of step kt (issued a step ago) have
    // returned, (3) barrier: both hold for every wave -- stage kt % 4 is free and stage (kt + 1) % 4 is complete, (4) the fragment
    // reads of step kt + 1 are issued and return while the matrix pipe runs step kt, (5) the copies of step kt + 4 go into the freed
    // stage (three steps of lead): the first right behind the barrier, the others after each third of the MFMAs.  An LDS-DMA issue holds
    // the wave's in-order instruction stream until the texture addresser takes it (100-150 cycles each); issued in a block in front of
    // the MFMAs that time ADDS to theirs when a SIMD has one wave (4 waves: 0.78 us per step, between the MFMAs 0.66).  Two waves per
    // SIMD: while one waits in an issue the other feeds the matrix pipe (0.64, wherever the issues sit).
    auto step = [&](v3::u32x4 (&cur)[12], v3::u32x4 (&nxt)[12], int kt) {
        if (kt + 1 < nk) wait_groups(nk - 2 - kt < 2 ? nk - 2 - kt : 2);
        lds_wait12(cur);
        __builtin_amdgcn_s_barrier();
        if (kt + kStagedStages < nk) {
            const int nx = kt + kStagedStages;
            const uint32_t st = lds0 + (nx & (kStagedStages - 1)) * kStagedStageBytes;
            const char* a = sgpr_ptr(rowA + (size_t)nx * KE * 2);
            const char* wgt = sgpr_ptr(rowW + (size_t)nx * KE * 2);
            v3::glds16(a, voff[0], st + ldst[0]);
            if (kt + 1 < nk) read_frags(nxt, kt + 1);
            __builtin_amdgcn_sched_barrier(0);
            multiply(cur, [&](auto gc) {
                constexpr int g = decltype(gc)::value;
                v3::glds16(g < 2 ? a : wgt, voff[g], st + ldst[g]);
            });
        } else {
            if (kt + 1 < nk) read_frags(nxt, kt + 1);
            __builtin_amdgcn_sched_barrier(0);
            multiply(cur, [](auto) {});
        }
        __builtin_amdgcn_sched_barrier(0);
    };

#pragma unroll
    for (int kt = 0; kt < kStagedStages; ++kt)
        if (kt < nk) issue(kt);
    wait_groups(nk - 1 < 3 ? nk - 1 : 3);
    __builtin_amdgcn_s_barrier();
    v3::u32x4 fA[12], fB[12];
    read_frags(fA, 0);
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        step(fA, fB, kt);
        step(fB, fA, kt + 1);
    }
    if (kt < nk) step(fA, fB, kt);
#undef TT_STAGED_RD4
#undef TT_STAGED_RD2
    if constexpr (X3) gemm_epilogue_tile_x3<EPI, 2, 4>(p, acc, m0 + wm * 64, n0 + wn * 32, lane);
    else gemm_epilogue_tile<EPI, 2, 4>(p, acc, m0 + wm * 64, n0 + wn * 32, lane);
}

// (diagnostic library: TT_GEMM_STAGED=0 puts the 256x256 split-plane kernel back; TT_GEMM_STAGED_MAX = the 128x128-tile count up to
// which the staged kernel takes a split-plane GEMM, default the CU count)
inline bool staged_enabled() {
    static const bool on = TT_DIAG_ENV_INT("TT_GEMM_STAGED", 1) != 0;
    return on;
}

// ---- relay form of the skinny kernel (round 6) ------------------------------------------------------------------------------
// The one-wave kernel above is a chain of memory round trips: a wave keeps PF K-steps of loads in flight and every refill waits
// for HBM again -- a K = 4096 projection is 128 (split planes: 384) steps = 6 (16) round trips of ~2.5 us while 255 of the
// CU's wave slots idle (split-plane FFN-down: 43 us for 16 MB of weights).  Bit-identity with the tiled kernels pins the MFMA
// CHAIN of an output tile (one accumulator, K ascending in steps of 32), not who issues it: here a workgroup of NW waves owns the
// 16 x 16 output tile, wave w loads the fragments of K-steps [t CH, (t + 1) CH) for its turns t = w, w + NW, ... -- ALL waves'
// loads are in flight from the first cycle -- and the accumulator is relayed through LDS (1 KiB) from turn to turn: wave t % NW
// reads it, runs its CH (x 3) MFMAs, writes it back, one workgroup barrier per turn.  Same instruction, operands and order per
// output element as gemm_skinny_kernel and the tiled kernels: the same bits (tests/test_configs_gpu.py, test_encoder_gpu.py).
template <int EPI, bool X3 = false, int NW = 8, int CH = 8>
__global__ __launch_bounds__(64 * NW) void gemm_relay_kernel(GemmParams p) {
    __shared__ f32x4 relay[64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
    const int frow = lane & 15, fchk = lane >> 4;
    const uint16_t* wp = p.W + (size_t)(n0 + frow) * (X3 ? p.ldw : p.K) + fchk * 8;
    const uint16_t* ap = p.A + (size_t)(m0 + frow) * p.lda + fchk * 8;
    const int nks1 = p.K / 32;                      // K-steps of 32 elements
    const int turns = (nks1 + CH - 1) / CH;
    const size_t lo = (size_t)nks1 * 32;            // split planes: the lo plane sits K elements behind the hi plane in a row
    ex8 wh[CH], xh[CH], wl[X3 ? CH : 1], xl[X3 ? CH : 1];
    auto issue = [&](int t) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int k = t * CH + c;
            if (k < nks1) {
                wh[c] = *reinterpret_cast<const ex8*>(wp + (size_t)k * 32);
                xh[c] = *reinterpret_cast<const ex8*>(ap + (size_t)k * 32);
                if constexpr (X3) {
                    wl[c] = *reinterpret_cast<const ex8*>(wp + lo + (size_t)k * 32);
                    xl[c] = *reinterpret_cast<const ex8*>(ap + lo + (size_t)k * 32);
                }
            }
        }
    };
    if (wave < turns) issue(wave);
    f32x4 acc[1][1] = {{f32x4{0.f, 0.f, 0.f, 0.f}}};
    for (int t = 0; t < turns; ++t) {
        if ((t % NW) == wave) {
            if (t > 0) acc[0][0] = relay[lane];
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (t * CH + c < nks1) {
                    acc[0][0] = TT_MFMA_16x16x32(wh[c], xh[c], acc[0][0]);
                    if constexpr (X3) {          // the tiled kernel's order per 32 K elements: hi.hi, x_hi.w_lo, x_lo.w_hi
                        acc[0][0] = TT_MFMA_16x16x32(wl[c], xh[c], acc[0][0]);
                        acc[0][0] = TT_MFMA_16x16x32(wh[c], xl[c], acc[0][0]);
                    }
                }
            }
            if (t + 1 < turns) relay[lane] = acc[0][0];
            if (t + NW < turns) issue(t + NW);
        }
        if (t + 1 < turns) __syncthreads();
    }
    if (((turns - 1) % NW) == wave) {
        if constexpr (X3) gemm_epilogue_tile_x3<EPI, 1, 1>(p, acc, m0, n0, lane);
        else gemm_epilogue_tile<EPI, 1, 1>(p, acc, m0, n0, lane);
    }
}

// (diagnostic library: TT_GEMM_RELAY=0 puts the one-wave kernel back -- the A/B switch)
inline bool skinny_relay() {
    static const bool on = TT_DIAG_ENV_INT("TT_GEMM_RELAY", 1) != 0;
    return on;
}
inline bool skinny_relay_16() {
    static const bool on = TT_DIAG_ENV_INT("TT_GEMM_RELAY", 1) == 2;
    return on;
}

// TT_GEMM_SKINNY_MT=4: round 2's shape (four row tiles per wave, 8 steps in flight), the A/B switch
inline bool skinny_mt4() {
    static const bool on = TT_DIAG_ENV_INT("TT_GEMM_SKINNY_MT", 0) == 4;
    return on;
}

// split-bf16 operands (GemmParams.x3): the 256x256 one-tile kernel, bias / GELU (planes out), residual (fp32 out), V^T
template <int EPI>
int launch_x3(const GemmParams& p, hipStream_t st) {
    if constexpr (EPI == TT_EPI_BIAS || EPI == TT_EPI_GELU || EPI == TT_EPI_VT || EPI == TT_EPI_RESIDUAL) {
        const int ldw = p.ldw ? p.ldw : p.K;
        // up to 256 rows (one query's embedding): the weight-streaming skinny kernel on the same virtual K stream -- a
        // 256-row tile grid would put 4-16 workgroups on the chip (11.5 ms per 24-layer forward instead of ~3)
        if (tt_gemm_skinny_enabled() && p.M > 0 && p.M <= 256 && p.M % 64 == 0 && p.N % 16 == 0 && p.K % 32 == 0 && p.lda >= 2 * p.K &&
            ldw >= 2 * p.K && p.lda % 8 == 0 && ldw % 8 == 0 && p.A && p.W && p.bias) {
            if constexpr (EPI == TT_EPI_RESIDUAL) {
                if (!p.C32 || (!p.res32 && !p.res_planes) || p.ldc % 4 || p.ldr % 4 || (p.res_planes && (p.ldr % 8 || p.res_lo_off % 8))) { tt_set_error("gemm x3: residual epilogue needs fp32 C32 and res32 or res_planes"); return TT_E_INVALID; }
            } else if constexpr (EPI == TT_EPI_VT) {
                if (!p.vt || !p.vt_lo) { tt_set_error("gemm x3: V^T epilogue needs vt / vt_lo"); return TT_E_INVALID; }
            } else {
                if (!p.C || p.ldc % 4 || p.c_lo_off % 4 || p.c_lo_off < p.N) { tt_set_error("gemm x3: planes output needs C, c_lo_off >= N"); return TT_E_INVALID; }
            }
            GemmParams q = p;
            q.ldw = ldw;
            {
                TtProfScope prof(TT_K_GEMM, st);
                if (skinny_mt4()) hipLaunchKernelGGL((gemm_skinny_kernel<EPI, true, 4, 8>), dim3(p.N / 16, p.M / 64), dim3(64), 0, st, q);
                else if (skinny_relay()) hipLaunchKernelGGL((gemm_relay_kernel<EPI, true>), dim3(p.N / 16, p.M / 16), dim3(64 * 8), 0, st, q);
                else hipLaunchKernelGGL((gemm_skinny_kernel<EPI, true>), dim3(p.N / 16, p.M / 16), dim3(64), 0, st, q);
            }
            TT_CHECK_LAUNCH();
            return TT_OK;
        }
        if (p.M % v3::BM3 || p.N % 64 || p.K % 64 || p.K < 128 || p.lda < 2 * p.K || ldw < 2 * p.K || p.lda % 8 || ldw % 8 || !p.A ||
            !p.W || !p.bias) {
            tt_set_error("gemm x3: M=%d N=%d K=%d lda=%d ldw=%d: M a multiple of 256, N and K of 64, planes [.][>= 2K]", p.M, p.N, p.K, p.lda, ldw);
            return TT_E_UNSUPPORTED;
        }
        if constexpr (EPI == TT_EPI_RESIDUAL) {
            if (!p.C32 || (!p.res32 && !p.res_planes) || p.ldc % 4 || p.ldr % 4 || (p.res_planes && (p.ldr % 8 || p.res_lo_off % 8))) { tt_set_error("gemm x3: residual epilogue needs fp32 C32 and res32 or res_planes"); return TT_E_INVALID; }
        } else if constexpr (EPI == TT_EPI_VT) {
            if (!p.vt || !p.vt_lo || p.ldvt % 8) { tt_set_error("gemm x3: V^T epilogue needs vt / vt_lo"); return TT_E_INVALID; }
        } else {
            if (!p.C || p.ldc % 8 || p.c_lo_off % 8 || p.c_lo_off < p.N) { tt_set_error("gemm x3: planes output needs C, c_lo_off >= N"); return TT_E_INVALID; }
        }
        // a lone caller's rerank (the reference's 10-20 pairs: M = 1-8 k rows): up to TWO rounds of 128x128 tiles the staged kernel beats
        // the 256x256 kernel's one round on a fraction of the CUs -- one round 2.1-2.4 x (M = 3072: 63 -> 28 us, 190 -> 88 us), two rounds
        // 1.1-1.2 x (M = 7424: 68 -> 57, 206 -> 182 us); beyond that the big tile's half fill per flop wins (profiles/r06_staged_ab.log)
        static const int staged_max = TT_DIAG_ENV_INT("TT_GEMM_STAGED_MAX", 0);
        const int mt1 = p.M / BM, nt1 = p.N / BN;
        if (staged_enabled() && p.N % BN == 0 && (long long)mt1 * nt1 <= (staged_max ? staged_max : 2 * tt_cu_count_cached())) {
            const int SN1 = nt1 < 8 ? nt1 : 8, SM1 = 8;
            int blocks1 = ((mt1 + SM1 - 1) / SM1) * ((nt1 + SN1 - 1) / SN1) * SM1 * SN1;
            blocks1 = (blocks1 + 7) / 8 * 8;
            TT_SET_MAX_LDS((gemm_staged_kernel<EPI, true>), kStagedLds);
            {
                TtProfScope prof(TT_K_GEMM, st);
                GemmParams q = p;
                q.ldw = ldw;
                hipLaunchKernelGGL((gemm_staged_kernel<EPI, true>), dim3(blocks1), dim3(kStagedThreads), kStagedLds, st, q);
            }
            TT_CHECK_LAUNCH();
            return TT_OK;
        }
        const int mt_n = p.M / v3::BM3, nt_n = (p.N + v3::BN3 - 1) / v3::BN3;      // (N % 64 == 0: the last column tile may be partial)
        const int SN = super_sn(nt_n), SM = 32 / SN;
        const int supers = ((mt_n + SM - 1) / SM) * ((nt_n + SN - 1) / SN);
        int blocks = supers * SM * SN;
        blocks = (blocks + 7) / 8 * 8;
        TT_SET_MAX_LDS((v3::gemm_kernel_v3<EPI, 4, false, true>), v3::kLds3);
        {
            TtProfScope prof(TT_K_GEMM, st);
            GemmParams q = p;
            q.sn = SN;
            q.ldw = ldw;
            if constexpr (EPI == TT_EPI_RESIDUAL) {
                if (p.res_planes) {
                    TT_SET_MAX_LDS((v3::gemm_kernel_v3<EPI, 4, false, true, false, true>), v3::kLds3);
                    hipLaunchKernelGGL((v3::gemm_kernel_v3<EPI, 4, false, true, false, true>), dim3(blocks), dim3(v3::kThreads3), v3::kLds3, st, q);
                } else {
                    hipLaunchKernelGGL((v3::gemm_kernel_v3<EPI, 4, false, true>), dim3(blocks), dim3(v3::kThreads3), v3::kLds3, st, q);
                }
            } else {
                hipLaunchKernelGGL((v3::gemm_kernel_v3<EPI, 4, false, true>), dim3(blocks), dim3(v3::kThreads3), v3::kLds3, st, q);
            }
        }
        TT_CHECK_LAUNCH();
        return TT_OK;
    } else {
        tt_set_error("gemm x3: epilogue %d has no split-bf16 form", EPI);
        return TT_E_UNSUPPORTED;
    }
}

// f16c operands (GemmParams.xc): the 256x256 one-tile kernel on the fp16 + scaled-e4m3 K stream; bias (fp16 out), GELU
// (c-planes out), residual (fp32 out), V^T (fp16 V8 out).  fp16 instantiation only.
template <int EPI>
int launch_xc(const GemmParams& p, hipStream_t st) {
    if constexpr (kF16 && (EPI == TT_EPI_BIAS || EPI == TT_EPI_GELU || EPI == TT_EPI_VT || EPI == TT_EPI_RESIDUAL)) {
        const int ldw = p.ldw ? p.ldw : 2 * p.K;
        if (p.M % v3::BM3 || p.N % v3::BN3 || p.K % 256 || p.K < 256 || p.lda < 2 * p.K || ldw < 2 * p.K || p.lda % 8 || ldw % 8 || !p.A ||
            !p.W || !p.bias || !p.a_scales || !p.w_scales) {
            tt_set_error("gemm f16c: M=%d N=%d K=%d lda=%d ldw=%d: M, N, K multiples of 256, c-planes [.][>= 2K uint16] with tiled scales",
                         p.M, p.N, p.K, p.lda, ldw);
            return TT_E_UNSUPPORTED;
        }
        if constexpr (EPI == TT_EPI_RESIDUAL) {
            if (!p.C32 || !p.res32 || p.ldc % 4 || p.ldr % 4) { tt_set_error("gemm f16c: residual epilogue needs fp32 C32 / res32"); return TT_E_INVALID; }
        } else if constexpr (EPI == TT_EPI_VT) {
            if (!p.vt || p.ldvt % 8) { tt_set_error("gemm f16c: V^T epilogue needs vt"); return TT_E_INVALID; }
        } else if constexpr (EPI == TT_EPI_GELU) {
            if (!p.C || !p.c_scales || p.ldc != 2 * p.N) { tt_set_error("gemm f16c: c-planes output needs C, c_scales, ldc = 2 N"); return TT_E_INVALID; }
        } else {
            if (!p.C || p.ldc % 8 || p.c_lo_off % 8 || (p.c_lo_off && p.c_lo_off < p.N)) {
                tt_set_error("gemm f16c: fp16 output needs C, ldc %% 8 == 0 (two planes: c_lo_off >= N, a multiple of 8)");
                return TT_E_INVALID;
            }
        }
        const int mt_n = p.M / v3::BM3, nt_n = p.N / v3::BN3;
        const int SN = super_sn(nt_n), SM = 32 / SN;
        const int supers = ((mt_n + SM - 1) / SM) * ((nt_n + SN - 1) / SN);
        int blocks = supers * SM * SN;
        blocks = (blocks + 7) / 8 * 8;
        TT_SET_MAX_LDS((v3::gemm_kernel_v3<EPI, 4, false, false, true>), v3::kLdsXc);
        {
            TtProfScope prof(TT_K_GEMM, st);
            GemmParams q = p;
            q.sn = SN;
            q.ldw = ldw;
            q.nt_store = 1;
            hipLaunchKernelGGL((v3::gemm_kernel_v3<EPI, 4, false, false, true>), dim3(blocks), dim3(v3::kThreads3), v3::kLdsXc, st, q);
        }
        TT_CHECK_LAUNCH();
        return TT_OK;
    } else {
        tt_set_error("gemm f16c: epilogue %d has no f16c form (fp16 instantiation: bias, GELU, residual, V^T)", EPI);
        return TT_E_UNSUPPORTED;
    }
}

bool skinny_shape(const GemmParams& p) {
    return !p.fp8 && tt_gemm_skinny_enabled() && p.M > 0 && p.M <= 256 && p.M % 64 == 0 && p.N % 16 == 0 && p.K % 32 == 0 && p.K > 0;
}

template <int EPI>
int launch_skinny(const GemmParams& p, hipStream_t st) {
    {
        TtProfScope prof(TT_K_GEMM, st);
#if TT_DIAG   // TT_GEMM_SKINNY_PF=32 / TT_GEMM_SKINNY_MT=2: shape experiments of round 5 (every K step of a K = 1024 projection in flight; two row tiles per wave)
        static const int pf_env = TT_DIAG_ENV_INT("TT_GEMM_SKINNY_PF", 0);
        static const int mt_env = TT_DIAG_ENV_INT("TT_GEMM_SKINNY_MT", 0);
        if (pf_env == 32) hipLaunchKernelGGL((gemm_skinny_kernel<EPI, false, 1, 32>), dim3(p.N / 16, p.M / 16), dim3(64), 0, st, p);
        else if (mt_env == 2 && p.M % 32 == 0) hipLaunchKernelGGL((gemm_skinny_kernel<EPI, false, 2, 16>), dim3(p.N / 16, p.M / 32), dim3(64), 0, st, p);
        else
#endif
        if (skinny_mt4()) hipLaunchKernelGGL((gemm_skinny_kernel<EPI, false, 4, 8>), dim3(p.N / 16, p.M / 64), dim3(64), 0, st, p);
        // (the relay form is for the split-plane stream, 3 K / 32 steps: on the 16-bit stream one wave's 24 steps in flight already cover
        // K = 1024 in two round trips and the relay's barriers cost 5-10 % -- profiles/r06_relay_ab.log; TT_GEMM_RELAY=2 forces it here)
#if TT_DIAG
        else if (skinny_relay_16()) hipLaunchKernelGGL((gemm_relay_kernel<EPI, false>), dim3(p.N / 16, p.M / 16), dim3(64 * 8), 0, st, p);
#endif
        else hipLaunchKernelGGL((gemm_skinny_kernel<EPI, false>), dim3(p.N / 16, p.M / 16), dim3(64), 0, st, p);
    }
    TT_CHECK_LAUNCH();
    return TT_OK;
}

template <int EPI>
int launch(const GemmParams& p, hipStream_t st) {
    if constexpr (!kF16) if (p.fp8) return launch_fp8<EPI>(p, st);
    static const bool trace = TT_DIAG_ENV_INT("TT_GEMM_TRACE", 0) == 1;
    if (trace) fprintf(stderr, "gemm launch<%d> M=%d N=%d K=%d lda=%d ldc=%d ldr=%d\n", EPI, p.M, p.N, p.K, p.lda, p.ldc, p.ldr);
    static const int variant = TT_DIAG_ENV_INT("TT_GEMM_VARIANT", 5);
    // (the residual epilogue stages the residual tile as two pseudo K-tiles: needs an even number of K-tiles)
    // Fewer than half a CU-wave of 256x256 tiles (query embedding, the CLS tail): the 128x128 kernel spreads the
    // work over 4x the workgroups and is 1.3-1.7x faster there (M = 768: 15 vs 23 us per K = 1024 GEMM).
    const bool small_grid = EPI != TT_EPI_VT && (long long)(p.M / v3::BM3) * (p.N / v3::BN3) < 128 && small_grid_v1();
    if (variant >= 4 && !small_grid && p.M % v3::BM3 == 0 && p.N % v3::BN3 == 0 && p.ldc % 8 == 0 &&
        (EPI != TT_EPI_RESIDUAL || (p.ldr % 8 == 0 && (p.K / BK) % 2 == 0))) {
        const int mt_n = p.M / v3::BM3, nt_n = p.N / v3::BN3;
        const int SN = super_sn(nt_n), SM = 32 / SN;
        const int supers = ((mt_n + SM - 1) / SM) * ((nt_n + SN - 1) / SN);
        int blocks = supers * SM * SN;
        blocks = (blocks + 7) / 8 * 8;
        // measured (M = 236800): the persistent kernel wins where the epilogue is VALU-heavy (GELU: 1.74 vs 1.78 ms), the
        // one-tile kernel with LDS-transposed full-line stores where it is store-bound (bias: 1.39 vs 1.40 ms)
        static const int xp = TT_DIAG_ENV_INT("TT_GEMM_XP", 0);
        if constexpr (EPI == TT_EPI_GELU || EPI == TT_EPI_BIAS) {
            const int cus = tt_cu_count_cached() / 8 * 8;
            if (variant == 5 && (EPI == TT_EPI_GELU || (xp & 1)) && (p.K / BK) % 2 == 0 && p.K / BK >= 2 && blocks > cus && cus >= 8) {
                constexpr int kLdsP = v3::kLds3 + (TT_DIAG ? 2048 : 0);     // (diagnostic build: room for the LayerNorm-folding experiment's strips)
                TT_SET_MAX_LDS(v3::gemm_kernel_p<EPI>, kLdsP);
                {
                    TtProfScope prof(TT_K_GEMM, st);
                    GemmParams q = p;
                    q.sn = SN;
                    q.xp = xp;
                    hipLaunchKernelGGL(v3::gemm_kernel_p<EPI>, dim3(cus), dim3(v3::kThreads3), kLdsP, st, q, blocks);
                }
                TT_CHECK_LAUNCH();
                return TT_OK;
            }
        }
        auto kern = v3::gemm_kernel_v3<EPI, 4>;
#if TT_DIAG
        if constexpr (EPI == TT_EPI_BIAS) {   // diagnostic build of the 4-slot loop with s_memtime stamps (tools/gemm_stamps)
            static const int abl = TT_DIAG_ENV_INT("TT_GEMM_ABLATE", 0);
            if (abl == 8) {   // stamped persistent kernel (tools/gemm_stamps_p)
                const int cus8 = tt_cu_count_cached() / 8 * 8;
                TT_SET_MAX_LDS((v3::gemm_kernel_p<EPI, false, true>), v3::kLds3);
                GemmParams q8 = p;
                q8.sn = SN;
                q8.xp = xp;
                hipLaunchKernelGGL((v3::gemm_kernel_p<EPI, false, true>), dim3(cus8), dim3(v3::kThreads3), v3::kLds3, st, q8, blocks);
                TT_CHECK_LAUNCH();
                return TT_OK;
            }
            if (abl == 6) kern = v3::gemm_kernel_v3<EPI, 46>;
            if (abl == 7) kern = v3::gemm_kernel_v3<EPI, 47>;   // chunk-major store experiment (output layout differs!)
            if (abl == 6 || abl == 7) TT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, v3::kLds3));
        }
#endif
        TT_SET_MAX_LDS((v3::gemm_kernel_v3<EPI, 4>), v3::kLds3);
        GemmParams q = p;
        // whole-line stores as streaming stores: +1...3 % on the bias-only shapes, -0.7 ms per bench step (with the old 32-byte
        // runs the same hint cost 18 %: no write combining in L2)
        static const int nts = TT_DIAG_ENV_INT("TT_GEMM_NT_STORE", 1);
        q.nt_store = nts;
        q.sn = SN;
        static const bool a0 = TT_DIAG_ENV_INT("TT_GEMM_DEBUG_A0", 0) == 1;      // (wrong results: diagnostic library only)
        if (a0) q.xp |= 0x20000;
        static const bool head_major = TT_DIAG_ENV_INT("TT_GEMM_HEAD_MAJOR", 0) == 1;   // (another output layout: diagnostic library only)
        if (head_major) q.xp |= 0x40000;
        static const int stamp_block = TT_DIAG_ENV_INT("TT_GEMM_STAMP_BLOCK", 0);
        q.xp |= stamp_block << 20;
        static const int energy = TT_DIAG_ENV_INT("TT_GEMM_ENERGY", 0);                 // (wrong results: diagnostic library only)
        if (energy) q.xp |= 0x80000 | ((energy & 15) << 4);
        {
            TtProfScope prof(TT_K_GEMM, st);
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(v3::kThreads3), v3::kLds3, st, q);
        }
        TT_CHECK_LAUNCH();
        return TT_OK;
    }
    if constexpr (EPI == TT_EPI_VT) {
        tt_set_error("gemm: internal V^T epilogue needs the 256x256 kernel");
        return TT_E_UNSUPPORTED;
    }
    const int mt_n = p.M / BM, nt_n = p.N / BN;
    const int SN = nt_n < 8 ? nt_n : 8, SM = 8;
    const int supers = ((mt_n + SM - 1) / SM) * ((nt_n + SN - 1) / SN);
    int blocks = supers * SM * SN;
    blocks = (blocks + 7) / 8 * 8;
    // ONE round of 128x128 tiles (a lone caller's 10-pair rerank): the staged kernel (four-stage ring, LDS-DMA issues between the MFMAs)
    // -- M = 3072: attention output 17 -> 15 us, FFN-down 49 -> 42 us; in two rounds the two-stage kernel's two workgroups per CU win
    static const int staged16 = TT_DIAG_ENV_INT("TT_GEMM_STAGED16", 1);
    if (staged16 && staged_enabled() && (long long)mt_n * nt_n <= (staged16 > 1 ? staged16 : tt_cu_count_cached()) && p.K % 64 == 0) {
        TT_SET_MAX_LDS((gemm_staged_kernel<EPI, false>), kStagedLds);
        {
            TtProfScope prof(TT_K_GEMM, st);
            hipLaunchKernelGGL((gemm_staged_kernel<EPI, false>), dim3(blocks), dim3(kStagedThreads), kStagedLds, st, p);
        }
        TT_CHECK_LAUNCH();
        return TT_OK;
    }
    auto kern = gemm_kernel<EPI>;
    TT_SET_MAX_LDS(kern, kGemmLds);
    {
        TtProfScope prof(TT_K_GEMM, st);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(kGemmThreads), kGemmLds, st, p);
    }
    TT_CHECK_LAUNCH();
    return TT_OK;
}

}  // namespace

bool tt_gemm_skinny_enabled() {
    static const bool on = TT_DIAG_ENV_INT("TT_GEMM_SKINNY", 1) != 0;
    return on;
}

int tt_gemm_launch(const GemmParams& p, int epilogue, hipStream_t st) {
    if (p.M <= 0 || p.N <= 0) return TT_OK;
    if constexpr (kF16) {      // the fp16 instantiation: the 16-bit path, split-fp16 planes (x3) and f16c; e4m3-only GEMMs are bf16-side
        if (p.fp8) { tt_set_error("gemm (fp16 build): fp8 operands belong to the bf16 instantiation"); return TT_E_UNSUPPORTED; }
        if (p.xc) {
            switch (epilogue) {
                case TT_EPI_BIAS: return launch_xc<TT_EPI_BIAS>(p, st);
                case TT_EPI_GELU: return launch_xc<TT_EPI_GELU>(p, st);
                case TT_EPI_RESIDUAL: return launch_xc<TT_EPI_RESIDUAL>(p, st);
                case TT_EPI_VT: return launch_xc<TT_EPI_VT>(p, st);
                default: tt_set_error("gemm f16c: epilogue %d has no f16c form", epilogue); return TT_E_UNSUPPORTED;
            }
        }
    } else if (p.xc) {
        tt_set_error("gemm (bf16 build): f16c operands belong to the fp16 instantiation");
        return TT_E_UNSUPPORTED;
    }
    if (p.x3) {
        switch (epilogue) {
            case TT_EPI_BIAS: return launch_x3<TT_EPI_BIAS>(p, st);
            case TT_EPI_GELU: return launch_x3<TT_EPI_GELU>(p, st);
            case TT_EPI_RESIDUAL: return launch_x3<TT_EPI_RESIDUAL>(p, st);
            case TT_EPI_VT: return launch_x3<TT_EPI_VT>(p, st);
            default: tt_set_error("gemm x3: epilogue %d has no split-bf16 form", epilogue); return TT_E_UNSUPPORTED;
        }
    }
    if constexpr (!kF16) if (p.fp8 && epilogue == TT_EPI_QKV) {
        // fp8 QKV projection: Q,K columns as a bias GEMM, V columns as un-swapped tiles stored transposed
        if (!p.vt || p.vt_col0 % v3::BN3 || (p.N - p.vt_col0) % v3::BN3 || p.ldvt % 8) {
            tt_set_error("gemm fp8: bad qkv split");
            return TT_E_INVALID;
        }
        GemmParams a = p;
        a.N = p.vt_col0;
        if (int rc = launch<TT_EPI_BIAS>(a, st)) return rc;
        GemmParams b = p;
        b.W = reinterpret_cast<const uint16_t*>(reinterpret_cast<const char*>(p.W) + (size_t)p.vt_col0 * p.K);
        b.bias = p.bias + p.vt_col0;
        b.w_scale = p.w_scale + p.vt_col0;
        b.N = p.N - p.vt_col0;
        b.vt_col0 = 0;
        return launch<TT_EPI_VT>(b, st);
    }
    if ((p.lda % 8) || (p.ldc % 4) || !p.A || !p.W || !p.C || !p.bias) {
        tt_set_error("gemm: bad leading dimension / null pointer");
        return TT_E_INVALID;
    }
    if (skinny_shape(p)) {
        switch (epilogue) {
            case TT_EPI_BIAS: return launch_skinny<TT_EPI_BIAS>(p, st);
            case TT_EPI_GELU: return launch_skinny<TT_EPI_GELU>(p, st);
            case TT_EPI_RESIDUAL:
                if (!p.residual || p.ldr % 4) { tt_set_error("gemm: residual epilogue without residual"); return TT_E_INVALID; }
                return launch_skinny<TT_EPI_RESIDUAL>(p, st);
            case TT_EPI_TANH: return launch_skinny<TT_EPI_TANH>(p, st);
            case TT_EPI_QKV:
                if (!p.vt || p.vt_col0 % 16) { tt_set_error("gemm: qkv epilogue without vt"); return TT_E_INVALID; }
                return launch_skinny<TT_EPI_QKV>(p, st);
            default: tt_set_error("gemm: unknown epilogue %d", epilogue); return TT_E_INVALID;
        }
    }
    if (p.M % BM || p.N % BN || p.K % BK || p.K <= 0) {
        tt_set_error("gemm: M=%d N=%d K=%d must be multiples of %d/%d/%d (or M a multiple of 64 up to 256)", p.M, p.N, p.K, BM, BN, BK);
        return TT_E_UNSUPPORTED;
    }
    switch (epilogue) {
        case TT_EPI_BIAS: return launch<TT_EPI_BIAS>(p, st);
        case TT_EPI_GELU: return launch<TT_EPI_GELU>(p, st);
        case TT_EPI_RESIDUAL:
            if (!p.residual) { tt_set_error("gemm: residual epilogue without residual"); return TT_E_INVALID; }
            return launch<TT_EPI_RESIDUAL>(p, st);
        case TT_EPI_TANH: return launch<TT_EPI_TANH>(p, st);
        case TT_EPI_QKV: {
            if (!p.vt) { tt_set_error("gemm: qkv epilogue without vt"); return TT_E_INVALID; }
            static const int variant = TT_DIAG_ENV_INT("TT_GEMM_VARIANT", 5);
            const int nv = p.N - p.vt_col0;
            // (TT_GEMM_QKV_SPLIT=0: one launch with the mixed epilogue -- measured 2 % slower end to end)
            static const int split = TT_DIAG_ENV_INT("TT_GEMM_QKV_SPLIT", 1);
            const bool small_grid = (long long)(p.M / v3::BM3) * (p.N / v3::BN3) < 128 && small_grid_v1();
            if (variant >= 3 && split && !small_grid && p.M % v3::BM3 == 0 && p.vt_col0 % v3::BN3 == 0 && nv % v3::BN3 == 0 &&
                nv > 0 && p.ldc % 8 == 0 && p.ldvt % 8 == 0) {
                // Q,K columns: plain bias GEMM; V columns: un-swapped tiles stored transposed (two launches,
                // same number of tile rounds as one)
                GemmParams a = p;
                a.N = p.vt_col0;
                if (int rc = launch<TT_EPI_BIAS>(a, st)) return rc;
                GemmParams b = p;
                b.W = p.W + (size_t)p.vt_col0 * p.K;
                b.bias = p.bias + p.vt_col0;
                b.N = nv;
                b.vt_col0 = 0;
                return launch<TT_EPI_VT>(b, st);
            }
            return launch<TT_EPI_QKV>(p, st);
        }
        default: tt_set_error("gemm: unknown epilogue %d", epilogue); return TT_E_INVALID;
    }
}

#ifndef TT_SCAN_PERSIST_DEFAULT
#define TT_SCAN_PERSIST_DEFAULT 1        // persistent + wave-private lists: -3.6 % per batch (profiles/r03_scan_wave_private_ab.log)
#endif
int tt_scan_gemm_sample_launch(const uint16_t* corpus, int tiles, int tile_stride, int dim, const uint16_t* queries256, const float* thr256,
                               float* dense, int dense_stride, hipStream_t st) {
    if (tiles <= 0) return TT_OK;
    if constexpr (kF16) { tt_set_error("scan gemm: the corpus is bf16 (bf16 instantiation only)"); return TT_E_UNSUPPORTED; }
    else
    if (dim % 128 || dim <= 0 || tile_stride < 1 || !dense || dense_stride < tiles * 8 || (int64_t)tiles * tile_stride * v3::BM3 > INT32_MAX) {
        tt_set_error("scan gemm sample: tiles=%d stride=%d dim=%d dense_stride=%d", tiles, tile_stride, dim, dense_stride);
        return TT_E_UNSUPPORTED;
    }
    GemmParams p{};
    p.A = corpus;
    p.W = queries256;
    p.bias = thr256;            // (staged like the filter pass's thresholds, not read)
    p.M = tiles * v3::BM3;
    p.N = v3::BN3;
    p.K = dim;
    p.lda = dim;
    p.ldc = 8;
    p.scan_dense = dense;
    p.scan_dense_stride = dense_stride;
    p.scan_tile_stride = tile_stride;
    p.sn = 1;
    const int blocks = (tiles + 31) / 32 * 32;
    constexpr int kLdsOne = v3::kLdsScan > v3::kLdsScanW ? v3::kLdsScan : v3::kLdsScanW;
    TT_SET_MAX_LDS((v3::gemm_kernel_v3<TT_EPI_SCAN, 4>), kLdsOne);
    {
        TtProfScope prof(TT_K_SCAN_SAMPLE, st);
        hipLaunchKernelGGL((v3::gemm_kernel_v3<TT_EPI_SCAN, 4>), dim3(blocks), dim3(v3::kThreads3), kLdsOne, st, p);
    }
    TT_CHECK_LAUNCH();
    return TT_OK;
}

int tt_scan_gemm_launch(const uint16_t* corpus, int64_t rows, int dim, const uint16_t* queries256, const float* thr256,
                        int32_t* cnt, float* cand_scores, int32_t* cand_idx, int cap, int32_t idx_base, hipStream_t st) {
    if (rows <= 0) return TT_OK;
    if constexpr (kF16) { tt_set_error("scan gemm: the corpus is bf16 (bf16 instantiation only)"); return TT_E_UNSUPPORTED; }
    else
    if (rows < v3::BM3 || dim % 128 || dim <= 0 || rows / v3::BM3 > (1 << 24)) {
        tt_set_error("scan gemm: rows=%lld must be at least 256, dim=%d a multiple of 128", (long long)rows, dim);
        return TT_E_UNSUPPORTED;
    }
    GemmParams p{};
    p.A = corpus;
    p.W = queries256;
    p.bias = thr256;
    p.M = (int)((rows + v3::BM3 - 1) / v3::BM3 * v3::BM3);
    p.scan_rows = (int)rows;
    p.N = v3::BN3;
    p.K = dim;
    p.lda = dim;
    p.ldc = 8;
    p.scan_cnt = cnt;
    p.scan_scores = cand_scores;
    p.scan_idx = cand_idx;
    p.scan_cap = cap;
    p.scan_idx_base = idx_base;
    p.sn = 1;
    const int mt_n = p.M / v3::BM3;
    int blocks = (mt_n + 31) / 32 * 32;     // super-tiles of 32 row tiles x 1 column tile
    // persistent form (one workgroup per CU walks the row tiles, the K stream never drains): no per-tile prologue and no
    // dispatch gap, and -- unlike the storing epilogues -- nothing of the filter epilogue sits in the vector-memory queue
    // (measured on 10M x 1024, 256 queries: 5.7 ms persistent vs 5.2 ms one tile per workgroup -- the storing epilogues'
    // problem in reverse: here the tile ends with a workgroup-wide survivor hand-off, which the two wave groups of the
    // persistent form cannot share)
    // Round 3: survivors go to WAVE-PRIVATE LDS lists (scan_filter_epilogue<2>: no workgroup barrier), which is what the
    // persistent form needs.  TT_SCAN_GEMM_PERSIST: 1 = persistent + wave-private lists, 2 = one tile per workgroup +
    // wave-private lists, 0 = one tile per workgroup + the workgroup-shared list (round 2's form), default = see below.
    static const int persist = TT_DIAG_ENV_INT("TT_SCAN_GEMM_PERSIST", TT_SCAN_PERSIST_DEFAULT);
    const int cus = tt_cu_count_cached() / 8 * 8;
    if (persist == 1 && blocks > cus && cus >= 8) {
        static const int sxp = TT_DIAG_ENV_INT("TT_SCAN_GEMM_XP", 0);
        p.xp = sxp;      // (experiment switches of the persistent kernel: bit 2 = wave groups NOT aligned at the tile boundary)
        TT_SET_MAX_LDS(v3::gemm_kernel_p<TT_EPI_SCAN>, v3::kLdsScanW);
        {
            TtProfScope prof(TT_K_SCAN_FILTER, st);
            hipLaunchKernelGGL(v3::gemm_kernel_p<TT_EPI_SCAN>, dim3(cus), dim3(v3::kThreads3), v3::kLdsScanW, st, p, blocks);
        }
        TT_CHECK_LAUNCH();
        return TT_OK;
    }
    if (persist != 0) p.xp |= 0x10000;      // wave-private lists in the one-tile kernel
    constexpr int kLdsOne = v3::kLdsScan > v3::kLdsScanW ? v3::kLdsScan : v3::kLdsScanW;
    TT_SET_MAX_LDS((v3::gemm_kernel_v3<TT_EPI_SCAN, 4>), kLdsOne);
    {
        TtProfScope prof(TT_K_SCAN_FILTER, st);
        hipLaunchKernelGGL((v3::gemm_kernel_v3<TT_EPI_SCAN, 4>), dim3(blocks), dim3(v3::kThreads3), kLdsOne, st, p);
    }
    TT_CHECK_LAUNCH();
    return TT_OK;
}
