// bf16 GEMM with fused epilogues for the encoder layers, gfx950 MFMA.
//
//   C[M,N] = epi(A[M,K] . W[N,K]^T + bias[N])      A, W, C bf16 row-major, fp32 accumulate
//
// W is the HF nn.Linear weight as stored ([out, in]), so both operands are K-contiguous and
// every MFMA fragment is one 16-byte read.  These are the dense contractions of the
// reference's bi-encoder / cross-encoder forward (transformers XLMRobertaLayer / BertLayer:
// QKV, attention output, FFN up, FFN down).
//
// Structure (v1): 128x128x64 block tile, 4 waves (2x2), 64x64 per wave as 4x4 tiles of
// v_mfma_f32_16x16x32_bf16; operands staged by global_load_lds_dwordx4 into a 2-deep LDS
// ring ([row][64] bf16 = 128-B rows, 16-B slots XOR-swizzled on the SOURCE address so the
// lane-linear LDS image is read conflict-free by ds_read_b128), one barrier per K-step.
// The MFMA is issued "swapped" (a = W fragment, b = A fragment) so each lane ends up with
// 4 consecutive N-columns of one row: the epilogue stores 8 bytes per lane.
// Block order: XCD-contiguous, 8x8 super-tiles, so an XCD's L2 sees each A/W panel 8 times.
//
// Roofline: MFMA-bound; 2*M*N*K flops per launch.
#include "common.h"
#include "encoder.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int kGemmThreads = 256;
constexpr int kTileBytes = BM * BK * 2;          // 16 KiB per operand tile
constexpr int kStageBytes = 2 * kTileBytes;      // A + W
constexpr int kGemmLds = 2 * kStageBytes;        // 64 KiB

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

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
            bf16x8 wf[4], xf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rw = wn * 64 + i * 16 + frow;
                wf[i] = *reinterpret_cast<const bf16x8*>(tW + rw * 128 + (((4 * ss + fchk) ^ ((rw >> 1) & 7)) << 4));
                const int ra = wm * 64 + i * 16 + frow;
                xf[i] = *reinterpret_cast<const bf16x8*>(tA + ra * 128 + (((4 * ss + fchk) ^ ((ra >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): next tile landed
        __syncthreads();
    }

    // ---- epilogue: lane holds C[m][n..n+3], m = tile row (l&15), n = 4*(l>>4) -----------
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + wn * 64 + i * 16 + (lane >> 4) * 4;
        const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + wm * 64 + j * 16 + (lane & 15);
            float v0 = acc[i][j][0] + b4.x, v1 = acc[i][j][1] + b4.y, v2 = acc[i][j][2] + b4.z, v3 = acc[i][j][3] + b4.w;
            if constexpr (EPI == TT_EPI_GELU) {
                v0 = gelu_erf(v0); v1 = gelu_erf(v1); v2 = gelu_erf(v2); v3 = gelu_erf(v3);
            } else if constexpr (EPI == TT_EPI_TANH) {
                v0 = tanhf(v0); v1 = tanhf(v1); v2 = tanhf(v2); v3 = tanhf(v3);
            } else if constexpr (EPI == TT_EPI_RESIDUAL) {
                const uint2 r = *reinterpret_cast<const uint2*>(p.residual + (size_t)m * p.ldr + n);
                v0 += __uint_as_float(r.x << 16);
                v1 += __uint_as_float(r.x & 0xFFFF0000u);
                v2 += __uint_as_float(r.y << 16);
                v3 += __uint_as_float(r.y & 0xFFFF0000u);
            }
            if constexpr (EPI == TT_EPI_QKV) {
                if (n >= p.vt_col0) {
                    // V third: store transposed, VT[n - vt_col0][m]
                    uint16_t* vt = p.vt + (size_t)(n - p.vt_col0) * p.ldvt + m;
                    vt[0] = f32_to_bf16_bits(v0);
                    vt[p.ldvt] = f32_to_bf16_bits(v1);
                    vt[2 * (size_t)p.ldvt] = f32_to_bf16_bits(v2);
                    vt[3 * (size_t)p.ldvt] = f32_to_bf16_bits(v3);
                    continue;
                }
            }
            uint2 o;
            o.x = pack_bf16x2(v0, v1);
            o.y = pack_bf16x2(v2, v3);
            *reinterpret_cast<uint2*>(p.C + (size_t)m * p.ldc + n) = o;
        }
    }
}

template <int EPI>
int launch(const GemmParams& p, hipStream_t st) {
    const int mt_n = p.M / BM, nt_n = p.N / BN;
    const int SN = nt_n < 8 ? nt_n : 8, SM = 8;
    const int supers = ((mt_n + SM - 1) / SM) * ((nt_n + SN - 1) / SN);
    int blocks = supers * SM * SN;
    blocks = (blocks + 7) / 8 * 8;
    auto kern = gemm_kernel<EPI>;
    static thread_local bool attr_set = false;
    if (!attr_set) {
        TT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, kGemmLds));
        attr_set = true;
    }
    {
        TtProfScope prof(TT_K_GEMM, st);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(kGemmThreads), kGemmLds, st, p);
    }
    TT_CHECK_LAUNCH();
    return TT_OK;
}

}  // namespace

int tt_gemm_launch(const GemmParams& p, int epilogue, hipStream_t st) {
    if (p.M <= 0 || p.N <= 0) return TT_OK;
    if (p.M % BM || p.N % BN || p.K % BK || p.K <= 0) {
        tt_set_error("gemm: M=%d N=%d K=%d must be multiples of %d/%d/%d", p.M, p.N, p.K, BM, BN, BK);
        return TT_E_UNSUPPORTED;
    }
    if ((p.lda % 8) || (p.ldc % 4) || !p.A || !p.W || !p.C || !p.bias) {
        tt_set_error("gemm: bad leading dimension / null pointer");
        return TT_E_INVALID;
    }
    switch (epilogue) {
        case TT_EPI_BIAS: return launch<TT_EPI_BIAS>(p, st);
        case TT_EPI_GELU: return launch<TT_EPI_GELU>(p, st);
        case TT_EPI_RESIDUAL:
            if (!p.residual) { tt_set_error("gemm: residual epilogue without residual"); return TT_E_INVALID; }
            return launch<TT_EPI_RESIDUAL>(p, st);
        case TT_EPI_TANH: return launch<TT_EPI_TANH>(p, st);
        case TT_EPI_QKV:
            if (!p.vt) { tt_set_error("gemm: qkv epilogue without vt"); return TT_E_INVALID; }
            return launch<TT_EPI_QKV>(p, st);
        default: tt_set_error("gemm: unknown epilogue %d", epilogue); return TT_E_INVALID;
    }
}
