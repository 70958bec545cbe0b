// Row-wise (HBM-bound) encoder kernels: embedding gather + LayerNorm, LayerNorm, CLS pooling +
// L2 normalisation, row gather, classifier output projection + sigmoid.  One 64-lane wave
// per row, 16-byte bf16x8 accesses, fp32 statistics (mean, then centred variance -- the
// same two-pass formula as the oracle / torch.nn.LayerNorm).
//
// Reference arithmetic: XLM-R / BERT embeddings
// LN(word[id] + pos[p] + type[t]); post-LN residual blocks; sentence-transformers
// Pooling(cls) + Normalize; XLMRobertaClassificationHead out_proj + CrossEncoder sigmoid.
#include "common.h"
#include "encoder.h"

namespace {

constexpr int kRowThreads = 256;  // 4 rows per block
constexpr int kMaxChunks = 2;     // H <= 1024: at most 2 x (8 bf16) per lane

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ void unpack8(const uint4& u, float (&f)[8]) {
    f[0] = elo(u.x); f[1] = ehi(u.x);
    f[2] = elo(u.y); f[3] = ehi(u.y);
    f[4] = elo(u.z); f[5] = ehi(u.z);
    f[6] = elo(u.w); f[7] = ehi(u.w);
}

__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
    uint4 u;
    u.x = pack_e2(f[0], f[1]); u.y = pack_e2(f[2], f[3]);
    u.z = pack_e2(f[4], f[5]); u.w = pack_e2(f[6], f[7]);
    return u;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// 8 floats (already multiplied by the row's 448 / absmax) -> 8 OCP e4m3 bytes, round to nearest even
__device__ __forceinline__ uint2 pack_fp8x8(const float (&f)[8]) {
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
    return make_uint2((unsigned)lo, (unsigned)hi);
}

// normalise x[kMaxChunks][8] (this lane's share of a row of H values) and store bf16; optionally also the
// per-row fp8 quantisation of the bf16-rounded result (q8_row, *q8_scale = absmax / 448): the A operand of the
// fp8 GEMMs, produced while the row is in registers
// a * b + c with the product ROUNDED before the add, whatever -ffp-contract says: the sums of squares and the head's dot product round
// like the CPU oracle's (x * x).sum() -- and like this file's earlier builds, in which the SLP vectoriser happened to keep these products
// apart from their adds (the file is now built without it: Makefile).  The empty asm makes the product opaque to the contraction pass.
__device__ __forceinline__ float mul_then_add(float a, float b, float c) {
    float p = a * b;
    asm("" : "+v"(p));
    return p + c;
}

template <bool NTS = false>
__device__ __forceinline__ void ln_finish(float (&x)[kMaxChunks][8], int nchunk_total, int lane, int H, float eps,
                                          const float* gamma, const float* beta, uint16_t* out_row,
                                          uint8_t* q8_row = nullptr, float* q8_scale = nullptr) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < kMaxChunks; ++c)
        if (lane + 64 * c < nchunk_total)
#pragma unroll
            for (int i = 0; i < 8; ++i) s += x[c][i];
    const float mean = wave_sum(s) / (float)H;
    float v = 0.f;
#pragma unroll
    for (int c = 0; c < kMaxChunks; ++c)
        if (lane + 64 * c < nchunk_total)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float d = x[c][i] - mean;
                v = mul_then_add(d, d, v);
            }
    const float rstd = rsqrtf(wave_sum(v) / (float)H + eps);
#pragma unroll
    for (int c = 0; c < kMaxChunks; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nchunk_total) {
            const float4 g0 = *reinterpret_cast<const float4*>(gamma + ch * 8);
            const float4 g1 = *reinterpret_cast<const float4*>(gamma + ch * 8 + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(beta + ch * 8);
            const float4 b1 = *reinterpret_cast<const float4*>(beta + ch * 8 + 4);
            const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            const float b[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            float y[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) y[i] = (x[c][i] - mean) * rstd * g[i] + b[i];
            const uint4 packed = pack8(y);
            if constexpr (NTS) {
                typedef unsigned int u4v __attribute__((ext_vector_type(4)));
                __builtin_nontemporal_store(u4v{packed.x, packed.y, packed.z, packed.w}, reinterpret_cast<u4v*>(out_row + ch * 8));
            } else {
                *reinterpret_cast<uint4*>(out_row + ch * 8) = packed;
            }
            if (q8_row) unpack8(packed, x[c]);   // keep the bf16-rounded values for the quantisation pass
        }
    }
    if (q8_row) {
        float amax = 0.f;
#pragma unroll
        for (int c = 0; c < kMaxChunks; ++c)
            if (lane + 64 * c < nchunk_total)
#pragma unroll
                for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(x[c][i]));
        amax = wave_max(amax);
        const float inv = amax > 0.f ? __fdiv_rn(448.0f, amax) : 0.f   /* correctly rounded: the oracle divides in IEEE fp32 */;
        if (lane == 0) *q8_scale = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
#pragma unroll
        for (int c = 0; c < kMaxChunks; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nchunk_total) {
                float q[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) q[i] = x[c][i] * inv;
                *reinterpret_cast<uint2*>(q8_row + ch * 8) = pack_fp8x8(q);
            }
        }
    }
}

// NT bit 0: streaming (non-temporal) loads of the input, which is read exactly once (+7.5 %); bit 1: two rows per
// wave, both rows' loads issued before either is reduced (measured slower); bit 2: streaming stores (+6 %).
template <int NT>
__global__ __launch_bounds__(kRowThreads) void layernorm_kernel(const uint16_t* in, uint16_t* out, const float* gamma,
                                                               const float* beta, int rows, int H, float eps, uint8_t* q8,
                                                               float* q8_scale) {
    constexpr int RPW = (NT & 2) ? 2 : 1;
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * (kRowThreads / 64) + (threadIdx.x >> 6)) * RPW;
    if (row0 >= rows) return;
    const int nch = H / 8;
    float x[RPW][kMaxChunks][8];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = row0 + r < rows ? row0 + r : rows - 1;
#pragma unroll
        for (int c = 0; c < kMaxChunks; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nch) {
                uint4 u;
                if constexpr (NT & 1) {
                    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
                    const u4v t = __builtin_nontemporal_load(reinterpret_cast<const u4v*>(in + (size_t)row * H + ch * 8));
                    u = make_uint4(t.x, t.y, t.z, t.w);
                } else {
                    u = *reinterpret_cast<const uint4*>(in + (size_t)row * H + ch * 8);
                }
                unpack8(u, x[r][c]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = row0 + r;
        if (row < rows)
            ln_finish<(NT & 4) != 0>(x[r], nch, lane, H, eps, gamma, beta, out + (size_t)row * H,
                                     q8 ? q8 + (size_t)row * H : nullptr, q8 ? q8_scale + row : nullptr);
    }
}

__global__ __launch_bounds__(kRowThreads) void embed_ln_kernel(EmbedParams p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (kRowThreads / 64) + (threadIdx.x >> 6);
    if (row >= p.T) return;
    int id = p.ids[row], ps = p.pos[row], ty = p.type ? p.type[row] : 0;
    // out-of-range ids would read outside the tables: clamp (the host validates too)
    id = id < 0 ? 0 : (id >= p.vocab ? p.vocab - 1 : id);
    ps = ps < 0 ? 0 : (ps >= p.max_pos ? p.max_pos - 1 : ps);
    ty = ty < 0 ? 0 : (ty >= p.type_vocab ? p.type_vocab - 1 : ty);
    const int nch = p.H / 8;
    float x[kMaxChunks][8];
#pragma unroll
    for (int c = 0; c < kMaxChunks; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nch) {
            float a[8], b[8], t[8];
            unpack8(*reinterpret_cast<const uint4*>(p.word + (size_t)id * p.H + ch * 8), a);
            unpack8(*reinterpret_cast<const uint4*>(p.posemb + (size_t)ps * p.H + ch * 8), b);
            unpack8(*reinterpret_cast<const uint4*>(p.typeemb + (size_t)ty * p.H + ch * 8), t);
#pragma unroll
            for (int i = 0; i < 8; ++i) x[c][i] = a[i] + b[i] + t[i];
        }
    }
    ln_finish(x, nch, lane, p.H, p.eps, p.gamma, p.beta, p.out + (size_t)row * p.H, p.q8 ? p.q8 + (size_t)row * p.H : nullptr,
              p.q8 ? p.q8_scale + row : nullptr);
}

__global__ __launch_bounds__(kRowThreads) void cls_pool_kernel(const uint16_t* hidden, int ld, const int32_t* rows, int n,
                                                              int H, float* out_f32, uint16_t* out_bf16) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * (kRowThreads / 64) + (threadIdx.x >> 6);
    if (b >= n) return;
    const int nch = H / 8;
    const uint16_t* src = hidden + (size_t)rows[b] * ld;
    float x[kMaxChunks][8];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < kMaxChunks; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nch) {
            unpack8(*reinterpret_cast<const uint4*>(src + ch * 8), x[c]);
#pragma unroll
            for (int i = 0; i < 8; ++i) ss = mul_then_add(x[c][i], x[c][i], ss);
        }
    }
    const float nrm = fmaxf(sqrtf(wave_sum(ss)), 1e-12f);
    const float inv = 1.0f / nrm;
#pragma unroll
    for (int c = 0; c < kMaxChunks; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nch) {
            float y[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) y[i] = x[c][i] * inv;
            float* o = out_f32 + (size_t)b * H + ch * 8;
            *reinterpret_cast<float4*>(o) = make_float4(y[0], y[1], y[2], y[3]);
            *reinterpret_cast<float4*>(o + 4) = make_float4(y[4], y[5], y[6], y[7]);
            if (out_bf16) *reinterpret_cast<uint4*>(out_bf16 + (size_t)b * H + ch * 8) = pack8(y);
        }
    }
}

// sentence-transformers Pooling(mean) + Normalize: the mean of a sequence's token rows (padding never exists here: sequences are
// packed), L2-normalised.  One wave per sequence walks its rows (a 256-token chunk is 512 KiB), fp32 sums in the token order.
template <bool F32>
__global__ __launch_bounds__(kRowThreads) void mean_pool_kernel(const void* hidden, int ld, const int32_t* seq_start,
                                                               const int32_t* seq_len, int n, int H, float* out_f32,
                                                               uint16_t* out_bf16) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * (kRowThreads / 64) + (threadIdx.x >> 6);
    if (b >= n) return;
    const int nch = H / 8;
    const int t0 = seq_start[b], len = seq_len[b];
    float acc[kMaxChunks][8];
#pragma unroll
    for (int c = 0; c < kMaxChunks; ++c)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[c][i] = 0.f;
    for (int t = 0; t < len; ++t) {
#pragma unroll
        for (int c = 0; c < kMaxChunks; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nch) {
                float x[8];
                if constexpr (F32) {
                    const float* src = static_cast<const float*>(hidden) + (size_t)(t0 + t) * ld + ch * 8;
                    const float4 a = *reinterpret_cast<const float4*>(src), bq = *reinterpret_cast<const float4*>(src + 4);
                    x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = bq.x; x[5] = bq.y; x[6] = bq.z; x[7] = bq.w;
                } else {
                    unpack8(*reinterpret_cast<const uint4*>(static_cast<const uint16_t*>(hidden) + (size_t)(t0 + t) * ld + ch * 8), x);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[c][i] += x[i];
            }
        }
    }
    const float inv_len = 1.0f / (float)(len > 0 ? len : 1);
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < kMaxChunks; ++c)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc[c][i] *= inv_len;
            if (lane + 64 * c < nch) ss = mul_then_add(acc[c][i], acc[c][i], ss);
        }
    const float inv = 1.0f / fmaxf(sqrtf(wave_sum(ss)), 1e-12f);
#pragma unroll
    for (int c = 0; c < kMaxChunks; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nch) {
            float y[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) y[i] = acc[c][i] * inv;
            float* o = out_f32 + (size_t)b * H + ch * 8;
            *reinterpret_cast<float4*>(o) = make_float4(y[0], y[1], y[2], y[3]);
            *reinterpret_cast<float4*>(o + 4) = make_float4(y[4], y[5], y[6], y[7]);
            if (out_bf16) *reinterpret_cast<uint4*>(out_bf16 + (size_t)b * H + ch * 8) = pack8(y);
        }
    }
}

__global__ __launch_bounds__(kRowThreads) void gather_rows_kernel(const uint16_t* src, int ld, const int32_t* rows, int n,
                                                                 int n_pad, int H, uint16_t* dst) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * (kRowThreads / 64) + (threadIdx.x >> 6);
    if (b >= n_pad) return;
    const int nch = H / 8;
    for (int ch = lane; ch < nch; ch += 64) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (b < n) v = *reinterpret_cast<const uint4*>(src + (size_t)rows[b] * ld + ch * 8);
        *reinterpret_cast<uint4*>(dst + (size_t)b * H + ch * 8) = v;
    }
}

__global__ __launch_bounds__(kRowThreads) void head_out_kernel(const uint16_t* t, int ld, const uint16_t* w,
                                                              const float* bias, int n, int H, float* scores,
                                                              float* logits) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * (kRowThreads / 64) + (threadIdx.x >> 6);
    if (b >= n) return;
    const int nch = H / 8;
    float acc = 0.f;
    for (int ch = lane; ch < nch; ch += 64) {
        float a[8], ww[8];
        unpack8(*reinterpret_cast<const uint4*>(t + (size_t)b * ld + ch * 8), a);
        unpack8(*reinterpret_cast<const uint4*>(w + ch * 8), ww);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc = mul_then_add(a[i], ww[i], acc);
    }
    acc = wave_sum(acc) + bias[0];
    if (lane == 0) {
        if (logits) logits[b] = acc;
        scores[b] = 1.0f / (1.0f + expf(-acc));
    }
}

// dist[i] = 1 - cos(e[i], e[i+1]) for consecutive rows of an fp32 [n][H] matrix (semantic splitter)
__global__ __launch_bounds__(kRowThreads) void adjacent_cosine_kernel(const float* e, int n, int H, float* dist) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * (kRowThreads / 64) + (threadIdx.x >> 6);
    if (i >= n - 1) return;
    const float* a = e + (size_t)i * H;
    const float* b = a + H;
    float ab = 0.f, aa = 0.f, bb = 0.f;
    for (int c = lane * 4; c < H; c += 256) {
        const float4 x = *reinterpret_cast<const float4*>(a + c);
        const float4 y = *reinterpret_cast<const float4*>(b + c);
        ab += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
        aa += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
        bb += y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w;
    }
    ab = wave_sum(ab); aa = wave_sum(aa); bb = wave_sum(bb);
    if (lane == 0) dist[i] = 1.0f - ab / fmaxf(sqrtf(aa) * sqrtf(bb), 1e-30f);
}

int check_h(int H) {
    if (H <= 0 || H % 8 || H > 64 * 8 * kMaxChunks) {
        tt_set_error("row op: hidden size %d unsupported (multiple of 8, <= %d)", H, 64 * 8 * kMaxChunks);
        return TT_E_UNSUPPORTED;
    }
    return TT_OK;
}

inline dim3 row_grid(int rows) { return dim3((rows + kRowThreads / 64 - 1) / (kRowThreads / 64)); }

}  // namespace

int tt_layernorm_launch(const uint16_t* in, uint16_t* out, const float* gamma, const float* beta, int rows, int H,
                        float eps, hipStream_t st, uint8_t* q8, float* q8_scale) {
    if (rows <= 0) return TT_OK;
    if (int rc = check_h(H)) return rc;
    // default 5 = streaming loads + streaming stores: 0.197 -> 0.172 ms for 236800 x 1024 (4.9 -> 5.6 TB/s) with the
    // caches cold as they are between two GEMMs, -0.8 ms per bench step; two rows per wave (3) measured slower
    static const int nt = TT_DIAG_ENV_INT("TT_LN_NT", 5);
    if (nt == 3)
        hipLaunchKernelGGL(layernorm_kernel<3>, row_grid((rows + 1) / 2), dim3(kRowThreads), 0, st, in, out, gamma, beta, rows, H,
                           eps, q8, q8_scale);
    else if (nt == 5)
        hipLaunchKernelGGL(layernorm_kernel<5>, row_grid(rows), dim3(kRowThreads), 0, st, in, out, gamma, beta, rows, H, eps, q8,
                           q8_scale);
    else if (nt == 1)
        hipLaunchKernelGGL(layernorm_kernel<1>, row_grid(rows), dim3(kRowThreads), 0, st, in, out, gamma, beta, rows, H, eps, q8,
                           q8_scale);
    else
        hipLaunchKernelGGL(layernorm_kernel<0>, row_grid(rows), dim3(kRowThreads), 0, st, in, out, gamma, beta, rows, H, eps, q8,
                           q8_scale);
    TT_CHECK_LAUNCH();
    return TT_OK;
}

// per-row fp8 quantisation of a bf16 matrix (cols a multiple of 8): q = e4m3(x * 448 / absmax_row), scale = absmax / 448
namespace {
__global__ __launch_bounds__(kRowThreads) void quantize_rows_kernel(const uint16_t* in, int ld, int rows, int cols, uint8_t* q8,
                                                                   float* scale) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (kRowThreads / 64) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const uint16_t* src = in + (size_t)row * ld;
    float amax = 0.f;
    for (int ch = lane; ch * 8 < cols; ch += 64) {
        float f[8];
        unpack8(*reinterpret_cast<const uint4*>(src + ch * 8), f);
#pragma unroll
        for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(f[i]));
    }
    amax = wave_max(amax);
    const float inv = amax > 0.f ? __fdiv_rn(448.0f, amax) : 0.f   /* correctly rounded: the oracle divides in IEEE fp32 */;
    if (lane == 0) scale[row] = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
    for (int ch = lane; ch * 8 < cols; ch += 64) {
        float f[8];
        unpack8(*reinterpret_cast<const uint4*>(src + ch * 8), f);
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] *= inv;
        *reinterpret_cast<uint2*>(q8 + (size_t)row * cols + ch * 8) = pack_fp8x8(f);
    }
}
}  // namespace

namespace {
__global__ __launch_bounds__(256) void absmax_kernel(const uint16_t* x, size_t n8, float* out) {
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        float f[8];
        unpack8(*reinterpret_cast<const uint4*>(x + i * 8), f);
#pragma unroll
        for (int k = 0; k < 8; ++k) m = fmaxf(m, fabsf(f[k]));
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(reinterpret_cast<unsigned int*>(out), __float_as_uint(m));   // m >= 0: bit order = value order
}
}  // namespace

// *out = max(*out, max |x|) over n bf16 values (n a multiple of 8); NaNs are ignored
int tt_absmax_launch(const uint16_t* x, size_t n, float* out, hipStream_t st) {
    if (n == 0) return TT_OK;
    hipLaunchKernelGGL(absmax_kernel, dim3(2048), dim3(256), 0, st, x, n / 8, out);
    TT_CHECK_LAUNCH();
    return TT_OK;
}

int tt_quantize_rows_launch(const uint16_t* in, int ld, int rows, int cols, uint8_t* q8, float* scale, hipStream_t st) {
    if (rows <= 0) return TT_OK;
    if (cols <= 0 || cols % 8 || ld % 8) {
        tt_set_error("quantize_rows: cols=%d ld=%d must be multiples of 8", cols, ld);
        return TT_E_INVALID;
    }
    hipLaunchKernelGGL(quantize_rows_kernel, row_grid(rows), dim3(kRowThreads), 0, st, in, ld, rows, cols, q8, scale);
    TT_CHECK_LAUNCH();
    return TT_OK;
}

int tt_embed_ln_launch(const EmbedParams& p, hipStream_t st) {
    if (p.T <= 0) return TT_OK;
    if (int rc = check_h(p.H)) return rc;
    hipLaunchKernelGGL(embed_ln_kernel, row_grid(p.T), dim3(kRowThreads), 0, st, p);
    TT_CHECK_LAUNCH();
    return TT_OK;
}

int tt_cls_pool_l2norm_launch(const uint16_t* hidden, int ld, const int32_t* rows, int n, int H, float* out_f32,
                              uint16_t* out_bf16, hipStream_t st) {
    if (n <= 0) return TT_OK;
    if (int rc = check_h(H)) return rc;
    hipLaunchKernelGGL(cls_pool_kernel, row_grid(n), dim3(kRowThreads), 0, st, hidden, ld, rows, n, H, out_f32, out_bf16);
    TT_CHECK_LAUNCH();
    return TT_OK;
}

int tt_mean_pool_l2norm_launch(const void* hidden, int is_f32, int ld, const int32_t* seq_start, const int32_t* seq_len, int n, int H,
                               float* out_f32, uint16_t* out_bf16, hipStream_t st) {
    if (n <= 0) return TT_OK;
    if (int rc = check_h(H)) return rc;
    if (is_f32) hipLaunchKernelGGL(mean_pool_kernel<true>, row_grid(n), dim3(kRowThreads), 0, st, hidden, ld, seq_start, seq_len, n, H, out_f32, out_bf16);
    else hipLaunchKernelGGL(mean_pool_kernel<false>, row_grid(n), dim3(kRowThreads), 0, st, hidden, ld, seq_start, seq_len, n, H, out_f32, out_bf16);
    TT_CHECK_LAUNCH();
    return TT_OK;
}

int tt_gather_rows_launch(const uint16_t* src, int ld, const int32_t* rows, int n, int n_pad, int H, uint16_t* dst,
                          hipStream_t st) {
    if (n_pad <= 0) return TT_OK;
    if (int rc = check_h(H)) return rc;
    hipLaunchKernelGGL(gather_rows_kernel, row_grid(n_pad), dim3(kRowThreads), 0, st, src, ld, rows, n, n_pad, H, dst);
    TT_CHECK_LAUNCH();
    return TT_OK;
}

int tt_head_out_sigmoid_launch(const uint16_t* t, int ld, const uint16_t* w, const float* bias, int n, int H,
                               float* scores, float* logits, hipStream_t st) {
    if (n <= 0) return TT_OK;
    if (int rc = check_h(H)) return rc;
    hipLaunchKernelGGL(head_out_kernel, row_grid(n), dim3(kRowThreads), 0, st, t, ld, w, bias, n, H, scores, logits);
    TT_CHECK_LAUNCH();
    return TT_OK;
}

int tt_adjacent_cosine_launch(const float* e, int n, int H, float* dist, hipStream_t st) {
    if (n <= 1) return TT_OK;
    if (H <= 0 || H % 4) { tt_set_error("adjacent cosine: hidden %d must be a multiple of 4", H); return TT_E_UNSUPPORTED; }
    hipLaunchKernelGGL(adjacent_cosine_kernel, row_grid(n - 1), dim3(kRowThreads), 0, st, e, n, H, dist);
    TT_CHECK_LAUNCH();
    return TT_OK;
}
