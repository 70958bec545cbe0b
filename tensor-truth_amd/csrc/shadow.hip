// fp8 shadow of the corpus as an EXACT prefilter for a lone caller's scan (round 6).
//
// One query over a 10 M x 1024 bf16 corpus is a 20.5 GB read: 2.97 ms at 6.9 TB/s, a fifth of an un-batched query (the reference's
// own usage: README.md:13, rag_engine.py:420-424).  Half the bytes carry enough to rule almost every row out:
//     shadow[n][d] = e4m3(256 c[n][d])          one byte per element, made once when rows are added (tt_scan_shadow_build)
//     be[n] = || c[n] - shadow[n] / 256 ||_2     what the rounding took, exactly, per row        (rounded UP)
//     dn[n] = || shadow[n] / 256 ||_2
// For a query q (bf16, as the bf16 pass reads it):
//     | q.c - s8 |  <=  ||q|| be            (Cauchy-Schwarz on c = d + e),     s8 = q . d,   d = shadow[n] / 256
// so with thr = the k-th best EXACT score of a sample of rows (the bf16 pass's own threshold: k distinct rows score >= thr), every
// true top-k row satisfies s8 + bound >= thr.  Pass 1 streams the shadow: every e4m3 byte is converted to bf16 EXACTLY (e4m3's 4
// significant bits fit bf16's 8: v_cvt_pk_f32_fp8 + one v_perm per pair, ~1 VALU operation per element, a fifth of the chip's VALU rate
// at HBM speed) and contracted with the bf16 query on v_mfma_f32_16x16x32_bf16 -- exact products, fp32 accumulation, the arithmetic
// model of the bf16 pass -- and the rows whose upper bound reaches thr are listed (~1 % of random rows at 10 M).  (The block-scaled
// e4m3 MFMA was tried first and dropped: v_mfma_scale_f32_16x16x128_f8f6f4 does not accumulate its 128 products at fp32 precision --
// tools/probes/shadow_mfma_probe.cpp: 4e-3 relative on wide-range operands -- so no rigorous bound could be stated for it.)
// Pass 2 re-scores THOSE rows from the
// bf16 matrix with the streaming kernel's own fragments and MFMA order (scan.hip, gather table) -- the same score bits as a full
// bf16 scan --; the exact selection (select.hip: score desc, row asc) is unchanged.  Results are identical to tt_scan_topk's; a
// survivor list that overflows raises the status flag like any other candidate overflow (the caller re-runs the bf16 path).
// NaN rows (tombstones) have NaN bounds and never pass, as in the bf16 filter.
//
// Roofline: HBM-bound; algorithmic bytes = rows * D (shadow) + 8 rows (bounds) per launch.
#include "common.h"
#include "scan.h"
#include "shadow.h"

namespace {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr float kShadowScale = 256.0f;
constexpr int kSWaves = 8, kSThreads = 64 * kSWaves;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float bf16_to_f32(uint16_t b) { return __uint_as_float((uint32_t)b << 16); }
// e4m3 byte of x (saturating at +-448; NaN stays NaN) and its value back in fp32
__device__ __forceinline__ uint32_t to_e4m3(float x) {
    const float c = __builtin_amdgcn_fmed3f(x, -448.0f, 448.0f);     // (NaN: fmed3 returns NaN -> the conversion's NaN code)
    return (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(c, c, 0, false) & 0xFFu;
}
__device__ __forceinline__ float from_e4m3(uint32_t byte) { return __builtin_amdgcn_cvt_f32_fp8((int)byte, 0); }

// ---- build: one wave per row ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void shadow_build_kernel(const uint16_t* corpus, int64_t n_rows, int dim, uint8_t* shadow, float* be, float* dn) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n_rows) return;
    float se = 0.f, sd = 0.f;
    for (int c = lane * 8; c < dim; c += 512) {           // 8 consecutive elements per lane: one 16-byte load, one 8-byte store
        const uint4 v = *reinterpret_cast<const uint4*>(corpus + (size_t)row * dim + c);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t out[2] = {0u, 0u};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float x = bf16_to_f32((uint16_t)(i & 1 ? w[i >> 1] >> 16 : w[i >> 1] & 0xFFFFu));
            const uint32_t b = to_e4m3(x * kShadowScale);
            const float d = from_e4m3(b) * (1.0f / kShadowScale);
            const float e = x - d;
            se = fmaf(e, e, se);
            sd = fmaf(d, d, sd);
            out[i >> 2] |= b << (8 * (i & 3));
        }
        *reinterpret_cast<uint2*>(shadow + (size_t)row * dim + c) = make_uint2(out[0], out[1]);
    }
    se = wave_sum(se);
    sd = wave_sum(sd);
    if (lane == 0) {
        // rounded UP: fp32 sums of <= 1024 squares are within 1024 * 2^-24 relative; sqrt and the products below within 2^-22
        be[row] = sqrtf(se) * 1.0005f + 1e-9f;
        dn[row] = sqrtf(sd) * 1.0005f + 1e-9f;
    }
}

// ---- query fragments: q (bf16) in the order the filter kernel reads it, + ||q|| --------------------------------------------------------
// frags[((kt * 4 + f) * 64 + lane) * 8 + e], lane = g4 * 16 + query: element kt 128 + g4 32 + f 8 + e of that query (zero beyond the batch)
__global__ __launch_bounds__(64) void shadow_query_kernel(const uint16_t* queries, int n_queries, int dim, uint16_t* frags, float* qinfo) {
    const int q = blockIdx.x, lane = threadIdx.x;            // 16 blocks
    float sq = 0.f;
    for (int c = lane; c < dim; c += 64) {
        uint16_t v = 0;
        if (q < n_queries) v = queries[(size_t)q * dim + c];
        const float x = bf16_to_f32(v);
        sq = fmaf(x, x, sq);
        const int kt = c >> 7, g4 = (c & 127) >> 5, f = (c & 31) >> 3, e = c & 7;
        frags[((size_t)(kt * 4 + f) * 64 + (g4 * 16 + q)) * 8 + e] = v;
    }
    sq = wave_sum(sq);
    if (lane == 0) qinfo[q] = sqrtf(sq) * 1.0005f + 1e-9f;       // ||q||, rounded up
}

// ---- pass 1: stream the shadow, list the rows whose upper bound reaches the query's threshold -----------------------------------------
struct ShadowParams {
    const uint8_t* shadow;     // [N][D] e4m3
    const float* be;           // [round_up(N, 16)]
    const float* dn;
    const uint16_t* frags;     // shadow_query_kernel's output: (D / 128) * 4 * 64 fragments of 16 bytes
    const float* qinfo;        // [16]: ||q||
    const float* thr;          // [>= n_queries] exact thresholds of the sample pass
    int64_t n_rows;
    int n_queries;             // <= 16
    int32_t* list;             // [n_queries][n_waves][capw] survivors of every wave
    int32_t* wave_cnt;         // [n_queries][n_waves]
    int capw;
    int32_t* status_flag;
};

// two e4m3 bytes (word `hi` of w) -> two bf16 in one dword, exactly: fp32 via v_cvt_pk_f32_fp8, then the high halves (e4m3 needs 4 significant bits)
template <bool HI>
__device__ __forceinline__ uint32_t e4m3x2_to_bf16x2(uint32_t w) {
    const auto f = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, HI);
    return __builtin_amdgcn_perm(__float_as_uint(f[1]), __float_as_uint(f[0]), 0x07060302u);
}

// Loads: one wave-instruction = 8 rows x 128 B (lane = row lane >> 3, 16-byte piece lane & 7): whole 128-byte lines, as the bf16 pass's
// full-line mode; a K-tile's two instructions (rows 0-7, 8-15 of the 16-row group) go through a 2-KiB wave-private LDS scratch (16-byte
// slots XOR-swizzled by (row >> 1) & 7, conflict-free both ways) into the operand layout: lane (row j, slice g4) reads the 32 bytes
// [32 g4, 32 g4 + 32) of its row.  (First version: every lane loaded its own 32 bytes straight from global -- two instructions per K-tile
// each touching HALF of sixteen 128-byte lines: 5.5 TB/s; this form: see profiles/r06_shadow_scan.log.)
template <int D>
__global__ __launch_bounds__(kSThreads, 2) void shadow_filter_kernel(ShadowParams p) {
    constexpr int NKT = D / 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint4* qf = reinterpret_cast<uint4*>(smem);                              // NKT x 4 fragments x 64 lanes x 16 B
    int* lcnt = reinterpret_cast<int*>(smem + NKT * 4 * 1024);               // [kSWaves][16]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* scratch = smem + NKT * 4 * 1024 + kSWaves * 16 * sizeof(int) + wave * 2048;      // this wave's transpose scratch
    for (int i = tid; i < NKT * 4 * 64; i += kSThreads) qf[i] = reinterpret_cast<const uint4*>(p.frags)[i];
    if (tid < kSWaves * 16) lcnt[tid] = 0;
    __syncthreads();

    const int j = lane & 15, g4 = lane >> 4;             // operands: row / query j, 32-element K slice g4; results: query j, rows 4 g4 + r
    const bool qok = j < p.n_queries;
    const float thr = qok ? p.thr[j] : __builtin_inff();
    const float qn = p.qinfo[j];
    const int n_waves = gridDim.x * kSWaves;
    const int wg = blockIdx.x * kSWaves + wave;
    const int64_t n_groups = (p.n_rows + 15) / 16;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const int lrow = lane >> 3, lpiece = lane & 7;
    // scratch offsets: write (row lrow / 8 + lrow, piece) and read (row j, pieces 2 g4, 2 g4 + 1), slot = piece ^ ((row >> 1) & 7)
    const int woff0 = lrow * 128 + ((lpiece ^ ((lrow >> 1) & 7)) << 4);
    const int woff1 = (8 + lrow) * 128 + ((lpiece ^ (((8 + lrow) >> 1) & 7)) << 4);
    const int roff0 = j * 128 + (((2 * g4) ^ ((j >> 1) & 7)) << 4);
    const int roff1 = j * 128 + (((2 * g4 + 1) ^ ((j >> 1) & 7)) << 4);

    // a[kt][h]: rows 8 h + lrow of the group, bytes [128 kt + 16 lpiece, + 16)
    auto load_group = [&](u32x4 (&a)[NKT][2], int64_t grp) {
        int64_t r0 = grp * 16 + lrow, r1 = r0 + 8;
        r0 = r0 < p.n_rows ? r0 : p.n_rows - 1;
        r1 = r1 < p.n_rows ? r1 : p.n_rows - 1;
        const u32x4* s0 = reinterpret_cast<const u32x4*>(p.shadow + (size_t)r0 * D + lpiece * 16);
        const u32x4* s1 = reinterpret_cast<const u32x4*>(p.shadow + (size_t)r1 * D + lpiece * 16);
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            a[kt][0] = __builtin_nontemporal_load(s0 + kt * 8);
            a[kt][1] = __builtin_nontemporal_load(s1 + kt * 8);
        }
    };
    auto process = [&](const u32x4 (&a)[NKT][2], int64_t grp) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            *reinterpret_cast<u32x4*>(scratch + woff0) = a[kt][0];
            *reinterpret_cast<u32x4*>(scratch + woff1) = a[kt][1];
            const u32x4 x0 = *reinterpret_cast<const u32x4*>(scratch + roff0);      // (LDS operations of one wave execute in order)
            const u32x4 x1 = *reinterpret_cast<const u32x4*>(scratch + roff1);
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const uint32_t w0 = (f < 2 ? x0 : x1)[(f & 1) * 2], w1 = (f < 2 ? x0 : x1)[(f & 1) * 2 + 1];      // elements 8 f .. 8 f + 7 of the slice
                const uint4 av = make_uint4(e4m3x2_to_bf16x2<false>(w0), e4m3x2_to_bf16x2<true>(w0), e4m3x2_to_bf16x2<false>(w1),
                                            e4m3x2_to_bf16x2<true>(w1));
                const uint4 bv = qf[(kt * 4 + f) * 64 + lane];
                // a = corpus rows, b = queries: lane l holds D[4 (l >> 4) + r][l & 15] = rows 4 g4 + r of the group for query j
                acc = TT_MFMA_16x16x32(__builtin_bit_cast(ex8, av), __builtin_bit_cast(ex8, bv), acc);
            }
        }
        const int64_t r0 = grp * 16 + g4 * 4;
        const float4 be4 = *reinterpret_cast<const float4*>(p.be + r0);
        const float bes[4] = {be4.x, be4.y, be4.z, be4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            // s8 = q . (256 d) / 256; + 1e-4: the fp32 accumulation of <= 1024 exact products (<= 1024 * 2^-24 * sum |terms| <= 6e-5)
            const float ub = acc[r] * (1.0f / kShadowScale) + qn * bes[r] + 1e-4f;
            if (qok && ub >= thr && r0 + r < p.n_rows) {       // (NaN rows: NaN bound, the compare fails)
                const int pos = atomicAdd(&lcnt[wave * 16 + j], 1);
                if (pos < p.capw) p.list[((size_t)j * n_waves + wg) * p.capw + pos] = (int32_t)(r0 + r);
            }
        }
    };

    u32x4 a0[NKT][2], a1[NKT][2];
    int64_t grp = wg;
    if (grp < n_groups) load_group(a0, grp);
    while (grp < n_groups) {
        const int64_t g1 = grp + n_waves, g2 = g1 + n_waves;
        if (g1 < n_groups) load_group(a1, g1);
        process(a0, grp);
        if (g1 >= n_groups) break;
        if (g2 < n_groups) load_group(a0, g2);
        process(a1, g1);
        grp = g2;
    }
    // this wave's counts (one lane per query); a list that did not fit raises the flag: the caller re-runs the bf16 path
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (lane < p.n_queries) {
        const int c = reinterpret_cast<volatile int*>(lcnt)[wave * 16 + lane];
        p.wave_cnt[(size_t)lane * n_waves + wg] = c < p.capw ? c : p.capw;
        if (c > p.capw && p.status_flag) atomicOr(p.status_flag, 2);      // (bit 1: a wave's survivor list; bit 2: a query's table)
    }
}

// ---- the waves' lists -> one contiguous table per query (block = query) ----------------------------------------------------------------
__global__ __launch_bounds__(1024) void shadow_compact_kernel(const int32_t* list, const int32_t* wave_cnt, int n_waves, int capw, int32_t* table,
                                                              int cap, int32_t* table_cnt, int32_t* status_flag) {
    __shared__ int wsum[16];
    __shared__ int total;
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int32_t* cnts = wave_cnt + (size_t)q * n_waves;
    const int per = (n_waves + 1023) / 1024;                 // consecutive waves per thread
    int mine = 0;
    for (int i = 0; i < per; ++i) {
        const int w = tid * per + i;
        if (w < n_waves) mine += cnts[w];
    }
    int inc = mine;                                          // inclusive scan inside the wave, then over the 16 wave totals
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(inc, o, 64);
        if (lane >= o) inc += v;
    }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    int base = 0;
    for (int i = 0; i < wv; ++i) base += wsum[i];
    if (tid == 1023) total = base + inc;
    int off = base + inc - mine;
    for (int i = 0; i < per; ++i) {
        const int w = tid * per + i;
        if (w >= n_waves) break;
        const int c = cnts[w];
        const int32_t* src = list + ((size_t)q * n_waves + w) * capw;
        for (int e = 0; e < c; ++e)
            if (off + e < cap) table[(size_t)q * cap + off + e] = src[e];
        off += c;
    }
    __syncthreads();
    if (tid == 0) {
        table_cnt[q] = total < cap ? total : cap;
        if (total > cap && status_flag) atomicOr(status_flag, 4);
    }
}

}  // namespace

int tt_shadow_build_launch(const uint16_t* corpus, int64_t n_rows, int dim, uint8_t* shadow, float* be, float* dn, hipStream_t st) {
    if (n_rows <= 0) return TT_OK;
    hipLaunchKernelGGL(shadow_build_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, st, corpus, n_rows, dim, shadow, be, dn);
    TT_CHECK_LAUNCH();
    return TT_OK;
}

int tt_shadow_query_launch(const uint16_t* queries, int n_queries, int dim, uint16_t* frags, float* qinfo, hipStream_t st) {
    hipLaunchKernelGGL(shadow_query_kernel, dim3(16), dim3(64), 0, st, queries, n_queries, dim, frags, qinfo);
    TT_CHECK_LAUNCH();
    return TT_OK;
}

template <int D>
static int launch_filter(const ShadowParams& p, int blocks, hipStream_t st) {
    constexpr size_t lds = (D / 128) * 4 * 1024 + kSWaves * 16 * sizeof(int) + kSWaves * 2048;
    TT_SET_MAX_LDS(shadow_filter_kernel<D>, lds);
    {
        TtProfScope prof(TT_K_SCAN_FILTER, st);
        hipLaunchKernelGGL(shadow_filter_kernel<D>, dim3(blocks), dim3(kSThreads), lds, st, p);
    }
    TT_CHECK_LAUNCH();
    return TT_OK;
}

int tt_shadow_filter_launch(const uint8_t* shadow, const float* be, const float* dn, int64_t n_rows, int dim, const uint16_t* frags,
                            const float* qinfo, const float* thr, int n_queries, int blocks, int32_t* list, int32_t* wave_cnt, int capw,
                            int32_t* status_flag, hipStream_t st) {
    ShadowParams p{};
    p.shadow = shadow; p.be = be; p.dn = dn; p.frags = frags; p.qinfo = qinfo; p.thr = thr; p.n_rows = n_rows; p.n_queries = n_queries;
    p.list = list; p.wave_cnt = wave_cnt; p.capw = capw; p.status_flag = status_flag;
    switch (dim) {
        case 128: return launch_filter<128>(p, blocks, st);
        case 256: return launch_filter<256>(p, blocks, st);
        case 384: return launch_filter<384>(p, blocks, st);
        case 512: return launch_filter<512>(p, blocks, st);
        case 640: return launch_filter<640>(p, blocks, st);
        case 768: return launch_filter<768>(p, blocks, st);
        case 896: return launch_filter<896>(p, blocks, st);
        case 1024: return launch_filter<1024>(p, blocks, st);
        default:
            tt_set_error("shadow scan: dim %d not in the compiled set {128, 256, ..., 1024}", dim);
            return TT_E_UNSUPPORTED;
    }
}

int tt_shadow_compact_launch(const int32_t* list, const int32_t* wave_cnt, int n_waves, int capw, int n_queries, int32_t* table, int cap,
                             int32_t* table_cnt, int32_t* status_flag, hipStream_t st) {
    TtProfScope prof(TT_K_SELECT, st);
    hipLaunchKernelGGL(shadow_compact_kernel, dim3(n_queries), dim3(1024), 0, st, list, wave_cnt, n_waves, capw, table, cap, table_cnt, status_flag);
    TT_CHECK_LAUNCH();
    return TT_OK;
}
