// Exact similarity scan (Q x C^T) with fused candidate filtering, gfx950.
//
// Replaces the vector search behind VectorIndexRetriever.retrieve
// (reference: src/tensortruth/rag_engine.py:639 -> ChromaVectorStore.query).
//
// Data layout in HBM
//   corpus   [N][D] bf16 row-major (a shard of the corpus matrix), streamed once
//   queries  [Q][D] bf16 row-major (L2 resident, staged into LDS once per block)
//
// Kernel structure (one workgroup of 8 waves per CU, no barrier in the main loop)
//   * the block's 64 queries live in LDS for the whole kernel as a swizzled
//     [64][D] bf16 image (128 KiB at D=1024): B operand of v_mfma_f32_32x32x16_bf16
//     read with conflict-free ds_read_b128;
//   * every wave owns 32-row groups of the corpus (64 KiB each at D=1024) and
//     streams them through a register ring of 128-B row segments loaded with
//     full-line global_load_dwordx4 (8 rows x 128 B per wave instruction);
//     MODE 1 transposes each 32x128-B chunk through a 4-KiB wave-private LDS
//     scratch into the MFMA A-operand layout, MODE 0 loads A fragments directly
//     (32 rows x 32 B per instruction);
//   * the accumulator tile has the QUERY on the lane and 16 corpus rows in
//     registers, so filtering against the per-query threshold is a per-lane
//     compare; survivors are appended to a per-query candidate list in global
//     memory (rare: the threshold comes from an exact top-k over a sample of
//     the first n0 rows, see scan_api.cpp).
//   * OUT=1 writes every score instead (small shards), OUT=2 writes one max per
//     (32-row group, query): the sample phase that produces the thresholds.
//
// Roofline: HBM-bound; algorithmic bytes = rows * D * 2 per launch.
#include "common.h"
#include "scan.h"

namespace {

constexpr int kThreads = 512;
constexpr int kWaves = kThreads / TT_WAVE;
constexpr int kBM = 64;            // queries per block
constexpr int kNG = kBM / 32;      // 32-query groups per block
constexpr int kScratchPerWave = 4096;
constexpr int kPrivSlots = TT_SCAN_PRIV_SLOTS;  // private candidate slots per (wave, lane, query group)

template <int D>
struct Cfg {
    static constexpr int kRowBytes = D * 2;
    static constexpr int kQImageBytes = kBM * kRowBytes;
    static constexpr int NCH = D / 64;   // 128-B chunks per row
    static constexpr int NKS = D / 16;   // MFMA k-steps per row
    // register ring depth (in chunks) for MODE 1; must divide NCH
    static constexpr int P1 = (NCH % 4 == 0) ? 4 : ((NCH % 6 == 0) ? 6 : 2);
    // ring depth in k-steps for MODE 0; must divide NKS
    static constexpr int P0 = (NKS % 16 == 0) ? 16 : 8;
};

__device__ __forceinline__ uint4 ldg16(const uint16_t* p) {
    return *reinterpret_cast<const uint4*>(p);
}

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
template <bool NT>
__device__ __forceinline__ uint4 ldg16c(const uint16_t* p) {
    if constexpr (NT) {
        const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
        return make_uint4(v[0], v[1], v[2], v[3]);
    } else {
        return *reinterpret_cast<const uint4*>(p);
    }
}

// ---- epilogue: dense store or threshold filter + candidate append ---------------
// Survivors of the threshold filter go to a list PRIVATE to this lane (one per wave,
// lane and query group: kPrivSlots entries in global memory, fill count in a register),
// so the hot loop issues no atomic: a returning global atomic sits in the in-order
// vmcnt queue and stalls the whole prefetch ring for its (long) latency -- measured
// ~2.4 us per event, +65 % kernel time at 3 survivors per row group.  Only a lane
// whose private list is full (heavily clustered hits) falls back to the shared
// per-query overflow list, which is allocated with an atomic.
template <int OUT>
__device__ __forceinline__ void scan_epilogue(const ScanParams& p, f32x16 (&acc)[2], const float (&thr)[2],
                                              const bool (&qvalid)[2], int q0, int ql, int half, int64_t grp,
                                              int (&pcnt)[2], uint2* const (&pbase)[2]) {
    const int64_t row0 = p.row_lo + grp * 32;         // first row of this group
    const bool tail = row0 + 32 > p.row_hi;           // wave-uniform
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int q = q0 + g * 32 + ql;
        if constexpr (OUT == 2) {
            // max over this lane's 16 rows, then over the other half's 16 rows
            float mx = acc[g][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, acc[g][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            if (qvalid[g] && half == 0) p.dense[(size_t)q * p.dense_stride + (size_t)grp] = mx;
        } else if constexpr (OUT == 1) {
            if (qvalid[g]) {
                float* dst = p.dense + (size_t)q * p.dense_stride + (size_t)(row0 - p.row_lo);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // registers 4j..4j+3 are rows 8j + 4*half + 0..3 (consecutive)
                    float4 v = make_float4(acc[g][4 * j], acc[g][4 * j + 1], acc[g][4 * j + 2], acc[g][4 * j + 3]);
                    *reinterpret_cast<float4*>(dst + 8 * j + 4 * half) = v;
                }
            }
        } else {
            const float t = thr[g];
            float mx = acc[g][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, acc[g][r]);
            if (__any(mx >= t)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t row = row0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    bool ok = acc[g][r] >= t;
                    if (tail) ok = ok && (row < p.row_hi);
                    if (ok) {
                        const uint2 e = make_uint2(__float_as_uint(acc[g][r]), (uint32_t)(p.idx_base + (int32_t)row));
                        if (pcnt[g] < kPrivSlots) {
                            pbase[g][pcnt[g]] = e;
                        } else {
                            const int pos = atomicAdd(p.cnt + q, 1);
                            if (pos < p.cap) {
                                p.cand_scores[(size_t)q * p.cap + pos] = acc[g][r];
                                p.cand_idx[(size_t)q * p.cap + pos] = (int32_t)e.y;
                            }
                        }
                        ++pcnt[g];
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
    }
}

// VAR bits (ablation / tuning knobs, D=1024 filter kernels only): 1 = default-policy corpus
// loads instead of the non-temporal ones every shipped variant uses (+11 % kernel time), 2 = loads only (no LDS, no MFMA), 4 = loads + LDS transpose writes only,
// 8 = everything but the MFMAs, 16 = MFMAs without the query-image reads.
template <int D, int MODE, int OUT, int VAR>
__global__ __launch_bounds__(kThreads) void scan_kernel(ScanParams p) {
    using C = Cfg<D>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* q_img = smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    // wave id is wave-uniform; make that provable
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* scratch = smem + C::kQImageBytes + wave * kScratchPerWave;

    const int q0 = blockIdx.y * kBM;

    // ---- stage the query tile into LDS (swizzled 16-B slots) ----------------
    {
        constexpr int kPiecesPerRow = D / 8;
        constexpr int kPieces = kBM * kPiecesPerRow;
        for (int t = tid; t < kPieces; t += kThreads) {
            const int q = t / kPiecesPerRow;
            const int c = t % kPiecesPerRow;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (q0 + q < p.n_queries) v = ldg16(p.queries + (size_t)(q0 + q) * D + c * 8);
            *reinterpret_cast<uint4*>(q_img + q * C::kRowBytes + ((c ^ (q & 15)) << 4)) = v;
        }
    }
    __syncthreads();

    // ---- per-lane constants --------------------------------------------------
    const int ql = lane & 31;     // query column inside a 32-query group
    const int half = lane >> 5;   // which 8-element half of the MFMA k-step
    float thr[kNG];
    bool qvalid[kNG];
#pragma unroll
    for (int g = 0; g < kNG; ++g) {
        const int q = q0 + g * 32 + ql;
        qvalid[g] = q < p.n_queries;
        thr[g] = __builtin_inff();
        if (OUT == 0 && qvalid[g]) thr[g] = p.thr[q];
    }
    // byte offset of this lane's B fragment slot base in the Q image
    int qoff[kNG];
#pragma unroll
    for (int g = 0; g < kNG; ++g) qoff[g] = (g * 32 + ql) * C::kRowBytes;
    const int qswz = ql & 15;

    // ---- row-group partition: block owns [g_lo, g_hi), waves interleave -------
    if constexpr (MODE == 0) {
        if (p.row_cnt) {          // gathered launch: the table's length lives on the device (no host round trip between the passes)
            const int64_t n = *p.row_cnt;
            p.row_hi = p.row_lo + (n < p.row_hi - p.row_lo ? n : p.row_hi - p.row_lo);
            if (p.row_hi <= p.row_lo) return;
        }
    }
    const int64_t n_groups = (p.row_hi - p.row_lo + 31) / 32;
    const int64_t per_blk = n_groups / gridDim.x;
    const int64_t rem = n_groups % gridDim.x;
    const int64_t b = blockIdx.x;
    const int64_t g_lo = b * per_blk + (b < rem ? b : rem);
    const int64_t g_hi = g_lo + per_blk + (b < rem ? 1 : 0);

    const uint16_t* corpus = p.corpus;
    const int64_t gstride = (OUT == 2 && p.group_stride > 1) ? p.group_stride : 1;   // strided threshold sample
    const int64_t last_row = (OUT == 2 && p.group_stride > 1) ? p.phys_rows - 1 : p.row_hi - 1;

    f32x16 acc[kNG];
#pragma unroll
    for (int g = 0; g < kNG; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;

    int64_t grp = g_lo + wave;
    // private candidate lists: sub-list id = (global wave, half), one per query
    const int n_sub = gridDim.x * kWaves * 2;
    const int sub = (blockIdx.x * kWaves + wave) * 2 + half;
    int pcnt[kNG];
    uint2* pbase[kNG];
#pragma unroll
    for (int g = 0; g < kNG; ++g) {
        pcnt[g] = 0;
        pbase[g] = nullptr;
        if (OUT == 0) pbase[g] = p.priv + ((size_t)(q0 + g * 32 + ql) * n_sub + sub) * kPrivSlots;
    }

    if constexpr (MODE == 1) {
        constexpr int P = C::P1;
        uint4 ring[P][4];
        const int lrow = lane >> 3;   // row inside an 8-row load
        const int lpiece = lane & 7;  // 16-B piece inside the 128-B chunk
        auto issue = [&](uint4(&dst)[4], int64_t g, int c) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int64_t row = p.row_lo + g * gstride * 32 + 8 * j + lrow;
                row = row < last_row ? row : last_row;
                dst[j] = ldg16c<(VAR & 1) == 0>(corpus + (size_t)row * D + c * 64 + lpiece * 8);
            }
        };
        // scratch write offsets (row = 8j + lrow): slot = piece ^ ((row>>1)&7)
        int woff[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = 8 * j + lrow;
            woff[j] = row * 128 + ((lpiece ^ ((row >> 1) & 7)) << 4);
        }
        // scratch read: lane (r = ql, half) reads piece 2*ks + half of row r
        const int rbase = ql * 128;
        const int rswz = (ql >> 1) & 7;

        if (grp < g_hi) {
#pragma unroll
            for (int c = 0; c < P; ++c) issue(ring[c], grp, c);
        }
        while (grp < g_hi) {
            const int64_t nxt = grp + kWaves;
            const bool has_next = nxt < g_hi;
#pragma unroll
            for (int c = 0; c < C::NCH; ++c) {
                const int s = c % P;
                if constexpr ((VAR & 2) == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        *reinterpret_cast<uint4*>(scratch + woff[j]) = ring[s][j];
                } else {
                    // ablation: keep the loads alive without touching LDS
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        asm volatile("" ::"v"(ring[s][j].x), "v"(ring[s][j].y), "v"(ring[s][j].z), "v"(ring[s][j].w));
                }
                // pin the refill of this ring slot right behind its drain: without the
                // barriers hipcc sinks the loads next to their use and the prefetch
                // depth collapses to zero
                __builtin_amdgcn_sched_barrier(0);
                if (c + P < C::NCH) {
                    issue(ring[s], grp, c + P);
                } else if (has_next) {
                    issue(ring[s], nxt, c + P - C::NCH);
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr ((VAR & 6) == 0)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int piece = 2 * ks + half;
                    const uint4 av = *reinterpret_cast<const uint4*>(scratch + rbase + ((piece ^ rswz) << 4));
                    const bf16x8 a = __builtin_bit_cast(bf16x8, av);
                    const int kc = c * 8 + piece;  // 16-B slot index inside the query row
#pragma unroll
                    for (int g = 0; g < kNG; ++g) {
                        if constexpr ((VAR & 16) != 0) {  // ablation: no Q-image reads
                            acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc[g], 0, 0, 0);
                        } else {
                            const uint4 bv = *reinterpret_cast<const uint4*>(q_img + qoff[g] + ((kc ^ qswz) << 4));
                            if constexpr ((VAR & 8) != 0) {  // ablation: LDS reads, no MFMA
                                asm volatile("" ::"v"(bv.x), "v"(bv.y), "v"(bv.z), "v"(bv.w), "v"(av.x), "v"(av.w));
                            } else {
                                acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, bv), acc[g], 0, 0, 0);
                            }
                        }
                    }
                }
            }
            scan_epilogue<OUT>(p, acc, thr, qvalid, q0, ql, half, grp, pcnt, pbase);
            grp = nxt;
        }
    } else {
        constexpr int P = C::P0;
        uint4 ring[P];
        // physical row of this lane in group g: contiguous, or looked up in the gather table (entries beyond the table: its last row)
        auto phys = [&](int64_t g) {
            int64_t row = p.row_lo + g * gstride * 32 + ql;
            const int64_t last = p.row_table ? p.row_hi - 1 : last_row;
            row = row < last ? row : last;
            return p.row_table ? (int64_t)p.row_table[row] : row;
        };
        int64_t row_cur = grp < g_hi ? phys(grp) : 0, row_nxt = 0;
        auto issue = [&](uint4& dst, int64_t row, int ks) {
            dst = ldg16c<(VAR & 1) == 0>(corpus + (size_t)row * D + ks * 16 + half * 8);
        };
        if (grp < g_hi) {
#pragma unroll
            for (int k = 0; k < P; ++k) issue(ring[k], row_cur, k);
        }
        while (grp < g_hi) {
            const int64_t nxt = grp + kWaves;
            const bool has_next = nxt < g_hi;
            if (has_next) row_nxt = phys(nxt);
#pragma unroll
            for (int ks = 0; ks < C::NKS; ++ks) {
                const int s = ks % P;
                const bf16x8 a = __builtin_bit_cast(bf16x8, ring[s]);
                const int kc = 2 * ks + half;
                bf16x8 bfr[kNG];
#pragma unroll
                for (int g = 0; g < kNG; ++g) {
                    const uint4 bv = *reinterpret_cast<const uint4*>(q_img + qoff[g] + ((kc ^ qswz) << 4));
                    bfr[g] = __builtin_bit_cast(bf16x8, bv);
                }
#pragma unroll
                for (int g = 0; g < kNG; ++g)
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bfr[g], acc[g], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (ks + P < C::NKS) {
                    issue(ring[s], row_cur, ks + P);
                } else if (has_next) {
                    issue(ring[s], row_nxt, ks + P - C::NKS);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            scan_epilogue<OUT>(p, acc, thr, qvalid, q0, ql, half, grp, pcnt, pbase);
            grp = nxt;
            row_cur = row_nxt;
        }
    }
    if constexpr (OUT == 0) {
#pragma unroll
        for (int g = 0; g < kNG; ++g)
            p.priv_cnt[(size_t)(q0 + g * 32 + ql) * n_sub + sub] = pcnt[g] < kPrivSlots ? pcnt[g] : kPrivSlots;
    }
}

}  // namespace

// ---- host-side launcher ---------------------------------------------------------
template <int D, int MODE, int OUT, int VAR>
static int launch_one(const ScanParams& p, int blocks, int q_tiles, hipStream_t stream) {
    using C = Cfg<D>;
    const size_t lds = (size_t)C::kQImageBytes + (MODE == 1 ? (size_t)kWaves * kScratchPerWave : 0);
    auto kern = scan_kernel<D, MODE, OUT, VAR>;
    TT_SET_MAX_LDS(kern, lds);   // per instantiation, per thread, per device
    {
        TtProfScope prof(p.prof_id ? p.prof_id : (OUT == 0 ? TT_K_SCAN_FILTER : TT_K_SCAN_SAMPLE), stream);
        hipLaunchKernelGGL(kern, dim3(blocks, q_tiles), dim3(kThreads), lds, stream, p);
    }
    TT_CHECK_LAUNCH();
    return TT_OK;
}

template <int D>
static int launch_d(const ScanParams& p, int mode, int out, int blocks, int q_tiles, hipStream_t stream) {
    const int load_mode = mode & 15;
    const int var = mode >> 4;
    if constexpr (D == 1024) {
        // tuning / ablation variants exist for the headline shape only
        if (out == 0 && load_mode == 1) {
            switch (var) {
                case 1: return launch_one<D, 1, 0, 1>(p, blocks, q_tiles, stream);       // plain instead of non-temporal loads (same results)
#if TT_DIAG     // ablation variants (loads only / + LDS writes / MFMAs only: WRONG results by design): the diagnostic library only
                case 2: return launch_one<D, 1, 0, 2>(p, blocks, q_tiles, stream);
                case 4: return launch_one<D, 1, 0, 4>(p, blocks, q_tiles, stream);
                case 8: return launch_one<D, 1, 0, 8>(p, blocks, q_tiles, stream);
#endif
                default: break;
            }
        }
    }
    if (load_mode == 1) {
        switch (out) {
            case 0: return launch_one<D, 1, 0, 0>(p, blocks, q_tiles, stream);
            case 1: return launch_one<D, 1, 1, 0>(p, blocks, q_tiles, stream);
            default: return launch_one<D, 1, 2, 0>(p, blocks, q_tiles, stream);
        }
    }
    switch (out) {
        case 0: return launch_one<D, 0, 0, 0>(p, blocks, q_tiles, stream);
        case 1: return launch_one<D, 0, 1, 0>(p, blocks, q_tiles, stream);
        default: return launch_one<D, 0, 2, 0>(p, blocks, q_tiles, stream);
    }
}

int tt_scan_launch(const ScanParams& p, int dim, int mode, int out, int blocks, hipStream_t stream) {
    const int q_tiles = (p.n_queries + kBM - 1) / kBM;
    if (p.row_hi <= p.row_lo || q_tiles == 0) return TT_OK;
    const int64_t n_groups = (p.row_hi - p.row_lo + 31) / 32;
    // dense / group-max launches are short: spread the row groups over every CU;
    // the long filter pass keeps 8 waves per CU busy
    const int64_t max_blocks = out != 0 ? n_groups : (n_groups + kWaves - 1) / kWaves;
    if (blocks > max_blocks) blocks = (int)max_blocks;
    if (blocks < 1) blocks = 1;
    switch (dim) {
        case 128: return launch_d<128>(p, mode, out, blocks, q_tiles, stream);
        case 256: return launch_d<256>(p, mode, out, blocks, q_tiles, stream);
        case 384: return launch_d<384>(p, mode, out, blocks, q_tiles, stream);
        case 512: return launch_d<512>(p, mode, out, blocks, q_tiles, stream);
        case 640: return launch_d<640>(p, mode, out, blocks, q_tiles, stream);
        case 768: return launch_d<768>(p, mode, out, blocks, q_tiles, stream);
        case 896: return launch_d<896>(p, mode, out, blocks, q_tiles, stream);
        case 1024: return launch_d<1024>(p, mode, out, blocks, q_tiles, stream);
        default:
            tt_set_error("tt_scan: dim %d not in the compiled set {128,256,...,1024: multiples of 128}", dim);
            return TT_E_UNSUPPORTED;
    }
}
