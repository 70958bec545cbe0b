// Exact top-k selection over a per-query candidate list, gfx950.
//
// Two kernels.  k <= 64 (every call of the retrieval path: k = 50, the rerank top-10): select_wave_kernel, the
// wavefront-shuffle reduction -- see below.  k in (64, 1024]: select_kernel, the LDS network of rounds 1-3:
//
// One 1024-thread workgroup per query; everything happens in a 128-KiB LDS buffer
// of 64-bit composite keys  (order-preserving score key << 32 | ~index), so a plain
// unsigned compare IS the total order (score desc, index asc) and the result is
// exact and deterministic whatever order the candidates were appended in.
// Network: bitonic-sort every group of kpad (= pow2 >= k) keys, then log2(n/kpad)
// rounds of "elementwise max of A[i] and B[kpad-1-i]" (the top kpad of two sorted
// groups, as a bitonic sequence) + a log2(kpad)-stage bitonic merge.  O(n log^2 k)
// compare-exchanges, no atomics, no data-dependent control flow.  Inputs longer
// than the buffer are streamed in chunks with the running top-k carried along.
//
// Used for: (1) the sample phase of the scan (dense scores, implicit indices),
// (2) the final pass over the filtered candidates, (3) the multi-GPU / multi-
// index merge after the RCCL all-gather (tt_topk_merge).
// NaN scores and entries with a negative index never rank.
#include "common.h"
#include "scan.h"

namespace {

constexpr int kSelThreads = 1024;
constexpr int kMaxK = 1024;
constexpr int kChunk = 8192;   // composite keys staged in LDS per round (64 KiB)
constexpr int kAux = 1024;     // per-thread maxima of the prefilter (8 KiB)
constexpr int kAux2 = 4096;    // prefilter survivors (32 KiB)
typedef unsigned long long u64;

__device__ __forceinline__ uint32_t score_key(float f) {
    if (f != f) return 0u;           // NaN: never selected
    if (f == 0.f) f = 0.f;           // -0 -> +0 so equal scores have equal keys
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ float key_score(uint32_t k) {
    const uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __uint_as_float(u);
}

__device__ __forceinline__ u64 make_key(float s, int32_t id) {
    const uint32_t k = score_key(s);
    if (k == 0u || id < 0) return 0ull;
    return ((u64)k << 32) | (u64)(0xFFFFFFFFu - (uint32_t)id);
}

// One compare-exchange stage over n_ce pairs.  Pairs are processed in batches of kU per
// thread with all LDS reads issued before the first write, so the reads pipeline
// instead of paying one LDS round trip per pair (4-8x faster than the naive loop).
constexpr int kU = 4;

template <typename PairFn>
__device__ __forceinline__ void ce_stage(u64* buf, int n_ce, PairFn pair) {
    const int tid = threadIdx.x;
    const int wave_first = tid & ~63;  // first thread id of this wave (wave-uniform)
    for (int base = 0; base + wave_first < n_ce; base += kU * kSelThreads) {
        int lo[kU], hi[kU];
        bool desc[kU];
        u64 a[kU], b[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int t = base + u * kSelThreads + tid;
            lo[u] = -1;
            if (t < n_ce) {
                pair(t, lo[u], hi[u], desc[u]);
                a[u] = buf[lo[u]];
                b[u] = buf[hi[u]];
            }
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            if (lo[u] >= 0 && ((a[u] < b[u]) == desc[u])) {
                buf[lo[u]] = b[u];
                buf[hi[u]] = a[u];
            }
        }
    }
    __syncthreads();
}

// top-kpad of buf[0..n_act) -> buf[0..kpad) sorted descending.  n_act, kpad powers of two,
// n_act >= 2*kpad.  lk = log2(kpad).
__device__ __forceinline__ void topk_network(u64* buf, int n_act, int kpad, int lk) {
    const int tid = threadIdx.x;
    // 1) sort each kpad-group (last stage descending for every group)
    for (int size = 2; size <= kpad; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            ce_stage(buf, n_act >> 1, [=](int i, int& lo, int& hi, bool& desc) {
                lo = 2 * i - (i & (stride - 1));
                hi = lo + stride;
                desc = ((lo & size) == 0) || (size == kpad);
            });
        }
    }
    // 2) merge-and-halve rounds
    for (int span = kpad; span < n_act; span <<= 1) {
        const int npairs = n_act / (2 * span);
        for (int t = tid; t < npairs * kpad; t += kSelThreads) {
            const int pr = t >> lk, i = t & (kpad - 1);
            const int a0 = pr * 2 * span;
            const u64 a = buf[a0 + i], b = buf[a0 + span + kpad - 1 - i];
            buf[a0 + i] = a > b ? a : b;
        }
        __syncthreads();
        for (int stride = kpad >> 1; stride > 0; stride >>= 1) {
            ce_stage(buf, npairs * (kpad >> 1), [=](int t, int& lo, int& hi, bool& desc) {
                const int pr = t >> (lk - 1), i = t & ((kpad >> 1) - 1);
                lo = pr * 2 * span + 2 * i - (i & (stride - 1));
                hi = lo + stride;
                desc = true;
            });
        }
    }
}

// Fold the `filled` keys staged at buf[kpad ..) into the running top-k at buf[0 .. kpad).
//
// Prefilter (when many keys are staged): thread t takes the max of its strided share;
// the kpad-th best of those 1024 maxima, thr, is a lower bound of the kpad-th best staged
// key (kpad distinct keys are >= thr), and every key >= thr lies in one of the <= kpad
// shares whose max is >= thr -- at most kpad * ceil(filled/1024) survivors, which then
// go through the exact network instead of all `filled` keys.  Keys are unique (they
// embed the row index), so there are no ties to reason about.
__device__ __forceinline__ void flush_staged(u64* buf, int filled, int kpad, int lk) {
    u64* aux = buf + kChunk;
    u64* aux2 = aux + kAux;
    int* ctr = reinterpret_cast<int*>(aux2 + kAux2);
    const int tid = threadIdx.x;
    const int per_thread = (filled + kSelThreads - 1) / kSelThreads;
    if (filled >= 2048 && per_thread * kpad <= kAux2) {
        u64 mx = 0ull;
        for (int j = tid; j < filled; j += kSelThreads) {
            const u64 key = buf[kpad + j];
            mx = key > mx ? key : mx;
        }
        aux[tid] = mx;
        __syncthreads();
        topk_network(aux, kAux, kpad, lk);
        const u64 thr = aux[kpad - 1];
        if (tid == 0) ctr[2] = 0;
        __syncthreads();
        for (int j = tid; j < filled; j += kSelThreads) {
            const u64 key = buf[kpad + j];
            if (key != 0ull && key >= thr) {
                const int pos = atomicAdd(&ctr[2], 1);
                if (pos < kAux2) aux2[pos] = key;
            }
        }
        __syncthreads();
        const int ns = ctr[2];
        if (ns <= kAux2) {
            for (int j = tid; j < ns; j += kSelThreads) buf[kpad + j] = aux2[j];
            filled = ns;
        }
        __syncthreads();
    }
    int n_act = 2 * kpad;
    while (n_act < kpad + filled) n_act <<= 1;
    for (int i = kpad + filled + tid; i < n_act; i += kSelThreads) buf[i] = 0ull;
    __syncthreads();
    topk_network(buf, n_act, kpad, lk);
}

__global__ __launch_bounds__(kSelThreads) void select_kernel(SelectParams p) {
    extern __shared__ __attribute__((aligned(16))) u64 buf[];
    int* lds_ctr = reinterpret_cast<int*>(buf + kChunk + kAux + kAux2);  // 4 ints behind the key buffers
    const int q = blockIdx.x;
    const int tid = threadIdx.x;
    const float* sc = p.scores ? p.scores + (size_t)q * p.stride : nullptr;
    const int32_t* ix = p.idx ? p.idx + (size_t)q * p.stride : nullptr;
    int m = p.m_fixed;
    size_t out_row = q;
    int32_t idx_add = p.idx_base;
    if (p.n_seg > 0) {  // segmented list: blockIdx.y = segment
        const int s = blockIdx.y;
        const int lo = p.seg_off[s];
        m = p.seg_off[s + 1] - lo;
        if (sc) sc += lo;
        if (ix) ix += lo;
        idx_add = p.seg_add[s];
        out_row = (size_t)q * p.n_seg + s;
    }
    if (p.cnt) {
        const int c = p.cnt[q];
        m = c < p.cap ? c : p.cap;
        if (c > p.cap && p.overflow_flag && tid == 0) atomicOr(p.overflow_flag, 1);
    }
    if (!sc) m = 0;
    const int k = p.k;
    int kpad = 2, lk = 1;
    while (kpad < k) { kpad <<= 1; ++lk; }
    const int room = kChunk - kpad;

    for (int i = tid; i < kpad; i += kSelThreads) buf[i] = 0ull;
    int filled = 0;

    // ---- source 1: a flat list (dense scores with implicit indices, or score/index pairs)
    // Loads are issued kLd at a time per thread before the first use: this kernel runs on
    // 64 CUs with one block each, so memory-level parallelism has to come from the unroll.
    constexpr int kLd = 8;
    int pos = 0;
    while (pos < m) {
        const int n = (m - pos) < (room - filled) ? (m - pos) : (room - filled);
        for (int i0 = tid; i0 < n; i0 += kLd * kSelThreads) {
            float v[kLd];
            int32_t id[kLd];
#pragma unroll
            for (int u = 0; u < kLd; ++u) {
                const int i = i0 + u * kSelThreads;
                v[u] = 0.f;
                id[u] = -1;
                if (i < n) {
                    v[u] = sc[pos + i];
                    id[u] = ix ? ix[pos + i] : pos + i;
                }
            }
#pragma unroll
            for (int u = 0; u < kLd; ++u) {
                const int i = i0 + u * kSelThreads;
                if (i < n) buf[kpad + filled + i] = make_key(v[u], id[u]);
            }
        }
        filled += n;
        pos += n;
        if (filled == room) {
            flush_staged(buf, filled, kpad, lk);
            filled = 0;
        }
    }

    // ---- source 2: the filter pass's private sub-lists -------------------------------
    if (p.priv) {
        const uint2* pv = p.priv + (size_t)q * p.n_sub * TT_SCAN_PRIV_SLOTS;
        const int32_t* pc = p.priv_cnt + (size_t)q * p.n_sub;
        // rounds of kSub sub-lists per thread; all fill counts of a round are loaded at once,
        // then the entries in groups of kGrp per sub-list
        constexpr int kSub = 4, kGrp = 4;
        for (int b0 = 0; b0 < p.n_sub; b0 += kSub * kSelThreads) {
            int c[kSub], off[kSub];
#pragma unroll
            for (int u = 0; u < kSub; ++u) {
                const int sidx = b0 + u * kSelThreads + tid;
                c[u] = sidx < p.n_sub ? pc[sidx] : 0;
            }
            int mine = 0;
#pragma unroll
            for (int u = 0; u < kSub; ++u) {
                c[u] = c[u] < TT_SCAN_PRIV_SLOTS ? c[u] : TT_SCAN_PRIV_SLOTS;
                mine += c[u];
            }
            if (tid == 0) lds_ctr[0] = 0;
            __syncthreads();
            int my_off = mine ? atomicAdd(&lds_ctr[0], mine) : 0;
            __syncthreads();
            const int total = lds_ctr[0];
            if (total > room) {
                // pathological round (heavily clustered hits): stage it sub-list by sub-list
#pragma unroll
                for (int u = 0; u < kSub; ++u) {
                    const int sidx = b0 + u * kSelThreads + tid;
                    for (int quarter = 0; quarter < 4; ++quarter) {
                        const bool on = (tid >> 8) == quarter;  // 256 threads * 16 <= room
                        if (tid == 0) lds_ctr[0] = 0;
                        __syncthreads();
                        const int o = (on && c[u]) ? atomicAdd(&lds_ctr[0], c[u]) : 0;
                        __syncthreads();
                        const int tot = lds_ctr[0];
                        if (filled + tot > room) {
                            flush_staged(buf, filled, kpad, lk);
                            filled = 0;
                        }
                        if (on)
                            for (int j = 0; j < c[u]; ++j) {
                                const uint2 e = pv[(size_t)sidx * TT_SCAN_PRIV_SLOTS + j];
                                buf[kpad + filled + o + j] = make_key(__uint_as_float(e.x), (int32_t)e.y);
                            }
                        filled += tot;
                        __syncthreads();
                    }
                }
                continue;
            }
            if (filled + total > room) {  // block-uniform
                flush_staged(buf, filled, kpad, lk);
                filled = 0;
            }
#pragma unroll
            for (int u = 0; u < kSub; ++u) {
                off[u] = my_off;
                my_off += c[u];
            }
            for (int j0 = 0; j0 < TT_SCAN_PRIV_SLOTS; j0 += kGrp) {
                bool any = false;
#pragma unroll
                for (int u = 0; u < kSub; ++u) any = any || (c[u] > j0);
                if (!__syncthreads_or(any ? 1 : 0)) break;
                uint2 e[kSub][kGrp];
#pragma unroll
                for (int u = 0; u < kSub; ++u) {
                    const int sidx = b0 + u * kSelThreads + tid;
#pragma unroll
                    for (int j = 0; j < kGrp; ++j)
                        if (j0 + j < c[u]) e[u][j] = pv[(size_t)sidx * TT_SCAN_PRIV_SLOTS + j0 + j];
                }
#pragma unroll
                for (int u = 0; u < kSub; ++u)
#pragma unroll
                    for (int j = 0; j < kGrp; ++j)
                        if (j0 + j < c[u])
                            buf[kpad + filled + off[u] + j0 + j] =
                                make_key(__uint_as_float(e[u][j].x), (int32_t)e[u][j].y);
            }
            filled += total;
            __syncthreads();
        }
    }
    if (filled > 0) flush_staged(buf, filled, kpad, lk);
    __syncthreads();

    for (int i = tid; i < k; i += kSelThreads) {
        const u64 e = buf[i];
        float s = -__builtin_inff();
        int32_t id = -1;
        if (e != 0ull) {
            s = key_score((uint32_t)(e >> 32));
            id = (int32_t)(0xFFFFFFFFu - (uint32_t)(e & 0xFFFFFFFFull));
            if (!ix) id += idx_add;
        }
        p.out_scores[out_row * p.out_stride + i] = s;
        p.out_idx[out_row * p.out_stride + i] = id;
    }
    if (p.min_valid > 0 && p.overflow_flag && tid == 0 && buf[p.min_valid - 1] == 0ull) atomicOr(p.overflow_flag, 1);
    if (p.thr_out || p.cnt_out) {
        if (tid == 0) lds_ctr[1] = 0;
        __syncthreads();
        int local = 0;
        for (int i = tid; i < k; i += kSelThreads) local += buf[i] != 0ull ? 1 : 0;
        if (local) atomicAdd(&lds_ctr[1], local);
        __syncthreads();
        if (tid == 0) {
            const u64 e = buf[k - 1];
            if (p.thr_out) p.thr_out[out_row] = (e != 0ull) ? key_score((uint32_t)(e >> 32)) : -__builtin_inff();
            if (p.cnt_out) p.cnt_out[out_row] = lds_ctr[1];
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// select_wave_kernel: top-k for k <= 64 as a wavefront-shuffle reduction (round 4).
//
// The same 64-bit composite keys, but a list never goes through an LDS sorting network: every WAVE keeps a running
// top-64 in registers -- one key per lane, sorted descending, lane 0 the best -- and streams its share of the
// candidates through it in chunks of 64 (one key per lane):
//     skip     a chunk none of whose keys beats the wave's current 64th best changes nothing: one compare + ballot
//     sort     the chunk ascending: 21 compare-exchange stages, the partner's key fetched lane-to-lane with DPP
//              (quad_perm, row_mirror / row_half_mirror, row_ror:8), v_permlane16_swap and v_permlane32_swap -- VALU
//              cross-lane moves, no LDS access, no barrier
//     merge    max(run[i], chunk[i]) of a descending and an ascending list is a bitonic sequence holding the top 64 of
//              both: 6 more stages sort it
// One LDS exchange at the end: the W (1, 4 or 16) waves' lists are merged 4-to-1 per round, read back lane-reversed
// (so no cross-lane reversal is needed), at most two barriers per launch where the LDS network paid one per stage
// (~100 per launch).  Exact and deterministic for the same reason as before: the keys are unique and totally ordered.
// Measured (256 queries): threshold selection over 1024 / 4096 group maxima and final selection over ~2-4 k candidates
// per query in 5-8 us per launch (LDS network: 33-288 us).
// ---------------------------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ uint32_t dpp32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
}
template <int CTRL>
__device__ __forceinline__ u64 dpp64(u64 v) {
    return ((u64)dpp32<CTRL>((uint32_t)(v >> 32)) << 32) | (u64)dpp32<CTRL>((uint32_t)v);
}
constexpr int kDppXor1 = 0xB1;          // quad_perm [1, 0, 3, 2]
constexpr int kDppXor2 = 0x4E;          // quad_perm [2, 3, 0, 1]
constexpr int kDppXor3 = 0x1B;          // quad_perm [3, 2, 1, 0]
constexpr int kDppHalfMirror = 0x141;   // lane i <- lane 7 - i of its half row  (i ^ 7)
constexpr int kDppRor8 = 0x128;         // row_ror:8  (i ^ 8)

// the key held by lane (lane ^ S)
template <int S>
__device__ __forceinline__ u64 lane_xor(u64 x, int lane) {
    if constexpr (S == 1) return dpp64<kDppXor1>(x);
    else if constexpr (S == 2) return dpp64<kDppXor2>(x);
    else if constexpr (S == 4) return dpp64<kDppXor3>(dpp64<kDppHalfMirror>(x));      // (i ^ 7) ^ 3
    else if constexpr (S == 8) return dpp64<kDppRor8>(x);
    else if constexpr (S == 16) {
        // v_permlane16_swap: odd rows of vdst <-> even rows of src.  With both = x: r[0] holds, on odd rows, x of the row before;
        // r[1] holds, on even rows, x of the row after.
        const auto lo = __builtin_amdgcn_permlane16_swap((uint32_t)x, (uint32_t)x, false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap((uint32_t)(x >> 32), (uint32_t)(x >> 32), false, false);
        const bool odd = (lane & 16) != 0;
        return ((u64)(odd ? hi[0] : hi[1]) << 32) | (u64)(odd ? lo[0] : lo[1]);
    } else {
        static_assert(S == 32, "strides 1..32");
        // v_permlane32_swap: upper half of vdst <-> lower half of src
        const auto lo = __builtin_amdgcn_permlane32_swap((uint32_t)x, (uint32_t)x, false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap((uint32_t)(x >> 32), (uint32_t)(x >> 32), false, false);
        const bool up = (lane & 32) != 0;
        return ((u64)(up ? hi[0] : hi[1]) << 32) | (u64)(up ? lo[0] : lo[1]);
    }
}

// one compare-exchange stage: lanes whose `take_max` is set keep the larger of (own, partner's) key, the others the smaller
template <int S>
__device__ __forceinline__ u64 ce_lane(u64 x, int lane, bool take_max) {
    const u64 y = lane_xor<S>(x, lane);
    return ((x > y) == take_max) ? x : y;
}

// bitonic merge of a bitonic 64-sequence, result descending (DESC) or ascending over the lanes
template <bool DESC>
__device__ __forceinline__ u64 bitonic_merge64(u64 x, int lane) {
    x = ce_lane<32>(x, lane, ((lane & 32) == 0) == DESC);
    x = ce_lane<16>(x, lane, ((lane & 16) == 0) == DESC);
    x = ce_lane<8>(x, lane, ((lane & 8) == 0) == DESC);
    x = ce_lane<4>(x, lane, ((lane & 4) == 0) == DESC);
    x = ce_lane<2>(x, lane, ((lane & 2) == 0) == DESC);
    x = ce_lane<1>(x, lane, ((lane & 1) == 0) == DESC);
    return x;
}

// full bitonic sort of 64 keys, one per lane
template <bool DESC>
__device__ __forceinline__ u64 sort64(u64 x, int lane) {
    // size 2 .. 32: region (lane & size) == 0 sorts descending, the other ascending (so that the next size sees bitonic runs)
#define TT_STAGE(SIZE, S) x = ce_lane<S>(x, lane, ((lane & S) == 0) == (((lane & SIZE) == 0) == DESC))
    TT_STAGE(2, 1);
    TT_STAGE(4, 2); TT_STAGE(4, 1);
    TT_STAGE(8, 4); TT_STAGE(8, 2); TT_STAGE(8, 1);
    TT_STAGE(16, 8); TT_STAGE(16, 4); TT_STAGE(16, 2); TT_STAGE(16, 1);
    TT_STAGE(32, 16); TT_STAGE(32, 8); TT_STAGE(32, 4); TT_STAGE(32, 2); TT_STAGE(32, 1);
#undef TT_STAGE
    return bitonic_merge64<DESC>(x, lane);
}

__device__ __forceinline__ u64 bcast_lane63(u64 x) {
    return ((u64)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), 63) << 32) |
           (u64)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, 63);
}

// fold one chunk (a key per lane, any order, 0 = empty) into the wave's running top-64 (descending; thr = its lane-63 key)
__device__ __forceinline__ void fold_chunk(u64& run, u64& thr, u64 key, int lane) {
    if (__builtin_amdgcn_ballot_w64(key > thr) == 0ull) return;     // wave-uniform: nothing in the chunk can enter
    const u64 asc = sort64<false>(key, lane);
    const u64 mx = run > asc ? run : asc;                            // descending vs ascending: the top 64 of both, bitonic
    run = bitonic_merge64<true>(mx, lane);
    thr = bcast_lane63(run);
}

constexpr int kWaveSelMaxWaves = 16;

__global__ __launch_bounds__(64 * kWaveSelMaxWaves) void select_wave_kernel(SelectParams p) {
    __shared__ u64 lists[kWaveSelMaxWaves][64];
    const int q = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W = blockDim.x >> 6;
    size_t out_row = q;
    if (q >= p.n_real) {            // padding query of a 256-wide filter batch: nothing passes, nothing is selected
        if (tid == 0) {
            if (p.thr_out) p.thr_out[out_row] = __builtin_inff();
            if (p.zero_cnt) p.zero_cnt[q] = 0;
        }
        return;
    }
    const float* sc = p.scores ? p.scores + (size_t)q * p.stride : nullptr;
    const int32_t* ix = p.idx ? p.idx + (size_t)q * p.stride : nullptr;
    int m = p.m_fixed;
    int32_t idx_add = p.idx_base;
    if (p.n_seg > 0) {
        const int s = blockIdx.y;
        const int lo = p.seg_off[s];
        m = p.seg_off[s + 1] - lo;
        if (sc) sc += lo;
        if (ix) ix += lo;
        idx_add = p.seg_add[s];
        out_row = (size_t)q * p.n_seg + s;
    }
    if (p.cnt) {
        const int c = p.cnt[q];
        m = c < p.cap ? c : p.cap;
        if (c > p.cap && p.overflow_flag && tid == 0) atomicOr(p.overflow_flag, 1);
    }
    if (!sc) m = 0;
    const int k = p.k;

    u64 run = 0ull, thr = 0ull;
    // ---- source 1: the flat list; wave w takes chunks w, w + W, ...; kU chunks' loads are in flight together
    constexpr int kU = 4;
    for (int base = wave * 64; base < m; base += kU * W * 64) {
        float v[kU];
        int32_t id[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int i = base + u * W * 64 + lane;
            v[u] = 0.f;
            id[u] = -1;
            if (i < m) {
                v[u] = sc[i];
                id[u] = ix ? ix[i] : i;
            }
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            if (base + u * W * 64 >= m) break;                        // wave-uniform
            fold_chunk(run, thr, make_key(v[u], id[u]), lane);
        }
    }
    // ---- source 2: the filter pass's private sub-lists: a sub-list per lane, slot j of 64 sub-lists = one chunk
    if (p.priv) {
        const uint2* pv = p.priv + (size_t)q * p.n_sub * TT_SCAN_PRIV_SLOTS;
        const int32_t* pc = p.priv_cnt + (size_t)q * p.n_sub;
        for (int sb = wave * 64; sb < p.n_sub; sb += W * 64) {
            const int sidx = sb + lane;
            int c = sidx < p.n_sub ? pc[sidx] : 0;
            c = c < TT_SCAN_PRIV_SLOTS ? c : TT_SCAN_PRIV_SLOTS;
            int cmax = c;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const int other = __shfl_xor(cmax, o, 64);
                cmax = other > cmax ? other : cmax;
            }
            for (int j0 = 0; j0 < cmax; j0 += kU) {
                uint2 e[kU];
#pragma unroll
                for (int u = 0; u < kU; ++u)
                    if (j0 + u < c) e[u] = pv[(size_t)sidx * TT_SCAN_PRIV_SLOTS + j0 + u];
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    if (j0 + u >= cmax) break;
                    const u64 key = (j0 + u < c) ? make_key(__uint_as_float(e[u].x), (int32_t)e[u].y) : 0ull;
                    fold_chunk(run, thr, key, lane);
                }
            }
        }
    }
    // ---- merge the waves' lists 4-to-1 per round: the reader takes the others' lists lane-reversed (ascending)
    for (int step = 1; step < W; step *= 4) {
        lists[wave][lane] = run;
        __syncthreads();
        if (wave % (4 * step) == 0) {
#pragma unroll
            for (int o = 1; o < 4; ++o) {
                const int other = wave + o * step;
                if (other < W) {
                    const u64 asc = lists[other][63 - lane];
                    const u64 mx = run > asc ? run : asc;
                    run = bitonic_merge64<true>(mx, lane);
                }
            }
        }
        __syncthreads();
    }
    if (wave != 0) return;
    if (lane < k) {
        float s = -__builtin_inff();
        int32_t id = -1;
        if (run != 0ull) {
            s = key_score((uint32_t)(run >> 32));
            id = (int32_t)(0xFFFFFFFFu - (uint32_t)(run & 0xFFFFFFFFull));
            if (!ix) id += idx_add;
        }
        p.out_scores[out_row * p.out_stride + lane] = s;
        p.out_idx[out_row * p.out_stride + lane] = id;
    }
    const unsigned long long valid = __builtin_amdgcn_ballot_w64(run != 0ull && lane < k);
    if (p.min_valid > 0 && p.overflow_flag && lane == 0 && !((valid >> (p.min_valid - 1)) & 1ull)) atomicOr(p.overflow_flag, 1);
    if (lane == k - 1) {
        if (p.thr_out) {
            float t = (run != 0ull) ? key_score((uint32_t)(run >> 32)) : -__builtin_inff();
            // thr_relax: lower the threshold by more than two fp32 evaluations of the same K <= 1024 dot product of unit-norm
            // bf16 rows can differ by (scan_api.hip: the sample and the tiled filter pass sum in different orders)
            if (p.thr_relax && t - t == 0.f) t = t - (fabsf(t) * 1.220703125e-4f + 2e-6f);
            p.thr_out[out_row] = t;
        }
    }
    if (lane == 0) {
        if (p.cnt_out) p.cnt_out[out_row] = __builtin_popcountll(valid);
        if (p.zero_cnt) p.zero_cnt[q] = 0;
    }
}

}  // namespace

static bool select_wave_on() {
    // TT_SELECT_WAVE=0: the LDS network for every k (the A/B switch; both kernels are exact: the same outputs)
    static const bool on = TT_DIAG_ENV_INT("TT_SELECT_WAVE", 1) != 0;
    return on;
}

bool tt_select_fused_outputs(int k) { return select_wave_on() && k <= 64; }

int tt_select_launch(const SelectParams& p, int n_queries, hipStream_t stream) {
    if (n_queries <= 0) return TT_OK;
    if (p.k < 1 || p.k > kMaxK) {
        tt_set_error("top-k: k=%d outside [1,%d]", p.k, kMaxK);
        return TT_E_INVALID;
    }
    //  the wave kernel tests bit (min_valid - 1) of a 64-bit ballot and the LDS network reads buf[min_valid - 1]: a
    // request for more valid outputs than the kernel writes is a caller's error, not a shift by >= 64
    if (p.min_valid < 0 || p.min_valid > p.k) {
        tt_set_error("top-k: min_valid=%d outside [0, k=%d]", p.min_valid, p.k);
        return TT_E_INVALID;
    }
    // thr_out (and cnt_out / zero_cnt) are indexed by the GRID's query index: with n_real < n_queries the padding blocks write
    // thr_out[q] = +inf for q up to n_queries, so those buffers must hold n_queries entries (scan_api.hip's Plan sizes them by qpad)
    const bool wave_on = select_wave_on();
    SelectParams q = p;
    if (q.n_real <= 0) q.n_real = n_queries;
    if (wave_on && p.k <= 64) {
        // waves per query: one wave folds a 64-key chunk in ~1.5 us of dependent VALU work, and the 4-to-1 merge rounds cost a
        // barrier each: 1 wave up to 128 keys, 4 up to 1024, 16 beyond (an upper bound of the list length is all the host knows)
        long long bound = p.cnt ? (long long)p.cap : (long long)p.m_fixed;
        if (p.n_seg > 0) {
            bound = 0;
            for (int s = 0; s < p.n_seg; ++s) bound = bound > p.seg_off[s + 1] - p.seg_off[s] ? bound : p.seg_off[s + 1] - p.seg_off[s];
        }
        if (p.priv) bound += (long long)p.n_sub * TT_SCAN_PRIV_SLOTS;
        const int waves = bound <= 128 ? 1 : (bound <= 1024 ? 4 : kWaveSelMaxWaves);
        TtProfScope prof(TT_K_SELECT, stream);
        hipLaunchKernelGGL(select_wave_kernel, dim3(n_queries, p.n_seg > 0 ? p.n_seg : 1), dim3(64 * waves), 0, stream, q);
    } else {
        if (p.thr_relax || p.zero_cnt || q.n_real != n_queries) {
            tt_set_error("top-k: the fused threshold outputs exist in the k <= 64 kernel only");
            return TT_E_UNSUPPORTED;
        }
        const size_t lds = (size_t)(kChunk + kAux + kAux2) * sizeof(u64) + 16;
        TT_SET_MAX_LDS(select_kernel, lds);
        TtProfScope prof(TT_K_SELECT, stream);
        hipLaunchKernelGGL(select_kernel, dim3(n_queries, p.n_seg > 0 ? p.n_seg : 1), dim3(kSelThreads), lds, stream, q);
    }
    TT_CHECK_LAUNCH();
    return TT_OK;
}
