// Exact top-k selection over a per-query candidate list, gfx950.
//
// One 256-thread workgroup per query.  MSB-first radix select (4 x 8-bit passes
// over an order-preserving 32-bit key of the fp32 score) finds the k-th largest
// score; ties on that score are resolved by a second radix select on the row
// index (smallest first), so the result is the exact top-k under the total
// order (score desc, index asc) whatever the input order -- which is what makes
// the atomically-appended candidate lists of scan.hip deterministic.  The
// survivors (<= 1024) are bitonic-sorted in LDS on a 64-bit composite key.
//
// Used for: (1) the sample phase of the scan (dense scores, implicit indices),
// (2) the final pass over the filtered candidates, (3) the multi-GPU / multi-
// index merge after the RCCL all-gather (tt_topk_merge).
// NaN scores and entries with a negative index never rank.
#include "common.h"
#include "scan.h"

namespace {

constexpr int kSelThreads = 256;
constexpr int kMaxK = 1024;

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

// Finds the digit (searching from `from_top ? 255 : 0`) where the running count
// reaches `need`.  hist[] in LDS; returns via LDS words sel[0]=digit,
// sel[1]=count strictly before the digit, sel[2]=count in the digit.
__device__ __forceinline__ void pick_digit(const uint32_t* hist, uint32_t need, bool from_top, uint32_t* sel) {
    // 256 bins; a single wave does an inclusive scan with shuffles (4 bins per lane)
    const int tid = threadIdx.x;
    if (tid < 64) {
        uint32_t v[4];
        uint32_t s = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int bin = from_top ? 255 - (tid * 4 + i) : tid * 4 + i;
            v[i] = hist[bin];
            s += v[i];
        }
        uint32_t incl = s;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off, 64);
            if (tid >= off) incl += o;
        }
        uint32_t run = incl - s;  // exclusive prefix of this lane's 4 bins
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (run < need && run + v[i] >= need) {
                const int bin = from_top ? 255 - (tid * 4 + i) : tid * 4 + i;
                sel[0] = (uint32_t)bin;
                sel[1] = run;
                sel[2] = v[i];
            }
            run += v[i];
        }
    }
}

__global__ __launch_bounds__(kSelThreads) void select_kernel(SelectParams p) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t sel[4];
    __shared__ uint32_t n_out;
    __shared__ unsigned long long buf[kMaxK];

    const int q = blockIdx.x;
    const int tid = threadIdx.x;
    const float* sc = p.scores + (size_t)q * p.stride;
    const int32_t* ix = p.idx ? p.idx + (size_t)q * p.stride : nullptr;
    int m = p.m_fixed;
    if (p.cnt) {
        const int c = p.cnt[q];
        m = c < p.cap ? c : p.cap;
        if (c > p.cap && p.overflow_flag && tid == 0) atomicOr(p.overflow_flag, 1);
    }
    const int k = p.k;

    // ---- count valid entries ---------------------------------------------------
    if (tid == 0) { sel[3] = 0; n_out = 0; }
    __syncthreads();
    {
        uint32_t local = 0;
        for (int i = tid; i < m; i += kSelThreads) {
            const bool ok = score_key(sc[i]) != 0u && (!ix || ix[i] >= 0);
            local += ok ? 1u : 0u;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
        if ((tid & 63) == 0 && local) atomicAdd(&sel[3], local);
    }
    __syncthreads();
    const uint32_t n_valid = sel[3];
    const uint32_t k_eff = n_valid < (uint32_t)k ? n_valid : (uint32_t)k;

    uint32_t pivot = 0, idx_pivot = 0x7FFFFFFFu;
    if (k_eff > 0) {
        // ---- radix select on the score key: k_eff-th largest ------------------------
        uint32_t prefix = 0, mask = 0, need = k_eff, eq_count = 0;
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            hist[tid] = 0;
            __syncthreads();
            for (int i = tid; i < m; i += kSelThreads) {
                const uint32_t key = score_key(sc[i]);
                if (key != 0u && (!ix || ix[i] >= 0) && (key & mask) == prefix)
                    atomicAdd(&hist[(key >> shift) & 255u], 1u);
            }
            __syncthreads();
            pick_digit(hist, need, true, sel);
            __syncthreads();
            prefix |= sel[0] << shift;
            mask |= 255u << shift;
            need -= sel[1];
            eq_count = sel[2];
            __syncthreads();
        }
        pivot = prefix;
        // `need` entries with key == pivot are wanted out of eq_count
        if (eq_count > need) {
            // ---- radix select on the index among the ties: need-th smallest -------------
            uint32_t iprefix = 0, imask = 0, ineed = need;
            for (int pass = 0; pass < 4; ++pass) {
                const int shift = 24 - 8 * pass;
                hist[tid] = 0;
                __syncthreads();
                for (int i = tid; i < m; i += kSelThreads) {
                    if (score_key(sc[i]) != pivot) continue;
                    const int32_t id = ix ? ix[i] : i;
                    if (id < 0) continue;
                    if (((uint32_t)id & imask) == iprefix) atomicAdd(&hist[((uint32_t)id >> shift) & 255u], 1u);
                }
                __syncthreads();
                pick_digit(hist, ineed, false, sel);
                __syncthreads();
                iprefix |= sel[0] << shift;
                imask |= 255u << shift;
                ineed -= sel[1];
                __syncthreads();
            }
            idx_pivot = iprefix;
        }
        // ---- collect survivors ----------------------------------------------------------
        for (int i = tid; i < m; i += kSelThreads) {
            const uint32_t key = score_key(sc[i]);
            const int32_t id = ix ? ix[i] : i;
            if (key == 0u || id < 0) continue;
            if (key > pivot || (key == pivot && (uint32_t)id <= idx_pivot)) {
                const uint32_t pos = atomicAdd(&n_out, 1u);
                if (pos < (uint32_t)kMaxK)
                    buf[pos] = ((unsigned long long)key << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)id);
            }
        }
    }
    __syncthreads();

    // ---- bitonic sort (descending) of the composite keys, padded with 0 ----------------
    int kpad = 1;
    while (kpad < (int)k_eff) kpad <<= 1;
    for (int i = tid; i < kpad; i += kSelThreads)
        if (i >= (int)k_eff) buf[i] = 0ull;
    __syncthreads();
    for (int size = 2; size <= kpad; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < kpad / 2; i += kSelThreads) {
                const int lo = 2 * i - (i & (stride - 1));  // index with bit `stride` clear
                const int hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const unsigned long long a = buf[lo], b2 = buf[hi];
                if ((a < b2) == desc) {
                    buf[lo] = b2;
                    buf[hi] = a;
                }
            }
            __syncthreads();
        }
    }

    // ---- write results -----------------------------------------------------------------
    for (int i = tid; i < k; i += kSelThreads) {
        float s = -__builtin_inff();
        int32_t id = -1;
        if (i < (int)k_eff) {
            const unsigned long long e = buf[i];
            s = key_score((uint32_t)(e >> 32));
            id = (int32_t)(0xFFFFFFFFu - (uint32_t)(e & 0xFFFFFFFFull));
            if (!ix) id += p.idx_base;
        }
        p.out_scores[(size_t)q * p.out_stride + i] = s;
        p.out_idx[(size_t)q * p.out_stride + i] = id;
    }
    if (tid == 0) {
        if (p.thr_out) p.thr_out[q] = (k_eff == (uint32_t)k) ? key_score(pivot) : -__builtin_inff();
        if (p.cnt_out) p.cnt_out[q] = (int32_t)k_eff;
    }
}

}  // namespace

int tt_select_launch(const SelectParams& p, int n_queries, hipStream_t stream) {
    if (n_queries <= 0) return TT_OK;
    if (p.k < 1 || p.k > kMaxK) {
        tt_set_error("top-k: k=%d outside [1,%d]", p.k, kMaxK);
        return TT_E_INVALID;
    }
    hipLaunchKernelGGL(select_kernel, dim3(n_queries), dim3(kSelThreads), 0, stream, p);
    TT_CHECK_LAUNCH();
    return TT_OK;
}
