// C-ABI entry points for the similarity scan (see include/tt_hip.h).
//
// Pipeline of tt_scan_topk for a shard of N rows (all launches on one stream):
//   1. sample : dense scores of rows [0, n0)                    (scan_kernel DENSE)
//   2. select : exact top-k of the sample -> seeds the candidate lists,
//               thr[q] = k-th best sample score                 (select_kernel)
//   3. main   : rows [n0, N), append scores >= thr[q]           (scan_kernel filter)
//   4. select : exact top-k of the candidates                   (select_kernel)
// Exactness: every true top-k row either lies in the sample's top-k or has a
// score >= thr[q] (thr is the k-th best of a subset, hence <= the true k-th
// best).  The expected candidate volume is k * N / n0 per query; n0 is sized so
// that this is <= cap/4, and an overflow (adversarial score order) is reported
// through status_flag rather than silently truncated.
#include <limits.h>
#include <string.h>

#include "common.h"
#include "scan.h"

static thread_local char g_err[512] = "";

void tt_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int tt_cu_count_cached() {
    static thread_local int cached_dev = -1;
    static thread_local int cached_cus = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (dev != cached_dev) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
        cached_dev = dev;
        cached_cus = cus;
    }
    return cached_cus;
}

namespace {

constexpr int kCap = 16384;            // candidate slots per query
constexpr int64_t kDenseMaxRows = 65536;  // shards up to this size take the dense-only path
constexpr int64_t kSampleMin = 16384;
constexpr int64_t kSampleMax = 262144;

struct Plan {
    bool dense_only;
    int64_t n0;        // sample rows (== n_rows when dense_only)
    int64_t stride;    // dense row stride (floats)
    int qpad;
    size_t off_dense, off_cs, off_ci, off_cnt, off_thr, total;
};

Plan make_plan(int64_t n_rows, int n_queries, int k) {
    Plan pl{};
    pl.qpad = (n_queries + 63) / 64 * 64;
    if (n_rows <= kDenseMaxRows) {
        pl.dense_only = true;
        pl.n0 = n_rows;
    } else {
        pl.dense_only = false;
        // expected candidates/query = k * N / n0  <= cap / 4
        int64_t need = (4 * (int64_t)k * n_rows + kCap - 1) / kCap;
        need = (need + 8191) / 8192 * 8192;
        if (need < kSampleMin) need = kSampleMin;
        if (need > kSampleMax) need = kSampleMax;
        if (need > n_rows) need = n_rows;
        pl.n0 = need;
    }
    pl.stride = (pl.n0 + 31) / 32 * 32;
    size_t off = 0;
    pl.off_dense = off; off += tt_align_up((size_t)pl.qpad * pl.stride * sizeof(float), 256);
    pl.off_cs = off;    off += tt_align_up((size_t)pl.qpad * kCap * sizeof(float), 256);
    pl.off_ci = off;    off += tt_align_up((size_t)pl.qpad * kCap * sizeof(int32_t), 256);
    pl.off_cnt = off;   off += tt_align_up((size_t)pl.qpad * sizeof(int32_t), 256);
    pl.off_thr = off;   off += tt_align_up((size_t)pl.qpad * sizeof(float), 256);
    pl.total = off;
    return pl;
}

int check_common(const void* corpus, int64_t n_rows, int dim, const void* queries, int n_queries, int k,
                 const float* out_scores, const int32_t* out_idx) {
    TT_CHECK_ARG(n_rows >= 0 && n_rows < (int64_t)INT32_MAX, "n_rows=%lld out of range", (long long)n_rows);
    TT_CHECK_ARG(n_queries >= 0, "n_queries=%d", n_queries);
    TT_CHECK_ARG(k >= 1 && k <= 1024, "k=%d outside [1,1024]", k);
    TT_CHECK_ARG(dim > 0 && dim % 128 == 0 && dim <= 1024, "dim=%d must be a multiple of 128 and <= 1024", dim);
    if (n_queries == 0) return TT_OK;
    TT_CHECK_ARG(out_scores && out_idx, "null output pointer");
    TT_CHECK_ARG(queries && ((uintptr_t)queries % 16) == 0, "queries must be non-null and 16-byte aligned");
    if (n_rows > 0) TT_CHECK_ARG(corpus && ((uintptr_t)corpus % 16) == 0, "corpus must be non-null and 16-byte aligned");
    return TT_OK;
}

__global__ void fill_pad_kernel(float* s, int32_t* ix, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        s[i] = -__builtin_inff();
        ix[i] = -1;
    }
}

int scan_mode_from_env() {
    // TT_SCAN_MODE=0|1 selects the corpus load path (bench/ablation knob); default 1.
    const char* e = getenv("TT_SCAN_MODE");
    if (e && e[0] == '0') return 0;
    return 1;
}

}  // namespace

extern "C" {

int tt_version(void) { return 1; }
const char* tt_arch(void) { return "gfx950"; }
const char* tt_last_error(void) { return g_err; }
int tt_device_cu_count(void) { return tt_cu_count_cached(); }

size_t tt_scan_workspace_bytes(int64_t n_rows, int dim, int n_queries, int k) {
    (void)dim;
    if (n_rows < 0 || n_queries <= 0 || k < 1) return 0;
    return make_plan(n_rows, n_queries, k).total;
}

size_t tt_scan_exact_workspace_bytes(int64_t n_rows, int dim, int n_queries, int k) {
    (void)dim; (void)k;
    if (n_rows < 0 || n_queries <= 0) return 0;
    const int qpad = (n_queries + 63) / 64 * 64;
    const int64_t stride = (n_rows + 31) / 32 * 32;
    return tt_align_up((size_t)qpad * stride * sizeof(float), 256);
}

static int scan_dense_select(const void* corpus, int64_t n_rows, int dim, const void* queries, int n_queries,
                             int k, int32_t idx_base, float* out_scores, int32_t* out_idx, float* dense,
                             int64_t stride, hipStream_t st) {
    ScanParams sp{};
    sp.corpus = (const uint16_t*)corpus;
    sp.queries = (const uint16_t*)queries;
    sp.row_lo = 0;
    sp.row_hi = n_rows;
    sp.n_queries = n_queries;
    sp.idx_base = idx_base;
    sp.dense = dense;
    sp.dense_stride = stride;
    int rc = tt_scan_launch(sp, dim, scan_mode_from_env(), true, tt_cu_count_cached(), st);
    if (rc) return rc;
    SelectParams se{};
    se.scores = dense;
    se.idx = nullptr;
    se.stride = stride;
    se.cnt = nullptr;
    se.m_fixed = (int)n_rows;
    se.cap = INT_MAX;
    se.idx_base = idx_base;
    se.k = k;
    se.out_scores = out_scores;
    se.out_idx = out_idx;
    se.out_stride = k;
    return tt_select_launch(se, n_queries, st);
}

int tt_scan_topk_exact(const void* corpus_bf16, int64_t n_rows, int dim, const void* queries_bf16,
                       int n_queries, int k, int32_t idx_base, float* out_scores, int32_t* out_idx,
                       void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_common(corpus_bf16, n_rows, dim, queries_bf16, n_queries, k, out_scores, out_idx);
    if (rc) return rc;
    if (n_queries == 0) return TT_OK;
    hipStream_t st = (hipStream_t)stream;
    if (n_rows == 0) {
        const int64_t n = (int64_t)n_queries * k;
        hipLaunchKernelGGL(fill_pad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, out_scores, out_idx, n);
        TT_CHECK_LAUNCH();
        return TT_OK;
    }
    const size_t need = tt_scan_exact_workspace_bytes(n_rows, dim, n_queries, k);
    if (!workspace || workspace_bytes < need) {
        tt_set_error("tt_scan_topk_exact: workspace %zu < required %zu bytes", workspace_bytes, need);
        return TT_E_WORKSPACE;
    }
    const int64_t stride = (n_rows + 31) / 32 * 32;
    return scan_dense_select(corpus_bf16, n_rows, dim, queries_bf16, n_queries, k, idx_base, out_scores, out_idx,
                             (float*)workspace, stride, st);
}

int tt_scan_topk(const void* corpus_bf16, int64_t n_rows, int dim, const void* queries_bf16, int n_queries,
                 int k, int32_t idx_base, float* out_scores, int32_t* out_idx, void* workspace,
                 size_t workspace_bytes, int32_t* status_flag, void* stream) {
    int rc = check_common(corpus_bf16, n_rows, dim, queries_bf16, n_queries, k, out_scores, out_idx);
    if (rc) return rc;
    if (n_queries == 0) return TT_OK;
    hipStream_t st = (hipStream_t)stream;
    if (status_flag) TT_CHECK_HIP(hipMemsetAsync(status_flag, 0, sizeof(int32_t), st));
    if (n_rows == 0) {
        const int64_t n = (int64_t)n_queries * k;
        hipLaunchKernelGGL(fill_pad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, out_scores, out_idx, n);
        TT_CHECK_LAUNCH();
        return TT_OK;
    }
    const Plan pl = make_plan(n_rows, n_queries, k);
    if (!workspace || workspace_bytes < pl.total) {
        tt_set_error("tt_scan_topk: workspace %zu < required %zu bytes", workspace_bytes, pl.total);
        return TT_E_WORKSPACE;
    }
    TT_CHECK_ARG(((uintptr_t)workspace % 256) == 0, "workspace must be 256-byte aligned");
    char* ws = (char*)workspace;
    float* dense = (float*)(ws + pl.off_dense);
    if (pl.dense_only) {
        return scan_dense_select(corpus_bf16, n_rows, dim, queries_bf16, n_queries, k, idx_base, out_scores,
                                 out_idx, dense, pl.stride, st);
    }
    float* cs = (float*)(ws + pl.off_cs);
    int32_t* ci = (int32_t*)(ws + pl.off_ci);
    int32_t* cnt = (int32_t*)(ws + pl.off_cnt);
    float* thr = (float*)(ws + pl.off_thr);
    const int mode = scan_mode_from_env();
    const int cus = tt_cu_count_cached();

    // 1. sample: dense scores of rows [0, n0)
    ScanParams sp{};
    sp.corpus = (const uint16_t*)corpus_bf16;
    sp.queries = (const uint16_t*)queries_bf16;
    sp.row_lo = 0;
    sp.row_hi = pl.n0;
    sp.n_queries = n_queries;
    sp.idx_base = idx_base;
    sp.dense = dense;
    sp.dense_stride = pl.stride;
    rc = tt_scan_launch(sp, dim, mode, true, cus, st);
    if (rc) return rc;

    // 2. exact top-k of the sample seeds the candidate lists and the thresholds
    SelectParams s1{};
    s1.scores = dense;
    s1.idx = nullptr;
    s1.stride = pl.stride;
    s1.cnt = nullptr;
    s1.m_fixed = (int)pl.n0;
    s1.cap = INT_MAX;
    s1.idx_base = idx_base;
    s1.k = k;
    s1.out_scores = cs;
    s1.out_idx = ci;
    s1.out_stride = kCap;
    s1.thr_out = thr;
    s1.cnt_out = cnt;
    rc = tt_select_launch(s1, n_queries, st);
    if (rc) return rc;

    // 3. main pass over rows [n0, N): append scores >= thr[q]
    ScanParams mp = sp;
    mp.row_lo = pl.n0;
    mp.row_hi = n_rows;
    mp.dense = nullptr;
    mp.thr = thr;
    mp.cnt = cnt;
    mp.cand_scores = cs;
    mp.cand_idx = ci;
    mp.cap = kCap;
    rc = tt_scan_launch(mp, dim, mode, false, cus, st);
    if (rc) return rc;

    // 4. exact top-k of the candidates
    SelectParams s2{};
    s2.scores = cs;
    s2.idx = ci;
    s2.stride = kCap;
    s2.cnt = cnt;
    s2.m_fixed = 0;
    s2.cap = kCap;
    s2.k = k;
    s2.out_scores = out_scores;
    s2.out_idx = out_idx;
    s2.out_stride = k;
    s2.overflow_flag = status_flag;
    return tt_select_launch(s2, n_queries, st);
}

int tt_topk_merge(const float* in_scores, const int32_t* in_idx, int n_queries, int n_candidates, int k_out,
                  float* out_scores, int32_t* out_idx, void* stream) {
    TT_CHECK_ARG(n_queries >= 0 && n_candidates >= 0, "negative size");
    TT_CHECK_ARG(k_out >= 1 && k_out <= 1024, "k_out=%d outside [1,1024]", k_out);
    if (n_queries == 0) return TT_OK;
    TT_CHECK_ARG(in_scores && in_idx && out_scores && out_idx, "null pointer");
    SelectParams se{};
    se.scores = in_scores;
    se.idx = in_idx;
    se.stride = n_candidates;
    se.cnt = nullptr;
    se.m_fixed = n_candidates;
    se.cap = INT_MAX;
    se.k = k_out;
    se.out_scores = out_scores;
    se.out_idx = out_idx;
    se.out_stride = k_out;
    return tt_select_launch(se, n_queries, (hipStream_t)stream);
}

}  // extern "C"
