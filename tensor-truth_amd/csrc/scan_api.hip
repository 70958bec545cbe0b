// C-ABI entry points for the similarity scan (see include/tt_hip.h).
//
// Pipeline of tt_scan_topk for a shard of N rows (all launches on one stream):
//   1. sample : one max per (32-row group, query) over n0 / 32 groups spread evenly over the shard
//               (every (N / n0)-th group: ingest order clusters topics)   (scan_kernel OUT=2)
//               [65+ queries over a large shard: step 3 is ONE tiled MFMA contraction per 256 queries instead (gemm.hip,
//                TT_EPI_SCAN), its last tile shifted to the shard's last 256 rows when the row count is ragged]
//   2. select : thr[q] = k-th best group maximum                        (select_kernel)
//   3. filter : all N rows, scores >= thr[q] go to atomic-free private lists
//               (overflowing lanes: shared per-query list)               (scan_kernel OUT=0)
//   4. select : exact top-k of the candidates                           (select_kernel)
// Exactness: the k best group maxima are k distinct rows scoring >= thr, so thr is a
// lower bound of the true k-th best score and every true top-k row passes the filter
// (the sample rows are simply scanned again: n0/N extra traffic, 3 % at N = 1M).
// Expected candidate volume ~ k * N / n0 per query.  An overflow of the shared list
// (adversarial score order AND clustering) is reported through status_flag, never
// silently truncated.  Shards with fewer than ~4k groups take the dense path
// (every score written, exact selection).
#include <limits.h>
#include <string.h>

#include "common.h"
#include "scan.h"
#include "encoder.h"   // tt_scan_gemm_launch
#include "shadow.h"

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

constexpr int kCap = 16384;              // shared overflow-list slots per query
constexpr int64_t kSampleRows = 32768;   // default sample size (1024 group maxima per query)

struct Plan {
    bool gemm;         // more than 64 queries over a large shard: the filter pass is ONE tiled MFMA contraction per 256 queries
    int q256;          // queries rounded up to 256 (gemm path)
    int cap;           // shared candidate-list slots per query
    size_t off_q256;
    bool dense_only;
    int64_t n0;        // sample rows (dense_only: all rows)
    int64_t stride;    // floats per query in the dense / group-max buffer
    int qpad;
    int main_blocks;   // grid.x of the filter pass (fixes the private-list geometry)
    int n_sub;         // private sub-lists per query = main_blocks * waves * 2
    size_t off_dense, off_cs, off_ci, off_cnt, off_thr, off_priv, off_pcnt, total;
};

int filter_blocks(int64_t rows, int cus) {
    const int64_t groups = (rows + 31) / 32;
    const int64_t mb = (groups + TT_SCAN_WAVES_PER_BLOCK - 1) / TT_SCAN_WAVES_PER_BLOCK;
    int64_t b = cus > 0 ? cus : 256;
    if (b > mb) b = mb;
    if (b < 1) b = 1;
    return (int)b;
}

bool scan_gemm_enabled() {
    static const bool on = TT_DIAG_ENV_INT("TT_SCAN_GEMM", 1) != 0;
    return on;
}

Plan make_plan(int64_t n_rows, int dim, int n_queries, int k, int cus) {
    Plan pl{};
    pl.qpad = (n_queries + 63) / 64 * 64;
    pl.cap = kCap;
    // the sample needs >= 4k groups for a useful threshold
    int64_t n0 = kSampleRows;
    if (n0 < (int64_t)128 * k) n0 = (int64_t)128 * k;
    // More than one 64-query tile: the streaming kernel would re-read the shard once per tile.  From 65 queries on (and a
    // shard worth tiling) the filter pass runs as a 256-query-wide MFMA contraction instead: one pass over the rows per 256
    // queries.  It has no lane-private lists, so every survivor goes to the shared list: a larger sample (tighter threshold;
    // expected survivors ~ k * rows / sample rows per query) and a list sized for 4x that expectation.
    pl.gemm = scan_gemm_enabled() && n_queries > 64 && dim % 128 == 0 && n_rows >= 262144 && n_rows >= (int64_t)2048 * k;
    if (pl.gemm) {
        static const int64_t n0_env = TT_DIAG_ENV_INT("TT_SCAN_GEMM_N0", 0);
        n0 = n0_env > 0 ? n0_env : 131072;   // (measured, 10M x 1024 x 256 queries: 65536 -> 5.50, 131072 -> 5.35, 262144 -> 5.52 ms per batch)
        // ... in proportion to the shard: the sample pass and its selection are a fixed cost per batch (0.3 ms at 131072 rows),
        // a third of the whole batch on the 1.25M-row shard of an 8-GPU step; expected survivors per query ~ k * rows / n0 stay
        // at or below the 10M-row figure (1.25M rows: 32768 sample rows, 38 k survivors per query instead of 76 k)
        if (n0_env <= 0)
            while (n0 > 32768 && n0 * 64 > n_rows) n0 /= 2;
        if (n0 < (int64_t)512 * k) n0 = (int64_t)512 * k;
        if (n0 > n_rows / 2) n0 = n_rows / 2 / 32 * 32;
        n0 = n0 / 256 * 256;     // (the sample runs on the same 256-row tiles as the filter pass: tt_scan_gemm_sample_launch)
        pl.q256 = (n_queries + 255) / 256 * 256;
        pl.qpad = pl.q256;
        int64_t cap = 4 * (int64_t)k * n_rows / n0 + 4096;
        if (cap < kCap) cap = kCap;
        if (cap > (1 << 20)) cap = 1 << 20;
        pl.cap = (int)((cap + 63) / 64 * 64);
    }
    if (n_rows < (int64_t)128 * k || n_rows <= 8192) {
        pl.dense_only = true;
        pl.n0 = n_rows;
        pl.stride = (n_rows + 31) / 32 * 32;
    } else {
        pl.dense_only = false;
        pl.n0 = n0 < n_rows ? n0 : n_rows;
        pl.stride = ((pl.n0 + 31) / 32 + 31) / 32 * 32;
    }
    size_t off = 0;
    pl.off_dense = off; off += tt_align_up((size_t)pl.qpad * pl.stride * sizeof(float), 256);
    pl.off_cs = off;    off += tt_align_up((size_t)pl.qpad * pl.cap * sizeof(float), 256);
    pl.off_ci = off;    off += tt_align_up((size_t)pl.qpad * pl.cap * sizeof(int32_t), 256);
    pl.off_cnt = off;   off += tt_align_up((size_t)pl.qpad * sizeof(int32_t), 256);
    pl.off_thr = off;   off += tt_align_up((size_t)pl.qpad * sizeof(float), 256);
    if (pl.gemm) {
        pl.off_q256 = off; off += tt_align_up((size_t)pl.q256 * dim * sizeof(uint16_t), 256);
        pl.main_blocks = 1;                        // the < 256 tail rows go through the streaming kernel, one block
        pl.n_sub = TT_SCAN_WAVES_PER_BLOCK * 2;
        pl.off_priv = off; off += tt_align_up((size_t)pl.qpad * pl.n_sub * TT_SCAN_PRIV_SLOTS * sizeof(uint2), 256);
        pl.off_pcnt = off; off += tt_align_up((size_t)pl.qpad * pl.n_sub * sizeof(int32_t), 256);
    } else if (!pl.dense_only) {
        pl.main_blocks = filter_blocks(n_rows, cus);
        pl.n_sub = pl.main_blocks * TT_SCAN_WAVES_PER_BLOCK * 2;
        pl.off_priv = off; off += tt_align_up((size_t)pl.qpad * pl.n_sub * TT_SCAN_PRIV_SLOTS * sizeof(uint2), 256);
        pl.off_pcnt = off; off += tt_align_up((size_t)pl.qpad * pl.n_sub * sizeof(int32_t), 256);
    }
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

// thr[q] -= 2^-13 |thr[q]| + 2e-6: below anything two fp32 evaluations of the same K <= 1024 dot product of unit-norm bf16
// rows can differ by (see tt_scan_topk); -inf / +inf / NaN stay what they are
__global__ void relax_thr_kernel(float* thr, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float t = thr[i];
        if (t - t == 0.f) thr[i] = t - (fabsf(t) * 1.220703125e-4f + 2e-6f);
    }
}

__global__ void fill_f32_kernel(float* s, float v, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) s[i] = v;
}

__global__ void fill_pad_kernel(float* s, int32_t* ix, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        s[i] = -__builtin_inff();
        ix[i] = -1;
    }
}

int scan_mode_from_env() {
    // TT_SCAN_MODE = load path (0 direct fragments | 1 LDS transpose) + 16 * variant; default 1
    // (bench/ablation knob, see scan.hip).
#if TT_DIAG
    const char* e = getenv("TT_SCAN_MODE");
    if (e && e[0]) return atoi(e);
#endif
    return 1;  // LDS-transpose load path (non-temporal corpus loads)
}

// ---- fp8 shadow prefilter: block layout and workspace plan (shadow.hip) ----
struct ShadowLayout { size_t off_be, off_dn, total; };
ShadowLayout shadow_layout(int64_t cap_rows, int dim) {
    ShadowLayout l{};
    const size_t r16 = (size_t)(cap_rows + 15) / 16 * 16;
    size_t off = tt_align_up((size_t)cap_rows * dim, 256);
    l.off_be = off; off += tt_align_up(r16 * sizeof(float), 256);
    l.off_dn = off; off += tt_align_up(r16 * sizeof(float), 256);
    l.total = off;
    return l;
}
struct ShadowPlan {
    int64_t n0, stride;        // threshold sample: rows, floats per query
    int blocks, n_waves, capw, cap;
    size_t off_sample, off_ss, off_si, off_thr, off_planes, off_qinfo, off_list, off_wcnt, off_table, off_tcnt, off_dense, total;
};
ShadowPlan shadow_plan(int64_t n_rows, int dim, int n_queries, int k, int cus) {
    ShadowPlan pl{};
    // The threshold decides how many rows survive the shadow pass: the bound is ~0.04 for unit rows at 1024 dimensions against a
    // score spread of 1/32, so with the bf16 pass's 32 768-row sample (k-th best of it: 2.96 sigma at k = 50) 4.6 % of ALL rows would
    // survive -- 460 k rows to re-score at 10 M.  A sample of rows / 32 (3 % more bytes) puts the threshold at 3.6 sigma: ~1 % survive.
    int64_t n0 = kSampleRows;
    if (n0 < (int64_t)128 * k) n0 = (int64_t)128 * k;
    if (n0 < n_rows / 32) n0 = n_rows / 32 / 32 * 32;
    pl.n0 = n0 < n_rows ? n0 : n_rows;
    pl.stride = ((pl.n0 + 31) / 32 + 31) / 32 * 32;
    pl.blocks = cus > 0 ? cus : 256;
    const int64_t groups = (n_rows + 15) / 16;
    if ((int64_t)pl.blocks * kShadowWavesPerBlock > groups) pl.blocks = (int)((groups + kShadowWavesPerBlock - 1) / kShadowWavesPerBlock);
    if (pl.blocks < 1) pl.blocks = 1;
    pl.n_waves = pl.blocks * kShadowWavesPerBlock;
    // survivors: ~1 % of random unit rows at 10 M x 1024 (more on small shards, whose sample is small: 4 % at 1.3 M); sized for 6 % of the rows
    int64_t cap = n_rows / 16;
    if (cap < 65536) cap = 65536;
    if (cap > (1 << 20)) cap = 1 << 20;
    if (cap > (n_rows + 31) / 32 * 32) cap = (n_rows + 31) / 32 * 32;
    pl.cap = (int)((cap + 31) / 32 * 32);
    int64_t capw = 4 * (int64_t)pl.cap / pl.n_waves;
    if (capw < 64) capw = 64;
    pl.capw = (int)capw;
    const int qpad = 64;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += tt_align_up(bytes, 256); return o; };
    pl.off_sample = take((size_t)qpad * pl.stride * sizeof(float));
    pl.off_ss = take((size_t)qpad * k * sizeof(float));
    pl.off_si = take((size_t)qpad * k * sizeof(int32_t));
    pl.off_thr = take((size_t)qpad * sizeof(float));
    pl.off_planes = take((size_t)(dim / 128) * 4 * 1024);
    pl.off_qinfo = take(32 * sizeof(float));
    pl.off_list = take((size_t)n_queries * pl.n_waves * pl.capw * sizeof(int32_t));
    pl.off_wcnt = take((size_t)n_queries * pl.n_waves * sizeof(int32_t));
    pl.off_table = take((size_t)n_queries * pl.cap * sizeof(int32_t));
    pl.off_tcnt = take((size_t)qpad * sizeof(int32_t));
    pl.off_dense = take((size_t)pl.cap * sizeof(float));
    pl.total = off;
    return pl;
}
__global__ void add_idx_base_kernel(int32_t* idx, int n, int32_t base) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && idx[i] >= 0) idx[i] += base;
}
}  // namespace

extern "C" {

int tt_version(void) { return 1; }
const char* tt_arch(void) { return "gfx950"; }
const char* tt_last_error(void) { return g_err; }
int tt_device_cu_count(void) { return tt_cu_count_cached(); }

size_t tt_scan_workspace_bytes(int64_t n_rows, int dim, int n_queries, int k) {
    if (n_rows < 0 || n_queries <= 0 || k < 1) return 0;
    // sized for the largest grid the filter pass can use on any gfx950 part (256 CUs)
    int cus = tt_cu_count_cached();
    if (cus <= 0) cus = 256;
    return make_plan(n_rows, dim, n_queries, k, cus).total;
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
    int rc = tt_scan_launch(sp, dim, scan_mode_from_env(), 1, tt_cu_count_cached(), st);
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
    int cus_plan = tt_cu_count_cached();
    if (cus_plan <= 0) cus_plan = 256;
    const Plan pl = make_plan(n_rows, dim, n_queries, k, cus_plan);
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

    // (tiled path) the 256-query blocks the contraction reads: the batch itself when it is full, else a zero-padded copy
    const uint16_t* q256 = (const uint16_t*)queries_bf16;
    if (pl.gemm && n_queries != pl.q256) {
        uint16_t* qpad256 = (uint16_t*)(ws + pl.off_q256);
        TT_CHECK_HIP(hipMemsetAsync(qpad256 + (size_t)n_queries * dim, 0, (size_t)(pl.q256 - n_queries) * dim * sizeof(uint16_t), st));
        TT_CHECK_HIP(hipMemcpyAsync(qpad256, queries_bf16, (size_t)n_queries * dim * sizeof(uint16_t), hipMemcpyDeviceToDevice, st));
        q256 = qpad256;
    }
    // 1. sample: group maxima of rows [0, n0)
    ScanParams sp{};
    sp.corpus = (const uint16_t*)corpus_bf16;
    sp.queries = (const uint16_t*)queries_bf16;
    sp.row_lo = 0;
    sp.row_hi = pl.n0;
    sp.n_queries = n_queries;
    sp.idx_base = idx_base;
    sp.dense = dense;
    sp.dense_stride = pl.stride;
    // the n0 / 32 sampled groups are spread evenly over the shard (every group_stride-th 32-row group)
    sp.group_stride = (n_rows / 32) / ((pl.n0 + 31) / 32);
    sp.phys_rows = n_rows;
    // Tiled path (round 4): the sample runs on the filter pass's own contraction -- n0 / 256 row tiles spread evenly over the
    // shard, ONE read of the sample rows for all 256 queries and the same MFMA arithmetic as the pass that applies the
    // thresholds (the streaming sample kernel reads the rows once per 64-query tile: 63 of the shard batch's 730 us).
    // TT_SCAN_GEMM_SAMPLE=0: the streaming sample, the A/B switch.
    static const bool gemm_sample = TT_DIAG_ENV_INT("TT_SCAN_GEMM_SAMPLE", 1) != 0;
    if (pl.gemm && gemm_sample) {
        const int tiles = (int)(pl.n0 / 256);
        const int tile_stride = (int)((n_rows / 256) / tiles);
        for (int b = 0; b < pl.q256 / 256; ++b) {
            rc = tt_scan_gemm_sample_launch((const uint16_t*)corpus_bf16, tiles, tile_stride, dim, q256 + (size_t)b * 256 * dim, thr + b * 256,
                                            dense + (size_t)b * 256 * pl.stride, (int)pl.stride, st);
            if (rc) return rc;
        }
    } else {
        rc = tt_scan_launch(sp, dim, mode, 2, cus, st);
        if (rc) return rc;
    }

    // 2. thr[q] = k-th best group maximum (the selected rows themselves are re-found by
    //    the filter pass, so the outputs of this selection are scratch)
    SelectParams s1{};
    s1.scores = dense;
    s1.idx = nullptr;
    s1.stride = pl.stride;
    s1.cnt = nullptr;
    s1.m_fixed = (int)((pl.n0 + 31) / 32);
    s1.cap = INT_MAX;
    s1.idx_base = 0;
    s1.k = k;
    s1.out_scores = cs;
    s1.out_idx = ci;
    s1.out_stride = pl.cap;
    s1.thr_out = thr;
    s1.cnt_out = nullptr;
    // k <= 64: the launches around this selection are folded into it (one launch instead of four): it runs a block for every
    // slot of the padded query batch -- padding slots write thr = +inf (nothing passes) --, every block zeroes its candidate
    // counter, and the tiled path's threshold relaxation (below) is applied where thr is written
    const bool fused = tt_select_fused_outputs(k);
    if (fused) {
        s1.n_real = n_queries;
        s1.zero_cnt = cnt;
        s1.thr_relax = pl.gemm ? 1 : 0;
        rc = tt_select_launch(s1, pl.qpad, st);
        if (rc) return rc;
    } else {
        if (pl.gemm) {   // thresholds of the padding queries: +inf (nothing passes)
            hipLaunchKernelGGL(fill_f32_kernel, dim3((pl.q256 + 255) / 256), dim3(256), 0, st, thr, __builtin_inff(), pl.q256);
            TT_CHECK_LAUNCH();
        }
        rc = tt_select_launch(s1, n_queries, st);
        if (rc) return rc;
    }
    if (pl.gemm && !fused) {
        // The thresholds come from the streaming sample kernel (v_mfma_f32_32x32x16_bf16), the tiled filter pass recomputes the
        // scores on v_mfma_f32_16x16x32_bf16 with another K grouping and summation order: the two fp32 sums of one row are not
        // guaranteed to agree in the last bits, so the row that DEFINES thr could miss it by an ulp (k = 1 with the best row
        // inside the sample: fewer than k survivors).  Lower thr by more than the two sums can differ (each is within
        // K * 2^-24 * sum |q_i c_i| <= 6e-5 of the exact value for unit vectors); the handful of extra survivors are sorted
        // out by the exact selection.  The select after the filter pass also raises the status flag when a query ends with
        // fewer than min(k, rows) results (-> dense exact fallback in the caller).
        hipLaunchKernelGGL(relax_thr_kernel, dim3((n_queries + 255) / 256), dim3(256), 0, st, thr, n_queries);
        TT_CHECK_LAUNCH();
    }
    if (!fused) TT_CHECK_HIP(hipMemsetAsync(cnt, 0, (size_t)pl.qpad * sizeof(int32_t), st));

    if (pl.gemm) {
        // 3'. filter pass as a tiled contraction: ALL rows x 256 queries per launch, ONE pass over the corpus per 256 queries
        // (q256: the batch itself when it is full, else the zero-padded copy made before the sample)
        // (rows not a multiple of 256: the contraction's last tile is the shard's last 256 rows, overlapping its predecessor -- no
        //  separate pass over the tail.  Diagnostic library only, TT_SCAN_GEMM_TAIL=1: rounds 2-3's form, rows [0, rows256) tiled and
        //  the < 256 tail rows through the streaming kernel, the A/B switch)
        static const bool tail_pass = TT_DIAG_ENV_INT("TT_SCAN_GEMM_TAIL", 0) == 1;
        const int64_t rows256 = tail_pass ? n_rows / 256 * 256 : n_rows;
        for (int b = 0; b < pl.q256 / 256; ++b) {
            rc = tt_scan_gemm_launch((const uint16_t*)corpus_bf16, rows256, dim, q256 + (size_t)b * 256 * dim, thr + b * 256,
                                     cnt + b * 256, cs + (size_t)b * 256 * pl.cap, ci + (size_t)b * 256 * pl.cap, pl.cap, idx_base, st);
            if (rc) return rc;
        }
        SelectParams s2{};
        if (rows256 < n_rows) {
            ScanParams tp = sp;
            tp.group_stride = 0;
            tp.row_lo = rows256;
            tp.row_hi = n_rows;
            tp.dense = nullptr;
            tp.thr = thr;
            tp.cnt = cnt;
            tp.cand_scores = cs;
            tp.cand_idx = ci;
            tp.cap = pl.cap;
            tp.priv = (uint2*)(ws + pl.off_priv);
            tp.priv_cnt = (int32_t*)(ws + pl.off_pcnt);
            tp.prof_id = TT_K_SCAN_TAIL;
            rc = tt_scan_launch(tp, dim, mode, 0, 1, st);
            if (rc) return rc;
            s2.priv = tp.priv;
            s2.priv_cnt = tp.priv_cnt;
            s2.n_sub = pl.n_sub;
        }
        s2.scores = cs;
        s2.idx = ci;
        s2.stride = pl.cap;
        s2.cnt = cnt;
        s2.cap = pl.cap;
        s2.k = k;
        s2.out_scores = out_scores;
        s2.out_idx = out_idx;
        s2.out_stride = k;
        s2.overflow_flag = status_flag;
        s2.min_valid = (int)(n_rows < (int64_t)k ? n_rows : (int64_t)k);
        return tt_select_launch(s2, n_queries, st);
    }

    // 3. filter pass over ALL rows: scores >= thr[q] -> private lists (+ shared overflow list)
    ScanParams mp = sp;
    mp.group_stride = 0;
    mp.row_lo = 0;
    mp.row_hi = n_rows;
    mp.dense = nullptr;
    mp.thr = thr;
    mp.cnt = cnt;
    mp.cand_scores = cs;
    mp.cand_idx = ci;
    mp.cap = pl.cap;
    mp.priv = (uint2*)(ws + pl.off_priv);
    mp.priv_cnt = (int32_t*)(ws + pl.off_pcnt);
    rc = tt_scan_launch(mp, dim, mode, 0, pl.main_blocks, st);
    if (rc) return rc;

    // 4. exact top-k of the candidates
    SelectParams s2{};
    s2.scores = cs;
    s2.idx = ci;
    s2.stride = pl.cap;
    s2.cnt = cnt;
    s2.m_fixed = 0;
    s2.cap = pl.cap;
    s2.k = k;
    s2.out_scores = out_scores;
    s2.out_idx = out_idx;
    s2.out_stride = k;
    s2.overflow_flag = status_flag;
    s2.priv = mp.priv;
    s2.priv_cnt = mp.priv_cnt;
    s2.n_sub = pl.n_sub;
    return tt_select_launch(s2, n_queries, st);
}

// ---- fp8 shadow prefilter (shadow.hip; plan and layout helpers: the anonymous namespace above) ----
size_t tt_scan_shadow_bytes(int64_t cap_rows, int dim) {
    if (cap_rows <= 0 || dim <= 0 || dim % 128 || dim > 1024) return 0;
    return shadow_layout(cap_rows, dim).total;
}

int tt_scan_shadow_build(const void* corpus_bf16, int dim, int64_t row_lo, int64_t row_hi, void* shadow, int64_t cap_rows, void* stream) {
    TT_CHECK_ARG(dim > 0 && dim % 128 == 0 && dim <= 1024, "dim=%d must be a multiple of 128 and <= 1024", dim);
    TT_CHECK_ARG(row_lo >= 0 && row_lo <= row_hi && row_hi <= cap_rows, "rows [%lld, %lld) outside the shadow's capacity %lld", (long long)row_lo,
                 (long long)row_hi, (long long)cap_rows);
    if (row_hi == row_lo) return TT_OK;
    TT_CHECK_ARG(corpus_bf16 && shadow && ((uintptr_t)corpus_bf16 % 16) == 0 && ((uintptr_t)shadow % 256) == 0, "null or misaligned pointer");
    const ShadowLayout l = shadow_layout(cap_rows, dim);
    char* sb = (char*)shadow;
    return tt_shadow_build_launch((const uint16_t*)corpus_bf16 + (size_t)row_lo * dim, row_hi - row_lo, dim, (uint8_t*)sb + (size_t)row_lo * dim,
                                  (float*)(sb + l.off_be) + row_lo, (float*)(sb + l.off_dn) + row_lo, (hipStream_t)stream);
}

size_t tt_scan_shadow_workspace_bytes(int64_t n_rows, int dim, int n_queries, int k) {
    if (n_rows <= 0 || n_queries <= 0 || n_queries > 4 || k < 1 || dim <= 0 || dim % 128) return 0;
    int cus = tt_cu_count_cached();
    if (cus <= 0) cus = 256;
    return shadow_plan(n_rows, dim, n_queries, k, cus).total;
}

int tt_scan_topk_shadow(const void* corpus_bf16, const void* shadow, int64_t cap_rows, int64_t n_rows, int dim, const void* queries_bf16,
                        int n_queries, int k, int32_t idx_base, float* out_scores, int32_t* out_idx, void* workspace, size_t workspace_bytes,
                        int32_t* status_flag, void* stream) {
    int rc = check_common(corpus_bf16, n_rows, dim, queries_bf16, n_queries, k, out_scores, out_idx);
    if (rc) return rc;
    if (n_queries == 0) return TT_OK;
    TT_CHECK_ARG(n_queries <= 4, "n_queries=%d: the shadow prefilter serves lone callers (<= 4 queries per call)", n_queries);
    TT_CHECK_ARG(shadow && ((uintptr_t)shadow % 256) == 0 && n_rows <= cap_rows, "shadow missing, misaligned or smaller than n_rows");
    TT_CHECK_ARG(n_rows >= (int64_t)128 * k && n_rows > 8192, "n_rows=%lld: shards this small take tt_scan_topk's dense path", (long long)n_rows);
    hipStream_t st = (hipStream_t)stream;
    if (status_flag) TT_CHECK_HIP(hipMemsetAsync(status_flag, 0, sizeof(int32_t), st));
    int cus = tt_cu_count_cached();
    if (cus <= 0) cus = 256;
    const ShadowPlan pl = shadow_plan(n_rows, dim, n_queries, k, cus);
    if (!workspace || workspace_bytes < pl.total) {
        tt_set_error("tt_scan_topk_shadow: workspace %zu < required %zu bytes", workspace_bytes, pl.total);
        return TT_E_WORKSPACE;
    }
    TT_CHECK_ARG(((uintptr_t)workspace % 256) == 0, "workspace must be 256-byte aligned");
    char* ws = (char*)workspace;
    float* sample = (float*)(ws + pl.off_sample);
    float* thr = (float*)(ws + pl.off_thr);
    const int mode = scan_mode_from_env();
    // 1. exact thresholds: the bf16 pass's own sample (group maxima of n0 rows spread evenly over the shard) and selection
    ScanParams sp{};
    sp.corpus = (const uint16_t*)corpus_bf16;
    sp.queries = (const uint16_t*)queries_bf16;
    sp.row_lo = 0;
    sp.row_hi = pl.n0;
    sp.n_queries = n_queries;
    sp.dense = sample;
    sp.dense_stride = pl.stride;
    sp.group_stride = (n_rows / 32) / ((pl.n0 + 31) / 32);
    sp.phys_rows = n_rows;
    rc = tt_scan_launch(sp, dim, mode, 2, cus, st);
    if (rc) return rc;
    SelectParams s1{};
    s1.scores = sample;
    s1.stride = pl.stride;
    s1.m_fixed = (int)((pl.n0 + 31) / 32);
    s1.cap = INT_MAX;
    s1.k = k;
    s1.out_scores = (float*)(ws + pl.off_ss);
    s1.out_idx = (int32_t*)(ws + pl.off_si);
    s1.out_stride = k;
    s1.thr_out = thr;
    rc = tt_select_launch(s1, n_queries, st);
    if (rc) return rc;
    // 2. the queries as e4m3 planes; 3. one pass over the shadow: rows whose upper bound reaches thr; 4. one table per query
    const ShadowLayout l = shadow_layout(cap_rows, dim);
    const char* sb = (const char*)shadow;
    uint16_t* planes = (uint16_t*)(ws + pl.off_planes);
    float* qinfo = (float*)(ws + pl.off_qinfo);
    int32_t* list = (int32_t*)(ws + pl.off_list);
    int32_t* wcnt = (int32_t*)(ws + pl.off_wcnt);
    int32_t* table = (int32_t*)(ws + pl.off_table);
    int32_t* tcnt = (int32_t*)(ws + pl.off_tcnt);
    rc = tt_shadow_query_launch((const uint16_t*)queries_bf16, n_queries, dim, planes, qinfo, st);
    if (rc) return rc;
    rc = tt_shadow_filter_launch((const uint8_t*)sb, (const float*)(sb + l.off_be), (const float*)(sb + l.off_dn), n_rows, dim, planes, qinfo, thr,
                                 n_queries, pl.blocks, list, wcnt, pl.capw, status_flag, st);
    if (rc) return rc;
    rc = tt_shadow_compact_launch(list, wcnt, pl.n_waves, pl.capw, n_queries, table, pl.cap, tcnt, status_flag, st);
    if (rc) return rc;
    // 5. per query: the survivors' exact scores from the bf16 rows (the streaming kernel's fragments and MFMA order through a gather
    //    table: the same bits as tt_scan_topk), then the exact selection over (score, row)
    float* dense = (float*)(ws + pl.off_dense);
    for (int q = 0; q < n_queries; ++q) {
        ScanParams rp{};
        rp.corpus = (const uint16_t*)corpus_bf16;
        rp.queries = (const uint16_t*)queries_bf16 + (size_t)q * dim;
        rp.row_lo = 0;
        rp.row_hi = pl.cap;
        rp.n_queries = 1;
        rp.dense = dense;
        rp.dense_stride = pl.cap;
        rp.row_table = table + (size_t)q * pl.cap;
        rp.row_cnt = tcnt + q;
        rp.prof_id = TT_K_SCAN_TAIL;
        rc = tt_scan_launch(rp, dim, 0, 1, cus, st);
        if (rc) return rc;
        SelectParams s2{};
        s2.scores = dense;
        s2.idx = table + (size_t)q * pl.cap;
        s2.stride = pl.cap;
        s2.cnt = tcnt + q;
        s2.cap = pl.cap;
        s2.k = k;
        s2.out_scores = out_scores + (size_t)q * k;
        s2.out_idx = out_idx + (size_t)q * k;
        s2.out_stride = k;
        s2.overflow_flag = status_flag;
        rc = tt_select_launch(s2, 1, st);
        if (rc) return rc;
    }
    if (idx_base != 0) {
        const int n = n_queries * k;
        hipLaunchKernelGGL(add_idx_base_kernel, dim3((n + 255) / 256), dim3(256), 0, st, out_idx, n, idx_base);
        TT_CHECK_LAUNCH();
    }
    return TT_OK;
}

constexpr int64_t kPieceRows = 65536;   // longest row range one selection block walks (8 LDS rounds)

static size_t seg_dense_bytes(int64_t rows, int n_queries) {
    return tt_align_up((size_t)n_queries * (size_t)((rows + 31) / 32 * 32) * sizeof(float), 256);
}

size_t tt_scan_segmented_workspace_bytes(int64_t n_rows, int dim, int n_queries, int k) {
    (void)dim;
    if (n_rows < 0 || n_queries <= 0 || k < 1) return 0;
    // dense scores of the valid queries + per-piece partial top-k lists (long modules only)
    return seg_dense_bytes(n_rows, n_queries) + 2 * tt_align_up((size_t)n_queries * TT_SCAN_MAX_PIECES * k * 4, 256);
}

// One dense pass over rows [seg_offsets[0], seg_offsets[S]) (every score written once: Q * 4 bytes per
// row next to the 2 * dim bytes read) + ONE selection launch with a block per (query, module).  Modules
// longer than kPieceRows are cut into pieces (a block per piece) and a second launch merges the pieces.
int tt_scan_topk_segmented(const void* corpus_bf16, int64_t n_rows, int dim, const void* queries_bf16,
                           int n_queries, int k, const int64_t* seg_offsets, int n_segments, float* out_scores,
                           int32_t* out_idx, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_common(corpus_bf16, n_rows, dim, queries_bf16, n_queries, k, out_scores, out_idx);
    if (rc) return rc;
    TT_CHECK_ARG(n_segments >= 1 && n_segments <= TT_SCAN_MAX_SEGMENTS, "n_segments=%d outside [1,%d]", n_segments,
                 TT_SCAN_MAX_SEGMENTS);
    TT_CHECK_ARG(seg_offsets != nullptr, "null seg_offsets");
    TT_CHECK_ARG(seg_offsets[0] >= 0 && seg_offsets[n_segments] <= n_rows, "segment offsets outside [0, n_rows]");
    for (int s = 0; s < n_segments; ++s)
        TT_CHECK_ARG(seg_offsets[s] <= seg_offsets[s + 1], "segment offsets must be non-decreasing (segment %d)", s);
    if (n_queries == 0) return TT_OK;
    hipStream_t st = (hipStream_t)stream;
    const int64_t lo = seg_offsets[0], rows = seg_offsets[n_segments] - lo;
    if (rows == 0) {
        const int64_t n = (int64_t)n_queries * n_segments * k;
        hipLaunchKernelGGL(fill_pad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, out_scores, out_idx, n);
        TT_CHECK_LAUNCH();
        return TT_OK;
    }
    const size_t need = tt_scan_segmented_workspace_bytes(rows, dim, n_queries, k);
    if (!workspace || workspace_bytes < need) {
        tt_set_error("tt_scan_topk_segmented: workspace %zu < required %zu bytes", workspace_bytes, need);
        return TT_E_WORKSPACE;
    }
    TT_CHECK_ARG(((uintptr_t)workspace % 16) == 0, "workspace must be 16-byte aligned");
    const int64_t stride = (rows + 31) / 32 * 32;
    ScanParams sp{};
    sp.corpus = (const uint16_t*)corpus_bf16 + (size_t)lo * dim;
    sp.queries = (const uint16_t*)queries_bf16;
    sp.row_lo = 0;
    sp.row_hi = rows;
    sp.n_queries = n_queries;
    sp.dense = (float*)workspace;
    sp.dense_stride = stride;
    rc = tt_scan_launch(sp, dim, scan_mode_from_env(), 1, tt_cu_count_cached(), st);
    if (rc) return rc;

    // cut the modules into pieces of at most piece_rows rows, at most TT_SCAN_MAX_PIECES in all
    int64_t piece_rows = kPieceRows;
    const int64_t spare = TT_SCAN_MAX_PIECES - n_segments;
    if ((rows + piece_rows - 1) / piece_rows > spare) piece_rows = (rows + spare - 1) / spare;
    SelectParams se{};
    int first_piece[TT_SCAN_MAX_SEGMENTS + 1];
    int n_pieces = 0;
    for (int s = 0; s < n_segments; ++s) {
        first_piece[s] = n_pieces;
        const int64_t b = seg_offsets[s] - lo, e = seg_offsets[s + 1] - lo;
        int64_t pos = b;
        do {
            const int64_t end = (e - pos > piece_rows) ? pos + piece_rows : e;
            se.seg_off[n_pieces] = (int32_t)pos;
            se.seg_add[n_pieces] = (int32_t)(pos - b);
            ++n_pieces;
            pos = end;
        } while (pos < e);
    }
    first_piece[n_segments] = n_pieces;
    se.seg_off[n_pieces] = (int32_t)rows;
    // a piece ends where the next one starts, except the last piece of a module followed by a gap-free
    // neighbour -- which is the same thing: modules are contiguous in [lo, lo + rows)
    se.scores = sp.dense;
    se.stride = stride;
    se.cap = INT_MAX;
    se.k = k;
    se.out_stride = k;
    se.n_seg = n_pieces;
    if (n_pieces == n_segments) {
        se.out_scores = out_scores;
        se.out_idx = out_idx;
        return tt_select_launch(se, n_queries, st);
    }
    char* tmp = (char*)workspace + seg_dense_bytes(rows, n_queries);
    float* part_s = (float*)tmp;
    int32_t* part_i = (int32_t*)(tmp + tt_align_up((size_t)n_queries * TT_SCAN_MAX_PIECES * k * 4, 256));
    se.out_scores = part_s;
    se.out_idx = part_i;
    rc = tt_select_launch(se, n_queries, st);
    if (rc) return rc;
    SelectParams sm{};
    sm.scores = part_s;
    sm.idx = part_i;
    sm.stride = (int64_t)n_pieces * k;
    sm.cap = INT_MAX;
    sm.k = k;
    sm.out_scores = out_scores;
    sm.out_idx = out_idx;
    sm.out_stride = k;
    sm.n_seg = n_segments;
    for (int s = 0; s <= n_segments; ++s) sm.seg_off[s] = first_piece[s] * k;
    return tt_select_launch(sm, n_queries, st);
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
