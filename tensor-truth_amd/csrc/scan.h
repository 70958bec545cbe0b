// Internal interface between scan.hip (kernels), select.hip and scan_api.cpp.
#pragma once
#include "common.h"

struct ScanParams {
    const uint16_t* corpus;   // [N][D] bf16, shard base
    const uint16_t* queries;  // [Q][D] bf16
    int64_t row_lo, row_hi;   // rows of the shard this launch covers
    int n_queries;
    int32_t idx_base;         // emitted index = idx_base + shard row
    // out=1: every score -> dense[q * dense_stride + (row - row_lo)]; out=2: group maxima
    float* dense;
    int64_t dense_stride;     // >= round_up(row_hi - row_lo, 32), multiple of 4
    // out=0: scores >= thr[q] are appended to the candidate lists
    const float* thr;         // [Qpad]
    int32_t* cnt;             // [Qpad] running candidate counts (may exceed cap)
    float* cand_scores;       // [Qpad][cap]
    int32_t* cand_idx;        // [Qpad][cap]
    int cap;
    // private (atomic-free) candidate lists written by the filter pass:
    // priv[(q * n_sub + sub) * TT_SCAN_PRIV_SLOTS + j] = {score bits, index},
    // priv_cnt[q * n_sub + sub] = fill count, n_sub = grid.x * 16 (wave, lane half)
    uint2* priv;
    int32_t* priv_cnt;
    // out=2 only (threshold sample): group g of the launch is physical 32-row group g * group_stride, so the sample is
    // spread evenly over the shard instead of being its first rows (corpora are ingested document by document: the
    // first rows are one topic).  0 / 1 = contiguous.  phys_rows = rows of the shard (clamp for the last group).
    int64_t group_stride;
    int64_t phys_rows;
    int prof_id;              // 0 = TT_K_SCAN_FILTER / TT_K_SCAN_SAMPLE by `out`; else the timing id of this launch
    // mode 0 only (round 6, shadow.hip: exact re-scoring of the fp8 prefilter's survivors): launch row i is PHYSICAL row
    // row_table[i] of `corpus` (a gather by index, same fragments and MFMA order as a contiguous scan: the same score bits);
    // row_cnt (device) = the number of table entries, read by the kernel -- row_hi is then only the launch's upper bound
    const int32_t* row_table;
    const int32_t* row_cnt;
};

#define TT_SCAN_PRIV_SLOTS 16
#define TT_SCAN_WAVES_PER_BLOCK 8
#define TT_SCAN_MAX_SEGMENTS 64   // index modules per tt_scan_topk_segmented call
#define TT_SCAN_MAX_PIECES 256    // selection blocks per query (long modules are cut into pieces)

// mode 0: A fragments loaded directly from global; mode 1: full-line loads
// transposed through wave-private LDS.  blocks = grid.x (clamped to the work).
// out: 0 = threshold filter into the candidate lists, 1 = dense scores,
// 2 = one max per (32-row group, query) into dense[q * dense_stride + group].
int tt_scan_launch(const ScanParams& p, int dim, int mode, int out, int blocks, hipStream_t stream);

struct SelectParams {
    const float* scores;      // per query: scores + q * stride
    const int32_t* idx;       // per query: idx + q * stride; NULL => implicit position + idx_base
    int64_t stride;
    const int32_t* cnt;       // per-query candidate count (device), NULL => m_fixed
    int m_fixed;
    int cap;                  // cnt is clamped to cap; cnt > cap raises *overflow_flag
    int32_t idx_base;         // added to implicit indices only
    int k;
    float* out_scores;        // [Q][out_stride]
    int32_t* out_idx;         // [Q][out_stride]
    int64_t out_stride;
    // optional second source: private sub-lists of the filter pass (see ScanParams)
    const uint2* priv;
    const int32_t* priv_cnt;
    int n_sub;
    float* thr_out;           // optional [Q]: k-th best score (or -inf if fewer than k valid)
    int32_t* cnt_out;         // optional [Q]: number of valid outputs (<= k)
    int32_t* overflow_flag;   // optional
    int min_valid;            // > 0: fewer than min_valid valid outputs for a query also raise *overflow_flag
    // k <= 64 kernel only -- the small launches around the threshold selection of a tiled filter batch, folded into it:
    int thr_relax;            // thr_out is lowered by 2^-13 |thr| + 2e-6 (see scan_api.hip: two summation orders of one dot product)
    int32_t* zero_cnt;        // optional [grid.x]: zero_cnt[q] = 0 (the filter pass's candidate counters)
    int n_real;               // 0 = every block has a list; else blocks q >= n_real only write thr_out[q] = +inf, zero_cnt[q] = 0
    // optional segmentation of each query's list (tt_scan_topk_segmented): block (q, s) selects over
    // positions [seg_off[s], seg_off[s+1]) of query q's scores (and idx, when given), writes output row
    // q * n_seg + s; implicit indices are emitted as (position in the segment) + seg_add[s].
    // n_seg == 0: one block per query over m_fixed / cnt entries.
    int n_seg;
    int32_t seg_off[TT_SCAN_MAX_PIECES + 1];
    int32_t seg_add[TT_SCAN_MAX_PIECES];
};

int tt_select_launch(const SelectParams& p, int n_queries, hipStream_t stream);
// true when a selection with this k runs on the k <= 64 kernel, i.e. thr_relax / zero_cnt / n_real are available
bool tt_select_fused_outputs(int k);
