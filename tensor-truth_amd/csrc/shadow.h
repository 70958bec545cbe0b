// Internal interface of the fp8 shadow prefilter (shadow.hip) towards scan_api.hip.
#pragma once
#include "common.h"

constexpr int kShadowWavesPerBlock = 8;
int tt_shadow_build_launch(const uint16_t* corpus, int64_t n_rows, int dim, uint8_t* shadow, float* be, float* dn, hipStream_t st);
// frags: (dim / 128) * 4 * 1024 bytes; qinfo: 16 floats
int tt_shadow_query_launch(const uint16_t* queries, int n_queries, int dim, uint16_t* frags, float* qinfo, hipStream_t st);
// list [n_queries][blocks * 8][capw], wave_cnt [n_queries][blocks * 8]
int tt_shadow_filter_launch(const uint8_t* shadow, const float* be, const float* dn, int64_t n_rows, int dim, const uint16_t* frags,
                            const float* qinfo, const float* thr, int n_queries, int blocks, int32_t* list, int32_t* wave_cnt, int capw,
                            int32_t* status_flag, hipStream_t st);
int tt_shadow_compact_launch(const int32_t* list, const int32_t* wave_cnt, int n_waves, int capw, int n_queries, int32_t* table, int cap,
                             int32_t* table_cnt, int32_t* status_flag, hipStream_t st);
