/* CPU oracle (plain C): exact similarity scan + top-k.  TEST INFRASTRUCTURE ONLY.
 *
 * Independent restatement of oracle/scan.py used to cross-check it (and, built
 * with -fopenmp, available as a scalar "port" CPU baseline).  Restates the
 * vector search behind VectorIndexRetriever (reference:
 * src/tensortruth/rag_engine.py:628-639; upstream chromadb/hnswlib is
 * approximate -- BASELINE.json asks for the exact brute-force scan):
 *   score[q][n] = sum_d bf16(Q[q][d]) * bf16(C[n][d])   (fp32 accumulate, d ascending)
 *   top-K per query ordered by (score desc, row index asc); padding = (-inf, -1).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline float bf16_to_f32(uint16_t h) {
    uint32_t u = ((uint32_t)h) << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

/* 1 if (sa, ia) ranks strictly before (sb, ib). */
static inline int before(float sa, int32_t ia, float sb, int32_t ib) {
    if (sa > sb) return 1;
    if (sa < sb) return 0;
    return ia < ib;
}

int tt_oracle_scan_topk(const uint16_t* corpus, int64_t n_rows, int dim,
                        const uint16_t* queries, int n_q, int k,
                        float* out_scores, int32_t* out_idx) {
    if (dim <= 0 || k <= 0 || n_q < 0 || n_rows < 0) return -1;
    float* qf = (float*)malloc(sizeof(float) * (size_t)dim);
    if (!qf) return -2;
    for (int q = 0; q < n_q; ++q) {
        float* vs = out_scores + (size_t)q * k;
        int32_t* is = out_idx + (size_t)q * k;
        int filled = 0;
        for (int j = 0; j < k; ++j) { vs[j] = -INFINITY; is[j] = -1; }
        for (int d = 0; d < dim; ++d) qf[d] = bf16_to_f32(queries[(size_t)q * dim + d]);
        for (int64_t n = 0; n < n_rows; ++n) {
            const uint16_t* row = corpus + (size_t)n * dim;
            float acc = 0.0f;
            for (int d = 0; d < dim; ++d) acc += qf[d] * bf16_to_f32(row[d]);
            if (acc != acc) continue; /* NaN never ranks */
            if (filled == k && !before(acc, (int32_t)n, vs[k - 1], is[k - 1])) continue;
            int pos = filled < k ? filled : k - 1;
            while (pos > 0 && before(acc, (int32_t)n, vs[pos - 1], is[pos - 1])) {
                vs[pos] = vs[pos - 1];
                is[pos] = is[pos - 1];
                --pos;
            }
            vs[pos] = acc;
            is[pos] = (int32_t)n;
            if (filled < k) ++filled;
        }
    }
    free(qf);
    return 0;
}
