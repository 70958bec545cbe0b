// Variable-length bidirectional self-attention (flash-style, online softmax), gfx950 MFMA.
//
//   ctx[t][h*dh + d] = sum_k softmax_k(Q[t].K[k] / sqrt(dh)) V[k][d]   over the keys of t's sequence
//
// Replaces the scaled-dot-product attention inside the reference's XLM-R / BERT encoder
// layers (additive -inf key-padding mask == attending only to the sequence's own tokens
// in the packed varlen layout).
//
// One workgroup = 4 waves (one per SIMD: with 5 the fifth lands on SIMD 0 again and caps the CU at two
// workgroups) = 128 query rows of one (sequence, head); keys stream through LDS in tiles of 64, staged by
// LDS-DMA (global_load_lds_dwordx4) into two buffers: the copies of tile t+1 are in flight while tile t is
// computed and there is ONE barrier per tile.  Both products run "swapped" on v_mfma_f32_32x32x16_bf16 so
// that the QUERY sits on the lane for the whole kernel:
//   S^T[key][q] = K . Q^T      A = K fragment (ds_read_b128, XOR-swizzled rows), B = Q fragment (registers)
//   O^T[d][q]  += V^T . P^T    A = V^T fragment (one ds_read_b128), B = the S^T accumulator itself,
//                              converted to bf16 in place
// so the softmax row statistics are lane-local (one cross-half exchange per tile) and P never goes through
// LDS.  The K rows are fed to the first product in a permuted order (row bits 2 and 3 swapped) so that the
// eight accumulator registers a lane converts into one P fragment are eight CONSECUTIVE keys: exactly one
// 16-byte piece of the token-blocked V copy the QKV GEMM epilogue wrote (V8: [token/8][feature][8]), which
// therefore goes global -> LDS -> MFMA operand without any transposition or padding.
#include "common.h"
#include "encoder.h"
#include "f16c.h"

namespace {

#ifndef TT_ATT_RESIDENT_DEFAULT
#define TT_ATT_RESIDENT_DEFAULT 0        // measured neutral (-2 % at 292 tokens, -13 % at 200, +3 % at 64 / 130): profiles/r05_attention_resident_ab.log
#endif
#ifndef TT_ATT_RESIDENT_MIN_LEN
#define TT_ATT_RESIDENT_MIN_LEN 1
#endif
constexpr int kKTile = 64;    // keys per LDS tile
constexpr int kWaves = 4;     // 128 query rows per workgroup, one wave per SIMD

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// LDS reads hidden from the compiler's LDS-DMA tracking (it would put s_waitcnt vmcnt(0) in front of every
// ds_read while the next tile's copies are in flight); lds_wait4 is the matching lgkmcnt(0).
template <int OFF>
__device__ __forceinline__ u32x4 lds_read128_async(uint32_t addr) {
    u32x4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
// wait until at most N LDS reads of this wave are still in flight (they return in order)
template <int N>
__device__ __forceinline__ void lds_wait4n(u32x4& a, u32x4& b, u32x4& c, u32x4& d) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}
template <int N>
__device__ __forceinline__ void lds_wait2n(u32x4& a, u32x4& b) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}

// ABL (diagnostic builds, TT_ATT_ABLATE, wrong results by design -- what each part of the key-tile loop costs): 1 = no
// exponentials (the scaled difference goes on as the "probability"), 2 = no softmax at all (no mask, maximum, FMAs,
// exponentials, sums or rescale: the raw scores are packed as P), 3 = 2 without the V reads and the P.V MFMAs,
// 4 = everything, but every workgroup reads the K / V rows of sequence 0 (L2 hits: what the K / V misses cost)
// NW (round 4): waves = 32-row query blocks per workgroup.  4 is the shape of rounds 1-3; 8 (256 rows) is taken where a sequence's
// last workgroup stays mostly full (tt_attention_launch: the measured rule); 5 exists as an experiment (uneven piece map).  A
// row's arithmetic does not depend on NW.
template <int DH, bool STAMP = false, int ABL = 0, int NW = kWaves>
__global__ __launch_bounds__(64 * NW, 4) void attention_kernel(AttnParams p) {
    constexpr int RB = DH * 2;              // bytes per K row
    constexpr int CH = RB / 16;             // 16-B chunks per K row
    constexpr int RPB = 256 / RB;           // K rows per 256-B bank row
    constexpr int KS = DH / 16;             // k-steps of Q.K
    constexpr int DT = DH / 32;             // 32-row d tiles of O^T
    constexpr int NPK = kKTile * RB / 1024; // 1-KiB copy pieces of a K tile
    constexpr int NPV = 8 * DH * 16 / 1024; // ... of a V tile (8 token groups x DH features x 16 B)
    constexpr bool kEvenPieces = NPK % NW == 0 && NPV % NW == 0;     // NW = 5: piece j of the tile's NPK + NPV goes to wave j % NW
    constexpr int KPW = kEvenPieces ? NPK / NW : 1, VPW = kEvenPieces ? NPV / NW : 1;   // pieces per wave: NW = 4: 2 + 2 (dh 64), 1 + 1 (dh 32); NW = 8: 1 + 1
    static_assert(kEvenPieces || (DH == 64 && NPK == 8 && NPV == 8), "the uneven piece map is written for head_dim 64");
    constexpr int BUF = (NPK + NPV) * 1024;
    // Two buffers: tile kt + 1 is copied while tile kt is computed.  (Three -- copies two tiles ahead, a copy needs about 3 us
    // to land under load and a tile 2-2.5 us to compute -- cost a workgroup per CU at 48 KiB each and measured 20 % slower,
    // 1.62 vs 1.35 ms at 1600 x 292 tokens and at every other length tried: resident waves hide more than the deeper prefetch.)
    __shared__ __attribute__((aligned(1024))) char lds[2 * BUF];

    // Workgroup -> (query tile, head, sequence).  The query tiles of one (sequence, head) read the same K / V rows, and
    // every XCD has its own L2: in the plain order (tile fastest over all workgroups) the tiles of a pair are dealt
    // round-robin to DIFFERENT XCDs and each fetches K and V again (3.4 GB instead of 1.4 GB per launch at 800 x 292).
    // XCD-aware order (TT_ATT_XCD=1): linear id L goes to XCD L % 8, so slot t = L / 8 of an XCD walks (pair, query tile)
    // with the tile fastest -- the tiles of a pair run back to back on one XCD and share its L2.
    // diagnostic build (tools/att_stamps): waves 0..3 of ONE workgroup (the one in the middle of the grid) stamp their phases
    int st_i = 0;
    auto stamp = [&]() {
        if constexpr (STAMP) {
            if (p.dbg && blockIdx.x == gridDim.x / 2 && (threadIdx.x & 63) == 0 && st_i < 60) {
                __builtin_amdgcn_sched_barrier(0);
                p.dbg[(threadIdx.x >> 6) * 64 + st_i] = __builtin_amdgcn_s_memtime();
                __builtin_amdgcn_sched_barrier(0);
            }
            ++st_i;
        }
    };
    stamp();                                   // 0: entry
    const int nqt = p.n_qt > 0 ? p.n_qt : -p.n_qt;
    const int L = blockIdx.x, t = L >> 3;
    // (n_qt < 0: TT_ATT_XCD=0, the plain order -- tile fastest over ALL workgroups -- kept as the A/B switch)
    const int qt = p.n_qt > 0 ? t % nqt : L % nqt;
    const int pair = p.n_qt > 0 ? (t / nqt) * 8 + (L & 7) : L / nqt;
    if (pair >= p.heads * p.n_seq) return;
    const int head = pair % p.heads, seq = pair / p.heads;
    const int len = p.seq_len[seq];
    if (qt * 32 * NW >= len) return;
    const int t0 = p.seq_start[seq];
    // Sequences may start at any row.  Keys are walked in the ALIGNED frame of the V8 token groups: aligned key ka is
    // global row t0a + ka, t0a = t0 rounded down to 8; the off = t0 - t0a rows in front of the sequence (the tail of its
    // predecessor) and everything from off + len on are masked.
    const int t0a = t0 & ~7, off = t0 - t0a, alen = off + len;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & 31, hh = lane >> 5;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;

    // Which of the tile's four 32-row blocks this wave computes.  A sequence's LAST tile usually has fewer than four live blocks
    // (292 tokens: 36 rows = 2 of 4), and in wave order they always land on the same SIMDs (wave w of every workgroup runs on
    // SIMD w): SIMDs 0-1 carry every tail's work, SIMDs 2-3 idle through it.  TT_ATT_ROTATE=1 deals the live blocks to different
    // waves per (pair x live blocks); a row's arithmetic does not depend on the wave that runs it.  MEASURED NEUTRAL (1.291 vs
    // 1.290 ms at 1600 x 292 tokens, three rounds; 160 / 200 / 420 tokens within 2 %: profiles/r04_attention_tail_rotation_ab.log)
    // -- the kernel is not SIMD-issue bound: a key tile costs one LDS-DMA round trip whatever is computed in it (round 3's
    // ablations), so what a tail costs is its workgroup's walk over the key tiles, not the lanes it keeps busy.  Off by default.
    int wslot = wave;
    if (p.rotate && NW == kWaves) {
        const int left = len - qt * 32 * kWaves;
        const int n_live = left >= 32 * kWaves ? kWaves : (left + 31) >> 5;
        if (n_live < kWaves) wslot = (wave + ((pair * n_live) & (kWaves - 1))) & (kWaves - 1);
    }
    // ---- Q fragments (B operand), straight from global ---------------------------------
    const int q_row = (qt * NW + wslot) * 32 + ql;         // row inside the sequence
    const int q_row_c = q_row < len ? q_row : len - 1;     // clamp: result discarded
    const uint16_t* qp = p.qk + (size_t)(t0 + q_row_c) * p.ld_qk + p.q_col0 + head * DH + hh * 8;
    ex8 qf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) qf[s] = *reinterpret_cast<const ex8*>(qp + s * 16);

    const int n_kt = (alen + kKTile - 1) / kKTile;
    const int n_g8 = (alen + 7) >> 3;
    // ---- staging: wave w copies K pieces w*KPW.. and V pieces w*VPW.. of each tile.  Per-lane row / token-group
    // indices are loop constants; rows / groups beyond the sequence are clamped to its last one (finite values;
    // their probabilities are 0) -- only the last tile can need that.
    const int t0kv = (ABL == 4) ? (p.seq_start[0] & ~7) : t0a;
    const uint16_t* kbase = p.qk + (size_t)t0kv * p.ld_qk + p.k_col0 + head * DH;
    const uint16_t* vbase = p.vt + (size_t)(t0kv >> 3) * p.ldvt + (size_t)head * DH * 8;
    // Copies in the SGPR-base form (wave-uniform 64-bit base + per-lane 32-bit byte offset, as the GEMM's glds16), with ONE
    // per-lane offset register per operand.  History: through the builtin every piece carried a per-lane 64-bit address
    // (v_mad_i64_i32 + v_lshl_add_u64 and a VGPR pair per piece and tile); with per-piece offset arrays kept across the loop
    // the kernel, which sits at its 128-VGPR cap, spilled two of them and reloaded them INSIDE the key loop behind an
    // s_waitcnt vmcnt(0) -- a drain of the copy queue per tile, 6 % of the bench shape's time.  Piece i of a wave is piece 0
    // moved by 64 / CH rows (K) or 64 / DH token groups (V): a scalar step of the BASE; the K swizzle (r / RPB) & (CH - 1)
    // advances by 4 per piece, i.e. toggles bit 2 of the chunk index (byte offset ^ 64) for odd i when CH = 8.
    constexpr int kRowsPerPiece = 64 / CH, kGroupsPerPiece = 64 / DH > 0 ? 64 / DH : 1;
    static_assert(KPW == 1 || CH == 8, "piece-to-piece swizzle step is written for 128-byte K rows");
    static_assert(VPW == 1 || DH == 64, "one token group per V piece");
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    uint32_t kvoff0, vvoff0;             // byte offsets of this lane inside the wave's piece 0 of an unclamped tile
    {
        // (uneven map: the lane offsets of K piece 0 / V piece 0 -- the piece index is in the scalar base and, for K, in bit 2 of
        //  the chunk index of odd pieces, as below)
        const int e = (kEvenPieces ? wave * KPW * 64 : 0) + lane, r = e / CH, pos = e % CH;
        kvoff0 = ((uint32_t)r * (uint32_t)p.ld_qk + (uint32_t)((pos ^ ((r / RPB) & (CH - 1))) << 3)) * 2u;
        const int ev = (kEvenPieces ? wave * VPW * 64 : 0) + lane;
        vvoff0 = ((uint32_t)(ev / DH) * (uint32_t)p.ldvt + (uint32_t)((ev % DH) * 8)) * 2u;
    }
    auto sbase = [](const void* ptr) {      // keep a wave-uniform pointer in SGPRs
        const unsigned long long b = reinterpret_cast<unsigned long long>(ptr);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
        return reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
    };
    auto glds16 = [](const char* base, uint32_t voff, uint32_t lds_addr) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds_addr) : "memory", "m0");
    };
    auto issue_tile = [&](int kt) {
        const uint32_t buf = lds_base + (uint32_t)(kt & 1) * BUF;
        const bool clamp = (kt + 1) * kKTile > alen || (kt == 0 && off != 0);   // wave-uniform
        if (!clamp) {
            if constexpr (kEvenPieces) {
#pragma unroll
                for (int i = 0; i < KPW; ++i)
                    glds16(sbase(kbase + ((size_t)kt * kKTile + i * kRowsPerPiece) * p.ld_qk), (i & 1) ? (kvoff0 ^ 64u) : kvoff0,
                           buf + (uint32_t)(wave * KPW + i) * 1024u);
#pragma unroll
                for (int i = 0; i < VPW; ++i)
                    glds16(sbase(vbase + ((size_t)kt * 8 + i * kGroupsPerPiece) * p.ldvt), vvoff0, buf + (uint32_t)(NPK + wave * VPW + i) * 1024u);
            } else {
                for (int j = wave; j < NPK + NPV; j += NW) {          // wave-uniform: 3 or 4 pieces
                    if (j < NPK)
                        glds16(sbase(kbase + ((size_t)kt * kKTile + j * kRowsPerPiece) * p.ld_qk), (j & 1) ? (kvoff0 ^ 64u) : kvoff0,
                               buf + (uint32_t)j * 1024u);
                    else
                        glds16(sbase(vbase + ((size_t)kt * 8 + (j - NPK) * kGroupsPerPiece) * p.ldvt), vvoff0, buf + (uint32_t)j * 1024u);
                }
            }
            return;
        }
        // first / last tile: rows / token groups outside the sequence are clamped to its nearest one (finite values; their
        // probabilities are 0).  The per-lane indices are recomputed here instead of living in registers across the loop.
        const char* kb = sbase(kbase);
        const char* vb = sbase(vbase);
        int lane_c = lane;
        asm volatile("" : "+v"(lane_c));
        auto clamped_k = [&](int piece) {
            const int e = piece * 64 + lane_c, r = e / CH, pos = e % CH;
            int row = kt * kKTile + r;
            row = row < off ? off : (row < alen ? row : alen - 1);               // rows of this sequence only
            glds16(kb, ((uint32_t)row * (uint32_t)p.ld_qk + (uint32_t)((pos ^ ((r / RPB) & (CH - 1))) << 3)) * 2u,
                   buf + (uint32_t)piece * 1024u);
        };
        auto clamped_v = [&](int piece) {
            const int e = piece * 64 + lane_c;
            int g8 = kt * 8 + e / DH;
            g8 = g8 < n_g8 ? g8 : n_g8 - 1;
            glds16(vb, ((uint32_t)g8 * (uint32_t)p.ldvt + (uint32_t)((e % DH) * 8)) * 2u, buf + (uint32_t)(NPK + piece) * 1024u);
        };
        if constexpr (kEvenPieces) {
#pragma unroll
            for (int i = 0; i < KPW; ++i) clamped_k(wave * KPW + i);
#pragma unroll
            for (int i = 0; i < VPW; ++i) clamped_v(wave * VPW + i);
        } else {
            for (int j = wave; j < NPK + NPV; j += NW) {
                if (j < NPK) clamped_k(j);
                else clamped_v(j - NPK);
            }
        }
    };

    f32x16 acc_o[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[d][r] = 0.f;
    float m_run = -__builtin_inff();
    float l_run = 0.f;
    const float sc = p.scale * 1.4426950408889634f;  // fold log2(e): softmax via exp2
    const bool wave_active = (qt * NW + wslot) * 32 < len;      // wave-uniform; idle waves only help staging

    // per-lane LDS offsets: K row perm(ql) (bits 2 and 3 of the row swapped), chunk (2s + hh) ^ swizzle(row)
    const int krow = (ql & 0x13) | ((ql & 4) << 1) | ((ql & 8) >> 1);
    uint32_t koff[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) koff[s] = lds0 + krow * RB + (((2 * s + hh) ^ ((krow / RPB) & (CH - 1))) << 4);
    const uint32_t voff = lds0 + NPK * 1024 + (hh * DH + ql) * 16;    // + (4j + 2 s2) * DH*16 + 32*dt*16

    issue_tile(0);
    stamp();                                   // 1: Q loads + tile 0 copies issued
    for (int kt = 0; kt < n_kt; ++kt) {
        const int k0 = kt * kKTile;
        stamp();                               // 2 + 6 kt: tile top
        // This wave's pieces of tile kt have landed.  As the BUILTIN, not inline asm: the waitcnt pass must see a vmcnt(0)
        // on every path into the loop -- the Q fragments are ordinary global loads of the prologue, and otherwise it protects
        // their first use in EVERY iteration with s_waitcnt vmcnt(3..0), which at run time also counts the copies of tile
        // kt + 1 issued a moment earlier: a full drain of the prefetch per tile (the double buffer was single-buffered in
        // effect; -2...4 % on top of the spill fix above).
        __builtin_amdgcn_s_waitcnt(0x0F70);
        asm volatile("" ::: "memory");
        stamp();                               // +1: own copies landed
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                       // ... everyone's; and tile kt-1 is no longer read
        __builtin_amdgcn_sched_barrier(0);
        stamp();                               // +2: past the barrier
        if (kt + 1 < n_kt) issue_tile(kt + 1);
        if (!wave_active) continue;
        const uint32_t bufo = (kt & 1) * BUF;

        // ---- S^T = K . Q^T for two 32-key tiles -------------------------------------------------
        f32x16 acc_s[2];
        {
            u32x4 kf[2][4];
            uint32_t ka[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) ka[s] = koff[s] + bufo;
#pragma unroll
            for (int s = 0; s < KS; ++s) kf[0][s] = lds_read128_async<0>(ka[s]);
#pragma unroll
            for (int s = 0; s < KS; ++s) kf[1][s] = lds_read128_async<32 * RB>(ka[s]);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (j == 0) {
                    if constexpr (KS == 4) lds_wait4n<KS>(kf[0][0], kf[0][1], kf[0][2], kf[0][3]);
                    else lds_wait2n<KS>(kf[0][0], kf[0][1]);
                } else {
                    if constexpr (KS == 4) lds_wait4n<0>(kf[1][0], kf[1][1], kf[1][2], kf[1][3]);
                    else lds_wait2n<0>(kf[1][0], kf[1][1]);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) acc_s[j][r] = 0.f;
#pragma unroll
                for (int s = 0; s < KS; ++s)
                    acc_s[j] = TT_MFMA_32x32x16(__builtin_bit_cast(ex8, kf[j][s]), qf[s], acc_s[j]);
            }
        }
        if constexpr (STAMP) { asm volatile("" :: "v"(acc_s[0][0]), "v"(acc_s[1][15])); }
        stamp();                               // +3: S = K.Q^T issued (results consumed next)
        if constexpr (ABL == 3) {   // S only: keep the 32 scores alive, nothing else happens in this tile
            asm volatile("" :: "v"(acc_s[0][0]), "v"(acc_s[0][1]), "v"(acc_s[0][2]), "v"(acc_s[0][3]), "v"(acc_s[0][4]), "v"(acc_s[0][5]),
                         "v"(acc_s[0][6]), "v"(acc_s[0][7]), "v"(acc_s[0][8]), "v"(acc_s[0][9]), "v"(acc_s[0][10]), "v"(acc_s[0][11]),
                         "v"(acc_s[0][12]), "v"(acc_s[0][13]), "v"(acc_s[0][14]), "v"(acc_s[0][15]));
            asm volatile("" :: "v"(acc_s[1][0]), "v"(acc_s[1][1]), "v"(acc_s[1][2]), "v"(acc_s[1][3]), "v"(acc_s[1][4]), "v"(acc_s[1][5]),
                         "v"(acc_s[1][6]), "v"(acc_s[1][7]), "v"(acc_s[1][8]), "v"(acc_s[1][9]), "v"(acc_s[1][10]), "v"(acc_s[1][11]),
                         "v"(acc_s[1][12]), "v"(acc_s[1][13]), "v"(acc_s[1][14]), "v"(acc_s[1][15]));
            l_run = 1.f;
            continue;
        }
        // V fragments of the first 32 keys: in flight during the softmax
        u32x4 vf[2][2];   // [s2][dt]
        const uint32_t vaddr = voff + bufo;
        vf[0][0] = lds_read128_async<0>(vaddr);
        if constexpr (DT == 2) vf[0][1] = lds_read128_async<512>(vaddr);
        vf[1][0] = lds_read128_async<2 * DH * 16>(vaddr);
        if constexpr (DT == 2) vf[1][1] = lds_read128_async<2 * DH * 16 + 512>(vaddr);
        if constexpr (ABL == 2) l_run = 1.f;
        if constexpr (ABL != 2) {

        // ---- mask the tail, running max, exponentials --------------------------------------------
        // The softmax scale (and log2 e) is folded into one FMA per score: p = 2^(s*sc - m), with m
        // tracked in the scaled domain; v_exp_f32 is used raw (arguments are <= 0, a result that
        // underflows is 0 either way), the libm exp2f wraps it in 5 more instructions per value.
        // Register r of tile j is key k0 + 32 j + 16 (r>>3) + 8 hh + (r & 7)  (permuted K rows).
        if (kt == 0 && off != 0) {   // wave-uniform: the first off (< 8) aligned keys belong to the previous sequence
            if (hh == 0) {
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    if (r < off) acc_s[0][r] = -__builtin_inff();
            }
        }
        if (k0 + kKTile > alen) {   // wave-uniform; one compare per value against a lane constant, no index arithmetic
            const int lim = alen - k0 - 8 * hh;   // register r of tile j is masked iff 32 j + 16 (r>>3) + (r&7) >= lim
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (32 * j + 16 * (r >> 3) + (r & 7) >= lim) acc_s[j][r] = -__builtin_inff();
        }
        float mx = -__builtin_inff();
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, acc_s[j][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        // Lazy running maximum: m only moves when a tile's maximum exceeds it by more than 2^lazy (p.lazy = 8; it is a scaling
        // reference, not a bound: probabilities up to 2^lazy are exact in fp32 sums and keep their relative precision
        // as bf16 MFMA operands, and the final division by l uses the same reference).  After the first tile the
        // reference almost never moves, alpha is 1 on every lane and the 32-register rescale of O is skipped.
        const float mt = mx * sc;
        const float m_new = (mt > m_run + p.lazy) ? mt : m_run;       // first tile: m_run = -inf -> mt (finite: >= 1 valid key)
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // first tile: 2^-inf = 0
        m_run = m_new;
        f32x2 psum2 = f32x2{0.f, 0.f};
        const f32x2 sc2 = f32x2{sc, sc}, mn2 = f32x2{m_new, m_new};
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 t = f32x2{acc_s[j][r], acc_s[j][r + 1]} * sc2 - mn2;
                const f32x2 e = (ABL == 1) ? t : f32x2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
                acc_s[j][r] = e.x;
                acc_s[j][r + 1] = e.y;
                psum2 += e;
            }
        l_run = l_run * alpha + (psum2.x + psum2.y);
        if (kt > 0 && !__all(alpha == 1.0f)) {   // (first tile: O is still zero) the running max rarely moves later
#pragma unroll
            for (int d = 0; d < DT; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc_o[d][r] *= alpha;
        }
        }   // ABL != 2

        if constexpr (STAMP) { asm volatile("" :: "v"(acc_s[0][0]), "v"(acc_s[1][15]), "v"(l_run)); }
        stamp();                               // +4: softmax done
        // ---- O^T += V^T . P^T ----------------------------------------------------------------------
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if constexpr (DT == 2) lds_wait4n<0>(vf[0][0], vf[0][1], vf[1][0], vf[1][1]);
            else lds_wait2n<0>(vf[0][0], vf[1][0]);
            ex8 va[2][2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int d = 0; d < DT; ++d) va[s2][d] = __builtin_bit_cast(ex8, vf[s2][d]);
            if (j == 0) {   // next 32 keys' fragments, in flight during these MFMAs
                vf[0][0] = lds_read128_async<4 * DH * 16>(vaddr);
                if constexpr (DT == 2) vf[0][1] = lds_read128_async<4 * DH * 16 + 512>(vaddr);
                vf[1][0] = lds_read128_async<6 * DH * 16>(vaddr);
                if constexpr (DT == 2) vf[1][1] = lds_read128_async<6 * DH * 16 + 512>(vaddr);
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                uint4 pb;
                pb.x = pack_e2_inrange(acc_s[j][8 * s2 + 0], acc_s[j][8 * s2 + 1]);
                pb.y = pack_e2_inrange(acc_s[j][8 * s2 + 2], acc_s[j][8 * s2 + 3]);
                pb.z = pack_e2_inrange(acc_s[j][8 * s2 + 4], acc_s[j][8 * s2 + 5]);
                pb.w = pack_e2_inrange(acc_s[j][8 * s2 + 6], acc_s[j][8 * s2 + 7]);
                const ex8 pf = __builtin_bit_cast(ex8, pb);
#pragma unroll
                for (int d = 0; d < DT; ++d)
                    acc_o[d] = TT_MFMA_32x32x16(va[s2][d], pf, acc_o[d]);
            }
        }
    }

    stamp();                                   // loop done
    // ---- normalise and store: lane = query row, registers = 4 consecutive d ------------------
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q_row < len) {
        if (p.out_scales) {
            // f16c: c-planes.  A scale block = 32 consecutive columns = one d tile of this head: this lane's 16 values and those
            // of the lane holding the row's other column quads (hh ^ 1: same q_row, so both are inside this branch).
            char* orow = reinterpret_cast<char*>(p.out) + (size_t)(t0 + q_row) * p.ld_out * 2;
            const int W = p.out_width, nks = W >> 7;
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                float y[16];
                float amax = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    y[r] = acc_o[d][r] * inv;
                    asm("" : "+v"(y[r]));            // opaque: no fusing of "* inv" into the "y - hi" of the split
                    amax = fmaxf(amax, fabsf(y[r]));
                }
                amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
                int sbyte, sh;
                xc_block_scale(amax, sbyte, sh);
                const int col0 = head * DH + 32 * d;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint2 hi;
                    uint32_t x8, l8;
                    xc_split4(y[4 * g + 0], y[4 * g + 1], y[4 * g + 2], y[4 * g + 3], sh, sh + 11, hi, x8, l8);
                    const int col = col0 + 8 * g + 4 * hh;
                    *reinterpret_cast<uint2*>(orow + (size_t)col * 2) = hi;
                    *reinterpret_cast<uint32_t*>(orow + (size_t)2 * W + col) = x8;
                    *reinterpret_cast<uint32_t*>(orow + (size_t)3 * W + col) = l8;
                }
                if (hh == 0) p.out_scales[xc_a_scale_at(t0 + q_row, col0 >> 5, nks)] = (uint8_t)sbyte;
            }
            return;
        }
        uint16_t* op = p.out + (size_t)(t0 + q_row) * p.ld_out + head * DH;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 o;
                o.x = pack_e2(acc_o[d][4 * g + 0] * inv, acc_o[d][4 * g + 1] * inv);
                o.y = pack_e2(acc_o[d][4 * g + 2] * inv, acc_o[d][4 * g + 3] * inv);
                *reinterpret_cast<uint2*>(op + 32 * d + 8 * g + 4 * hh) = o;
            }
    }
}


#if TT_DIAG   // measured, not kept: the diagnostic library only (TT_ATT_RESIDENT=1; tools/gpu_att_resident.sh)
// ---- Resident form (round 5): the whole K and V8 of one (sequence, head) staged ONCE, no key-tile round trips -------------------
// The streaming kernel above pays one LDS-DMA round trip + one workgroup barrier per 64-key tile whatever is computed in it
// (round 3's ablations: a kernel that only copies, waits and issues the S MFMAs still takes 76 % of the time), three times over
// for a 292-token pair (three 128-row workgroups each walk all five key tiles).  For sequences whose aligned key frame fits
// kResMaxTiles tiles (<= 320 keys: 80 KiB of K + V8) ONE four-wave workgroup per (sequence, head) copies every tile up front
// (the same pieces, swizzle and clamping as the streaming kernel: tile kt lands at kt * BUF), waits once, passes ONE barrier, and
// its waves then walk the sequence's 32-row query blocks (block b -> wave (b + rot) % 4: 3 + 3 + 2 + 2 blocks at 292 tokens) with
// nothing but LDS reads, MFMAs and the softmax in the loop.  80 KiB per workgroup = two workgroups per CU, two waves per SIMD,
// out of phase: one loads while the other computes.  A row's arithmetic -- key-tile order, MFMA order, the lazy running reference,
// the P packing -- is the streaming kernel's, so its bits do not depend on which kernel served it (tests: the parity suite with
// either forced; a query alone vs in a batch).
constexpr int kResMaxTiles = 5;

template <int DH, bool PIPE = false, bool EARLY = false>
__global__ __launch_bounds__(64 * kWaves, 2) void attention_resident_kernel(AttnParams p) {
    constexpr int RB = DH * 2, CH = RB / 16, RPB = 256 / RB, KS = DH / 16, DT = DH / 32;
    constexpr int NPK = kKTile * RB / 1024, NPV = 8 * DH * 16 / 1024, NW = kWaves;
    constexpr int KPW = NPK / NW, VPW = NPV / NW;
    static_assert(DH == 64 && NPK % NW == 0 && NPV % NW == 0, "written for head_dim 64");
    constexpr int BUF = (NPK + NPV) * 1024;
    extern __shared__ __attribute__((aligned(1024))) char lds_dyn[];

    const int pair = blockIdx.x;
    if (pair >= p.heads * p.n_seq) return;
    const int head = pair % p.heads, seq = pair / p.heads;
    const int len = p.seq_len[seq];
    if (len <= 0) return;
    const int t0 = p.seq_start[seq];
    const int t0a = t0 & ~7, off = t0 - t0a, alen = off + len;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & 31, hh = lane >> 5;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds_dyn;
    const int n_kt = (alen + kKTile - 1) / kKTile;
    const int n_g8 = (alen + 7) >> 3;
    const int n_blk = (len + 31) >> 5;
    // which blocks this wave walks: b = first, first + 4, ... ; the rotation spreads the waves with one block more over the SIMDs
    // of the two workgroups that share a CU (wave w of every workgroup runs on SIMD w)
    const int rot = (int)((blockIdx.x >> 8) * 2 + (blockIdx.x & 1)) & 3;
    const int first = (wave - rot) & 3;

    auto q_ptr = [&](int b) {
        const int q_row = b * 32 + ql;
        const int q_row_c = q_row < len ? q_row : len - 1;
        return p.qk + (size_t)(t0 + q_row_c) * p.ld_qk + p.q_col0 + head * DH + hh * 8;
    };
    const int abl = p.rotate;      // diagnostic (TT_ATT_RES_ABL): 1 = no output stores, 2 = no Q loads, 4 = no K / V copies -- wrong results by design
    ex8 qf[KS] = {};
    if (first < n_blk && !(abl & 2)) {
        const uint16_t* qp = q_ptr(first);
#pragma unroll
        for (int s = 0; s < KS; ++s) qf[s] = *reinterpret_cast<const ex8*>(qp + s * 16);
    }

    // ---- stage every key tile (the streaming kernel's issue_tile with buffer kt instead of kt & 1)
    const uint16_t* kbase = p.qk + (size_t)t0a * p.ld_qk + p.k_col0 + head * DH;
    const uint16_t* vbase = p.vt + (size_t)(t0a >> 3) * p.ldvt + (size_t)head * DH * 8;
    constexpr int kRowsPerPiece = 64 / CH, kGroupsPerPiece = 1;
    uint32_t kvoff0, vvoff0;
    {
        const int e = wave * KPW * 64 + lane, r = e / CH, pos = e % CH;
        kvoff0 = ((uint32_t)r * (uint32_t)p.ld_qk + (uint32_t)((pos ^ ((r / RPB) & (CH - 1))) << 3)) * 2u;
        const int ev = wave * VPW * 64 + lane;
        vvoff0 = ((uint32_t)(ev / DH) * (uint32_t)p.ldvt + (uint32_t)((ev % DH) * 8)) * 2u;
    }
    auto sbase = [](const void* ptr) {
        const unsigned long long b = reinterpret_cast<unsigned long long>(ptr);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
        return reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
    };
    auto glds16 = [](const char* base, uint32_t voff, uint32_t lds_addr) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds_addr) : "memory", "m0");
    };
    for (int kt = 0; kt < ((abl & 4) ? 0 : n_kt); ++kt) {
        const uint32_t buf = lds0 + (uint32_t)kt * BUF;
        const bool clamp = (kt + 1) * kKTile > alen || (kt == 0 && off != 0);   // wave-uniform
        if (!clamp) {
#pragma unroll
            for (int i = 0; i < KPW; ++i)
                glds16(sbase(kbase + ((size_t)kt * kKTile + i * kRowsPerPiece) * p.ld_qk), (i & 1) ? (kvoff0 ^ 64u) : kvoff0,
                       buf + (uint32_t)(wave * KPW + i) * 1024u);
#pragma unroll
            for (int i = 0; i < VPW; ++i)
                glds16(sbase(vbase + ((size_t)kt * 8 + i * kGroupsPerPiece) * p.ldvt), vvoff0, buf + (uint32_t)(NPK + wave * VPW + i) * 1024u);
        } else {
            const char* kb = sbase(kbase);
            const char* vb = sbase(vbase);
#pragma unroll
            for (int i = 0; i < KPW; ++i) {
                const int piece = wave * KPW + i;
                const int e = piece * 64 + lane, r = e / CH, pos = e % CH;
                int row = kt * kKTile + r;
                row = row < off ? off : (row < alen ? row : alen - 1);
                glds16(kb, ((uint32_t)row * (uint32_t)p.ld_qk + (uint32_t)((pos ^ ((r / RPB) & (CH - 1))) << 3)) * 2u, buf + (uint32_t)piece * 1024u);
            }
#pragma unroll
            for (int i = 0; i < VPW; ++i) {
                const int piece = wave * VPW + i;
                const int e = piece * 64 + lane;
                int g8 = kt * 8 + e / DH;
                g8 = g8 < n_g8 ? g8 : n_g8 - 1;
                glds16(vb, ((uint32_t)g8 * (uint32_t)p.ldvt + (uint32_t)((e % DH) * 8)) * 2u, buf + (uint32_t)(NPK + piece) * 1024u);
            }
        }
    }
    if constexpr (!EARLY) {
        __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0): this wave's copies (and its Q fragments) have landed
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                // ... everyone's: the only barrier of the kernel
        __builtin_amdgcn_sched_barrier(0);
    }
    // EARLY (TT_ATT_RESIDENT=3): the FIRST walk starts on tile 0 as soon as tile 0 has landed -- a counted wait (this wave issued four
    // copies per later tile after it) + one barrier per tile, in the first walk only; every later block finds everything resident.
    auto tile_landed = [&](int kt) {
        switch (n_kt - 1 - kt) {
            case 0: __builtin_amdgcn_s_waitcnt(0x0F70); break;                       // vmcnt(0)
            case 1: __builtin_amdgcn_s_waitcnt(0x0F70 | 4); break;                   // vmcnt(4)
            case 2: __builtin_amdgcn_s_waitcnt(0x0F70 | 8); break;
            case 3: __builtin_amdgcn_s_waitcnt(0x0F70 | 12); break;
            default: __builtin_amdgcn_s_waitcnt(0x4F70 | 0); break;                  // vmcnt(16): bit 4 of the count lives in bit 14
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    bool first_walk = EARLY;
    if (EARLY && first >= n_blk) {                   // a wave without a block still owes the first walk its barriers
        for (int kt = 0; kt < n_kt; ++kt) tile_landed(kt);
        return;
    }

    const float sc = p.scale * 1.4426950408889634f;
    const int krow = (ql & 0x13) | ((ql & 4) << 1) | ((ql & 8) >> 1);
    uint32_t koff[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) koff[s] = lds0 + krow * RB + (((2 * s + hh) ^ ((krow / RPB) & (CH - 1))) << 4);
    const uint32_t voff = lds0 + NPK * 1024 + (hh * DH + ql) * 16;

    if constexpr (PIPE) {
        // ---- software-pipelined walk (TT_ATT_RESIDENT=2): S = K.Q^T of tile kt + 1 is ISSUED before the softmax of tile kt, so its eight
        // MFMAs (and the K / V fragment reads) run under the softmax's VALU instead of in front of it; a row's arithmetic is unchanged.
        auto issue_k = [&](int kt, u32x4 (&kf)[2][4]) {
            const uint32_t bufo = (uint32_t)kt * BUF;
#pragma unroll
            for (int s = 0; s < KS; ++s) kf[0][s] = lds_read128_async<0>(koff[s] + bufo);
#pragma unroll
            for (int s = 0; s < KS; ++s) kf[1][s] = lds_read128_async<32 * RB>(koff[s] + bufo);
        };
        for (int b = first; b < n_blk; b += NW) {
            ex8 qn[KS];
            const bool more = b + NW < n_blk;
            if (more && !(abl & 2)) {
                const uint16_t* qp = q_ptr(b + NW);
#pragma unroll
                for (int s = 0; s < KS; ++s) qn[s] = *reinterpret_cast<const ex8*>(qp + s * 16);
            }
            f32x16 acc_o[DT];
#pragma unroll
            for (int d = 0; d < DT; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc_o[d][r] = 0.f;
            float m_run = -__builtin_inff();
            float l_run = 0.f;
            f32x16 s_cur[2], s_nxt[2];
            u32x4 kf[2][4];
            issue_k(0, kf);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (j == 0) lds_wait4n<KS>(kf[0][0], kf[0][1], kf[0][2], kf[0][3]);
                else lds_wait4n<0>(kf[1][0], kf[1][1], kf[1][2], kf[1][3]);
#pragma unroll
                for (int r = 0; r < 16; ++r) s_cur[j][r] = 0.f;
#pragma unroll
                for (int s = 0; s < KS; ++s) s_cur[j] = TT_MFMA_32x32x16(__builtin_bit_cast(ex8, kf[j][s]), qf[s], s_cur[j]);
            }
            for (int kt = 0; kt < n_kt; ++kt) {
                const int k0 = kt * kKTile;
                const uint32_t vaddr = voff + (uint32_t)kt * BUF;
                const bool has_next = kt + 1 < n_kt;              // wave-uniform
                u32x4 vf[2][2];
                vf[0][0] = lds_read128_async<0>(vaddr);
                vf[0][1] = lds_read128_async<512>(vaddr);
                vf[1][0] = lds_read128_async<2 * DH * 16>(vaddr);
                vf[1][1] = lds_read128_async<2 * DH * 16 + 512>(vaddr);
                issue_k(has_next ? kt + 1 : kt, kf);              // (last tile: a harmless re-read, its S is never used)
                if (kt == 0 && off != 0) {
                    if (hh == 0) {
#pragma unroll
                        for (int r = 0; r < 8; ++r)
                            if (r < off) s_cur[0][r] = -__builtin_inff();
                    }
                }
                if (k0 + kKTile > alen) {
                    const int lim = alen - k0 - 8 * hh;
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (32 * j + 16 * (r >> 3) + (r & 7) >= lim) s_cur[j][r] = -__builtin_inff();
                }
                // ---- one straight-line region: next tile's S MFMAs beside this tile's softmax
                lds_wait4n<KS>(kf[0][0], kf[0][1], kf[0][2], kf[0][3]);        // V first half + K first 32 keys have landed (in-order returns)
#pragma unroll
                for (int r = 0; r < 16; ++r) s_nxt[0][r] = 0.f;
#pragma unroll
                for (int s = 0; s < KS; ++s) s_nxt[0] = TT_MFMA_32x32x16(__builtin_bit_cast(ex8, kf[0][s]), qf[s], s_nxt[0]);
                float mx = -__builtin_inff();
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s_cur[j][r]);
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float mt = mx * sc;
                const float m_new = (mt > m_run + p.lazy) ? mt : m_run;
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                m_run = m_new;
                lds_wait4n<0>(kf[1][0], kf[1][1], kf[1][2], kf[1][3]);
#pragma unroll
                for (int r = 0; r < 16; ++r) s_nxt[1][r] = 0.f;
#pragma unroll
                for (int s = 0; s < KS; ++s) s_nxt[1] = TT_MFMA_32x32x16(__builtin_bit_cast(ex8, kf[1][s]), qf[s], s_nxt[1]);
                f32x2 psum2 = f32x2{0.f, 0.f};
                const f32x2 sc2 = f32x2{sc, sc}, mn2 = f32x2{m_new, m_new};
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const f32x2 t = f32x2{s_cur[j][r], s_cur[j][r + 1]} * sc2 - mn2;
                        const f32x2 e = f32x2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
                        s_cur[j][r] = e.x;
                        s_cur[j][r + 1] = e.y;
                        psum2 += e;
                    }
                l_run = l_run * alpha + (psum2.x + psum2.y);
                if (kt > 0 && !__all(alpha == 1.0f)) {
#pragma unroll
                    for (int d = 0; d < DT; ++d)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc_o[d][r] *= alpha;
                }
                // ---- O^T += V^T . P^T
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    ex8 va[2][2];
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                        for (int d = 0; d < DT; ++d) va[s2][d] = __builtin_bit_cast(ex8, vf[s2][d]);
                    if (j == 0) {
                        vf[0][0] = lds_read128_async<4 * DH * 16>(vaddr);
                        vf[0][1] = lds_read128_async<4 * DH * 16 + 512>(vaddr);
                        vf[1][0] = lds_read128_async<6 * DH * 16>(vaddr);
                        vf[1][1] = lds_read128_async<6 * DH * 16 + 512>(vaddr);
                    }
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        uint4 pb;
                        pb.x = pack_e2_inrange(s_cur[j][8 * s2 + 0], s_cur[j][8 * s2 + 1]);
                        pb.y = pack_e2_inrange(s_cur[j][8 * s2 + 2], s_cur[j][8 * s2 + 3]);
                        pb.z = pack_e2_inrange(s_cur[j][8 * s2 + 4], s_cur[j][8 * s2 + 5]);
                        pb.w = pack_e2_inrange(s_cur[j][8 * s2 + 6], s_cur[j][8 * s2 + 7]);
                        const ex8 pf = __builtin_bit_cast(ex8, pb);
#pragma unroll
                        for (int d = 0; d < DT; ++d)
                            acc_o[d] = TT_MFMA_32x32x16(va[s2][d], pf, acc_o[d]);
                    }
                    if (j == 0) lds_wait4n<0>(vf[0][0], vf[0][1], vf[1][0], vf[1][1]);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) s_cur[j] = s_nxt[j];
            }
            const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
            const float inv = 1.0f / l_tot;
            const int q_row = b * 32 + ql;
            if (q_row < len && !(abl & 1)) {
                uint16_t* op = p.out + (size_t)(t0 + q_row) * p.ld_out + head * DH;
#pragma unroll
                for (int d = 0; d < DT; ++d)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        uint2 o;
                        o.x = pack_e2(acc_o[d][4 * g + 0] * inv, acc_o[d][4 * g + 1] * inv);
                        o.y = pack_e2(acc_o[d][4 * g + 2] * inv, acc_o[d][4 * g + 3] * inv);
                        *reinterpret_cast<uint2*>(op + 32 * d + 8 * g + 4 * hh) = o;
                    }
            }
            if (more) {
#pragma unroll
                for (int s = 0; s < KS; ++s) qf[s] = qn[s];
            }
        }
        return;
    }
    for (int b = first; b < n_blk; b += NW) {
        // the next block's Q fragments travel while this block is computed (EARLY, first walk: issued after the walk -- loads issued
        // now would sit behind the copies in the in-order vmcnt queue and be counted by the per-tile waits)
        ex8 qn[KS];
        const bool more = b + NW < n_blk;
        if (more && !(abl & 2) && !first_walk) {
            const uint16_t* qp = q_ptr(b + NW);
#pragma unroll
            for (int s = 0; s < KS; ++s) qn[s] = *reinterpret_cast<const ex8*>(qp + s * 16);
        }
        f32x16 acc_o[DT];
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_o[d][r] = 0.f;
        float m_run = -__builtin_inff();
        float l_run = 0.f;
        for (int kt = 0; kt < n_kt; ++kt) {
            const int k0 = kt * kKTile;
            const uint32_t bufo = (uint32_t)kt * BUF;
            if constexpr (EARLY) {
                if (first_walk) tile_landed(kt);
            }
            f32x16 acc_s[2];
            {
                u32x4 kf[2][4];
                uint32_t ka[KS];
#pragma unroll
                for (int s = 0; s < KS; ++s) ka[s] = koff[s] + bufo;
#pragma unroll
                for (int s = 0; s < KS; ++s) kf[0][s] = lds_read128_async<0>(ka[s]);
#pragma unroll
                for (int s = 0; s < KS; ++s) kf[1][s] = lds_read128_async<32 * RB>(ka[s]);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (j == 0) lds_wait4n<KS>(kf[0][0], kf[0][1], kf[0][2], kf[0][3]);
                    else lds_wait4n<0>(kf[1][0], kf[1][1], kf[1][2], kf[1][3]);
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc_s[j][r] = 0.f;
#pragma unroll
                    for (int s = 0; s < KS; ++s)
                        acc_s[j] = TT_MFMA_32x32x16(__builtin_bit_cast(ex8, kf[j][s]), qf[s], acc_s[j]);
                }
            }
            u32x4 vf[2][2];
            const uint32_t vaddr = voff + bufo;
            vf[0][0] = lds_read128_async<0>(vaddr);
            vf[0][1] = lds_read128_async<512>(vaddr);
            vf[1][0] = lds_read128_async<2 * DH * 16>(vaddr);
            vf[1][1] = lds_read128_async<2 * DH * 16 + 512>(vaddr);
            if (kt == 0 && off != 0) {
                if (hh == 0) {
#pragma unroll
                    for (int r = 0; r < 8; ++r)
                        if (r < off) acc_s[0][r] = -__builtin_inff();
                }
            }
            if (k0 + kKTile > alen) {
                const int lim = alen - k0 - 8 * hh;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (32 * j + 16 * (r >> 3) + (r & 7) >= lim) acc_s[j][r] = -__builtin_inff();
            }
            float mx = -__builtin_inff();
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, acc_s[j][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mt = mx * sc;
            const float m_new = (mt > m_run + p.lazy) ? mt : m_run;
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
            f32x2 psum2 = f32x2{0.f, 0.f};
            const f32x2 sc2 = f32x2{sc, sc}, mn2 = f32x2{m_new, m_new};
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2 t = f32x2{acc_s[j][r], acc_s[j][r + 1]} * sc2 - mn2;
                    const f32x2 e = f32x2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
                    acc_s[j][r] = e.x;
                    acc_s[j][r + 1] = e.y;
                    psum2 += e;
                }
            l_run = l_run * alpha + (psum2.x + psum2.y);
            if (kt > 0 && !__all(alpha == 1.0f)) {
#pragma unroll
                for (int d = 0; d < DT; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc_o[d][r] *= alpha;
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                lds_wait4n<0>(vf[0][0], vf[0][1], vf[1][0], vf[1][1]);
                ex8 va[2][2];
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int d = 0; d < DT; ++d) va[s2][d] = __builtin_bit_cast(ex8, vf[s2][d]);
                if (j == 0) {
                    vf[0][0] = lds_read128_async<4 * DH * 16>(vaddr);
                    vf[0][1] = lds_read128_async<4 * DH * 16 + 512>(vaddr);
                    vf[1][0] = lds_read128_async<6 * DH * 16>(vaddr);
                    vf[1][1] = lds_read128_async<6 * DH * 16 + 512>(vaddr);
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    uint4 pb;
                    pb.x = pack_e2_inrange(acc_s[j][8 * s2 + 0], acc_s[j][8 * s2 + 1]);
                    pb.y = pack_e2_inrange(acc_s[j][8 * s2 + 2], acc_s[j][8 * s2 + 3]);
                    pb.z = pack_e2_inrange(acc_s[j][8 * s2 + 4], acc_s[j][8 * s2 + 5]);
                    pb.w = pack_e2_inrange(acc_s[j][8 * s2 + 6], acc_s[j][8 * s2 + 7]);
                    const ex8 pf = __builtin_bit_cast(ex8, pb);
#pragma unroll
                    for (int d = 0; d < DT; ++d)
                        acc_o[d] = TT_MFMA_32x32x16(va[s2][d], pf, acc_o[d]);
                }
            }
        }
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        const float inv = 1.0f / l_tot;
        const int q_row = b * 32 + ql;
        if (q_row < len && !(abl & 1)) {
            uint16_t* op = p.out + (size_t)(t0 + q_row) * p.ld_out + head * DH;
#pragma unroll
            for (int d = 0; d < DT; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint2 o;
                    o.x = pack_e2(acc_o[d][4 * g + 0] * inv, acc_o[d][4 * g + 1] * inv);
                    o.y = pack_e2(acc_o[d][4 * g + 2] * inv, acc_o[d][4 * g + 3] * inv);
                    *reinterpret_cast<uint2*>(op + 32 * d + 8 * g + 4 * hh) = o;
                }
        }
        if (EARLY && first_walk) {
            first_walk = false;
            if (more && !(abl & 2)) {
                const uint16_t* qp = q_ptr(b + NW);
#pragma unroll
                for (int s = 0; s < KS; ++s) qn[s] = *reinterpret_cast<const ex8*>(qp + s * 16);
            }
        }
        if (more) {
#pragma unroll
            for (int s = 0; s < KS; ++s) qf[s] = qn[s];
        }
    }
}


#endif   // TT_DIAG
// ---- CLS-only attention (last layer): one wave per (sequence, head), the single query row t0 -------------
// Scores with the key on the lane (fp32 dot products, K rows read as 16-B pieces), softmax across the wave,
// then the value sum with the FEATURE on the lane (V8 layout: one 16-B load brings 8 keys of a feature).
// Only the CLS token of the last layer is ever read by the pooling / classification head, so the other
// rows' attention (and their output projection / FFN) is skipped: 1/24 of the encoder's flops.
template <int DH>
__global__ __launch_bounds__(64) void attention_cls_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) float probs[];   // [max_len rounded up to 8]
    const int seq = blockIdx.x, head = blockIdx.y, lane = threadIdx.x;
    const int len = p.seq_len[seq], t0 = p.seq_start[seq];
    const int t0a = t0 & ~7, off = t0 - t0a, alen = off + len;   // aligned key frame, as in attention_kernel
    // query row -> registers (every lane holds the whole q: uniform address)
    float q[DH];
    {
        const uint16_t* qp = p.q_rows ? p.q_rows + (size_t)seq * p.ld_q_rows + head * DH : p.qk + (size_t)t0 * p.ld_qk + p.q_col0 + head * DH;
#pragma unroll
        for (int c = 0; c < DH / 8; ++c) {
            const uint4 u = *reinterpret_cast<const uint4*>(qp + c * 8);
            q[c * 8 + 0] = elo(u.x); q[c * 8 + 1] = ehi(u.x);
            q[c * 8 + 2] = elo(u.y); q[c * 8 + 3] = ehi(u.y);
            q[c * 8 + 4] = elo(u.z); q[c * 8 + 5] = ehi(u.z);
            q[c * 8 + 6] = elo(u.w); q[c * 8 + 7] = ehi(u.w);
            if (p.qk_lo_off) {          // f16c: hi + lo planes
                const uint4 l = *reinterpret_cast<const uint4*>(qp + p.qk_lo_off + c * 8);
                q[c * 8 + 0] += elo(l.x); q[c * 8 + 1] += ehi(l.x);
                q[c * 8 + 2] += elo(l.y); q[c * 8 + 3] += ehi(l.y);
                q[c * 8 + 4] += elo(l.z); q[c * 8 + 5] += ehi(l.z);
                q[c * 8 + 6] += elo(l.w); q[c * 8 + 7] += ehi(l.w);
            }
        }
    }
    const float sc = p.scale * 1.4426950408889634f;
    float mx = -__builtin_inff();
    for (int j = off + lane; j < alen; j += 64) {
        const uint16_t* kp = p.qk + (size_t)(t0a + j) * p.ld_qk + p.k_col0 + head * DH;
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < DH / 8; ++c) {
            const uint4 u = *reinterpret_cast<const uint4*>(kp + c * 8);
            float kf[8] = {elo(u.x), ehi(u.x), elo(u.y), ehi(u.y), elo(u.z), ehi(u.z), elo(u.w), ehi(u.w)};
            if (p.qk_lo_off) {
                const uint4 l = *reinterpret_cast<const uint4*>(kp + p.qk_lo_off + c * 8);
                kf[0] += elo(l.x); kf[1] += ehi(l.x); kf[2] += elo(l.y); kf[3] += ehi(l.y);
                kf[4] += elo(l.z); kf[5] += ehi(l.z); kf[6] += elo(l.w); kf[7] += ehi(l.w);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) acc = fmaf(q[c * 8 + i], kf[i], acc);
        }
        acc *= sc;
        probs[j] = acc;
        mx = fmaxf(mx, acc);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    float sum = 0.f;
    const int len8 = (alen + 7) & ~7;
    for (int j = lane; j < len8; j += 64) {
        float e = 0.f;
        if (j >= off && j < alen) e = __builtin_amdgcn_exp2f(probs[j] - mx);
        probs[j] = e;      // aligned keys outside the sequence (same 8-groups) get probability 0
        sum += e;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    __syncthreads();
    // value sum: lane = feature (DH=64) or (feature, key-group parity) (DH=32)
    constexpr int PARTS = 64 / DH;
    const int d = lane % DH, part = lane / DH;
    float o = 0.f;
    for (int g8 = part; g8 * 8 < alen; g8 += PARTS) {
        const uint4 u = *reinterpret_cast<const uint4*>(p.vt + (size_t)(t0a / 8 + g8) * p.ldvt + (size_t)(head * DH + d) * 8);
        const float4 pa = *reinterpret_cast<const float4*>(probs + g8 * 8);
        const float4 pb = *reinterpret_cast<const float4*>(probs + g8 * 8 + 4);
        o = fmaf(pa.x, elo(u.x), o); o = fmaf(pa.y, ehi(u.x), o);
        o = fmaf(pa.z, elo(u.y), o); o = fmaf(pa.w, ehi(u.y), o);
        o = fmaf(pb.x, elo(u.z), o); o = fmaf(pb.y, ehi(u.z), o);
        o = fmaf(pb.z, elo(u.w), o); o = fmaf(pb.w, ehi(u.w), o);
    }
    if constexpr (PARTS == 2) o += __shfl_xor(o, 32, 64);
    if (p.out_scales) {
        // f16c: c-planes (row = sequence index).  DH = 64: lanes 0-31 / 32-63 are the two scale blocks of this head.
        float y = o / sum;
        asm("" : "+v"(y));
        float amax = fabsf(y);
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off, 64));
        int sbyte, sh;
        xc_block_scale(amax, sbyte, sh);
        if (part == 0) {
            const int W = p.out_width, col = head * DH + d;
            char* orow = reinterpret_cast<char*>(p.out) + (size_t)seq * p.ld_out * 2;
            const uint16_t hb = f32_to_ebits(y);
            *reinterpret_cast<uint16_t*>(orow + (size_t)col * 2) = hb;
            orow[(size_t)2 * W + col] = (char)(xc_pack4(xc_sat(ldexpf(y, sh)), 0.f, 0.f, 0.f) & 0xFFu);
            orow[(size_t)3 * W + col] = (char)(xc_pack4(xc_sat(ldexpf(y - ebits_to_f32(hb), sh + 11)), 0.f, 0.f, 0.f) & 0xFFu);
            if ((d & 31) == 0) p.out_scales[xc_a_scale_at(seq, col >> 5, W >> 7)] = (uint8_t)sbyte;
        }
        return;
    }
    if (part == 0) p.out[(size_t)seq * p.ld_out + head * DH + d] = f32_to_ebits(o / sum);
}

}  // namespace

int tt_attention_launch(const AttnParams& p, hipStream_t st) {
    if (p.n_seq <= 0 || p.max_len <= 0) return TT_OK;
    if ((p.ld_qk % 8) || (p.q_col0 % 8) || (p.k_col0 % 8) || (p.ldvt % 8) || (p.ld_out % 4)) {
        tt_set_error("attention: leading dimensions / column offsets must keep 16-byte alignment");
        return TT_E_INVALID;
    }
    // Waves (32-row query blocks) per workgroup, chosen per batch from B = ceil(max_len / 32) (measured at 1600 sequences x 16 heads,
    // profiles/r04_attention_five_waves_ab.log, ...eight_waves...): EIGHT waves (256 rows: half the K / V copies and barriers per
    // query row, two waves per SIMD like two four-wave workgroups) win where the last workgroup of a sequence is at least
    // five-eighths full -- B = 5..8 (130 tokens +32 %, 160 +26 %, 200 +13 %) and B = 13..16 (420 +6 %, 512 +5 %) -- and lose where
    // it is nearly empty (B = 9, 10: 258 / 292 / 320 tokens -20...-23 %); FIVE waves (two of them on one SIMD) never beat the
    // better of four and eight.  TT_ATT_WAVES=4|5|8 forces one (the A/B switch; a row's arithmetic does not depend on it).
    static const int waves_env = TT_DIAG_ENV_INT("TT_ATT_WAVES", 0);
    int nw = kWaves;
    if (p.head_dim == 64) {
        const int B = (p.max_len + 31) / 32, r8 = B % 8;
        if (waves_env == 4 || waves_env == 5 || waves_env == 8) nw = waves_env;
        // ... and only for batches of SIMILAR lengths (rerank pairs, length-sorted embedding windows): a short sequence in an
        // eight-wave launch is one workgroup with mostly idle waves (64 tokens: +20 %), so the mean length must be near max_len
        // (rows are padded to 8 per sequence; unknown total: the caller is a test or a tool with uniform lengths)
        else if (B >= 5 && (r8 == 0 || r8 >= 5) && (p.total_rows <= 0 || (long long)p.total_rows * 10 >= (long long)p.n_seq * p.max_len * 7)) nw = 8;
    }
    const int n_qt = (p.max_len + 32 * nw - 1) / (32 * nw);
    const long long pairs8 = ((long long)p.heads * p.n_seq + 7) / 8 * 8;
    if (pairs8 * n_qt > 0x7FFFFFFFLL) {
        tt_set_error("attention: %lld workgroups exceed the grid limit", pairs8 * n_qt);
        return TT_E_UNSUPPORTED;
    }
    const dim3 grid((unsigned)(pairs8 * n_qt));
    // log2 slack of the running softmax reference (TT_ATT_LAZY=0: classic running maximum, the A/B switch)
    static const float lazy = TT_DIAG_ENV_FLOAT("TT_ATT_LAZY", 8.0f);
    AttnParams q = p;
    q.lazy = lazy;
    // default: the plain order.  Stand-alone the XCD-aware order is 2-9 % faster, inside the encoder (K / V fresh from the
    // QKV GEMM) it measured 1.8 % slower -- see DESIGN_HISTORY.md section 4.4.
    static const bool xcd = TT_DIAG_ENV_INT("TT_ATT_XCD", 0) == 1;
    q.n_qt = xcd ? n_qt : -n_qt;
    // TT_ATT_ROTATE=1: tail tiles deal their live row blocks to different waves (A/B switch, measured neutral; same bits either way)
    static const bool rotate = TT_DIAG_ENV_INT("TT_ATT_ROTATE", 0) == 1;
    q.rotate = rotate ? 1 : 0;
    TtProfScope prof(TT_K_ATTENTION, st);
#if TT_DIAG
    // Resident form (round 5): every sequence's aligned key frame (up to 7 rows of its predecessor in front) fits kResMaxTiles key tiles
    // -- one workgroup per (sequence, head), K and V8 staged once.  TT_ATT_RESIDENT (diagnostic library): 0 = never, 1 = whenever it fits.
    static const int resident = TT_DIAG_ENV_INT("TT_ATT_RESIDENT", TT_ATT_RESIDENT_DEFAULT);
    if (resident && p.head_dim == 64 && !p.out_scales && !p.dbg && p.max_len + 7 <= kResMaxTiles * kKTile && p.max_len >= TT_ATT_RESIDENT_MIN_LEN) {
        const int tiles = (p.max_len + 7 + kKTile - 1) / kKTile;
        const size_t lds = (size_t)tiles * 16384;
        q.rotate = TT_DIAG_ENV_INT("TT_ATT_RES_ABL", 0);
        if (resident == 3) {
            TT_SET_MAX_LDS((attention_resident_kernel<64, false, true>), 160 * 1024);
            hipLaunchKernelGGL((attention_resident_kernel<64, false, true>), dim3((unsigned)(p.heads * p.n_seq)), dim3(64 * kWaves), lds, st, q);
        } else if (resident == 2) {
            TT_SET_MAX_LDS((attention_resident_kernel<64, true>), 160 * 1024);
            hipLaunchKernelGGL((attention_resident_kernel<64, true>), dim3((unsigned)(p.heads * p.n_seq)), dim3(64 * kWaves), lds, st, q);
        } else {
            TT_SET_MAX_LDS(attention_resident_kernel<64>, 160 * 1024);
            hipLaunchKernelGGL(attention_resident_kernel<64>, dim3((unsigned)(p.heads * p.n_seq)), dim3(64 * kWaves), lds, st, q);
        }
        TT_CHECK_LAUNCH();
        return TT_OK;
    }
#endif
#if TT_DIAG   // stamped / ablated instantiations: the diagnostic library only (tools/att_stamps, tools/gpu_att_ablate.sh)
    if (p.head_dim == 64 && (p.dbg || TT_DIAG_ENV_INT("TT_ATT_ABLATE", 0) != 0) && nw != kWaves) {
        tt_set_error("attention: the stamped / ablated kernels are four-wave (set TT_ATT_WAVES=4)");
        return TT_E_UNSUPPORTED;
    }
    if (p.head_dim == 64 && p.dbg) {
        hipLaunchKernelGGL((attention_kernel<64, true>), grid, dim3(64 * kWaves), 0, st, q);
    } else if (p.head_dim == 64 && TT_DIAG_ENV_INT("TT_ATT_ABLATE", 0) != 0) {
        static const int abl = TT_DIAG_ENV_INT("TT_ATT_ABLATE", 0);
        if (abl == 1) hipLaunchKernelGGL((attention_kernel<64, false, 1>), grid, dim3(64 * kWaves), 0, st, q);
        else if (abl == 2) hipLaunchKernelGGL((attention_kernel<64, false, 2>), grid, dim3(64 * kWaves), 0, st, q);
        else if (abl == 3) hipLaunchKernelGGL((attention_kernel<64, false, 3>), grid, dim3(64 * kWaves), 0, st, q);
        else hipLaunchKernelGGL((attention_kernel<64, false, 4>), grid, dim3(64 * kWaves), 0, st, q);
    } else
#endif
    if (p.head_dim == 64 && nw == 8) {
        hipLaunchKernelGGL((attention_kernel<64, false, 0, 8>), grid, dim3(64 * 8), 0, st, q);
#if TT_DIAG   // five waves: measured slower than the better of four and eight at every length (EXPERIMENTS.md round 4)
    } else if (p.head_dim == 64 && nw == 5) {
        hipLaunchKernelGGL((attention_kernel<64, false, 0, 5>), grid, dim3(64 * 5), 0, st, q);
#endif
    } else if (p.head_dim == 64) {
        hipLaunchKernelGGL(attention_kernel<64>, grid, dim3(64 * kWaves), 0, st, q);
    } else if (p.head_dim == 32) {
        hipLaunchKernelGGL(attention_kernel<32>, grid, dim3(64 * kWaves), 0, st, q);
    } else {
        tt_set_error("attention: head_dim %d not in {32, 64}", p.head_dim);
        return TT_E_UNSUPPORTED;
    }
    TT_CHECK_LAUNCH();
    return TT_OK;
}

// out: [n_seq][ld_out] (row = sequence index, NOT token row)
int tt_attention_cls_launch(const AttnParams& p, hipStream_t st) {
    if (p.n_seq <= 0) return TT_OK;
    const size_t lds = (size_t)((p.max_len + 14) / 8 * 8) * sizeof(float);   // aligned key frame: up to 7 leading slots
    if (lds > 160 * 1024) {
        tt_set_error("attention_cls: max_len %d exceeds the LDS score buffer", p.max_len);
        return TT_E_UNSUPPORTED;
    }
    TtProfScope prof(TT_K_ATTENTION, st);
    const dim3 grid(p.n_seq, p.heads);
    if (p.head_dim == 64) {
        TT_SET_MAX_LDS(attention_cls_kernel<64>, 160 * 1024);
        hipLaunchKernelGGL(attention_cls_kernel<64>, grid, dim3(64), lds, st, p);
    } else if (p.head_dim == 32) {
        TT_SET_MAX_LDS(attention_cls_kernel<32>, 160 * 1024);
        hipLaunchKernelGGL(attention_cls_kernel<32>, grid, dim3(64), lds, st, p);
    } else {
        tt_set_error("attention: head_dim %d not in {32, 64}", p.head_dim);
        return TT_E_UNSUPPORTED;
    }
    TT_CHECK_LAUNCH();
    return TT_OK;
}
