// "c-planes": the operand format of the f16c path (reference precision on two matrix-time units; encoder.h GemmParams.xc,
// f16c_path.hip).  Device helpers shared by the producers of c-planes: the GELU epilogue of the GEMM, the LayerNorm /
// embedding kernels, the attention epilogues, the stand-alone quantisers.  Used by the fp16 instantiation (TT_F16) only.
//
// A value x is carried as   hi = fp16(x)   x8 = e4m3(x 2^-s)   lo8 = e4m3((x - hi) 2^-(s - 11))
// with one E8M0 block exponent per 32 consecutive elements of the contraction axis: s = E - 7, E = exponent of the block's
// absmax (so |x| 2^-s < 2^8 <= 448, and |x - hi| <= 2^(E - 11) lands on the same grid 11 binades down).
// Row of an ACTIVATION operand with K elements (4 K bytes):  [ hi: 2 K | x8: K | lo8: K ]
// Row of a WEIGHT operand:                                   [ hi: 2 K | lo8: K | x8: K ]     (each e4m3 plane meets the other
//                                                              operand's opposite plane at the same byte offset)
// Scales are stored TILED: 1 KiB per (256-row block, 128-element K-tile), in the order the GEMM's lanes read them.
#pragma once
#include "common.h"

// (the helpers compile in both instantiations; only the fp16 one calls them)
// activation scale byte of (row, 32-element block blk) for an operand with nks = K / 128 K-tiles
__device__ __forceinline__ int xc_a_scale_image(int r, int g) {   // r = row inside the 256-row block, g = block inside the K-tile
    return ((((r >> 7) * 2 + ((r >> 6) & 1)) * 16 + (r & 15)) * 16) + g * 4 + ((r >> 4) & 3);
}
__device__ __forceinline__ size_t xc_a_scale_at(int row, int blk, int nks) {
    return ((size_t)(row >> 8) * nks + (blk >> 2)) * 1024 + xc_a_scale_image(row & 255, blk & 3);
}
// weight scale byte of (output column n, part, block blk): part 0 = the lo8 plane's own exponents, part 1 = the x8 plane's - 11
__device__ __forceinline__ int xc_w_scale_image(int c, int g) {   // c = column inside the 256-column block
    return ((((c >> 6) * 16 + (c & 15)) * 4 + g) * 4) + ((c >> 5) & 1) * 2 + ((c >> 4) & 1);
}
__device__ __forceinline__ size_t xc_w_scale_at(int n, int part, int blk, int nks) {
    return (((size_t)(n >> 8) * 2 + part) * nks + (blk >> 2)) * 1024 + xc_w_scale_image(n & 255, blk & 3);
}

// E8M0 byte of a block with absolute maximum amax, and the shift that takes its values to the e4m3 grid (x 2^sh < 2^8)
__device__ __forceinline__ void xc_block_scale(float amax, int& sbyte, int& sh) {
    int eb = (int)((__float_as_uint(amax) >> 23) & 0xFFu);      // biased exponent (0: zero / denormal block)
    eb = eb > 254 ? 254 : eb;                                    // (inf / NaN: saturate)
    sbyte = eb > 7 ? eb - 7 : 0;
    sh = 127 - sbyte;
}

__device__ __forceinline__ uint32_t xc_pack4(float a, float b, float c, float d) {
    int r = 0;
    r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, r, false);
    r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
    return (uint32_t)r;
}
__device__ __forceinline__ float xc_sat(float f) { return __builtin_amdgcn_fmed3f(f, -448.0f, 448.0f); }

// four consecutive values -> their pieces of the three planes (lo_sh = the lo8 plane's shift: sh + 11 for activations)
__device__ __forceinline__ void xc_split4(float v0, float v1, float v2, float v3, int sh, int lo_sh, uint2& hi, uint32_t& x8, uint32_t& l8) {
    hi.x = pack_e2(v0, v1);
    hi.y = pack_e2(v2, v3);
    x8 = xc_pack4(xc_sat(ldexpf(v0, sh)), xc_sat(ldexpf(v1, sh)), xc_sat(ldexpf(v2, sh)), xc_sat(ldexpf(v3, sh)));
    l8 = xc_pack4(xc_sat(ldexpf(v0 - elo(hi.x), lo_sh)), xc_sat(ldexpf(v1 - ehi(hi.x), lo_sh)), xc_sat(ldexpf(v2 - elo(hi.y), lo_sh)),
                  xc_sat(ldexpf(v3 - ehi(hi.y), lo_sh)));
}
