// Shared helpers for libtt_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/tt_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define TT_WAVE 64

// thread-local error text (tt_last_error)
void tt_set_error(const char* fmt, ...);

#define TT_CHECK_ARG(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            tt_set_error(__VA_ARGS__);     \
            return TT_E_INVALID;           \
        }                                  \
    } while (0)

#define TT_CHECK_HIP(expr)                                                              \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            tt_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return TT_E_HIP;                                                            \
        }                                                                               \
    } while (0)

#define TT_CHECK_LAUNCH()                                                               \
    do {                                                                                \
        hipError_t _e = hipGetLastError();                                              \
        if (_e != hipSuccess) {                                                         \
            tt_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
            return TT_E_HIP;                                                            \
        }                                                                               \
    } while (0)

int tt_cu_count_cached();

static inline size_t tt_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- device helpers ---------------------------------------------------------
__device__ __forceinline__ float bf16_bits_to_f32(uint16_t h) {
    return __uint_as_float(((uint32_t)h) << 16);
}

// round-to-nearest-even f32 -> bf16 bits (NaN stays NaN through the compiler cast)
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(uint16_t, b);
}

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return (uint32_t)f32_to_bf16_bits(lo) | ((uint32_t)f32_to_bf16_bits(hi) << 16);
}
