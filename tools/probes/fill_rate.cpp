// Round 6 probe: how fast can ONE workgroup per CU fill LDS from L2-resident global memory -- LDS-DMA (global_load_lds_dwordx4) against
// register staging (global_load_dwordx4 -> VGPR -> ds_write_b128) -- with the access shape of gemm_staged_kernel: 4 waves, 8 x 1 KiB
// per wave and step (rows of 128 bytes, 8 rows per wave-instruction), a ring of 4 x 32 KiB in LDS, three steps in flight, nothing
// computed.  Every workgroup walks its own 128-row A window and a W window shared by 8 workgroups, as the tiles of one XCD do.
//   hipcc -O2 --offload-arch=gfx950 -o fill_rate fill_rate.cpp && ./fill_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const void* base, uint32_t voff, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(base), "s"(lds_addr) : "memory", "m0");
}

constexpr int kStage = 32768, kStages = 4;

// MODE 0: LDS-DMA, three steps in flight.  MODE 1: register staging, three steps in flight (96 VGPRs of landing space).
// MODE 2: LDS-DMA, ONE step in flight (the two-stage kernel's depth).
template <int MODE>
__global__ __launch_bounds__(256, 1) void fill_kernel(const char* a, const char* w, int k_bytes, int lda, int steps, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int tile = blockIdx.x;
    const char* rowA = a + (size_t)(tile >> 3) * 128 * lda;            // 8 tiles share an A window ...
    const char* rowW = w + (size_t)(tile & 7) * 128 * lda;             // ... and every eighth tile a W window
    uint32_t voff[8], ldst[8];
    const int lrow = lane >> 3, slot = lane & 7;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = 32 * wave + 8 * j + lrow;
        voff[j] = voff[4 + j] = (uint32_t)row * (uint32_t)lda + ((slot ^ ((row >> 1) & 7)) << 4);
        ldst[j] = (32 * wave + 8 * j) * 128;
        ldst[4 + j] = 16384 + ldst[j];
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    auto sgpr = [](const char* p) {
        const unsigned long long b = reinterpret_cast<unsigned long long>(p);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
        return reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);     // (unsigned: the builtin returns int)
    };
    unsigned acc = 0;
    if constexpr (MODE != 1) {
        constexpr int AHEAD = MODE == 0 ? 3 : 1;
        auto issue = [&](int s) {
            const uint32_t st = lds0 + (s & (kStages - 1)) * kStage;
            const int off = (s * 128) % k_bytes;
            const char* pa = sgpr(rowA + off);
            const char* pw = sgpr(rowW + off);
#pragma unroll
            for (int c = 0; c < 4; ++c) glds16(pa, voff[c], st + ldst[c]);
#pragma unroll
            for (int c = 4; c < 8; ++c) glds16(pw, voff[c], st + ldst[c]);
        };
        for (int s = 0; s < AHEAD && s < steps; ++s) issue(s);
        for (int s = 0; s < steps; ++s) {
            const int ahead = steps - 1 - s < AHEAD - 1 ? steps - 1 - s : AHEAD - 1;
            if (ahead >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (s + AHEAD < steps) issue(s + AHEAD);
            // touch the stage (one ds_read per wave) so the data is really consumed
            uint32_t v;
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lds0 + (s & (kStages - 1)) * kStage + threadIdx.x * 4));
            acc += v;
        }
    } else {
        // loads and waits written as asm: through plain C++ loads hipcc counts vmcnt conservatively across the branches and waits for
        // ALL steps in flight before the first store (vmcnt(7) ... vmcnt(0)), i.e. runs at depth one
        u32x4 r[3][8];
        auto issue = [&](u32x4 (&dst)[8], int s) {
            const int off = (s * 128) % k_bytes;
            const char* pa = sgpr(rowA + off);
            const char* pw = sgpr(rowW + off);
#pragma unroll
            for (int c = 0; c < 4; ++c) asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(dst[c]) : "v"(voff[c]), "s"(pa) : "memory");
#pragma unroll
            for (int c = 4; c < 8; ++c) asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(dst[c]) : "v"(voff[c]), "s"(pw) : "memory");
        };
        auto wait = [&](u32x4 (&d)[8], int ahead) {         // the eight oldest loads have returned; `ahead` younger steps stay in flight
            if (ahead >= 2) asm volatile("s_waitcnt vmcnt(16)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]));
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(8)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]));
            else asm volatile("s_waitcnt vmcnt(0)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]));
        };
        auto land = [&](const u32x4 (&src)[8], int s) {
            const uint32_t st = lds0 + (s & (kStages - 1)) * kStage + lane * 16;
#pragma unroll
            for (int c = 0; c < 8; ++c) asm volatile("ds_write_b128 %0, %1" :: "v"(st + ldst[c]), "v"(src[c]) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };
        if (0 < steps) issue(r[0], 0);
        if (1 < steps) issue(r[1], 1);
        if (2 < steps) issue(r[2], 2);
        for (int s = 0; s < steps; s += 3) {
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                if (s + u < steps) {
                    const int left = steps - 1 - (s + u);
                    wait(r[u], left < 2 ? left : 2);
                    land(r[u], s + u);
                    __builtin_amdgcn_s_barrier();
                    if (s + u + 3 < steps) issue(r[u], s + u + 3);
                    uint32_t v;
                    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lds0 + ((s + u) & (kStages - 1)) * kStage + threadIdx.x * 4));
                    acc += v;
                }
            }
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int wgs = argc > 1 ? atoi(argv[1]) : 192;          // workgroups = CUs kept busy
    const int k_bytes = argc > 2 ? atoi(argv[2]) : 8192;     // bytes of a row walked before wrapping (K = 4096 bf16)
    const int steps = argc > 3 ? atoi(argv[3]) : 2048;
    const int modes = argc > 4 ? atoi(argv[4]) : 7;          // bit mask of the modes to run
    const int lda = k_bytes;
    const size_t a_bytes = (size_t)((wgs + 7) / 8) * 128 * lda, w_bytes = (size_t)8 * 128 * lda;
    char *a, *w;
    unsigned* sink;
    CK(hipMalloc(&a, a_bytes));
    CK(hipMalloc(&w, w_bytes));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(a, 1, a_bytes));
    CK(hipMemset(w, 2, w_bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const char* names[3] = {"LDS-DMA, 3 steps in flight", "register staging, 3 steps in flight", "LDS-DMA, 1 step in flight"};
    printf("fill_rate: %d workgroups x 4 waves, %d steps of 32 KiB, rows of %d bytes (A %.1f MB + W %.1f MB resident)\n", wgs, steps, k_bytes,
           a_bytes / 1e6, w_bytes / 1e6);
    for (int mode = 0; mode < 3; ++mode) {
        if (!((modes >> mode) & 1)) continue;
        auto launch = [&]() {
            if (mode == 0) hipLaunchKernelGGL(fill_kernel<0>, dim3(wgs), dim3(256), kStage * kStages, 0, a, w, k_bytes, lda, steps, sink);
            else if (mode == 1) hipLaunchKernelGGL(fill_kernel<1>, dim3(wgs), dim3(256), kStage * kStages, 0, a, w, k_bytes, lda, steps, sink);
            else hipLaunchKernelGGL(fill_kernel<2>, dim3(wgs), dim3(256), kStage * kStages, 0, a, w, k_bytes, lda, steps, sink);
        };
        if (mode == 0) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fill_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, kStage * kStages));
        if (mode == 1) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fill_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, kStage * kStages));
        if (mode == 2) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fill_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, kStage * kStages));
        launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < 5; ++i) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 5;
        const double bytes = (double)wgs * steps * kStage;
        printf("  %-38s %8.3f ms   %6.1f GB/s per CU   %5.2f TB/s over %d CUs   %.2f us per step\n", names[mode], ms, bytes / wgs / (ms * 1e-3) / 1e9,
               bytes / (ms * 1e-3) / 1e12, wgs, ms * 1e3 / steps);
    }
    return 0;
}
