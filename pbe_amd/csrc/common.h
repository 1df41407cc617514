// Shared device/host helpers for libpbe_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef _Float16 h16;
typedef h16 h16x2 __attribute__((ext_vector_type(2)));
typedef h16 h16x4 __attribute__((ext_vector_type(4)));
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- error plumbing (include/pbe_hip.h) ----------------------------------------------------
#define PBE_OK 0
#define PBE_EINVAL (-1)
#define PBE_ELAUNCH (-2)
#define PBE_ENOTSUP (-3)

extern thread_local char g_pbe_err[512];
int pbe_set_error(int code, const char* fmt, ...);

#define PBE_REQUIRE(cond, ...)                                   \
    do {                                                         \
        if (!(cond)) return pbe_set_error(PBE_EINVAL, __VA_ARGS__); \
    } while (0)

#define PBE_LAUNCH_CHECK(name)                                                              \
    do {                                                                                    \
        hipError_t e__ = hipGetLastError();                                                 \
        if (e__ != hipSuccess)                                                              \
            return pbe_set_error(PBE_ELAUNCH, "%s: launch failed: %s", name, hipGetErrorString(e__)); \
    } while (0)

// optional per-kernel-class event timing (pbe_prof_*), see profile.cpp
void pbe_prof_begin(int klass, hipStream_t s);
void pbe_prof_end(int klass, hipStream_t s, double work);
enum { PBE_K_CONV3 = 0, PBE_K_GEMM = 1, PBE_K_ATTN = 2, PBE_K_GNORM = 3, PBE_K_LNORM = 4, PBE_K_ELEM = 5, PBE_K_SOFTMAX = 6, PBE_K_COUNT = 7 };

// ---- device helpers -------------------------------------------------------------------------
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float quick_gelu_f(float x) { return x / (1.0f + __expf(-1.702f * x)); }

__device__ __forceinline__ float apply_act(float x, int act) {
    switch (act) {
        case 1: return silu_f(x);
        case 2: return gelu_erf_f(x);
        case 3: return quick_gelu_f(x);
        default: return x;
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// activation on a register quad with ONE uniform branch per quad (keeps unrolled epilogues small)
__device__ __forceinline__ void apply_act4(float (&v)[4], int act) {
    if (act == 1 || act == 3) {
        const float k = act == 1 ? 1.0f : 1.702f;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = v[r] / (1.0f + __expf(-k * v[r]));
    } else if (act == 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_erf_f(v[r]);
    }
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
