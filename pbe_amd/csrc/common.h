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
void pbe_prof_end(int klass, hipStream_t s, double work, double bytes = 0.0);   // work = FLOP (MFMA classes) or bytes; bytes = algorithmic HBM bytes
enum { PBE_K_CONV3 = 0, PBE_K_GEMM = 1, PBE_K_ATTN = 2, PBE_K_GNORM = 3, PBE_K_LNORM = 4, PBE_K_ELEM = 5, PBE_K_SOFTMAX = 6, PBE_K_SPLITK = 7, PBE_K_COUNT = 8 };

// ---- device helpers -------------------------------------------------------------------------
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }              // v_rcp_f32, 1 ulp
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }            // v_exp_f32
__device__ __forceinline__ float silu_f(float x) { return x * fast_rcp(1.0f + fast_exp2(-1.4426950408889634f * x)); }
__device__ __forceinline__ float quick_gelu_f(float x) { return x * fast_rcp(1.0f + fast_exp2(-1.702f * 1.4426950408889634f * x)); }
// exact (erf) GELU as x * Phi(x) = relu(x) - |x| * Phi(-|x|), with Phi(-u) = exp2(-u R(u) - 1): R = -(log2 Phi(-u) + 1) / u is smooth
// (1.151 at 0, ~ u / (2 ln 2) far out) and a degree-4 fit of it, weighted by what an error in R does to the result, leaves |gelu error|
// <= 7.1e-7 absolute over the whole fp32 range (oracle-side check: tests/test_oracle_golden.py::test_gelu_fit_constants), far below the
// fp16 rounding of the stored result.  7 full-rate instructions + one v_exp (|x| rides on the operand modifiers): the erfc form of
// Abramowitz & Stegun 7.1.26 used before (v_rcp + v_exp + 16 others, 4.2e-7) cost the GEGLU epilogue more VALU cycles than its tile has
// MFMA cycles at K = 320.  The leading coefficient is positive, so q -> 0 for every |x| beyond the fitted range (no clamp).
#define PBE_GELU_R0 1.1510006189346313f
#define PBE_GELU_R1 0.4595957100391388f
#define PBE_GELU_R2 0.052146803587675095f
#define PBE_GELU_R3 (-0.007198805455118418f)
#define PBE_GELU_R4 0.00048811722081154585f
__device__ __forceinline__ float gelu_erf_f(float x) {
    const float u = fabsf(x);
    float r = fmaf(PBE_GELU_R4, u, PBE_GELU_R3);
    r = fmaf(r, u, PBE_GELU_R2);
    r = fmaf(r, u, PBE_GELU_R1);
    r = fmaf(r, u, PBE_GELU_R0);
    const float q = fast_exp2(fmaf(-u, r, -1.0f));
    return fmaf(-u, q, fmaxf(x, 0.0f));
}

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
        const float k = (act == 1 ? 1.0f : 1.702f) * 1.4426950408889634f;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = v[r] * fast_rcp(1.0f + fast_exp2(-k * v[r]));
    } else if (act == 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_erf_f(v[r]);
    }
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel.  One bit per device remembers where it
// has been raised; the call is idempotent, so two threads racing on the same bit merely repeat it (no lock, no UB).
#include <atomic>
static inline void pbe_raise_dynamic_lds(std::atomic<uint64_t>& done, const void* kernel, int bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return;
    (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    done.fetch_or(bit, std::memory_order_release);
}
