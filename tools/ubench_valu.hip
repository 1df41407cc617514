// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU ops the attention
// softmax is made of, on gfx950.  hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o tools/ubench_valu.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

// 16 independent chains per lane, ITER x 16 ops of one kind
template <int OP>
__global__ __launch_bounds__(256) void valu_kernel(float* out, int iters, float seed) {
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = seed + threadIdx.x * 1e-3f + i;
    f16v acc = {0};
    h8 ha = {1, 2, 3, 4, 5, 6, 7, 8}, hb = {1, 1, 1, 1, 1, 1, 1, 1};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (OP == 0) x[i] = __builtin_amdgcn_exp2f(x[i]);
            if (OP == 1) x[i] = __builtin_fmaf(x[i], 1.0001f, 0.5f);
            if (OP == 2) { if (i % 2 == 0) { f2 v = {x[i], x[i + 1]}; f2 c = {1.0001f, 1.0002f}, d = {0.5f, 0.25f}; asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(v) : "v"(v), "v"(c), "v"(d)); x[i] = v[0]; x[i + 1] = v[1]; } }
            if (OP == 3) x[i] = __builtin_fmaxf(__builtin_fmaxf(x[i], x[(i + 1) & 15]), seed);
            if (OP == 4) { auto h = __builtin_amdgcn_cvt_pkrtz(x[i], x[(i + 1) & 15]); x[i] += (float)h[0]; }
            if (OP == 5) x[i] = __builtin_amdgcn_rcpf(x[i]);
            if (OP == 6) { _Float16 h = (_Float16)x[i]; asm volatile("v_exp_f16 %0, %1" : "=v"(h) : "v"(h)); x[i] = (float)h; }
        }
        if (OP == 7) {                      // exp + MFMA interleaved in one wave: does the MFMA hide under the exps?
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = __builtin_amdgcn_exp2f(x[i]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(hb, ha, acc, 0, 0, 0);
        }
        if (OP == 8) {                      // MFMA only (2 per iteration, dependent)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(hb, ha, acc, 0, 0, 0);
        }
    }
    float s = acc[0] + acc[7];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    if (s == 123.456f) out[blockIdx.x] = s;
}

// Attention-shaped dependency chain: 6 MFMAs -> (their accumulators feed) ~105 VALU ops -> (whose fp16 results feed) 8 MFMAs.
// How much of the MFMA time hides under the VALU time with 1 / 2 / 4 waves per SIMD?   MODE 0 both, 1 VALU only, 2 MFMA only
template <int MODE>
__global__ __launch_bounds__(256) void phase_kernel(float* out, int iters, float seed) {
    h8 q = {1, 2, 3, 4, 5, 6, 7, 8}, k = {1, 1, 1, 1, 1, 1, 1, 1};
    f16v o0 = {0}, o1 = {0};
    float m = seed;
    for (int it = 0; it < iters; ++it) {
        f16v s0 = {0}, s1 = {0};
        if (MODE != 1) {
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k, q, s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(q, k, s1, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) { s0[r] = m + r; s1[r] = m - r; }
        }
        h8 pf[4];
        if (MODE != 2) {
            float mx = s0[0];
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(fmaxf(mx, s0[r]), s1[r]);
            m = fmaxf(m, mx * 0.01f);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s0[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[r], 0.01f, -m));
                s1[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[r], 0.01f, -m));
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { pf[0][j] = (_Float16)s0[j]; pf[1][j] = (_Float16)s0[8 + j]; pf[2][j] = (_Float16)s1[j]; pf[3][j] = (_Float16)s1[8 + j]; }
        if (MODE != 1) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k, pf[ks], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(q, pf[ks], o1, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) { o0[ks] += (float)pf[ks][0]; o1[ks] += (float)pf[ks][7]; }
        }
    }
    float sres = m;
#pragma unroll
    for (int i = 0; i < 16; ++i) sres += o0[i] + o1[i];
    if (sres == 123.456f) out[blockIdx.x] = sres;
}

// Same work, software-pipelined inside the wave: the S MFMAs of unit u+1 are issued BEFORE the softmax of unit u,
// the PV MFMAs of unit u right after it, so every VALU phase has independent MFMAs in flight.
__global__ __launch_bounds__(256) void phase_pipe_kernel(float* out, int iters, float seed) {
    h8 q = {1, 2, 3, 4, 5, 6, 7, 8}, k = {1, 1, 1, 1, 1, 1, 1, 1};
    f16v o0 = {0}, o1 = {0};
    float m = seed;
    f16v sa0 = {0}, sa1 = {0}, sb0 = {0}, sb1 = {0};
    auto qk = [&](f16v& s0, f16v& s1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k, q, s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(q, k, s1, 0, 0, 0);
        }
    };
    auto softmax_pv = [&](f16v& s0, f16v& s1) {
        float mx = s0[0];
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(fmaxf(mx, s0[r]), s1[r]);
        m = fmaxf(m, mx * 0.01f);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s0[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[r], 0.01f, -m));
            s1[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[r], 0.01f, -m));
        }
        h8 pf[4];
#pragma unroll
        for (int j = 0; j < 8; ++j) { pf[0][j] = (_Float16)s0[j]; pf[1][j] = (_Float16)s0[8 + j]; pf[2][j] = (_Float16)s1[j]; pf[3][j] = (_Float16)s1[8 + j]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k, pf[ks], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(q, pf[ks], o1, 0, 0, 0);
        }
    };
    qk(sa0, sa1);
    for (int it = 0; it < iters; it += 2) {
        qk(sb0, sb1);
        __builtin_amdgcn_sched_barrier(0);
        softmax_pv(sa0, sa1);
        __builtin_amdgcn_sched_barrier(0);
        qk(sa0, sa1);
        __builtin_amdgcn_sched_barrier(0);
        softmax_pv(sb0, sb1);
        __builtin_amdgcn_sched_barrier(0);
    }
    float sres = m;
#pragma unroll
    for (int i = 0; i < 16; ++i) sres += o0[i] + o1[i] + sa0[i];
    if (sres == 123.456f) out[blockIdx.x] = sres;
}

static void run_phase_pipe(const char* name, int waves_per_simd, float* out) {
    const int iters = 2048;
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    phase_pipe_kernel<<<blocks, 256>>>(out, iters, 0.001f);
    CK(hipEventRecord(e0));
    phase_pipe_kernel<<<blocks, 256>>>(out, iters, 0.001f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-46s waves/SIMD %d  %8.1f us  -> %7.1f ns per unit per SIMD\n", name, waves_per_simd, ms * 1e3, ms * 1e6 / ((double)iters * waves_per_simd));
}

template <int MODE>
static void run_phase(const char* name, int waves_per_simd, float* out) {
    const int iters = 2048;
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    phase_kernel<MODE><<<blocks, 256>>>(out, iters, 0.001f);
    CK(hipEventRecord(e0));
    phase_kernel<MODE><<<blocks, 256>>>(out, iters, 0.001f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-46s waves/SIMD %d  %8.1f us  -> %7.1f ns per unit per SIMD\n", name, waves_per_simd, ms * 1e3, ms * 1e6 / ((double)iters * waves_per_simd));
}

template <int OP>
static void run(const char* name, int waves_per_simd, float* out, double ops_per_iter) {
    const int iters = 4096;
    const int blocks = 256 * waves_per_simd;          // 256 threads = one wave per SIMD per block
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    valu_kernel<OP><<<blocks, 256>>>(out, iters, 0.001f);
    CK(hipEventRecord(e0));
    valu_kernel<OP><<<blocks, 256>>>(out, iters, 0.001f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double inst_per_simd = (double)iters * ops_per_iter * waves_per_simd;
    printf("%-46s waves/SIMD %d  %8.1f us  -> %6.2f ns per wave-instruction per SIMD (x2.4 GHz = %5.1f cycles)\n", name, waves_per_simd, ms * 1e3,
           ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4);
}

int main() {
    float* out; CK(hipMalloc(&out, 1 << 20));
    for (int w : {1, 2, 3, 4}) {
        run_phase<0>("attention-shaped unit: 14 MFMA + softmax VALU", w, out);
        run_phase_pipe("  same, S(u+1) issued before softmax(u)", w, out);
        run_phase<1>("  softmax VALU only", w, out);
        run_phase<2>("  14 MFMA only", w, out);
    }
    for (int w = 4; w < 4; ++w) {
        run<0>("v_exp_f32", w, out, 16);
        run<1>("v_fma_f32", w, out, 16);
        run<2>("v_pk_fma_f32 (2 values per lane)", w, out, 8);
        run<3>("v_max3_f32", w, out, 16);
        run<4>("v_cvt_pkrtz_f16_f32 + cvt + add (3 ops)", w, out, 48);
        run<5>("v_rcp_f32", w, out, 16);
        run<6>("v_exp_f16 (+2 cvt)", w, out, 48);
        run<7>("16 v_exp_f32 + 2 mfma_32x32x16 (per 18 instr)", w, out, 18);
        run<8>("2 mfma_32x32x16 dependent", w, out, 2);
    }
    return 0;
}
