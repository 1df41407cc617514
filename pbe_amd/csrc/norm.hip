// HBM-bound normalisation kernels: GroupNorm(+SiLU) over NHWC, LayerNorm, row softmax.
// All loads/stores are 16 bytes per lane (8 fp16), statistics in fp32 (finalised in fp64).
#include "common.h"
#include "../../include/pbe_hip.h"

// ------------------------------------------------------------------------------------------------
// GroupNorm.  NHWC rows are contiguous, so thread (ty, tx) with tx = 8-channel vector index and
// ty = row-in-pass addresses x + (row * C8 + tx) * 16 B: consecutive lanes -> consecutive 16-B
// segments.  Pass 1 (gn_stats): per-block partial (sum, sumsq) per group -> workspace.
// Pass 2 (gn_apply): finalise mean / rstd per (b, group) from the partials, then stream
// y = silu((x - mean) * rstd * gamma + beta).
// The input may be a channel concat of two tensors (vector index < C1/8 -> X, else X2).
// ------------------------------------------------------------------------------------------------
#define GN_MAX_C 4096
#define GN_MAX_CHUNKS 256

__device__ __forceinline__ const h16* gn_src(const h16* X, const h16* X2, int C1, int C2, long row, int v) {
    const int c = v * 8;
    return (c < C1) ? X + row * C1 + c : X2 + row * C2 + (c - C1);
}

__global__ void __launch_bounds__(256) gn_stats_kernel(const h16* X, const h16* X2, float* part, int HW, int C1, int C2,
                                                        int groups, int rows_per_block, int TX, int TY) {
    // Deterministic reduction (bit-identical run to run): per-thread partials go to LDS and are
    // summed over ty in a fixed order; no atomics.
    __shared__ float s_sum[GN_MAX_C], s_sq[GN_MAX_C];
    __shared__ float s_pa[256 * 8], s_pq[256 * 8];
    const int C = C1 + C2, C8 = C >> 3, cg = C / groups;
    const int tid = threadIdx.x, tx = tid % TX, ty = tid / TX;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int r0 = chunk * rows_per_block;
    const int r1 = min(HW, r0 + rows_per_block);
    if (ty < TY) {
        for (int v = tx; v < C8; v += TX) {            // TY > 1 implies C8 == TX: exactly one pass
            float a[8], q[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { a[e] = 0.f; q[e] = 0.f; }
            int r = r0 + ty;
            for (; r + 3 * TY < r1; r += 4 * TY) {      // 4 independent 16-byte loads in flight per lane
                h16x8 x[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) x[u] = *reinterpret_cast<const h16x8*>(gn_src(X, X2, C1, C2, (long)b * HW + r + u * TY, v));
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float f = (float)x[u][e]; a[e] += f; q[e] += f * f; }
            }
            for (; r < r1; r += TY) {
                const h16x8 x = *reinterpret_cast<const h16x8*>(gn_src(X, X2, C1, C2, (long)b * HW + r, v));
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float f = (float)x[e]; a[e] += f; q[e] += f * f; }
            }
            if (TY == 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { s_sum[v * 8 + e] = a[e]; s_sq[v * 8 + e] = q[e]; }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) { s_pa[tid * 8 + e] = a[e]; s_pq[tid * 8 + e] = q[e]; }
            }
        }
    }
    __syncthreads();
    if (TY > 1) {
        for (int c = tid; c < C; c += 256) {
            const int v = c >> 3, e = c & 7;
            float a = 0.f, q = 0.f;
            for (int y = 0; y < TY; ++y) { a += s_pa[(y * TX + v) * 8 + e]; q += s_pq[(y * TX + v) * 8 + e]; }
            s_sum[c] = a; s_sq[c] = q;
        }
        __syncthreads();
    }
    if (tid < groups) {
        float a = 0.f, q = 0.f;
        for (int c = tid * cg; c < (tid + 1) * cg; ++c) { a += s_sum[c]; q += s_sq[c]; }
        float* dst = part + (((long)b * gridDim.x + chunk) * groups + tid) * 2;
        dst[0] = a; dst[1] = q;
    }
}

__global__ void __launch_bounds__(256) gn_apply_kernel(const h16* X, const h16* X2, const float* part, const float* gamma,
                                                        const float* beta, h16* Y, int HW, int C1, int C2, int groups,
                                                        int nchunks, int rows_per_block, int TX, int TY, float eps, int silu) {
    __shared__ float s_mean[64], s_rstd[64];
    __shared__ double s_a[256], s_q[256];
    const int C = C1 + C2, C8 = C >> 3, cg = C / groups;
    const int tid = threadIdx.x, tx = tid % TX, ty = tid / TX;
    const int b = blockIdx.y;
    {   // finalise statistics: thread -> (group g, slice sl of the partial list)
        const int slices = 256 / groups;                 // groups <= 64
        const int g = tid % groups, sl = tid / groups;
        double a = 0.0, q = 0.0;
        if (sl < slices)
            for (int c = sl; c < nchunks; c += slices) {
                const float* src = part + (((long)b * nchunks + c) * groups + g) * 2;
                a += (double)src[0]; q += (double)src[1];
            }
        s_a[tid] = a; s_q[tid] = q;
        __syncthreads();
        if (tid < groups) {
            double ta = 0.0, tq = 0.0;
            for (int s = 0; s < slices; ++s) { ta += s_a[s * groups + tid]; tq += s_q[s * groups + tid]; }
            const double n = (double)HW * cg;
            const double mean = ta / n;
            double var = tq / n - mean * mean;
            if (var < 0.0) var = 0.0;
            s_mean[tid] = (float)mean;
            s_rstd[tid] = (float)(1.0 / sqrt(var + (double)eps));
        }
        __syncthreads();
    }
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(HW, r0 + rows_per_block);
    if (ty >= TY) return;
    for (int v = tx; v < C8; v += TX) {
        float sc[8], sh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = v * 8 + e, g = c / cg;
            const float a = s_rstd[g] * gamma[c];
            sc[e] = a; sh[e] = beta[c] - s_mean[g] * a;
        }
        int r = r0 + ty;
        for (; r + 3 * TY < r1; r += 4 * TY) {
            h16x8 x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) x[u] = *reinterpret_cast<const h16x8*>(gn_src(X, X2, C1, C2, (long)b * HW + r + u * TY, v));
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                h16x8 y;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float f = (float)x[u][e] * sc[e] + sh[e];
                    if (silu) f = silu_f(f);
                    y[e] = (h16)f;
                }
                *reinterpret_cast<h16x8*>(Y + ((long)b * HW + r + u * TY) * C + v * 8) = y;
            }
        }
        for (; r < r1; r += TY) {
            const long row = (long)b * HW + r;
            const h16x8 x = *reinterpret_cast<const h16x8*>(gn_src(X, X2, C1, C2, row, v));
            h16x8 y;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float f = (float)x[e] * sc[e] + sh[e];
                if (silu) f = silu_f(f);
                y[e] = (h16)f;
            }
            *reinterpret_cast<h16x8*>(Y + row * C + v * 8) = y;
        }
    }
}

// Small feature maps (the 16x16 / 8x8 U-Net levels): one workgroup per (sample, group) keeps its
// HW x (C/groups) slice in registers, so statistics and normalisation are ONE launch and one read.
// Needs C/groups % 8 == 0 and HW <= GN_SMALL_ROWS * (256 / (C/groups/8)).
#define GN_SMALL_ROWS 12
__global__ void __launch_bounds__(256) gn_small_kernel(const h16* X, const h16* X2, const float* gamma, const float* beta, h16* Y,
                                                        int HW, int C1, int C2, int groups, float eps, int silu) {
    __shared__ float s_a[4], s_q[4];
    const int C = C1 + C2, cg = C / groups, TX = cg >> 3, TY = 256 / TX;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tx = tid % TX, ty = tid / TX;
    const int g = blockIdx.x, b = blockIdx.y;
    const int v = (g * cg >> 3) + tx;
    const bool act = ty < TY;
    h16x8 x[GN_SMALL_ROWS];
    float a = 0.f, q = 0.f;
#pragma unroll
    for (int u = 0; u < GN_SMALL_ROWS; ++u) {
        const int r = ty + u * TY;
        if (act && r < HW) {
            x[u] = *reinterpret_cast<const h16x8*>(gn_src(X, X2, C1, C2, (long)b * HW + r, v));
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float f = (float)x[u][e]; a += f; q += f * f; }
        }
    }
    a = wave_sum(a); q = wave_sum(q);
    if (lane == 0) { s_a[wave] = a; s_q[wave] = q; }
    __syncthreads();
    const double n = (double)HW * cg;
    const double mean = ((double)s_a[0] + s_a[1] + s_a[2] + s_a[3]) / n;
    double var = ((double)s_q[0] + s_q[1] + s_q[2] + s_q[3]) / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float mu = (float)mean, rstd = (float)(1.0 / sqrt(var + (double)eps));
    if (!act) return;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = v * 8 + e;
        sc[e] = rstd * gamma[c];
        sh[e] = beta[c] - mu * sc[e];
    }
#pragma unroll
    for (int u = 0; u < GN_SMALL_ROWS; ++u) {
        const int r = ty + u * TY;
        if (r < HW) {
            h16x8 y;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float f = (float)x[u][e] * sc[e] + sh[e];
                if (silu) f = silu_f(f);
                y[e] = (h16)f;
            }
            *reinterpret_cast<h16x8*>(Y + ((long)b * HW + r) * C + v * 8) = y;
        }
    }
}

int g_pbe_gn_rows = 16;      // pbe_tune(7, n): rows per thread of the two-pass GroupNorm kernels (developer knob)
int g_pbe_gn_apply2_rows = 8; // pbe_tune(10, n): rows per thread of the two-pass norm's normalisation pass (0 = the statistics pass's geometry); 8: -2..4 % with the input resident
int g_pbe_gn_apply_rows = 4; // pbe_tune(9, n): rows per thread of the normalisation pass when the statistics come from the producing conv
static void gn_geometry(int HW, int C, int* nchunks, int* rpb, int* TX, int* TY, int rows_per_thread = 0) {
    const int C8 = C / 8;
    *TX = C8 < 256 ? C8 : 256;
    *TY = 256 / *TX;
    int rows = *TY * (rows_per_thread > 0 ? rows_per_thread : g_pbe_gn_rows);           // 16 rows per thread: four 4-deep load batches
    if (rows < 32) rows = 32;
    int n = (HW + rows - 1) / rows;
    if (n > GN_MAX_CHUNKS) { n = GN_MAX_CHUNKS; rows = (HW + n - 1) / n; n = (HW + rows - 1) / rows; }
    *nchunks = n; *rpb = rows;
}

extern "C" size_t pbe_groupnorm_workspace_bytes(int32_t B, int32_t HW) {
    (void)HW;
    return (size_t)B * GN_MAX_CHUNKS * 64 * 2 * sizeof(float);
}

extern "C" int pbe_groupnorm_f16(const void* X, const void* X2, const float* gamma, const float* beta, void* Y, int32_t B,
                                 int32_t HW, int32_t C1, int32_t C2, int32_t groups, float eps, int32_t silu, void* workspace,
                                 size_t workspace_bytes, pbe_stream_t stream) {
    const int C = C1 + C2;
    PBE_REQUIRE(X && gamma && beta && Y && workspace, "pbe_groupnorm_f16: null operand");
    PBE_REQUIRE(B > 0 && HW > 0 && C1 > 0 && C2 >= 0, "pbe_groupnorm_f16: bad dims");
    PBE_REQUIRE((C2 == 0) == (X2 == nullptr), "pbe_groupnorm_f16: X2 / C2 mismatch");
    PBE_REQUIRE(C1 % 8 == 0 && C2 % 8 == 0 && C <= GN_MAX_C, "pbe_groupnorm_f16: C1=%d C2=%d must be multiples of 8, total <= %d", C1, C2, GN_MAX_C);
    PBE_REQUIRE(groups > 0 && groups <= 64 && C % groups == 0, "pbe_groupnorm_f16: groups=%d must divide C=%d (<= 64)", groups, C);
    PBE_REQUIRE(B <= 65535, "pbe_groupnorm_f16: batch too large");
    PBE_REQUIRE(workspace_bytes >= pbe_groupnorm_workspace_bytes(B, HW), "pbe_groupnorm_f16: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int cg = C / groups;
    if (cg % 8 == 0 && cg / 8 <= 64 && HW <= GN_SMALL_ROWS * (256 / (cg / 8))) {
        pbe_prof_begin(PBE_K_GNORM, s);
        hipLaunchKernelGGL(gn_small_kernel, dim3(groups, B), dim3(256), 0, s, (const h16*)X, (const h16*)X2, gamma, beta, (h16*)Y, HW, C1, C2,
                           groups, eps, silu);
        pbe_prof_end(PBE_K_GNORM, s, 4.0 * B * (double)HW * C);
        PBE_LAUNCH_CHECK("pbe_groupnorm_f16");
        return PBE_OK;
    }
    int nchunks, rpb, TX, TY;
    gn_geometry(HW, C, &nchunks, &rpb, &TX, &TY);
    pbe_prof_begin(PBE_K_GNORM, s);
    hipLaunchKernelGGL(gn_stats_kernel, dim3(nchunks, B), dim3(256), 0, s, (const h16*)X, (const h16*)X2, (float*)workspace, HW, C1, C2,
                       groups, rpb, TX, TY);
    {   // the normalisation pass runs a finer grid than the statistics pass (pbe_tune(10, rows per thread), 0 = the statistics pass's geometry)
        int ac = nchunks, arpb = rpb, aTX = TX, aTY = TY;
        if (g_pbe_gn_apply2_rows > 0) gn_geometry(HW, C, &ac, &arpb, &aTX, &aTY, g_pbe_gn_apply2_rows);
        hipLaunchKernelGGL(gn_apply_kernel, dim3(ac, B), dim3(256), 0, s, (const h16*)X, (const h16*)X2, (const float*)workspace, gamma,
                           beta, (h16*)Y, HW, C1, C2, groups, nchunks, arpb, aTX, aTY, eps, silu);
    }
    pbe_prof_end(PBE_K_GNORM, s, 6.0 * B * (double)HW * C);   // bytes: read x twice, write y once
    PBE_LAUNCH_CHECK("pbe_groupnorm_f16");
    return PBE_OK;
}

extern "C" int pbe_groupnorm_apply_f16(const void* X, const float* partials, int32_t blocks, const float* gamma, const float* beta, void* Y,
                                       int32_t B, int32_t HW, int32_t C, int32_t groups, float eps, int32_t silu, pbe_stream_t stream) {
    PBE_REQUIRE(X && partials && gamma && beta && Y, "pbe_groupnorm_apply_f16: null operand");
    PBE_REQUIRE(B > 0 && B <= 65535 && HW > 0 && C > 0 && C % 8 == 0 && C <= GN_MAX_C && blocks > 0, "pbe_groupnorm_apply_f16: bad dims");
    PBE_REQUIRE(groups > 0 && groups <= 64 && C % groups == 0, "pbe_groupnorm_apply_f16: groups=%d must divide C=%d (<= 64)", groups, C);
    hipStream_t s = (hipStream_t)stream;
    // (the finalise prologue of a workgroup reads `blocks` partials per group - 16 at 64x64, not one per chunk of this grid - so short
    //  workgroups are cheap here: 4 rows per thread = 5 workgroups per CU at [8, 4096, 320] against 1.3 with the two-pass geometry)
    int nchunks, rpb, TX, TY;
    gn_geometry(HW, C, &nchunks, &rpb, &TX, &TY, g_pbe_gn_apply_rows);
    pbe_prof_begin(PBE_K_GNORM, s);
    hipLaunchKernelGGL(gn_apply_kernel, dim3(nchunks, B), dim3(256), 0, s, (const h16*)X, (const h16*)nullptr, partials, gamma, beta, (h16*)Y, HW, C, 0,
                       groups, blocks, rpb, TX, TY, eps, silu);
    pbe_prof_end(PBE_K_GNORM, s, 4.0 * B * (double)HW * C);   // bytes: read x once, write y once
    PBE_LAUNCH_CHECK("pbe_groupnorm_apply_f16");
    return PBE_OK;
}

// ------------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, the row held in registers (<= 4 x 8 halfs per lane), two wave
// reductions (mean, then centred variance — same two-pass form as torch's CPU kernel).
// ------------------------------------------------------------------------------------------------
template <int VPL>
__global__ void __launch_bounds__(256) layernorm_kernel(const h16* X, const float* gamma, const float* beta, h16* Y, long rows, int C,
                                                         long ldx, long ldy, float eps) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int C8 = C >> 3;
    h16x8 x[VPL];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int v = lane + 64 * i;
        if (v < C8) {
            x[i] = *reinterpret_cast<const h16x8*>(X + row * ldx + v * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += (float)x[i][e];
        }
    }
    const float mean = wave_sum(sum) / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int v = lane + 64 * i;
        if (v < C8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = (float)x[i][e] - mean; sq += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int v = lane + 64 * i;
        if (v < C8) {
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + v * 8), g1 = *reinterpret_cast<const f32x4*>(gamma + v * 8 + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + v * 8), b1 = *reinterpret_cast<const f32x4*>(beta + v * 8 + 4);
            h16x8 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y[e] = (h16)(((float)x[i][e] - mean) * rstd * g0[e] + b0[e]);
                y[4 + e] = (h16)(((float)x[i][4 + e] - mean) * rstd * g1[e] + b1[e]);
            }
            *reinterpret_cast<h16x8*>(Y + row * ldy + v * 8) = y;
        }
    }
}

extern "C" int pbe_layernorm_f16(const void* X, const float* gamma, const float* beta, void* Y, int64_t rows, int32_t C, int64_t ldx,
                                 int64_t ldy, float eps, pbe_stream_t stream) {
    PBE_REQUIRE(X && gamma && beta && Y, "pbe_layernorm_f16: null operand");
    PBE_REQUIRE(rows > 0 && C > 0 && C % 8 == 0 && C <= 2048, "pbe_layernorm_f16: C=%d must be a multiple of 8, <= 2048", C);
    PBE_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0 && ldx >= C && ldy >= C, "pbe_layernorm_f16: bad leading dims");
    PBE_REQUIRE(((uintptr_t)gamma & 15) == 0 && ((uintptr_t)beta & 15) == 0 && ((uintptr_t)X & 15) == 0 && ((uintptr_t)Y & 15) == 0, "pbe_layernorm_f16: alignment");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    pbe_prof_begin(PBE_K_LNORM, s);
    const int vpl = (C / 8 + 63) / 64;
    if (vpl <= 1) hipLaunchKernelGGL(layernorm_kernel<1>, grid, block, 0, s, (const h16*)X, gamma, beta, (h16*)Y, (long)rows, C, (long)ldx, (long)ldy, eps);
    else if (vpl == 2) hipLaunchKernelGGL(layernorm_kernel<2>, grid, block, 0, s, (const h16*)X, gamma, beta, (h16*)Y, (long)rows, C, (long)ldx, (long)ldy, eps);
    else if (vpl == 3) hipLaunchKernelGGL(layernorm_kernel<3>, grid, block, 0, s, (const h16*)X, gamma, beta, (h16*)Y, (long)rows, C, (long)ldx, (long)ldy, eps);
    else hipLaunchKernelGGL(layernorm_kernel<4>, grid, block, 0, s, (const h16*)X, gamma, beta, (h16*)Y, (long)rows, C, (long)ldx, (long)ldy, eps);
    pbe_prof_end(PBE_K_LNORM, s, 4.0 * (double)rows * C);
    PBE_LAUNCH_CHECK("pbe_layernorm_f16");
    return PBE_OK;
}

// Row statistics alone: (sum, sum of squares) of every fp16 row as ONE float2 partial - what the LayerNorm-folding GEMM
// (pbe_gemm_desc.ln_stats) needs when the producer of x did not emit them from its own epilogue (row_stats_out).
template <int VPL>
__global__ void __launch_bounds__(256) row_stats_kernel(const h16* X, float2* out, long rows, int C, long ldx) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int C8 = C >> 3;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int v = lane + 64 * i;
        if (v < C8) {
            const h16x8 x = *reinterpret_cast<const h16x8*>(X + row * ldx + v * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float f = (float)x[e]; s1 += f; s2 = __builtin_fmaf(f, f, s2); }
        }
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    if (lane == 0) out[row] = make_float2(s1, s2);
}

extern "C" int pbe_row_stats_f16(const void* X, float* out, int64_t rows, int32_t C, int64_t ldx, pbe_stream_t stream) {
    PBE_REQUIRE(X && out && rows > 0 && C > 0 && C % 8 == 0 && C <= 2048 && ldx % 8 == 0 && ldx >= C && ((uintptr_t)X & 15) == 0 && ((uintptr_t)out & 7) == 0,
                "pbe_row_stats_f16: bad arguments (C %% 8 == 0, <= 2048, aligned rows)");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    pbe_prof_begin(PBE_K_LNORM, s);
    const int vpl = (C / 8 + 63) / 64;
#define PBE_RS(V) hipLaunchKernelGGL(row_stats_kernel<V>, grid, block, 0, s, (const h16*)X, (float2*)out, (long)rows, C, (long)ldx)
    if (vpl <= 1) PBE_RS(1); else if (vpl == 2) PBE_RS(2); else if (vpl == 3) PBE_RS(3); else PBE_RS(4);
#undef PBE_RS
    pbe_prof_end(PBE_K_LNORM, s, 2.0 * (double)rows * C);
    PBE_LAUNCH_CHECK("pbe_row_stats_f16");
    return PBE_OK;
}

// LayerNorm with an fp8 (OCP e4m3) output and one scale per row: y8[r, :] = e4m3(LN(x[r, :]) / s[r]), s[r] = max|LN(x[r, :])| / 448.
// Feeds the fp8 form of pbe_gemm_f16 (a_scale = s): the normalised row never exists in fp16 in HBM (half the bytes of the
// fp16 LayerNorm's write and of the GEMM's A read).  Same two-pass statistics as layernorm_kernel.
template <int VPL>
__global__ void __launch_bounds__(256) layernorm_f8_kernel(const h16* X, const float* gamma, const float* beta, unsigned char* Y, float* S,
                                                            long rows, int C, long ldx, long ldy, float eps) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int C8 = C >> 3;
    float y[VPL][8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int v = lane + 64 * i;
        if (v < C8) {
            const h16x8 x = *reinterpret_cast<const h16x8*>(X + row * ldx + v * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) { y[i][e] = (float)x[e]; sum += y[i][e]; }
        }
    }
    const float mean = wave_sum(sum) / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int v = lane + 64 * i;
        if (v < C8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = y[i][e] - mean; sq += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int v = lane + 64 * i;
        if (v < C8) {
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + v * 8), g1 = *reinterpret_cast<const f32x4*>(gamma + v * 8 + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + v * 8), b1 = *reinterpret_cast<const f32x4*>(beta + v * 8 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y[i][e] = (y[i][e] - mean) * rstd * g0[e] + b0[e];
                y[i][4 + e] = (y[i][4 + e] - mean) * rstd * g1[e] + b1[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(y[i][e]));
        }
    }
    amax = wave_max(amax);
    const float scale = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
    const float inv = 1.0f / scale;
    if (lane == 0) S[row] = scale;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int v = lane + 64 * i;
        if (v < C8) {
            unsigned int w0 = 0, w1 = 0;
            w0 = __builtin_amdgcn_cvt_pk_fp8_f32(y[i][0] * inv, y[i][1] * inv, w0, false);
            w0 = __builtin_amdgcn_cvt_pk_fp8_f32(y[i][2] * inv, y[i][3] * inv, w0, true);
            w1 = __builtin_amdgcn_cvt_pk_fp8_f32(y[i][4] * inv, y[i][5] * inv, w1, false);
            w1 = __builtin_amdgcn_cvt_pk_fp8_f32(y[i][6] * inv, y[i][7] * inv, w1, true);
            *reinterpret_cast<uint2*>(Y + row * ldy + v * 8) = make_uint2(w0, w1);
        }
    }
}

extern "C" int pbe_layernorm_f8(const void* X, const float* gamma, const float* beta, void* Y, float* row_scale, int64_t rows, int32_t C,
                                int64_t ldx, int64_t ldy, float eps, pbe_stream_t stream) {
    PBE_REQUIRE(X && gamma && beta && Y && row_scale, "pbe_layernorm_f8: null operand");
    PBE_REQUIRE(rows > 0 && C > 0 && C % 16 == 0 && C <= 2048, "pbe_layernorm_f8: C=%d must be a multiple of 16, <= 2048", C);
    PBE_REQUIRE(ldx % 8 == 0 && ldy % 16 == 0 && ldx >= C && ldy >= C, "pbe_layernorm_f8: bad leading dims (ldy in bytes, multiple of 16)");
    PBE_REQUIRE(((uintptr_t)gamma & 15) == 0 && ((uintptr_t)beta & 15) == 0 && ((uintptr_t)X & 15) == 0 && ((uintptr_t)Y & 15) == 0, "pbe_layernorm_f8: alignment");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    pbe_prof_begin(PBE_K_LNORM, s);
    const int vpl = (C / 8 + 63) / 64;
#define PBE_LN8(V) hipLaunchKernelGGL(layernorm_f8_kernel<V>, grid, block, 0, s, (const h16*)X, gamma, beta, (unsigned char*)Y, row_scale, (long)rows, C, (long)ldx, (long)ldy, eps)
    if (vpl <= 1) PBE_LN8(1); else if (vpl == 2) PBE_LN8(2); else if (vpl == 3) PBE_LN8(3); else PBE_LN8(4);
#undef PBE_LN8
    pbe_prof_end(PBE_K_LNORM, s, 3.0 * (double)rows * C);
    PBE_LAUNCH_CHECK("pbe_layernorm_f8");
    return PBE_OK;
}

// ------------------------------------------------------------------------------------------------
// Row softmax with a scale (VAE mid-block attention scores): one workgroup per row, three
// sweeps over the row (max, sum, write); the 8-18 KB row stays in L2 between sweeps.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) softmax_rows_kernel(const h16* X, h16* Y, int cols, long ldx, long ldy, float scale_log2e) {
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const h16* x = X + (long)blockIdx.x * ldx;
    h16* y = Y + (long)blockIdx.x * ldy;
    const int nv = cols >> 3;
    float mx = -INFINITY;
    for (int v = tid; v < nv; v += 256) {
        const h16x8 a = *reinterpret_cast<const h16x8*>(x + v * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) mx = fmaxf(mx, (float)a[e]);
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * scale_log2e;
    __syncthreads();
    float sum = 0.f;
    for (int v = tid; v < nv; v += 256) {
        const h16x8 a = *reinterpret_cast<const h16x8*>(x + v * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) sum += __builtin_amdgcn_exp2f((float)a[e] * scale_log2e - mx);
    }
    sum = wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
    for (int v = tid; v < nv; v += 256) {
        const h16x8 a = *reinterpret_cast<const h16x8*>(x + v * 8);
        h16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (h16)(__builtin_amdgcn_exp2f((float)a[e] * scale_log2e - mx) * inv);
        *reinterpret_cast<h16x8*>(y + v * 8) = o;
    }
}

extern "C" int pbe_softmax_rows_f16(const void* X, void* Y, int64_t rows, int32_t cols, int64_t ldx, int64_t ldy, float scale,
                                    pbe_stream_t stream) {
    PBE_REQUIRE(X && Y && rows > 0 && cols > 0, "pbe_softmax_rows_f16: bad arguments");
    PBE_REQUIRE(cols % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0, "pbe_softmax_rows_f16: cols / ld must be multiples of 8");
    PBE_REQUIRE(scale > 0.f, "pbe_softmax_rows_f16: scale must be positive");
    PBE_REQUIRE(rows < (1L << 31), "pbe_softmax_rows_f16: too many rows");
    hipStream_t s = (hipStream_t)stream;
    pbe_prof_begin(PBE_K_SOFTMAX, s);
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, s, (const h16*)X, (h16*)Y, cols, (long)ldx, (long)ldy,
                       scale * 1.4426950408889634f);
    pbe_prof_end(PBE_K_SOFTMAX, s, 4.0 * (double)rows * cols);
    PBE_LAUNCH_CHECK("pbe_softmax_rows_f16");
    return PBE_OK;
}
