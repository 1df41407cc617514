// Small HBM / launch-latency bound kernels around the matrix-core ops: layout conversion at the
// NCHW API boundary, im2col for the three tiny-Cin convs, GEGLU, timestep embedding, the fused
// PLMS sampler update, VAE posterior sampling, CLIP patchify.
#include "common.h"
#include "../../include/pbe_hip.h"

#define EW_GRID(n) dim3((unsigned)(((n) + 255) / 256))
#define EW_BEGIN(s) pbe_prof_begin(PBE_K_ELEM, s)
#define EW_END(s, bytes, name) \
    pbe_prof_end(PBE_K_ELEM, s, bytes); \
    PBE_LAUNCH_CHECK(name); \
    return PBE_OK

// ---- NCHW fp32 <-> NHWC fp16 -------------------------------------------------------------------
__global__ void nchw2nhwc_kernel(const float* src, h16* dst, int C, int HW, int Cp, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;     // over B*HW pixels
    if (i >= total) return;
    const long b = i / HW, px = i - b * HW;
    h16* d = dst + i * Cp;
    for (int c = 0; c < Cp; ++c) d[c] = (h16)(c < C ? src[(b * C + c) * HW + px] : 0.f);
}
extern "C" int pbe_nchw_f32_to_nhwc_f16(const float* src, void* dst, int32_t B, int32_t C, int32_t HW, int32_t Cp, pbe_stream_t stream) {
    PBE_REQUIRE(src && dst && B > 0 && C > 0 && HW > 0 && Cp >= C, "pbe_nchw_f32_to_nhwc_f16: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)B * HW;
    EW_BEGIN(s);
    hipLaunchKernelGGL(nchw2nhwc_kernel, EW_GRID(total), dim3(256), 0, s, src, (h16*)dst, C, HW, Cp, total);
    EW_END(s, (double)total * (4.0 * C + 2.0 * Cp), "pbe_nchw_f32_to_nhwc_f16");
}

__global__ void nhwc2nchw_kernel(const h16* src, float* dst, int C, int HW, int ld, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long b = i / HW, px = i - b * HW;
    const h16* s = src + i * ld;
    for (int c = 0; c < C; ++c) dst[(b * C + c) * HW + px] = (float)s[c];
}
extern "C" int pbe_nhwc_f16_to_nchw_f32(const void* src, float* dst, int32_t B, int32_t C, int32_t HW, int32_t ld, pbe_stream_t stream) {
    PBE_REQUIRE(src && dst && B > 0 && C > 0 && HW > 0 && ld >= C, "pbe_nhwc_f16_to_nchw_f32: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)B * HW;
    EW_BEGIN(s);
    hipLaunchKernelGGL(nhwc2nchw_kernel, EW_GRID(total), dim3(256), 0, s, (const h16*)src, dst, C, HW, ld, total);
    EW_END(s, (double)total * 6.0 * C, "pbe_nhwc_f16_to_nchw_f32");
}

// ---- im2col 3x3 for tiny Cin (Cp in {8,16}) ------------------------------------------------------
__global__ void im2col3x3_kernel(const h16* X, h16* out, int H, int W, int Cp, int Ho, int Wo, int stride, int pad, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;     // over M * 9 (pixel, tap)
    if (i >= total) return;
    const long m = i / 9;
    const int tap = (int)(i - m * 9), dy = tap / 3, dx = tap - dy * 3;
    const int hw = Ho * Wo;
    const long b = m / hw;
    const int rem = (int)(m - b * hw), oy = rem / Wo, ox = rem - oy * Wo;
    const int iy = oy * stride + dy - pad, ix = ox * stride + dx - pad;
    const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    const h16x8* src = reinterpret_cast<const h16x8*>(X + ((b * H + (ok ? iy : 0)) * W + (ok ? ix : 0)) * Cp);
    h16x8* dst = reinterpret_cast<h16x8*>(out + (m * 9 + tap) * Cp);
    const h16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int v = 0; v < Cp / 8; ++v) dst[v] = ok ? src[v] : z;
}
extern "C" int pbe_im2col3x3_f16(const void* X, void* out, int32_t B, int32_t H, int32_t W, int32_t Cp, int32_t stride, int32_t pad,
                                 pbe_stream_t stream) {
    PBE_REQUIRE(X && out && B > 0 && H > 0 && W > 0, "pbe_im2col3x3_f16: bad arguments");
    PBE_REQUIRE(Cp % 8 == 0 && Cp > 0, "pbe_im2col3x3_f16: Cp must be a multiple of 8");
    PBE_REQUIRE((stride == 1 || stride == 2) && (pad == 0 || pad == 1), "pbe_im2col3x3_f16: stride/pad");
    const int extra = pad ? 2 : 1;
    const int Ho = (H + extra - 3) / stride + 1, Wo = (W + extra - 3) / stride + 1;
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)B * Ho * Wo * 9;
    EW_BEGIN(s);
    hipLaunchKernelGGL(im2col3x3_kernel, EW_GRID(total), dim3(256), 0, s, (const h16*)X, (h16*)out, H, W, Cp, Ho, Wo, stride, pad, total);
    EW_END(s, (double)total * Cp * 4.0, "pbe_im2col3x3_f16");
}

// ---- GEGLU ---------------------------------------------------------------------------------------
__global__ void geglu_kernel(const h16* Hh, h16* Y, int F8, long F, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;     // over M * F/8
    if (i >= total) return;
    const long m = i / F8;
    const int v = (int)(i - m * F8);
    const h16x8 a = *reinterpret_cast<const h16x8*>(Hh + m * 2 * F + v * 8);
    const h16x8 g = *reinterpret_cast<const h16x8*>(Hh + m * 2 * F + F + v * 8);
    h16x8 y;
#pragma unroll
    for (int e = 0; e < 8; ++e) y[e] = (h16)((float)a[e] * gelu_erf_f((float)g[e]));
    *reinterpret_cast<h16x8*>(Y + m * F + v * 8) = y;
}
extern "C" int pbe_geglu_f16(const void* Hh, void* Y, int64_t M, int32_t F, pbe_stream_t stream) {
    PBE_REQUIRE(Hh && Y && M > 0 && F > 0 && F % 8 == 0, "pbe_geglu_f16: bad arguments (F %% 8 == 0)");
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)M * (F / 8);
    EW_BEGIN(s);
    hipLaunchKernelGGL(geglu_kernel, EW_GRID(total), dim3(256), 0, s, (const h16*)Hh, (h16*)Y, F / 8, (long)F, total);
    EW_END(s, (double)M * F * 6.0, "pbe_geglu_f16");
}

// ---- timestep embedding --------------------------------------------------------------------------
__global__ void temb_kernel(const int64_t* t, h16* out, int dim, float neg_log_period_over_half, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;     // over B * dim
    if (i >= total) return;
    const int b = (int)(i / dim), j = (int)(i - (long)b * dim), half = dim / 2;
    float v = 0.f;
    if (j < 2 * half) {
        const int k = j < half ? j : j - half;
        const float freq = expf(neg_log_period_over_half * (float)k);
        const float arg = (float)t[b] * freq;
        v = j < half ? cosf(arg) : sinf(arg);
    }
    out[i] = (h16)v;
}
extern "C" int pbe_timestep_embedding_f16(const int64_t* t, void* out, int32_t B, int32_t dim, float max_period, pbe_stream_t stream) {
    PBE_REQUIRE(t && out && B > 0 && dim >= 2 && max_period > 1.f, "pbe_timestep_embedding_f16: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)B * dim;
    EW_BEGIN(s);
    hipLaunchKernelGGL(temb_kernel, EW_GRID(total), dim3(256), 0, s, t, (h16*)out, dim, -logf(max_period) / (float)(dim / 2), total);
    EW_END(s, (double)total * 2.0, "pbe_timestep_embedding_f16");
}

// ---- PLMS sampler ------------------------------------------------------------------------------
__global__ void plms_pack_kernel(const float* x, const float* z, const float* mask, h16* x9, int B, int HW, int dup, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;     // over B*HW
    if (i >= total) return;
    const long b = i / HW, px = i - b * HW;
    h16 v[16];
#pragma unroll
    for (int c = 0; c < 4; ++c) { v[c] = (h16)x[(b * 4 + c) * HW + px]; v[4 + c] = (h16)z[(b * 4 + c) * HW + px]; }
    v[8] = (h16)mask[b * HW + px];
#pragma unroll
    for (int c = 9; c < 16; ++c) v[c] = (h16)0.f;
    const h16x8 lo = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
    const h16x8 hi = {v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15]};
    for (int d = 0; d < dup; ++d) {
        h16x8* dst = reinterpret_cast<h16x8*>(x9 + ((long)d * B * HW + i) * 16);
        dst[0] = lo; dst[1] = hi;
    }
}
extern "C" int pbe_plms_pack_input(const float* x, const float* z_inpaint, const float* mask, void* x9, int32_t B, int32_t HW, int32_t dup,
                                   pbe_stream_t stream) {
    PBE_REQUIRE(x && z_inpaint && mask && x9 && B > 0 && HW > 0 && (dup == 1 || dup == 2), "pbe_plms_pack_input: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)B * HW;
    EW_BEGIN(s);
    hipLaunchKernelGGL(plms_pack_kernel, EW_GRID(total), dim3(256), 0, s, x, z_inpaint, mask, (h16*)x9, B, HW, dup, total);
    EW_END(s, (double)total * (36.0 + 32.0 * dup), "pbe_plms_pack_input");
}

struct PlmsCoef { float c[8]; };
__global__ void plms_update_kernel(const h16* eps, int ld, int dup, float cfg, const float* x, const float* h1, const float* h2,
                                   const float* h3, PlmsCoef k, float* e_t, float* x_prev, float* pred_x0, int B, int HW, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;     // over B*4*HW (NCHW order)
    if (i >= total) return;
    const long b = i / (4L * HW);
    const int rem = (int)(i - b * 4L * HW), c = rem / HW, px = rem - c * HW;
    const long tok = b * HW + px;
    float e;
    if (dup == 2) {
        const float eu = (float)eps[tok * ld + c], ec = (float)eps[((long)B * HW + tok) * ld + c];
        e = eu + cfg * (ec - eu);
    } else {
        e = (float)eps[tok * ld + c];
    }
    float ep = k.c[0] * e;
    if (h1) ep += k.c[1] * h1[i];
    if (h2) ep += k.c[2] * h2[i];
    if (h3) ep += k.c[3] * h3[i];
    const float px0 = (x[i] - k.c[4] * ep) * k.c[5];
    if (e_t) e_t[i] = e;
    if (pred_x0) pred_x0[i] = px0;
    x_prev[i] = k.c[6] * px0 + k.c[7] * ep;
}
extern "C" int pbe_plms_update(const void* eps_out, int32_t ld, int32_t dup, float cfg_scale, const float* x, const float* h1, const float* h2,
                               const float* h3, const float* coef8, float* e_t, float* x_prev, float* pred_x0, int32_t B, int32_t HW,
                               pbe_stream_t stream) {
    PBE_REQUIRE(eps_out && x && coef8 && x_prev && B > 0 && HW > 0 && ld >= 4 && (dup == 1 || dup == 2), "pbe_plms_update: bad arguments");
    PlmsCoef k;
    for (int i = 0; i < 8; ++i) k.c[i] = coef8[i];
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)B * 4 * HW;
    EW_BEGIN(s);
    hipLaunchKernelGGL(plms_update_kernel, EW_GRID(total), dim3(256), 0, s, (const h16*)eps_out, ld, dup, cfg_scale, x, h1, h2, h3, k, e_t,
                       x_prev, pred_x0, B, HW, total);
    EW_END(s, (double)total * 32.0, "pbe_plms_update");
}

// ---- stochastic sampler options: DDIM eta > 0 noise term (ddim.py:226-238), mask / x0 blending through q_sample (plms.py:150-153) ----
__global__ void axpy_kernel(float* y, float a, const float* x, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < total) y[i] = __builtin_fmaf(a, x[i], y[i]);
}
extern "C" int pbe_axpy_f32(float* y, float a, const float* x, int64_t n, pbe_stream_t stream) {
    PBE_REQUIRE(y && x && n > 0, "pbe_axpy_f32: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    EW_BEGIN(s);
    hipLaunchKernelGGL(axpy_kernel, EW_GRID(n), dim3(256), 0, s, y, a, x, (long)n);
    EW_END(s, (double)n * 12.0, "pbe_axpy_f32");
}
__global__ void qsample_blend_kernel(const float* x0, const float* noise, const float* mask, const float* img, float a, float b, float* out, int C,
                                     int HW, int mask_c, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;     // over B*C*HW (NCHW)
    if (i >= total) return;
    const long bi = i / ((long)C * HW);
    const int rem = (int)(i - bi * (long)C * HW), c = rem / HW, px = rem - c * HW;
    const float m = mask[(bi * mask_c + (mask_c == 1 ? 0 : c)) * HW + px];
    const float orig = a * x0[i] + b * noise[i];             // q_sample(x0, t) (ddpm.py:337-341)
    out[i] = orig * m + (1.f - m) * img[i];
}
extern "C" int pbe_qsample_blend_f32(const float* x0, const float* noise, const float* mask, const float* img, float sqrt_ac, float sqrt_1m_ac,
                                     float* out, int32_t B, int32_t C, int32_t HW, int32_t mask_channels, pbe_stream_t stream) {
    PBE_REQUIRE(x0 && noise && mask && img && out && B > 0 && C > 0 && HW > 0 && (mask_channels == 1 || mask_channels == C), "pbe_qsample_blend_f32: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)B * C * HW;
    EW_BEGIN(s);
    hipLaunchKernelGGL(qsample_blend_kernel, EW_GRID(total), dim3(256), 0, s, x0, noise, mask, img, sqrt_ac, sqrt_1m_ac, out, C, HW, mask_channels, total);
    EW_END(s, (double)total * 20.0, "pbe_qsample_blend_f32");
}

// ---- VAE posterior sample / latent un-scale / image post ---------------------------------------
__global__ void posterior_kernel(const h16* mom, int ld, const float* eps, float* z, int HW, float scale, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;     // over B*4*HW
    if (i >= total) return;
    const long b = i / (4L * HW);
    const int rem = (int)(i - b * 4L * HW), c = rem / HW, px = rem - c * HW;
    const h16* m = mom + (b * HW + px) * ld;
    const float mean = (float)m[c];
    const float logvar = fminf(fmaxf((float)m[4 + c], -30.f), 20.f);
    z[i] = scale * (mean + expf(0.5f * logvar) * eps[i]);
}
extern "C" int pbe_posterior_sample(const void* moments, int32_t ld, const float* eps, float* z, int32_t B, int32_t HW, float scale,
                                    pbe_stream_t stream) {
    PBE_REQUIRE(moments && eps && z && B > 0 && HW > 0 && ld >= 8, "pbe_posterior_sample: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)B * 4 * HW;
    EW_BEGIN(s);
    hipLaunchKernelGGL(posterior_kernel, EW_GRID(total), dim3(256), 0, s, (const h16*)moments, ld, eps, z, HW, scale, total);
    EW_END(s, (double)total * 12.0, "pbe_posterior_sample");
}

__global__ void scale_latent_kernel(const float* z, h16* out, int C, int HW, float inv_scale, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;     // over B*HW
    if (i >= total) return;
    const long b = i / HW, px = i - b * HW;
    h16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = (h16)(z[(b * C + c) * HW + px] * inv_scale);
    *reinterpret_cast<h16x8*>(out + i * 8) = v;
}
extern "C" int pbe_scale_latent_f16(const float* z, void* out, int32_t B, int32_t C, int32_t HW, float inv_scale, pbe_stream_t stream) {
    PBE_REQUIRE(z && out && B > 0 && C >= 4 && HW > 0, "pbe_scale_latent_f16: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)B * HW;
    EW_BEGIN(s);
    hipLaunchKernelGGL(scale_latent_kernel, EW_GRID(total), dim3(256), 0, s, z, (h16*)out, C, HW, inv_scale, total);
    EW_END(s, (double)total * 32.0, "pbe_scale_latent_f16");
}

__global__ void image_post_kernel(const h16* src, float* dst, int HW, int ld, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;     // over B*HW
    if (i >= total) return;
    const long b = i / HW, px = i - b * HW;
#pragma unroll
    for (int c = 0; c < 3; ++c) dst[(b * 3 + c) * HW + px] = fminf(fmaxf(((float)src[i * ld + c] + 1.f) * 0.5f, 0.f), 1.f);
}
extern "C" int pbe_image_post_f32(const void* src, float* dst, int32_t B, int32_t HW, int32_t ld, pbe_stream_t stream) {
    PBE_REQUIRE(src && dst && B > 0 && HW > 0 && ld >= 3, "pbe_image_post_f32: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)B * HW;
    EW_BEGIN(s);
    hipLaunchKernelGGL(image_post_kernel, EW_GRID(total), dim3(256), 0, s, (const h16*)src, dst, HW, ld, total);
    EW_END(s, (double)total * 18.0, "pbe_image_post_f32");
}

// ---- CLIP patchify / class-token row ------------------------------------------------------------
__global__ void clip_patchify_kernel(const float* px, h16* out, int S, int P, int Kp, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;     // over B*G*G*Kp
    if (i >= total) return;
    const int G = S / P;
    const long tokn = i / Kp;
    const int k = (int)(i - tokn * Kp);
    float v = 0.f;
    if (k < 3 * P * P) {
        const long b = tokn / (G * G);
        const int t = (int)(tokn - b * G * G), gy = t / G, gx = t - gy * G;
        const int c = k / (P * P), r = k - c * P * P, ky = r / P, kx = r - ky * P;
        v = px[((b * 3 + c) * S + gy * P + ky) * S + gx * P + kx];
    }
    out[i] = (h16)v;
}
extern "C" int pbe_clip_patchify_f16(const float* pixels, void* out, int32_t B, int32_t S, int32_t P, int32_t Kp, pbe_stream_t stream) {
    PBE_REQUIRE(pixels && out && B > 0 && S > 0 && P > 0 && S % P == 0 && Kp >= 3 * P * P && Kp % 8 == 0, "pbe_clip_patchify_f16: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)B * (S / P) * (S / P) * Kp;
    EW_BEGIN(s);
    hipLaunchKernelGGL(clip_patchify_kernel, EW_GRID(total), dim3(256), 0, s, pixels, (h16*)out, S, P, Kp, total);
    EW_END(s, (double)total * 6.0, "pbe_clip_patchify_f16");
}

__global__ void bcast_row_kernel(const h16* a, const h16* b, h16* Y, int C, long y_bs, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;     // over B*C
    if (i >= total) return;
    const long bb = i / C;
    const int c = (int)(i - bb * C);
    Y[bb * y_bs + c] = (h16)((float)a[c] + (float)b[c]);
}
extern "C" int pbe_bcast_row_f16(const void* a, const void* b, void* Y, int32_t B, int32_t C, int64_t y_bs, pbe_stream_t stream) {
    PBE_REQUIRE(a && b && Y && B > 0 && C > 0, "pbe_bcast_row_f16: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)B * C;
    EW_BEGIN(s);
    hipLaunchKernelGGL(bcast_row_kernel, EW_GRID(total), dim3(256), 0, s, (const h16*)a, (const h16*)b, (h16*)Y, C, (long)y_bs, total);
    EW_END(s, (double)total * 6.0, "pbe_bcast_row_f16");
}

// ---- bilinear resize of fp32 planes (mask 512x512 -> 64x64: scripts/inference.py:332 `Resize([h, w])`) -------------------------
// antialias = 1: the triangle filter torchvision >= 0.17 applies to tensors (support = scale, weights normalised per output
// pixel; row sums first, then the column combination - the order of ATen's upsample_bilinear2d_aa); antialias = 0: plain 2-tap
// bilinear, align_corners = False (torchvision 0.12, the reference's pinned version).  One thread per output pixel.
#define RS_MAX_TAPS 64
__global__ void resize_bilinear_kernel(const float* src, float* dst, int Hin, int Win, int Hout, int Wout, float sh, float sw, int aa, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int ox = (int)(i % Wout);
    const long t = i / Wout;
    const int oy = (int)(t % Hout);
    const float* pl = src + (t / Hout) * (long)Hin * Win;
    if (!aa) {
        const float fy = fmaxf(sh * (oy + 0.5f) - 0.5f, 0.f), fx = fmaxf(sw * (ox + 0.5f) - 0.5f, 0.f);
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < Hin - 1 ? 1 : 0), x1 = x0 + (x0 < Win - 1 ? 1 : 0);
        const float ly = fy - y0, lx = fx - x0;
        dst[i] = (1.f - ly) * ((1.f - lx) * pl[(long)y0 * Win + x0] + lx * pl[(long)y0 * Win + x1]) +
                 ly * ((1.f - lx) * pl[(long)y1 * Win + x0] + lx * pl[(long)y1 * Win + x1]);
        return;
    }
    const float sup_h = sh >= 1.f ? sh : 1.f, sup_w = sw >= 1.f ? sw : 1.f;
    const float inv_h = sh >= 1.f ? 1.f / sh : 1.f, inv_w = sw >= 1.f ? 1.f / sw : 1.f;
    const float cy = sh * (oy + 0.5f), cx = sw * (ox + 0.5f);
    const int ymin = max((int)(cy - sup_h + 0.5f), 0), ysize = min((int)(cy + sup_h + 0.5f), Hin) - ymin;
    const int xmin = max((int)(cx - sup_w + 0.5f), 0), xsize = min((int)(cx + sup_w + 0.5f), Win) - xmin;
    float wx[RS_MAX_TAPS];
    float tw = 0.f;
    for (int j = 0; j < xsize; ++j) {
        const float a = fabsf((j + xmin - cx + 0.5f) * inv_w);
        wx[j] = a < 1.f ? 1.f - a : 0.f;
        tw += wx[j];
    }
    const float nx = tw != 0.f ? 1.f / tw : 0.f;
    float wy_tot = 0.f, acc = 0.f;
    for (int k = 0; k < ysize; ++k) {
        const float a = fabsf((k + ymin - cy + 0.5f) * inv_h);
        wy_tot += a < 1.f ? 1.f - a : 0.f;
    }
    const float ny = wy_tot != 0.f ? 1.f / wy_tot : 0.f;
    for (int k = 0; k < ysize; ++k) {
        const float a = fabsf((k + ymin - cy + 0.5f) * inv_h);
        const float wy = (a < 1.f ? 1.f - a : 0.f) * ny;
        const float* row = pl + (long)(ymin + k) * Win + xmin;
        float r = 0.f;
        for (int j = 0; j < xsize; ++j) r += row[j] * (wx[j] * nx);
        acc += r * wy;
    }
    dst[i] = acc;
}
extern "C" int pbe_resize_bilinear_f32(const float* src, float* dst, int32_t planes, int32_t Hin, int32_t Win, int32_t Hout, int32_t Wout,
                                       int32_t antialias, pbe_stream_t stream) {
    PBE_REQUIRE(src && dst && planes > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0, "pbe_resize_bilinear_f32: bad arguments");
    const float sh = (float)Hin / (float)Hout, sw = (float)Win / (float)Wout;
    PBE_REQUIRE(!antialias || (2.f * fmaxf(sw, 1.f) + 2.f <= (float)RS_MAX_TAPS), "pbe_resize_bilinear_f32: horizontal scale %.2f needs more than %d taps", sw, RS_MAX_TAPS);
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)planes * Hout * Wout;
    EW_BEGIN(s);
    hipLaunchKernelGGL(resize_bilinear_kernel, EW_GRID(total), dim3(256), 0, s, src, dst, Hin, Win, Hout, Wout, sh, sw, antialias ? 1 : 0, total);
    EW_END(s, 4.0 * ((double)planes * Hin * Win + (double)total), "pbe_resize_bilinear_f32");
}

// ---- image I/O on the device (scripts/inference.py:305-322, 346-399) ---------------------------------
// u8 pixels travel both ways; normalisation, masking, un-normalisation, clamp, the uint8 pack and the 4-tile grid run here.
// pbe_u8_to_planes_f32: dst[b, c, y, x] = f(src[b, y, x, c] / 255) with f(v) = v * a[c] + b[c]   (ToTensor + Normalize)
//                       or, with binarize = 1, f(v) = (1 - v) < 0.5 ? 0 : 1                      (inference.py:311-315 mask)
//                       or, with binarize = 2, f(v) = 1 - v                                       (test_bench_dataset.py: not thresholded)
// The float ops are the reference's, in its order (divide by 255, subtract mean, divide by std): (v - mean) / std, not an FMA.
__global__ void u8_to_planes_kernel(const unsigned char* src, float* dst, int C, int HW, float m0, float m1, float m2, float s0, float s1, float s2,
                                    int binarize, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;           // over B * HW pixels
    if (i >= total) return;
    const long b = i / HW, px = i - b * HW;
    const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
    for (int c = 0; c < C; ++c) {
        float v = (float)src[i * C + c] / 255.0f;
        if (binarize == 1) v = (1.0f - v) < 0.5f ? 0.0f : 1.0f;
        else if (binarize == 2) v = 1.0f - v;
        else { v = v - mean[c]; asm volatile("" : "+v"(v)); v = v / sd[c]; }
        dst[(b * C + c) * HW + px] = v;
    }
}
extern "C" int pbe_u8_to_planes_f32(const void* src, float* dst, int32_t B, int32_t C, int32_t HW, const float* mean3, const float* std3,
                                    int32_t binarize, pbe_stream_t stream) {
    PBE_REQUIRE(src && dst && B > 0 && (C == 1 || C == 3) && HW > 0, "pbe_u8_to_planes_f32: bad arguments");
    PBE_REQUIRE(binarize || (mean3 && std3), "pbe_u8_to_planes_f32: mean / std needed");
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)B * HW;
    const float m[3] = {mean3 ? mean3[0] : 0.f, mean3 && C > 1 ? mean3[1] : 0.f, mean3 && C > 2 ? mean3[2] : 0.f};
    const float d[3] = {std3 ? std3[0] : 1.f, std3 && C > 1 ? std3[1] : 1.f, std3 && C > 2 ? std3[2] : 1.f};
    EW_BEGIN(s);
    hipLaunchKernelGGL(u8_to_planes_kernel, EW_GRID(total), dim3(256), 0, s, (const unsigned char*)src, dst, C, HW, m[0], m[1], m[2], d[0], d[1], d[2], binarize, total);
    EW_END(s, (double)total * C * 5.0, "pbe_u8_to_planes_f32");
}

// pbe_mul_planes_f32: out[b, c, :] = x[b, c, :] * m[b, 0, :]   (inpaint_image = image * mask, inference.py:319)
__global__ void mul_planes_kernel(const float* x, const float* m, float* out, int C, int HW, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;           // over B * C * HW
    if (i >= total) return;
    const long bc = i / HW, px = i - bc * HW, b = bc / C;
    out[i] = x[i] * m[b * HW + px];
}
extern "C" int pbe_mul_planes_f32(const float* x, const float* m, float* out, int32_t B, int32_t C, int32_t HW, pbe_stream_t stream) {
    PBE_REQUIRE(x && m && out && B > 0 && C > 0 && HW > 0, "pbe_mul_planes_f32: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)B * C * HW;
    EW_BEGIN(s);
    hipLaunchKernelGGL(mul_planes_kernel, EW_GRID(total), dim3(256), 0, s, x, m, out, C, HW, total);
    EW_END(s, (double)total * 12.0, "pbe_mul_planes_f32");
}

// pbe_planes_to_u8_canvas: canvas[y0 + y, x0 + x, c] = (uint8) trunc(255 * clamp(src[c, y, x] * a[c] + b[c], 0, 1))  for one CHW plane
// set (src channel stride HW; `bcast` = 1 repeats channel 0 into the 3 output channels: the mask file).  Truncation, multiply THEN
// add (two roundings), like the reference's `(255. * x.numpy()).astype(np.uint8)` after `x * std + mean` / `(x + 1) / 2`.
__global__ void planes_to_canvas_kernel(const float* src, unsigned char* canvas, int H, int W, int Wc, int y0, int x0, float a0, float a1, float a2,
                                        float b0, float b1, float b2, int bcast, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;           // over H * W pixels
    if (i >= total) return;
    const int y = (int)(i / W), x = (int)(i - (long)y * W);
    const float a[3] = {a0, a1, a2}, b[3] = {b0, b1, b2};
    unsigned char* d = canvas + ((long)(y0 + y) * Wc + (x0 + x)) * 3;
    for (int c = 0; c < 3; ++c) {
        float v = src[(long)(bcast ? 0 : c) * H * W + i] * a[c];
        asm volatile("" : "+v"(v));
        v = v + b[c];
        v = fminf(fmaxf(v, 0.0f), 1.0f);
        d[c] = (unsigned char)(255.0f * v);
    }
}
extern "C" int pbe_planes_to_u8_canvas(const float* src, void* canvas, int32_t H, int32_t W, int32_t Hc, int32_t Wc, int32_t y0, int32_t x0,
                                       const float* a3, const float* b3, int32_t bcast, pbe_stream_t stream) {
    PBE_REQUIRE(src && canvas && a3 && b3 && H > 0 && W > 0 && y0 >= 0 && x0 >= 0 && y0 + H <= Hc && x0 + W <= Wc, "pbe_planes_to_u8_canvas: tile outside the canvas");
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)H * W;
    EW_BEGIN(s);
    hipLaunchKernelGGL(planes_to_canvas_kernel, EW_GRID(total), dim3(256), 0, s, src, (unsigned char*)canvas, H, W, Wc, y0, x0, a3[0], a3[1], a3[2], b3[0], b3[1],
                       b3[2], bcast, total);
    EW_END(s, (double)total * 15.0, "pbe_planes_to_u8_canvas");
}
