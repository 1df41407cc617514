// Implicit-GEMM on the gfx950 matrix cores: one kernel body, two activation loaders.
//
//   MODE 0 (dense)   C[m,n] = sum_k A[m,k] W[n,k]          Linear / 1x1 conv / bmm
//   MODE 1 (conv3x3) m = (b,oy,ox), k = (tap, ci)          NHWC 3x3 conv gather, zero padding,
//                                                          optional fused nearest-2x upsample and
//                                                          two-source channel concat
//
// Tiling (CDNA4, wave64): 256 threads = 4 waves as 2(m) x 2(n); block tile BM x BN x 64;
// each wave owns (BM/2) x (BN/2) as 16x16 MFMA tiles of v_mfma_f32_16x16x32_f16.  The weights
// are the MFMA "A" operand and the activations the "B" operand, so an accumulator register
// quad holds 4 consecutive n for one m (packs to one 8-byte fp16 store).
// LDS: rows of 64 halfs (128 B), 16-B chunks XOR-swizzled by (row & 7) -> ds_read_b128 fragment
// reads and ds_write_b128 staging writes are bank-conflict free.  Global -> register -> LDS
// staging with the next k-tile's loads in flight under the current tile's MFMAs, two LDS
// buffers, one barrier per k-tile.  The C tile is staged through LDS so HBM sees whole
// 16-byte-per-lane row segments (bias / row-broadcast / activation applied in fp32 registers
// first, the residual added on the way out).
#include "common.h"
#include "../../include/pbe_hip.h"

struct IGemmP {
    const h16* A; const h16* A2; const h16* W; h16* C;
    const float* bias; const h16* rowvec; const h16* resid;
    int M, N, K, K1;
    long lda, lda2, ldw, ldc, ldr;
    int ldv, group_rows;
    long sA, sW, sC, sR;
    float alpha; int act; int bias_row; int vec;
    // conv gather
    int H, Wd, C1, C2, Ho, Wo, cstride, pad, ups;
};

template <int BM, int BN, int MODE>
__global__ void __launch_bounds__(256) igemm_kernel(const IGemmP p) {
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 16, TN = WN / 16;
    constexpr int NA = BM * 8 / 256, NW = BN * 8 / 256;
    constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, BUF = A_BYTES + W_BYTES;
    constexpr int CLD = BN + 8;  // C tile leading dim (halfs)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const long bz = blockIdx.z;

    const int chunk = tid & 7, rbase = tid >> 3;                 // this thread's 16-B column / first row
    const int st_off = rbase * 128 + ((chunk ^ (rbase & 7)) << 4);  // + 32*128*i per extra row

    // ---- per-row state of the activation loader ----
    bool a_ok[NA];
    const h16* a_row[NA];   // dense: row base (A)      conv: unused
    const h16* a_row2[NA];  // dense: row base (A2)
    int cb[NA], cy[NA], cx[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int m = m0 + rbase + 32 * i;
        a_ok[i] = m < p.M;
        if (MODE == 0) {
            a_row[i] = p.A + bz * p.sA + (long)m * p.lda;
            a_row2[i] = p.A2 ? p.A2 + (long)m * p.lda2 : p.A;
            cb[i] = cy[i] = cx[i] = 0;
        } else {
            const int hw = p.Ho * p.Wo;
            const int b = m / hw, rem = m - b * hw;
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            cb[i] = b; cy[i] = oy * p.cstride - p.pad; cx[i] = ox * p.cstride - p.pad;
            a_row[i] = a_row2[i] = p.A;
        }
    }
    bool w_ok[NW];
    const h16* w_row[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int n = n0 + rbase + 32 * i;
        w_ok[i] = n < p.N;
        w_row[i] = p.W + bz * p.sW + (long)(w_ok[i] ? n : 0) * p.ldw;
    }

    const int nk = (p.K + 63) >> 6;
    const int Cin = p.C1 + p.C2;
    const h16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    h16x8 ra[NA], rw[NW];
    int tap = 0, c0 = 0;  // conv: position of the NEXT tile to load

    auto load_tile = [&](int kt) {
        const int k = kt * 64 + chunk * 8;
        const bool kok = k < p.K;
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const bool ok = a_ok[i] && kok;
                const h16* src = (k < p.K1) ? a_row[i] + k : a_row2[i] + (k - p.K1);
                src = ok ? src : p.W;
                h16x8 v = *reinterpret_cast<const h16x8*>(src);
                ra[i] = ok ? v : zero8;
            }
        } else {
            const int dy = tap / 3, dx = tap - 3 * dy;
            const int cc = c0 + chunk * 8;
            const int Hv = p.H << p.ups, Wv = p.Wd << p.ups;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int iy = cy[i] + dy, ix = cx[i] + dx;
                const bool ok = a_ok[i] && kok && (unsigned)iy < (unsigned)Hv && (unsigned)ix < (unsigned)Wv;
                const long pix = ((long)cb[i] * p.H + (iy >> p.ups)) * p.Wd + (ix >> p.ups);
                const h16* src = (cc < p.C1) ? p.A + pix * p.C1 + cc : p.A2 + pix * p.C2 + (cc - p.C1);
                src = ok ? src : p.W;
                h16x8 v = *reinterpret_cast<const h16x8*>(src);
                ra[i] = ok ? v : zero8;
            }
            c0 += 64;
            if (c0 >= Cin) { c0 = 0; ++tap; }
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const bool ok = w_ok[i] && kok;
            const h16* src = ok ? w_row[i] + k : p.W;
            h16x8 v = *reinterpret_cast<const h16x8*>(src);
            rw[i] = ok ? v : zero8;
        }
    };
    auto store_tile = [&](int buf) {
        unsigned char* sa = smem + buf * BUF;
        unsigned char* sw = sa + A_BYTES;
#pragma unroll
        for (int i = 0; i < NA; ++i) *reinterpret_cast<h16x8*>(sa + st_off + i * 32 * 128) = ra[i];
#pragma unroll
        for (int i = 0; i < NW; ++i) *reinterpret_cast<h16x8*>(sw + st_off + i * 32 * 128) = rw[i];
    };

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragment read offsets: row = base + (lane&15), 16-B chunk (ks*4 + lane>>4) ^ (row&7)
    const int fr = lane & 15, fq = lane >> 4;
    const int a_rd = (wm * WM + fr) * 128, w_rd = (wn * WN + fr) * 128;
    const int sw0 = ((0 + fq) ^ (fr & 7)) << 4, sw1 = ((4 + fq) ^ (fr & 7)) << 4;

    load_tile(0);
    store_tile(0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        const unsigned char* sa = smem + cur * BUF;
        const unsigned char* sw = sa + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int so = ks ? sw1 : sw0;
            h16x8 fa[TM], fw[TN];
#pragma unroll
            for (int j = 0; j < TM; ++j) fa[j] = *reinterpret_cast<const h16x8*>(sa + a_rd + j * 16 * 128 + so);
#pragma unroll
            for (int i = 0; i < TN; ++i) fw[i] = *reinterpret_cast<const h16x8*>(sw + w_rd + i * 16 * 128 + so);
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[i], fa[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tile(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: registers (alpha, bias, row-broadcast, act) -> fp16 C tile in LDS ----
    h16* sC = reinterpret_cast<h16*>(smem);
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const int nl = wn * WN + i * 16 + fq * 4;
        const int n = n0 + nl;
        float bn[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.bias && !p.bias_row) {
#pragma unroll
            for (int r = 0; r < 4; ++r) bn[r] = (n + r < p.N) ? p.bias[n + r] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int ml = wm * WM + j * 16 + fr;
            const int m = m0 + ml;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] * p.alpha + bn[r];
            if (p.bias && p.bias_row) {
                const float bm = (m < p.M) ? p.bias[m] : 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += bm;
            }
            if (p.rowvec && m < p.M) {
                const h16* rv = p.rowvec + (long)(m / p.group_rows) * p.ldv + n;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < p.N) v[r] += (float)rv[r];
            }
            h16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (h16)apply_act(v[r], p.act);
            *reinterpret_cast<h16x4*>(sC + ml * CLD + nl) = o;
        }
    }
    __syncthreads();

    // ---- C tile -> global, whole 16-byte row segments, residual fused ----
    constexpr int CPR = BN / 8;
    constexpr int PER = BM * CPR / 256;
    h16* Cb = p.C + bz * p.sC;
    const h16* Rb = p.resid ? p.resid + bz * p.sR : nullptr;
#pragma unroll
    for (int it = 0; it < PER; ++it) {
        const int idx = tid + 256 * it;
        const int row = idx / CPR, ch = idx - row * CPR;
        const int m = m0 + row, n = n0 + ch * 8;
        if (m >= p.M || n >= p.N) continue;
        h16x8 v = *reinterpret_cast<const h16x8*>(sC + row * CLD + ch * 8);
        if (p.vec) {
            if (Rb) {
                const h16x8 r = *reinterpret_cast<const h16x8*>(Rb + (long)m * p.ldr + n);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (h16)((float)v[e] + (float)r[e]);
            }
            *reinterpret_cast<h16x8*>(Cb + (long)m * p.ldc + n) = v;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (n + e < p.N) {
                    float f = (float)v[e];
                    if (Rb) f += (float)Rb[(long)m * p.ldr + n + e];
                    Cb[(long)m * p.ldc + n + e] = (h16)f;
                }
            }
        }
    }
}

// ---- host side --------------------------------------------------------------------------------

template <int BM, int BN, int MODE>
static void launch_igemm(const IGemmP& p, int batch, hipStream_t s) {
    constexpr size_t main_bytes = 2 * (BM + BN) * 128;
    constexpr size_t c_bytes = (size_t)BM * (BN + 8) * 2;
    constexpr size_t lds = main_bytes > c_bytes ? main_bytes : c_bytes;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<BM, BN, MODE>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    dim3 grid(cdiv(p.M, BM), cdiv(p.N, BN), batch);
    hipLaunchKernelGGL((igemm_kernel<BM, BN, MODE>), grid, dim3(256), lds, s, p);
}

// Pick the block tile: prefer big tiles, but not when padding waste or a part-filled last
// wave of workgroups (256 CUs x 2 resident blocks) costs more than the smaller tile's lower
// MFMA:LDS ratio.
static void pick_tile(long M, long N, int batch, int* bm, int* bn) {
    static const int cand[4][2] = {{128, 128}, {128, 64}, {64, 128}, {64, 64}};
    static const double eff[4] = {1.0, 0.86, 0.86, 0.72};
    double best = -1.0;
    for (int c = 0; c < 4; ++c) {
        const long tm = (M + cand[c][0] - 1) / cand[c][0], tn = (N + cand[c][1] - 1) / cand[c][1];
        const double tiles = (double)tm * tn * batch;
        const double useful = (double)M * N * batch / (tiles * cand[c][0] * cand[c][1]);
        const double slots = 512.0;
        const double rounds = (double)((long)((tiles + slots - 1) / slots));
        const double quant = tiles / (rounds * slots);
        const double score = eff[c] * useful * (0.35 + 0.65 * quant);
        if (score > best) { best = score; *bm = cand[c][0]; *bn = cand[c][1]; }
    }
}

template <int MODE>
static int dispatch_igemm(const IGemmP& p, int batch, hipStream_t s) {
    int bm = 128, bn = 128;
    pick_tile(p.M, p.N, batch, &bm, &bn);
    if (bm == 128 && bn == 128) launch_igemm<128, 128, MODE>(p, batch, s);
    else if (bm == 128 && bn == 64) launch_igemm<128, 64, MODE>(p, batch, s);
    else if (bm == 64 && bn == 128) launch_igemm<64, 128, MODE>(p, batch, s);
    else launch_igemm<64, 64, MODE>(p, batch, s);
    return 0;
}

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int pbe_gemm_f16(const pbe_gemm_desc* d, pbe_stream_t stream) {
    PBE_REQUIRE(d && d->A && d->W && d->C, "pbe_gemm_f16: null operand");
    PBE_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0 && d->batch >= 1, "pbe_gemm_f16: bad dims M=%d N=%d K=%d batch=%d", d->M, d->N, d->K, d->batch);
    PBE_REQUIRE(d->K % 8 == 0, "pbe_gemm_f16: K=%d must be a multiple of 8", d->K);
    PBE_REQUIRE(d->lda % 8 == 0 && d->ldw % 8 == 0 && al16(d->A) && al16(d->W), "pbe_gemm_f16: A/W must be 16-byte aligned with ld %% 8 == 0");
    PBE_REQUIRE(d->strideA % 8 == 0 && d->strideW % 8 == 0, "pbe_gemm_f16: batch strides of A/W must be multiples of 8");
    const int K1 = d->A2 ? d->K1 : d->K;
    if (d->A2) {
        PBE_REQUIRE(K1 > 0 && K1 < d->K && K1 % 64 == 0 && d->lda2 % 8 == 0 && al16(d->A2) && d->batch == 1,
                    "pbe_gemm_f16: split-K source needs K1 %% 64 == 0 (K1=%d), aligned A2, batch 1", K1);
    }
    PBE_REQUIRE(d->lda >= (d->A2 ? K1 : d->K) && d->ldw >= d->K && d->ldc >= d->N, "pbe_gemm_f16: leading dims too small");
    PBE_REQUIRE(!d->rowvec || d->group_rows > 0, "pbe_gemm_f16: rowvec needs group_rows > 0");
    IGemmP p;
    memset(&p, 0, sizeof(p));
    p.A = (const h16*)d->A; p.A2 = (const h16*)d->A2; p.W = (const h16*)d->W; p.C = (h16*)d->C;
    p.bias = d->bias; p.rowvec = (const h16*)d->rowvec; p.resid = (const h16*)d->resid;
    p.M = d->M; p.N = d->N; p.K = d->K; p.K1 = K1;
    p.lda = d->lda; p.lda2 = d->lda2; p.ldw = d->ldw; p.ldc = d->ldc; p.ldr = d->ldr;
    p.ldv = d->ldv; p.group_rows = d->group_rows > 0 ? d->group_rows : 1;
    p.sA = d->strideA; p.sW = d->strideW; p.sC = d->strideC; p.sR = d->strideR;
    p.alpha = d->alpha; p.act = d->act; p.bias_row = d->bias_per_row;
    p.vec = (d->N % 8 == 0) && (d->ldc % 8 == 0) && al16(d->C) && (d->strideC % 8 == 0) &&
            (!d->resid || ((d->ldr % 8 == 0) && al16(d->resid) && (d->strideR % 8 == 0)));
    hipStream_t s = (hipStream_t)stream;
    pbe_prof_begin(PBE_K_GEMM, s);
    dispatch_igemm<0>(p, d->batch, s);
    pbe_prof_end(PBE_K_GEMM, s, 2.0 * d->M * (double)d->N * d->K * d->batch);
    PBE_LAUNCH_CHECK("pbe_gemm_f16");
    return PBE_OK;
}

extern "C" int pbe_conv3x3_f16(const pbe_conv3x3_desc* d, pbe_stream_t stream) {
    PBE_REQUIRE(d && d->X && d->Wp && d->Y, "pbe_conv3x3_f16: null operand");
    const int Cin = d->C1 + d->C2;
    PBE_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Cout > 0, "pbe_conv3x3_f16: bad dims");
    PBE_REQUIRE(d->C1 > 0 && d->C1 % 64 == 0 && d->C2 >= 0 && d->C2 % 64 == 0, "pbe_conv3x3_f16: C1=%d C2=%d must be multiples of 64 (use pbe_im2col3x3_f16 + pbe_gemm_f16 for small Cin)", d->C1, d->C2);
    PBE_REQUIRE((d->C2 == 0) == (d->X2 == nullptr), "pbe_conv3x3_f16: X2 / C2 mismatch");
    PBE_REQUIRE(d->stride == 1 || d->stride == 2, "pbe_conv3x3_f16: stride must be 1 or 2");
    PBE_REQUIRE(d->pad == 0 || d->pad == 1, "pbe_conv3x3_f16: pad must be 0 or 1");
    PBE_REQUIRE(d->upsample == 0 || (d->upsample == 1 && d->stride == 1), "pbe_conv3x3_f16: upsample only with stride 1");
    PBE_REQUIRE(al16(d->X) && al16(d->Wp) && al16(d->Y) && (!d->X2 || al16(d->X2)), "pbe_conv3x3_f16: 16-byte alignment");
    const int Hv = d->H << d->upsample, Wv = d->W << d->upsample;
    // output size: pad=1 -> floor((Hv + 2 - 3)/s) + 1 ; pad=0 is the VAE (0,1,0,1) pad: floor((Hv + 1 - 3)/s) + 1
    const int extra = d->pad ? 2 : 1;
    const int Ho = (Hv + extra - 3) / d->stride + 1, Wo = (Wv + extra - 3) / d->stride + 1;
    PBE_REQUIRE(Ho > 0 && Wo > 0, "pbe_conv3x3_f16: empty output");
    const long M = (long)d->B * Ho * Wo;
    PBE_REQUIRE(M < (1L << 31), "pbe_conv3x3_f16: too many output pixels");
    IGemmP p;
    memset(&p, 0, sizeof(p));
    p.A = (const h16*)d->X; p.A2 = (const h16*)d->X2; p.W = (const h16*)d->Wp; p.C = (h16*)d->Y;
    p.bias = d->bias; p.rowvec = (const h16*)d->rowvec; p.resid = (const h16*)d->resid;
    p.M = (int)M; p.N = d->Cout; p.K = 9 * Cin; p.K1 = p.K;
    p.ldw = p.K; p.ldc = d->Cout; p.ldr = d->Cout;
    p.ldv = d->ldv; p.group_rows = Ho * Wo;
    p.alpha = 1.f; p.act = d->act; p.bias_row = 0;
    p.vec = (d->Cout % 8 == 0) && (!d->resid || al16(d->resid));
    p.H = d->H; p.Wd = d->W; p.C1 = d->C1; p.C2 = d->C2; p.Ho = Ho; p.Wo = Wo;
    p.cstride = d->stride; p.pad = d->pad; p.ups = d->upsample;
    hipStream_t s = (hipStream_t)stream;
    pbe_prof_begin(PBE_K_CONV3, s);
    dispatch_igemm<1>(p, 1, s);
    pbe_prof_end(PBE_K_CONV3, s, 2.0 * (double)M * d->Cout * 9.0 * Cin);
    PBE_LAUNCH_CHECK("pbe_conv3x3_f16");
    return PBE_OK;
}
