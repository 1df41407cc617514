// Host side of the implicit-GEMM entry points: descriptor checks, plan queries, developer knobs.  The kernel and its planner are in
// igemm_kernel.h; the instantiations in igemm_{dense,conv,halo,f8,ex}.hip.
#include "igemm_kernel.h"

#ifdef PBE_STAMPS
unsigned long long* g_pbe_stamps = nullptr;
extern "C" int pbe_debug_set_stamps(void* buf) { g_pbe_stamps = (unsigned long long*)buf; return PBE_OK; }
#endif
int g_pbe_force_cfg = -1;        // pbe_tune(1, cfg index [| splits << 8]) forces a tile config (and split-K factor); -1 = heuristic
int g_pbe_allow_splitk = 1;      // pbe_tune(2, 0/1)
int g_pbe_pingpong = 1;          // pbe_tune(4, 0/1): ping-pong main loop of the halo-resident conv tiles
int g_pbe_mfast = 1;             // pbe_tune(5, 0/1): let a launch walk its tiles m fastest per XCD when that fetches fewer bytes

extern "C" int pbe_tune(int32_t key, int32_t value) {
    if (key == 1) { g_pbe_force_cfg = (value >= 0 && (value & 255) < kNCfg) ? value : -1; return PBE_OK; }
    if (key == 2) { g_pbe_allow_splitk = value ? 1 : 0; return PBE_OK; }
    if (key == 3) { extern int g_pbe_attn_qw; g_pbe_attn_qw = value; return PBE_OK; }
    if (key == 4) { g_pbe_pingpong = value ? 1 : 0; return PBE_OK; }
    if (key == 5) { g_pbe_mfast = value ? 1 : 0; return PBE_OK; }
    if (key == 8) { extern int g_pbe_attn_pad_lds; g_pbe_attn_pad_lds = value > 0 ? value : 0; return PBE_OK; }
    if (key == 7) { extern int g_pbe_gn_rows; g_pbe_gn_rows = value >= 1 ? value : 16; return PBE_OK; }
    if (key == 10) { extern int g_pbe_gn_apply2_rows; g_pbe_gn_apply2_rows = value >= 0 ? value : 8; return PBE_OK; }
    if (key == 9) { extern int g_pbe_gn_apply_rows; g_pbe_gn_apply_rows = value >= 1 ? value : 4; return PBE_OK; }
    if (key == 6) { extern int g_pbe_attn_mpad; g_pbe_attn_mpad = value ? 1 : 0; return PBE_OK; }
    return pbe_set_error(PBE_EINVAL, "pbe_tune: unknown key %d", key);
}

static int fill_gemm(const pbe_gemm_desc* d, IGemmP& p, const char* who) {
    PBE_REQUIRE(d && d->A && d->W && d->C, "%s: null operand", who);
    PBE_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0 && d->batch >= 1 && d->batch <= 65535, "%s: bad dims M=%d N=%d K=%d batch=%d", who, d->M, d->N, d->K, d->batch);
    PBE_REQUIRE(d->K % 8 == 0, "%s: K=%d must be a multiple of 8", who, d->K);
    PBE_REQUIRE(d->lda % 8 == 0 && d->ldw % 8 == 0 && al16(d->A) && al16(d->W), "%s: A/W must be 16-byte aligned with ld %% 8 == 0", who);
    PBE_REQUIRE(d->strideA % 8 == 0 && d->strideW % 8 == 0, "%s: batch strides of A/W must be multiples of 8", who);
    const int K1 = d->A2 ? d->K1 : d->K;
    if (d->A2) {
        PBE_REQUIRE(K1 > 0 && K1 < d->K && K1 % 32 == 0 && d->lda2 % 8 == 0 && al16(d->A2) && d->batch == 1,
                    "%s: split-K source needs K1 %% 32 == 0 (K1=%d), aligned A2, batch 1", who, K1);
    }
    PBE_REQUIRE(d->lda >= (d->A2 ? K1 : d->K) && d->ldw >= d->K, "%s: leading dims too small", who);
    PBE_REQUIRE(!d->rowvec || d->group_rows > 0, "%s: rowvec needs group_rows > 0", who);
    const bool geglu = d->act == PBE_ACT_GEGLU;
    const int Nout = geglu ? d->N / 2 : d->N;
    PBE_REQUIRE(!geglu || (d->N % 16 == 0 && !d->resid && !d->rowvec && !d->bias_per_row), "%s: GEGLU epilogue needs N %% 16 == 0 and no resid/rowvec", who);
    memset(&p, 0, sizeof(p));
    p.A = (const h16*)d->A; p.A2 = (const h16*)d->A2; p.W = (const h16*)d->W; p.C = (h16*)d->C;
    p.bias = d->bias; p.rowvec = (const h16*)d->rowvec; p.resid = (const h16*)d->resid;
    p.M = d->M; p.N = d->N; p.K = d->K; p.K1 = K1;
    p.lda = d->lda; p.lda2 = d->lda2; p.ldw = d->ldw; p.ldc = d->ldc; p.ldr = d->ldr;
    p.ldv = d->ldv; p.group_rows = d->group_rows > 0 ? d->group_rows : 1;
    p.sA = d->strideA; p.sW = d->strideW; p.sC = d->strideC; p.sR = d->strideR;
    p.alpha = d->alpha; p.act = d->act; p.bias_row = d->bias_per_row;
    PBE_REQUIRE(d->ldc >= (d->VT ? d->vt_col0 : Nout), "%s: ldc too small", who);
    p.vec = (Nout % 8 == 0) && (d->ldc % 8 == 0) && al16(d->C) && (d->strideC % 8 == 0) &&
            (!d->resid || ((d->ldr % 8 == 0) && al16(d->resid) && (d->strideR % 8 == 0)));
    p.ws = (float*)d->workspace;
    // ---- extended epilogue ----
    p.alpha_cols = d->alpha_cols;
    PBE_REQUIRE(d->alpha_cols >= 0 && d->alpha_cols % 4 == 0, "%s: alpha_cols=%d must be a multiple of 4", who, d->alpha_cols);
    if (d->ln_stats) {
        PBE_REQUIRE(d->ln_colsum && d->ln_parts >= 1 && d->ln_stats_ld >= d->M && !d->A2 && d->batch == 1 && d->operand_dtype == PBE_DTYPE_F16,
                    "%s: LayerNorm fold needs ln_colsum, ln_parts >= 1, ln_stats_ld >= M, a single fp16 A source, batch 1", who);
        p.ln_stat = d->ln_stats; p.ln_parts = d->ln_parts; p.ln_ld = d->ln_stats_ld; p.ln_c1 = d->ln_colsum; p.ln_eps = d->ln_eps; p.ln_inv_k = 1.0 / (double)d->K;
    }
    if (d->row_stats_out) {
        PBE_REQUIRE(p.vec && d->batch == 1 && !geglu && !d->VT, "%s: row_stats_out needs 16-byte aligned C / resid rows, batch 1, no GEGLU / VT", who);
        PBE_REQUIRE(d->row_stats_ld == 0 || d->row_stats_ld >= d->M, "%s: row_stats_ld < M", who);
        p.rstat = d->row_stats_out; p.rstat_ld = d->row_stats_ld > 0 ? d->row_stats_ld : d->M;
    }
    if (d->VT) {
        PBE_REQUIRE(d->vt_col0 > 0 && d->vt_col0 < d->N && d->vt_tokens > 0 && d->vt_tokens % 8 == 0 && d->M % d->vt_tokens == 0 && d->vt_rs % 8 == 0 &&
                    d->vt_bs % 8 == 0 && d->vt_rs >= d->vt_tokens && al16(d->VT) && !d->resid && !geglu && d->batch == 1 && p.vec,
                    "%s: VT needs 0 < vt_col0 < N, vt_tokens %% 8 == 0 dividing M, strides %% 8 == 0, no resid / GEGLU, batch 1", who);
        p.vt = (h16*)d->VT; p.vt_col0 = d->vt_col0; p.vt_tok = d->vt_tokens; p.vt_bs = d->vt_bs; p.vt_rs = d->vt_rs;
    }
    PBE_REQUIRE(!ex_needed(p) || d->operand_dtype == PBE_DTYPE_F16, "%s: the extended epilogue takes fp16 operands", who);
    if (d->operand_dtype == PBE_DTYPE_F8E4M3) {
        // A / W are bytes: [M, K] and [N, K] e4m3, leading dims and batch strides in BYTES.  The loader addresses halfs, so K, K1 and
        // the operand strides are halved (K % 16 == 0 keeps every 16-byte chunk whole).
        PBE_REQUIRE(!d->A2 && d->K % 16 == 0 && d->lda % 16 == 0 && d->ldw % 16 == 0 && d->strideA % 16 == 0 && d->strideW % 16 == 0,
                    "%s: fp8 operands need K, lda, ldw and batch strides in multiples of 16 bytes and a single A source", who);
        PBE_REQUIRE(d->a_scale && d->w_scale, "%s: fp8 operands need a_scale [M] and w_scale [N]", who);
        p.K = d->K / 2; p.K1 = p.K; p.lda = d->lda / 2; p.ldw = d->ldw / 2; p.sA = d->strideA / 2; p.sW = d->strideW / 2;
        p.sa = d->a_scale; p.sw = d->w_scale; p.ssa = d->a_scale_stride; p.ssw = d->w_scale_stride;
    } else {
        PBE_REQUIRE(d->operand_dtype == PBE_DTYPE_F16, "%s: unknown operand_dtype %d", who, d->operand_dtype);
    }
    return PBE_OK;
}

extern "C" int pbe_gemm_f16(const pbe_gemm_desc* d, pbe_stream_t stream) {
    IGemmP p;
    const int rc = fill_gemm(d, p, "pbe_gemm_f16");
    if (rc != PBE_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (d->operand_dtype == PBE_DTYPE_F8E4M3) pbe_dispatch_f8(p, d->batch, s, d->tile_cfg);
    else if (ex_needed(p)) pbe_dispatch_ex(p, d->batch, s, d->tile_cfg);
    else pbe_dispatch_dense(p, d->batch, s, d->workspace ? d->workspace_bytes : 0, d->tile_cfg);
    PBE_LAUNCH_CHECK("pbe_gemm_f16");
    return PBE_OK;
}

static void report_plan(const IGemmP& p, int batch, size_t ws_bytes, int want_cfg, int32_t* out, int mode) {
    const Plan pl = plan_igemm(p, batch, ws_bytes, want_cfg, mode);
    const TileCfg& t = kCfg[pl.cfg];
    out[0] = pl.cfg; out[1] = pl.splits; out[2] = t.bm; out[3] = t.bn;
    out[4] = cdiv(p.M, t.bm) * cdiv(p.N, t.bn) * batch * pl.splits;                       // workgroups launched
    out[5] = 0;
}

extern "C" int pbe_gemm_plan(const pbe_gemm_desc* d, int32_t* out6, size_t* workspace_needed) {
    PBE_REQUIRE(out6 && workspace_needed, "pbe_gemm_plan: null output");
    IGemmP p;
    const int rc = fill_gemm(d, p, "pbe_gemm_plan");
    if (rc != PBE_OK) return rc;
    if (ex_needed(p) || d->operand_dtype == PBE_DTYPE_F8E4M3) p.ws = nullptr;            // these forms never split K
    report_plan(p, d->batch, p.ws && d->workspace ? d->workspace_bytes : 0, d->tile_cfg, out6, 0);
    out6[5] = cdiv(p.N, out6[3]);                                                       // column tiles = row-statistics partials per row
    *workspace_needed = out6[1] > 1 ? (size_t)out6[1] * p.M * p.N * sizeof(float) : 0;
    return PBE_OK;
}

static int fill_conv(const pbe_conv3x3_desc* d, IGemmP& p, const char* who) {
    PBE_REQUIRE(d && d->X && d->Wp && d->Y, "%s: null operand", who);
    const int Cin = d->C1 + d->C2;
    PBE_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Cout > 0, "%s: bad dims", who);
    PBE_REQUIRE(d->C1 > 0 && d->C1 % 64 == 0 && d->C2 >= 0 && d->C2 % 64 == 0, "%s: C1=%d C2=%d must be multiples of 64 (use pbe_im2col3x3_f16 + pbe_gemm_f16 for small Cin)", who, d->C1, d->C2);
    PBE_REQUIRE((d->C2 == 0) == (d->X2 == nullptr), "%s: X2 / C2 mismatch", who);
    PBE_REQUIRE(d->stride == 1 || d->stride == 2, "%s: stride must be 1 or 2", who);
    PBE_REQUIRE(d->pad == 0 || d->pad == 1, "%s: pad must be 0 or 1", who);
    PBE_REQUIRE(d->upsample == 0 || ((d->upsample == 1 || d->upsample == 2) && d->stride == 1), "%s: upsample only with stride 1", who);
    PBE_REQUIRE(d->upsample != 2 || (d->pad == 1 && !d->resid && !d->rowvec), "%s: the phase form of the upsampling conv takes pad 1, no resid / rowvec", who);
    PBE_REQUIRE(al16(d->X) && al16(d->Wp) && al16(d->Y) && (!d->X2 || al16(d->X2)), "%s: 16-byte alignment", who);
    const bool phase = d->upsample == 2;          // nearest-2x + 3x3 as four 2x2 convs on the source grid (Wp = the four summed weight sets)
    const int Hv = phase ? d->H : d->H << d->upsample, Wv = phase ? d->W : d->W << d->upsample;
    // output size: pad=1 -> floor((Hv + 2 - 3)/s) + 1 ; pad=0 is the VAE (0,1,0,1) pad: floor((Hv + 1 - 3)/s) + 1
    const int extra = d->pad ? 2 : 1;
    const int Ho = (Hv + extra - 3) / d->stride + 1, Wo = (Wv + extra - 3) / d->stride + 1;
    PBE_REQUIRE(Ho > 0 && Wo > 0, "%s: empty output", who);
    const long M = (long)d->B * Ho * Wo;
    PBE_REQUIRE(M < (1L << 31), "%s: too many output pixels", who);
    memset(&p, 0, sizeof(p));
    p.A = (const h16*)d->X; p.A2 = (const h16*)d->X2; p.W = (const h16*)d->Wp; p.C = (h16*)d->Y;
    p.bias = d->bias; p.rowvec = (const h16*)d->rowvec; p.resid = (const h16*)d->resid;
    p.M = (int)M; p.N = d->Cout; p.K = (phase ? 4 : 9) * Cin; p.K1 = p.K;
    p.phase = phase ? 1 : 0;
    if (phase) p.sW = (long)d->Cout * p.K;         // one weight set per output phase; blockIdx.y = phase
    p.ldw = p.K; p.ldc = d->Cout; p.ldr = d->Cout;
    p.ldv = d->ldv; p.group_rows = Ho * Wo;
    p.alpha = 1.f; p.act = d->act; p.bias_row = 0;
    p.vec = (d->Cout % 8 == 0) && (!d->resid || al16(d->resid));
    p.H = d->H; p.Wd = d->W; p.C1 = d->C1; p.C2 = d->C2; p.Ho = Ho; p.Wo = Wo;
    p.cstride = d->stride; p.pad = d->pad; p.ups = phase ? 0 : d->upsample;
    p.cb = d->kblock > 0 ? d->kblock : 64;
    PBE_REQUIRE(p.cb % 64 == 0 && d->C1 % p.cb == 0 && d->C2 % p.cb == 0, "%s: kblock=%d must be a multiple of 64 dividing C1=%d and C2=%d", who, p.cb, d->C1, d->C2);
    p.ws = (float*)d->workspace;
    if (d->group_stats_blocks) *d->group_stats_blocks = 0;
    if (d->group_stats_out) {
        PBE_REQUIRE(d->group_stats_blocks && d->group_stats_groups > 0 && d->group_stats_groups <= 64 && d->Cout % d->group_stats_groups == 0,
                    "%s: group_stats_out needs group_stats_blocks and 0 < group_stats_groups <= 64 dividing Cout", who);
        p.gstat = d->group_stats_out; p.gs_groups = d->group_stats_groups; p.gs_cg = d->Cout / d->group_stats_groups; p.gs_hw = Ho * Wo;
        p.gs_report = d->group_stats_blocks;
    }
    return PBE_OK;
}

extern "C" int pbe_conv3x3_f16(const pbe_conv3x3_desc* d, pbe_stream_t stream) {
    IGemmP p;
    const int rc = fill_conv(d, p, "pbe_conv3x3_f16");
    if (rc != PBE_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    pbe_dispatch_conv(p, p.phase ? 4 : 1, s, d->workspace ? d->workspace_bytes : 0, d->tile_cfg);
    PBE_LAUNCH_CHECK("pbe_conv3x3_f16");
    return PBE_OK;
}

extern "C" int pbe_conv3x3_plan(const pbe_conv3x3_desc* d, int32_t* out6, size_t* workspace_needed) {
    PBE_REQUIRE(out6 && workspace_needed, "pbe_conv3x3_plan: null output");
    IGemmP p;
    const int rc = fill_conv(d, p, "pbe_conv3x3_plan");
    if (rc != PBE_OK) return rc;
    report_plan(p, p.phase ? 4 : 1, d->workspace ? d->workspace_bytes : 0, d->tile_cfg, out6, 1);
    *workspace_needed = out6[1] > 1 ? (size_t)out6[1] * p.M * p.N * sizeof(float) : 0;
    return PBE_OK;
}
