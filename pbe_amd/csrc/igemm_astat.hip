// A-stationary persistent GEMM tiles for the K = 320 token GEMMs of the 64x64 level (ldm/modules/attention.py:198-252 in the reference:
// the GEGLU projection of FeedForward after norm3) - tile configs 19 / 20 of the igemm planner.
//
// Why: with K = 320 an output tile of the streaming kernel (igemm_kernel.h) lives for 5 k-tiles; its ring (one k-tile in flight per
// workgroup) drains at every tile, the first tile waits a full memory round trip, and every k-tile costs each wave 9 LDS-DMA pieces
// (A and W) at ~110 cycles of issue each against 40 MFMAs (tools/astat_stamps.py).  Here a workgroup keeps its 128 rows of A for a RUN
// of column tiles and streams only the weights (L2-resident: 1.6 MB for N = 2 560): 4 pieces per wave and k-tile against 32 MFMAs, a
// ring that never drains (the next tile's first two weight k-tiles land during the epilogue), no per-tile prologue.
//
// The A block lives in REGISTERS: a wave's 32 rows x 320 k are 2 x 10 MFMA operand fragments = 80 VGPRs, loaded once; the LDS holds only
// the weight ring (3 slots), the per-wave staging rows and the epilogue vectors, so TWO independent 4-wave workgroups fit a CU (59 KiB
// each at 128 columns): while one runs its epilogue (VALU) the other's MFMAs own the matrix pipe - what the two waves of a SIMD cannot do
// inside one workgroup, whose barriers keep them in step.  (First form, A block in LDS with one 8-wave workgroup per CU: 86.5 us against
// 72.7 us for this one and 107 us for the streaming 128x160 tile on [32768, 320] x [2560, 320]^T, profiles/r03_astat_lds_vs_regs_ab.txt.)
// Wave tile 32 x (16 TN): every wave spans the tile's columns, so a row of the GEGLU output leaves as one 16 TN-byte run.
// Rows of 128 B with the chunk swizzle of igemm_kernel.h (position p of row r holds chunk p ^ (r & 7)): the same conflict-free
// ds_read_b128 fragment reads.  The arithmetic (accumulation order, epilogue association) is that of the streaming kernel: bit-identical
// output, checked by tests/test_ops_gpu.py::test_gemm_a_stationary_matches_streaming_tile.
#include "igemm_kernel.h"

namespace {
constexpr int ABM = 128, AKT = 5;
}  // namespace

// FORM 0: GEGLU output (the FeedForward projection).  FORM 1: q | k | v^T of the fused self-attention projection - plain fp16 columns, alpha on
// the columns below alpha_cols only, the tiles at and beyond vt_col0 stored transposed (IGemmP.vt), as the EX_LN | EX_VT streaming form does.
template <int TN, int FORM>
__global__ void __launch_bounds__(256, 2) astat_regs_kernel(const IGemmP p, int run, int tiles_m) {
    constexpr int TM = 2, BN = 16 * TN, PW = BN / 8 / 4;          // PW = weight pieces (8 rows x 128 B) per wave and k-tile
    constexpr int SLOT = BN * 128, RS = 3;      // (a ring of 4 fits the 128-column form: measured slower, 0.70 -> 0.73 of the streaming tile's time cold)
    constexpr int STGR = BN / 2 + 8;                               // halfs per staging row
    constexpr int STGW = 16 * STGR * 2;
    constexpr int NST = (16 * (BN / 16) + 63) / 64;                // store instructions per 16 rows x BN / 2 columns (chunks of 16 bytes)
    constexpr int VROW = 40;                                       // halfs per row of the transposed staging (32 tokens + pad)
    // stores a tile's epilogue leaves in the vmcnt queue: GEGLU 2 NST; plain 4 NST, transposed TN (the counted waits take the smaller)
    constexpr int EST = FORM == 0 ? 2 * NST : (4 * NST < TN ? 4 * NST : TN);
    static_assert(FORM == 0 || (TN % 2 == 0 && 32 * VROW * 2 <= 16 * (BN / 2 + 8) * 2), "transposed staging: two 16-column groups at a time");
    static_assert(BN % 32 == 0 && RS * SLOT + 4 * STGW + 2 * 2 * BN * 4 <= 80 * 1024, "two workgroups per CU");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wm = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mb = blockIdx.x % tiles_m, rn = blockIdx.x / tiles_m;
    const int m0 = mb * ABM, nt0 = rn * run;
    const int Q = run * AKT;
    h16* stg = reinterpret_cast<h16*>(smem + RS * SLOT + wm * STGW);
    float* svec = reinterpret_cast<float*>(smem + RS * SLOT + 4 * STGW);       // [2 buffers][bias | colsum][BN]
    const int fr = lane & 15, fq = lane >> 4;
    PBE_ACC_DECL;
    PBE_STAMP(0);
    PBE_STAMP(7);

    // ---- prologue, in vmcnt order: [W0] [LayerNorm statistics] [A fragments, 20 loads] [W1]: the first k-tile needs W0 and the k-tile 0
    //      fragments only, the other 16 fragment loads and W1 stay in flight behind it ----
    const int lrow = lane >> 3, gch = (lane & 7) ^ lrow;
    // every workgroup of the chip streams the SAME weight tiles: each starts its run at another tile (rot) so that at one moment the
    // workgroups read different lines of the L2 (pbe_tune-free A/B: PBE_ASTAT_NOROT)
#ifdef PBE_ASTAT_NOROT
    const int rot = 0;
#else
    const int rot = (mb + (mb >> 3)) % run;
#endif
    const h16* w_src[PW];
#pragma unroll
    for (int i = 0; i < PW; ++i) w_src[i] = p.W + ((long)(nt0 + rot) * BN + (wm + 4 * i) * 8 + lrow) * p.ldw + gch * 8;
    const long w_next_tile = (long)BN * p.ldw - (AKT - 1) * 64, w_wrap = -(long)run * BN * p.ldw;
    int wq_kt = 0, wq_slot = 0, wq_tile = rot;
    auto issue_w = [&]() {
        unsigned char* dst = smem + wq_slot * SLOT;
#pragma unroll
        for (int i = 0; i < PW; ++i) PBE_GLDS16(w_src[i], dst + (wm + 4 * i) * 1024);
        long adv = 64;
        if (wq_kt == AKT - 1) {
            adv = w_next_tile;
            if (++wq_tile == run) { wq_tile = 0; adv += w_wrap; }
        }
#pragma unroll
        for (int i = 0; i < PW; ++i) w_src[i] += adv;
        wq_kt = wq_kt == AKT - 1 ? 0 : wq_kt + 1;
        wq_slot = wq_slot == RS - 1 ? 0 : wq_slot + 1;
    };
    issue_w();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    float ln_rs[TM], ln_nm[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = m0 + wm * 32 + j * 16 + fr;
        float a = 0.f, q = 0.f;
        for (int z = 0; z < p.ln_parts; ++z) {
            const float2 t = *reinterpret_cast<const float2*>(p.ln_stat + 2 * ((long)z * p.ln_ld + m));
            a += t.x; q += t.y;
        }
        const double mean = (double)a * p.ln_inv_k;
        const float var = (float)((double)q * p.ln_inv_k - mean * mean);
        const float rstd = __builtin_amdgcn_rsqf(fmaxf(var, 0.f) + p.ln_eps);
        ln_rs[j] = rstd; ln_nm[j] = -(float)mean * rstd;
    }
    // this wave's A fragments: row (j, fr), k = kt * 64 + ks * 32 + fq * 8 .. + 8 (the MFMA "B" operand of igemm_kernel.h)
    h16x8 fa[AKT][2][TM];
#pragma unroll
    for (int kt = 0; kt < AKT; ++kt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < TM; ++j)
                fa[kt][ks][j] = *reinterpret_cast<const h16x8*>(p.A + (long)(m0 + wm * 32 + j * 16 + fr) * p.lda + kt * 64 + ks * 32 + fq * 8);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    issue_w();
    PBE_STAMP(1);

    f32x4 acc[TN][TM];
    const int rsw = (fq ^ (fr & 7)) << 4, w_rd = fr * 128;
    int q = 0, slot_rd = 0;
    auto step = [&](auto WAITC, auto KTC, int n0, int sbuf) {
        constexpr int WAIT = decltype(WAITC)::value, kt = decltype(KTC)::value;
        PBE_ACC_T0();
        if (q == Q - 1) wait_vmcnt<0>(); else wait_vmcnt<WAIT>();
        PBE_ACC(acc_w_);
        PBE_ACC_T0();
        __builtin_amdgcn_s_barrier();
        PBE_ACC(acc_b2_);
        PBE_ACC_T0();
        const unsigned char* sw = smem + slot_rd * SLOT;
        h16x8 fw[TN];
#pragma unroll
        for (int i = 0; i < TN; ++i) fw[i] = *reinterpret_cast<const h16x8*>(sw + w_rd + rsw + i * 16 * 128);
        __builtin_amdgcn_sched_barrier(0);
        // the DMA pieces of k-tile q + 2 go out behind the fragment reads (in the shadow of the first k-step's MFMAs, as a burst or one piece
        // per four MFMAs, they cost the wave the same ~110 cycles each - tools/astat_stamps.py - and delay the second k-step's reads)
        if (q + 2 < Q) issue_w();
        float vb = 0.f, vc = 0.f;
        if (kt == 2 && tid < BN) {                        // this tile's epilogue vectors -> LDS (read after the barriers of k-tiles 3 and 4)
            const int n = n0 + tid;
            vb = p.bias ? p.bias[n] : 0.f;
            vc = ((p.alpha_cols > 0 && n >= p.alpha_cols) ? 1.f : p.alpha) * p.ln_c1[n];      // alpha * colsum, as the streaming kernel's epilogue forms it
        }
        __builtin_amdgcn_sched_barrier(0);
        PBE_ACC(acc_r_);
        PBE_ACC_T0();
        // (reading the 160-column tile's ten weight fragments five at a time does not lower the register count: the scheduler hoists the
        //  second five - 26-31 spilled registers against 4-14)
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[i], fa[kt][0][j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TN; ++i) fw[i] = *reinterpret_cast<const h16x8*>(sw + w_rd + (rsw ^ 64) + i * 16 * 128);
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[i], fa[kt][1][j], acc[i][j], 0, 0, 0);
        if (kt == 2 && tid < BN) {
            __builtin_amdgcn_sched_barrier(0);
            svec[sbuf * 2 * BN + tid] = vb;
            svec[sbuf * 2 * BN + BN + tid] = vc;
        }
        PBE_ACC(acc_m_);
        ++q;
        slot_rd = slot_rd == RS - 1 ? 0 : slot_rd + 1;
    };
    using std::integral_constant;
    for (int t = 0; t < run; ++t) {
        const int n0 = (nt0 + (t + rot >= run ? t + rot - run : t + rot)) * BN, sbuf = t & 1;
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // behind W(q) in this wave's queue at each wait: the next weight k-tile (PW pieces) and, for the first two k-tiles of a later tile, the
        // previous epilogue's EST stores (stores count in vmcnt on gfx950)
        if (t == 0) {
            step(integral_constant<int, PW + 16>{}, integral_constant<int, 0>{}, n0, sbuf);
            step(integral_constant<int, PW>{}, integral_constant<int, 1>{}, n0, sbuf);
        } else {
            step(integral_constant<int, PW + EST>{}, integral_constant<int, 0>{}, n0, sbuf);
            step(integral_constant<int, PW + EST>{}, integral_constant<int, 1>{}, n0, sbuf);
        }
        step(integral_constant<int, PW>{}, integral_constant<int, 2>{}, n0, sbuf);
        step(integral_constant<int, PW>{}, integral_constant<int, 3>{}, n0, sbuf);
        step(integral_constant<int, PW>{}, integral_constant<int, 4>{}, n0, sbuf);
        // ---- epilogue: LayerNorm fold + bias (+ GEGLU), 16 rows at a time through the wave's own staging rows (no workgroup barrier) ----
        PBE_ACC_T0();
        const float al = p.alpha;
        const float* sb = svec + sbuf * 2 * BN;
        constexpr int CPR = BN / 16;                       // 16-byte chunks per staged row (BN / 2 halfs)
        if constexpr (FORM == 0) {
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const float ars = al * ln_rs[j];
#pragma unroll
                for (int i = 0; i < TN; ++i) {
                    const f32x4 bn = *reinterpret_cast<const f32x4*>(sb + i * 16 + fq * 4);
                    const f32x4 c1 = *reinterpret_cast<const f32x4*>(sb + BN + i * 16 + fq * 4);
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf(acc[i][j][r], ars, __builtin_fmaf(ln_nm[j], c1[r], bn[r]));
                    const h16x2 o2 = {(h16)(v[0] * gelu_erf_f(v[1])), (h16)(v[2] * gelu_erf_f(v[3]))};
                    *reinterpret_cast<h16x2*>(stg + fr * STGR + i * 8 + fq * 2) = o2;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int h = 0; h < NST; ++h) {
                    const int idx = lane + 64 * h;
                    if (idx < 16 * CPR) {
                        const int row = idx / CPR, ch = idx - row * CPR;
                        const h16x8 val = *reinterpret_cast<const h16x8*>(stg + row * STGR + ch * 8);
                        *reinterpret_cast<h16x8*>(p.C + (long)(m0 + wm * 32 + j * 16 + row) * p.ldc + (n0 >> 1) + ch * 8) = val;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        } else {
            auto value = [&](int i, int j, float (&v)[4]) {           // the streaming kernel's EX_LN | EX_VT expression, same association
                const int nl = i * 16 + fq * 4;
                const float ali = (p.alpha_cols > 0 && n0 + nl >= p.alpha_cols) ? 1.f : al;
                const f32x4 bn = *reinterpret_cast<const f32x4*>(sb + nl);
                const f32x4 c1 = *reinterpret_cast<const f32x4*>(sb + BN + nl);
                const float ars = ali * ln_rs[j];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf(acc[i][j][r], ars, __builtin_fmaf(ln_nm[j], c1[r], bn[r]));
            };
            if (!(p.vt && n0 >= p.vt_col0)) {                        // q | k columns: rows of BN / 2 columns at a time
#pragma unroll
                for (int j = 0; j < TM; ++j)
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
                        for (int ii = 0; ii < TN / 2; ++ii) {
                            float v[4];
                            value(hf * (TN / 2) + ii, j, v);
                            const h16x4 o = {(h16)v[0], (h16)v[1], (h16)v[2], (h16)v[3]};
                            *reinterpret_cast<h16x4*>(stg + fr * STGR + ii * 16 + fq * 4) = o;
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                        for (int h = 0; h < NST; ++h) {
                            const int idx = lane + 64 * h;
                            if (idx < 16 * CPR) {
                                const int row = idx / CPR, ch = idx - row * CPR;
                                const h16x8 val = *reinterpret_cast<const h16x8*>(stg + row * STGR + ch * 8);
                                *reinterpret_cast<h16x8*>(p.C + (long)(m0 + wm * 32 + j * 16 + row) * p.ldc + n0 + hf * (BN / 2) + ch * 8) = val;
                            }
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
            } else {                                                  // v columns: [channel][token] through the staging, 32 channels x the wave's 32 tokens at a time
                const int mw = m0 + wm * 32, bsmp = mw / p.vt_tok, tok0 = mw - bsmp * p.vt_tok;      // (32 | vt_tok: one sample per wave)
                h16* vdst = p.vt + (long)bsmp * p.vt_bs + tok0;
#pragma unroll
                for (int ps = 0; ps < TN / 2; ++ps) {
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                        for (int j = 0; j < TM; ++j) {
                            float v[4];
                            value(ps * 2 + ii, j, v);
#pragma unroll
                            for (int r = 0; r < 4; ++r) stg[(ii * 16 + fq * 4 + r) * VROW + j * 16 + fr] = (h16)v[r];
                        }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int idx = lane + 64 * h, crow = idx >> 2, tch = idx & 3;
                        const h16x8 val = *reinterpret_cast<const h16x8*>(stg + crow * VROW + tch * 8);
                        *reinterpret_cast<h16x8*>(vdst + (long)(n0 - p.vt_col0 + ps * 32 + crow) * p.vt_rs + tch * 8) = val;
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
            }
        }
        PBE_ACC(acc_b1_);
    }
    PBE_ACC_STORE();
    PBE_STAMP(6);
    PBE_STAMP(8);
}

template <int TN, int FORM>
static void launch_regs(IGemmP p, hipStream_t s) {
    constexpr int BN = 16 * TN;
    constexpr int lds = 3 * BN * 128 + 4 * 16 * (BN / 2 + 8) * 2 + 2 * 2 * BN * 4;
    const int tiles_m = p.M / ABM, tiles_n = p.N / BN;
    int run = tiles_n;                                     // the longest run of column tiles that still gives every CU its two workgroups
    while (run > 1 && ((long)tiles_m * (tiles_n / run) < 512 || tiles_n % run)) --run;
    static std::atomic<uint64_t> attr_done{0};
    pbe_raise_dynamic_lds(attr_done, reinterpret_cast<const void*>(&astat_regs_kernel<TN, FORM>), lds);
    pbe_prof_begin(PBE_K_GEMM, s);
    hipLaunchKernelGGL((astat_regs_kernel<TN, FORM>), dim3((unsigned)(tiles_m * (tiles_n / run))), dim3(256), lds, s, p, run, tiles_m);
    const double nout = FORM == 0 ? 0.5 * p.N : (double)p.N;
    pbe_prof_end(PBE_K_GEMM, s, 2.0 * p.M * (double)p.N * p.K, 2.0 * ((double)p.M * p.K + (double)p.N * p.K + (double)p.M * nout));
}

// Tile 19 = 128 columns, tile 20 = 160 columns.  GEGLU with the LayerNorm fold, or (tile 20) the fused q | k | v^T projection.
bool pbe_astat_ok(const IGemmP& p, int batch, int cfg) {
    const int bn = cfg == 19 ? 128 : 160;
    if (!(batch == 1 && !p.A2 && p.K == 64 * AKT && p.M % ABM == 0 && p.N % bn == 0 && p.ln_stat && !p.rstat && !p.resid && !p.rowvec && !p.bias_row && p.vec &&
          (p.lda & 7) == 0 && (p.ldw & 7) == 0))
        return false;
    if (p.act == PBE_ACT_GEGLU) return !p.vt && p.alpha_cols == 0;
    return cfg == 20 && p.act == PBE_ACT_NONE && p.vt && p.vt_col0 % bn == 0 && p.vt_tok % 32 == 0 && (p.alpha_cols & 3) == 0 && (p.ldc & 7) == 0;
}

void pbe_launch_astat(int cfg, IGemmP p, hipStream_t s) {
    if (p.act != PBE_ACT_GEGLU) launch_regs<10, 1>(p, s);
    else if (cfg == 19) launch_regs<8, 0>(p, s);
    else launch_regs<10, 0>(p, s);
}
