// Fused multi-head attention forward, softmax(Q K^T * scale) V, for gfx950 (wave64, MFMA 32x32x16 f16).
//
// One workgroup = 4 waves = 128 queries of one (batch, head); each wave owns 32 queries for the
// whole K/V sweep.  K/V^T tiles of 64 keys are staged global -> registers -> LDS (next tile's
// loads in flight under the current tile's MFMAs, two LDS buffers, one barrier per tile).
//
// The score tile is computed TRANSPOSED, S^T = K Q^T, so the query index sits on the lane
// (column of the 32x32 accumulator) and the 32 keys of a sub-tile sit in the lane's registers:
// the row max / row sum are register-local plus ONE cross-half exchange, and the exponentiated
// accumulator registers feed the second product O^T = V^T P^T directly as its B operand (no LDS
// round trip, no shuffles) — the k order inside each 16-key step is permuted
// (key = 16s + 8(j>>2) + 4h + (j&3)), so the V^T fragment is read with the same permutation.
// V therefore arrives already transposed ([head*D + d][key], produced for free by the
// projection GEMM with swapped operands), its tile rows padded to 136 B (conflict-free
// ds_read_b64); K tile rows padded to an odd number of 16-B units (conflict-free ds_read_b128).
#include "common.h"
#include "../../include/pbe_hip.h"

struct AttnP {
    const h16* Q; const h16* K; const h16* VT; h16* O;
    int B, H, Nq, Nk, D;
    long q_bs, q_rs, k_bs, k_rs, vt_bs, vt_rs, o_bs, o_rs;
    float scale_log2e;
};

template <int DP, int QW>
__global__ void __launch_bounds__(256) attn_kernel(const AttnP p) {
    // QW = 32-query sub-blocks per wave (1 or 2).  QW = 2 halves the global K/V traffic, the LDS staging and the
    // barriers per query (more work between barriers); used for long sequences where the grid still fills the chip.
    constexpr int KT = 64;                // keys per staged tile (one barrier per tile)
    constexpr int DV = (DP + 31) / 32 * 32;
    constexpr int NDS = DP / 16;          // k-steps of the QK^T product
    constexpr int NDT = DV / 32;          // 32-wide d tiles of the output
    constexpr int DC = DP / 8;            // 16-B chunks per K row
    constexpr int KSTR = (DC | 1) * 16;   // K tile row stride, bytes (odd number of 16-B units)
    constexpr int VSTR = KT * 2 + 8;      // V^T tile row stride, bytes (8 mod 128: conflict-free ds_read_b64)
    constexpr int K_BYTES = KT * KSTR;
    constexpr int V_BYTES = DV * VSTR;
    constexpr int BUF = (K_BYTES + V_BYTES + 15) / 16 * 16;
    constexpr int KCH = KT * DC, NKL = (KCH + 255) / 256;
    constexpr int VPR = KT / 8;           // 16-B chunks per V^T row
    constexpr int VCH = DV * VPR, NVL = (VCH + 255) / 256;
    constexpr int BQ = 128 * QW;          // queries per workgroup
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h5 = lane >> 5;
    const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
    const int q0 = blockIdx.x * BQ + wave * 32 * QW + l31;          // sub-block i holds query q0 + 32 i
    const int D = p.D;
    const h16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const h16x8 one8 = {1, 1, 1, 1, 1, 1, 1, 1};
    // When the d-tile padding leaves a free row (DV > DP), row DV-1 of the V^T tile is all ones: the PV
    // MFMA then accumulates the softmax denominator sum_k P[k] in O^T[DV-1] for free (same rescaling as O),
    // and the 32 VALU adds per tile disappear.  Otherwise the denominator is summed on the VALU.
    constexpr bool ONES = DV > DP;

    const h16* Qb = p.Q + (long)b * p.q_bs + (long)h * D;
    const h16* Kb = p.K + (long)b * p.k_bs + (long)h * D;
    const h16* Vb = p.VT + (long)b * p.vt_bs + (long)h * D * p.vt_rs;

    // Q fragments (B operand of S^T = K Q^T): lane = query column, 8 consecutive d per k-step half
    h16x8 qf[QW][NDS];
#pragma unroll
    for (int i = 0; i < QW; ++i)
#pragma unroll
        for (int ds = 0; ds < NDS; ++ds) {
            const int d0 = ds * 16 + 8 * h5;
            const bool ok = (q0 + 32 * i) < p.Nq && d0 < D;
            const h16* src = ok ? Qb + (long)(q0 + 32 * i) * p.q_rs + d0 : p.Q;
            h16x8 v = *reinterpret_cast<const h16x8*>(src);
            qf[i][ds] = ok ? v : zero8;
        }

    h16x8 rk[NKL], rv[NVL];
    auto load_tile = [&](int t) {
        const int kv0 = t * KT;
#pragma unroll
        for (int i = 0; i < NKL; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx / DC, ch = idx - row * DC;
            const bool ok = idx < KCH && (kv0 + row) < p.Nk && ch * 8 < D;
            const h16* src = ok ? Kb + (long)(kv0 + row) * p.k_rs + ch * 8 : p.K;
            h16x8 v = *reinterpret_cast<const h16x8*>(src);
            rk[i] = ok ? v : zero8;
        }
#pragma unroll
        for (int i = 0; i < NVL; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx / VPR, ch = idx - row * VPR;
            const int key0 = kv0 + ch * 8;
            const bool ok = idx < VCH && row < D && key0 < p.Nk;
            const h16* src = ok ? Vb + (long)row * p.vt_rs + key0 : p.VT;
            h16x8 v = *reinterpret_cast<const h16x8*>(src);
            v = ok ? v : zero8;
            if (ONES && idx < VCH && row == DV - 1 && key0 < p.Nk) v = one8;      // denominator row
            if (key0 < p.Nk && key0 + 8 > p.Nk) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (key0 + e >= p.Nk) v[e] = (h16)0.f;
            }
            rv[i] = v;
        }
    };
    auto store_tile = [&](int buf) {
        unsigned char* sk = smem + buf * BUF;
        unsigned char* sv = sk + K_BYTES;
#pragma unroll
        for (int i = 0; i < NKL; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx / DC, ch = idx - row * DC;
            if (idx < KCH) *reinterpret_cast<h16x8*>(sk + row * KSTR + ch * 16) = rk[i];
        }
#pragma unroll
        for (int i = 0; i < NVL; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx / VPR, ch = idx - row * VPR;
            if (idx < VCH) {
                h16x4 lo = {rv[i][0], rv[i][1], rv[i][2], rv[i][3]};
                h16x4 hi = {rv[i][4], rv[i][5], rv[i][6], rv[i][7]};
                *reinterpret_cast<h16x4*>(sv + row * VSTR + ch * 16) = lo;
                *reinterpret_cast<h16x4*>(sv + row * VSTR + ch * 16 + 8) = hi;
            }
        }
    };

    f32x16 o[QW][NDT];
    float m_run[QW], l_run[QW];
#pragma unroll
    for (int i = 0; i < QW; ++i) {
        m_run[i] = -INFINITY; l_run[i] = 0.f;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][dt][r] = 0.f;
    }

    const int nt = (p.Nk + KT - 1) / KT;
    load_tile(0);
    store_tile(0);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        if (t + 1 < nt) load_tile(t + 1);
        const unsigned char* sk = smem + cur * BUF;
        const unsigned char* sv = sk + K_BYTES;
        const int kv0 = t * KT;

        // One query sub-block at a time (registers for S / P are reused); the staged K / V^T tile, its
        // global loads and the barrier are shared by all QW sub-blocks.
#pragma unroll
        for (int i = 0; i < QW; ++i) {
            // ---- S^T = K Q^T : two 32-key sub-tiles ----
            f32x16 s0, s1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
            for (int ds = 0; ds < NDS; ++ds) {
                const h16x8 k0 = *reinterpret_cast<const h16x8*>(sk + l31 * KSTR + (ds * 2 + h5) * 16);
                const h16x8 k1 = *reinterpret_cast<const h16x8*>(sk + (32 + l31) * KSTR + (ds * 2 + h5) * 16);
                s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0, qf[i][ds], s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k1, qf[i][ds], s1, 0, 0, 0);
            }
            // ---- online softmax over this lane's 32 keys (+ the other half-wave's 32) ----
            // VALU budget matters here (d = 40: 14 MFMAs per tile vs ~200 VALU ops): the max runs on RAW scores,
            // the scale is folded into the exp2 argument (one FMA), O is rescaled only when some lane's max grew.
            if (kv0 + 64 > p.Nk) {                      // tail tile only (wave-uniform branch, selects inside)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kv0 + (r & 3) + 8 * (r >> 2) + 4 * h5;
                    s0[r] = (key >= p.Nk) ? -INFINITY : s0[r];
                    s1[r] = (key + 32 >= p.Nk) ? -INFINITY : s1[r];
                }
            }
            float mx = fmaxf(s0[0], s1[0]);
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, s0[r]), s1[r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run[i], mx * p.scale_log2e);
            if (__builtin_amdgcn_ballot_w64(m_new > m_run[i]) != 0) {      // some query's running max grew in this wave
                const float alpha = __builtin_amdgcn_exp2f(m_run[i] - m_new);
                m_run[i] = m_new;
                l_run[i] *= alpha;
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[i][dt][r] *= alpha;
            }
            const float neg_m = -m_run[i];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s0[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[r], p.scale_log2e, neg_m));
                s1[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[r], p.scale_log2e, neg_m));
            }
            if (!ONES) {
                float psum = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) psum += s0[r] + s1[r];
                l_run[i] += psum;
            }
            // ---- P^T fragments straight from the accumulator registers (permuted k order) ----
            h16x8 pf[4];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                pf[0][j] = (h16)s0[j];
                pf[1][j] = (h16)s0[8 + j];
                pf[2][j] = (h16)s1[j];
                pf[3][j] = (h16)s1[8 + j];
            }
            // ---- O^T += V^T P^T ----
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const unsigned char* vrow = sv + (dt * 32 + l31) * VSTR + 8 * h5;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {   // ks = sub*2 + s : keys sub*32 + 16 s + {4h..4h+3, 8+4h..}
                    const h16x4 lo = *reinterpret_cast<const h16x4*>(vrow + ks * 32);
                    const h16x4 hi = *reinterpret_cast<const h16x4*>(vrow + ks * 32 + 16);
                    const h16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    o[i][dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[ks], o[i][dt], 0, 0, 0);
                }
            }
            if (QW > 1) __builtin_amdgcn_sched_barrier(0);      // keep the sub-blocks sequential: their S / P registers are shared
        }
        if (t + 1 < nt) store_tile(cur ^ 1);
        __syncthreads();
    }

    // ---- normalise and store O[q, h*D + d] ----
#pragma unroll
    for (int i = 0; i < QW; ++i) {
        float l;
        if (ONES) {                                       // O^T row DV-1 lives in register 15 of the upper half-wave
            const float mine = o[i][NDT - 1][15];
            const float other = __shfl_xor(mine, 32, 64);
            l = h5 ? mine : other;
        } else {
            l = l_run[i] + __shfl_xor(l_run[i], 32, 64);
        }
        const float inv = 1.0f / l;
        const int q = q0 + 32 * i;
        if (q < p.Nq) {
            h16* Ob = p.O + (long)b * p.o_bs + (long)q * p.o_rs + (long)h * D;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d0 = dt * 32 + 8 * g + 4 * h5;
                    if (d0 < D) {
                        h16x4 v;
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = (h16)(o[i][dt][4 * g + r] * inv);
                        *reinterpret_cast<h16x4*>(Ob + d0) = v;
                    }
                }
        }
    }
}

template <int DP, int QW>
static void launch_attn(const AttnP& p, hipStream_t s) {
    constexpr int DV = (DP + 31) / 32 * 32;
    constexpr int KSTR = ((DP / 8) | 1) * 16;
    constexpr int BUF = (64 * KSTR + DV * 136 + 15) / 16 * 16;
    constexpr size_t lds = 2 * BUF;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_kernel<DP, QW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    dim3 grid(cdiv(p.Nq, 128 * QW), p.B * p.H);
    hipLaunchKernelGGL((attn_kernel<DP, QW>), grid, dim3(256), lds, s, p);
}

int g_pbe_attn_qw = 0;       // pbe_tune(3, v): 0 = heuristic, 1 / 2 = force queries-per-wave factor

extern "C" int pbe_attention_f16(const pbe_attn_desc* d, pbe_stream_t stream) {
    PBE_REQUIRE(d && d->Q && d->K && d->VT && d->O, "pbe_attention_f16: null operand");
    PBE_REQUIRE(d->B > 0 && d->H > 0 && d->Nq > 0 && d->Nk > 0, "pbe_attention_f16: bad dims");
    PBE_REQUIRE(d->D % 8 == 0 && d->D >= 8 && d->D <= 160, "pbe_attention_f16: head dim %d unsupported (multiple of 8, <= 160)", d->D);
    PBE_REQUIRE(d->q_rs % 8 == 0 && d->k_rs % 8 == 0 && d->vt_rs % 8 == 0 && d->o_rs % 8 == 0 &&
                d->q_bs % 8 == 0 && d->k_bs % 8 == 0 && d->vt_bs % 8 == 0 && d->o_bs % 8 == 0,
                "pbe_attention_f16: strides must be multiples of 8 elements");
    PBE_REQUIRE(d->vt_rs >= (d->Nk + 7) / 8 * 8, "pbe_attention_f16: vt_rs must cover Nk rounded up to 8");
    PBE_REQUIRE(((uintptr_t)d->Q & 15) == 0 && ((uintptr_t)d->K & 15) == 0 && ((uintptr_t)d->VT & 15) == 0 && ((uintptr_t)d->O & 15) == 0,
                "pbe_attention_f16: 16-byte alignment");
    PBE_REQUIRE((long)d->B * d->H <= 65535, "pbe_attention_f16: B*H too large");
    AttnP p;
    p.Q = (const h16*)d->Q; p.K = (const h16*)d->K; p.VT = (const h16*)d->VT; p.O = (h16*)d->O;
    p.B = d->B; p.H = d->H; p.Nq = d->Nq; p.Nk = d->Nk; p.D = d->D;
    p.q_bs = d->q_bs; p.q_rs = d->q_rs; p.k_bs = d->k_bs; p.k_rs = d->k_rs;
    p.vt_bs = d->vt_bs; p.vt_rs = d->vt_rs; p.o_bs = d->o_bs; p.o_rs = d->o_rs;
    p.scale_log2e = d->scale * 1.4426950408889634f;
    hipStream_t s = (hipStream_t)stream;
    pbe_prof_begin(PBE_K_ATTN, s);
    const int D = d->D;
    // 64 queries per wave when the grid still covers the chip (256 CUs x ~3 resident workgroups)
    const long blocks2 = (long)cdiv(d->Nq, 256) * d->B * d->H;
    const bool two = g_pbe_attn_qw ? g_pbe_attn_qw == 2 : (blocks2 >= 512 && D <= 80);
    if (D <= 16) launch_attn<16, 1>(p, s);
    else if (D <= 32) launch_attn<32, 1>(p, s);
    else if (D <= 48) { if (two) launch_attn<48, 2>(p, s); else launch_attn<48, 1>(p, s); }
    else if (D <= 64) { if (two) launch_attn<64, 2>(p, s); else launch_attn<64, 1>(p, s); }
    else if (D <= 80) { if (two) launch_attn<80, 2>(p, s); else launch_attn<80, 1>(p, s); }
    else if (D <= 128) launch_attn<128, 1>(p, s);
    else launch_attn<160, 1>(p, s);
    pbe_prof_end(PBE_K_ATTN, s, 4.0 * d->B * d->H * (double)d->Nq * d->Nk * d->D);
    PBE_LAUNCH_CHECK("pbe_attention_f16");
    return PBE_OK;
}
