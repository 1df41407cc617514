// Fused multi-head attention forward, softmax(Q K^T * scale) V, for gfx950 (wave64, MFMA 32x32x16 f16).
//
// One workgroup = 4 waves = 128 (or 256) queries of one (batch, head); each wave owns 32 (or 64)
// queries for the whole K/V sweep.  K / V^T tiles of 64 keys go global -> LDS by LDS-DMA
// (global_load_lds, 16 B per lane, no VGPR staging, no ds_write): the per-lane source pointers are
// computed once and advance by a constant per tile, so the staging costs ~3 VALU ops per 1 KB.
// Two LDS buffers, tile t+1 in flight under tile t's MFMAs, one barrier per tile.
//
// The score tile is computed TRANSPOSED, S^T = K Q^T, so the query index sits on the lane
// (column of the 32x32 accumulator) and the 32 keys of a sub-tile sit in the lane's registers:
// the row max / row sum are register-local plus ONE cross-half exchange, and the exponentiated
// accumulator registers feed the second product O^T = V^T P^T directly as its B operand (no LDS
// round trip, no shuffles).  A 32x32 accumulator keeps rows {4h..4h+3, 8+4h.., 16+4h.., 24+4h..}
// in half-wave h; the K tile is therefore staged with its rows PERMUTED (the DMA source address is
// per lane, so this is free): LDS row 4q+a of every 16-row group holds key 4*pi(q)+a, pi = (0,2,1,3).
// With that, the 8 accumulator registers a lane feeds to one PV k-step are keys 16s + 8h + 0..7 in
// natural order, and the matching V^T fragment is ONE 16-byte LDS read.
// V arrives already transposed ([head*D + d][key], produced for free by the projection GEMM with
// swapped operands).  Tile rows are padded to an odd number of 16-B slots (K: DP/8 | 1, V^T: 9):
// conflict-free ds_read_b128; the pad slots and the d-padding are zeroed once and never written.
#include "common.h"
#include "../../include/pbe_hip.h"

struct AttnP {
    const h16* Q; const h16* K; const h16* VT; h16* O;
    int B, H, Nq, Nk, D;
    long q_bs, q_rs, k_bs, k_rs, vt_bs, vt_rs, o_bs, o_rs;
    float scale_log2e;
    int q_pre;          // Q already multiplied by scale log2e (then scale_log2e == 1)
    int nqb;            // query blocks per (batch, head): the grid is 1-D, nqb * B * H workgroups
#ifdef PBE_ATTN_STAMPS
    unsigned long long* stamps;     // diagnostic build only (tools/attn_stamps.py): 4 words per workgroup, written by lane 0 of wave 0, read by nothing
#endif
};

#define PBE_GLDS16(gsrc, ldst)                                                                     \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc),         \
                                     (__attribute__((address_space(3))) void*)(ldst), 16, 0, 0)

template <int DP>
struct AttnTile {
    static constexpr int KT = 64;                 // keys per staged tile (one barrier per tile)
    static constexpr int DV = (DP + 31) / 32 * 32;
    static constexpr int DC = DP / 8;             // 16-B chunks per K row
    static constexpr int KR = DC | 1;             // slots per K row (odd)
    static constexpr int VR = KT / 8 + 1;         // slots per V^T row (odd)
    static constexpr int KSTR = KR * 16, VSTR = VR * 16;
    static constexpr int KSLOTS = KT * KR, VSLOTS = DV * VR, SLOTS = KSLOTS + VSLOTS;
    static constexpr int NI = (SLOTS + 63) / 64;  // LDS-DMA instructions (64 slots each) per tile
    static constexpr int NPW = (NI + 3) / 4;      // ... per wave
    static constexpr int BUF = NI * 1024;         // bytes per buffer
};

// key held by LDS row r of the K tile (middle quads of every 16 rows swapped)
__device__ __forceinline__ int attn_key_of_row(int r) { return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1); }

// Online softmax with a DEFERRED reference maximum (cdna_hip_programming.md T13): a query keeps a reference m (log2 domain) that is
// raised only when a tile's largest score exceeds it by more than ATTN_THR; P = exp2(s - m) is then bounded by 2^ATTN_THR = 256
// instead of 1 - exact in fp16 (11 significant bits at any magnitude, fewer subnormal P), and numerator (P V) and denominator (the
// ones row of V^T, or the VALU row sum) see the same rounded P.  The rescale of O - and, in the MPAD form, the subtraction of m -
// leave the per-tile path.
#define ATTN_THR 8.0f
#ifndef PBE_ATTN_PRIO
#define PBE_ATTN_PRIO 0      // 1: s_setprio 1 around the MFMA phases (A/B build, tools/attn_ab.py with PBE_LIB_PATH)
#endif

template <int DP, int QW, int KH, bool MPAD = false>
__global__ void __launch_bounds__(256, (QW == 2 && DP <= 48) ? 2 : 1) attn_kernel(const AttnP p) {
    // MPAD (D == DP - 8: the U-Net's d = 40 heads): the reference maximum rides in the padding of the head dimension.  Column D
    // of every K row is the constant 64, column D of the (pre-scaled) Q row holds -m / 64 as fp16, so the QK^T MFMA itself returns
    // s = scale log2e q.k - m and the softmax is max -> exp2 -> cvt with no multiply-add per score (32 of ~130 VALU issues per
    // unit at d = 40, where the VALU, not the matrix pipe, is the bound).  m is whatever fp16 value the kernel chose (a reference,
    // not the true maximum: |s| stays within the rounding of m, <= 2^-11 |m|), the factor 64 keeps |m| <= 4.1e6 representable.
    // QW = 32-query sub-blocks per wave (1 or 2).  QW = 2 halves the global K/V traffic, the LDS staging and the
    // barriers per query (more work between barriers); used for long sequences where the grid still fills the chip.
    // KH = 64-key tiles staged per barrier (1 or 2): the 4 waves of a workgroup sit on 4 different SIMDs and
    // re-synchronise at every barrier, so fewer barriers means less time lost to the slowest wave.
    using T = AttnTile<DP>;
    constexpr int KT = T::KT, DV = T::DV, KR = T::KR, VR = T::VR, KSTR = T::KSTR, VSTR = T::VSTR;
    constexpr int KSLOTS = T::KSLOTS, SLOTS = T::SLOTS, NPW = T::NPW, BUF = T::BUF;
    constexpr int K_BYTES = KSLOTS * 16;
    constexpr int NDS = DP / 16;          // k-steps of the QK^T product
    constexpr int NDT = DV / 32;          // 32-wide d tiles of the output
    constexpr int BQ = 128 * QW;          // queries per workgroup
    constexpr int NSLOT = 2 * KH;         // tile images in LDS: one group of KH in use, one in flight
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h5 = lane >> 5;
    // Workgroups are dealt round-robin over the 8 XCDs (ids i and i + 8 share one; speed only, never correctness): the query blocks of ONE
    // (batch, head) all stream the same K / V^T, so they are given ids of one residue mod 8 - that XCD's L2 then serves the 2nd .. 16th
    // sweep.  With the plain (q block, head) grid every XCD fetched every head's K / V^T: 172 MB left L2 per launch against 43 MB
    // algorithmic (profiles/r03 PMC passes).
    int bh, qblk;
    {
        const int nq = p.nqb, nbh = p.B * p.H, lin = blockIdx.x;
        if ((nbh & 7) == 0) { const int xcd = lin & 7, j = lin >> 3; qblk = j % nq; bh = (j / nq) * 8 + xcd; }
        else { qblk = lin % nq; bh = lin / nq; }
    }
    const int b = bh / p.H, h = bh - b * p.H;
    const int q0 = qblk * BQ + wave * 32 * QW + l31;          // sub-block i holds query q0 + 32 i
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

    // ---- LDS init: zero every tile image (pad slots, d padding), then the ones rows ----
    for (int i = tid; i < NSLOT * BUF / 16; i += 256) *reinterpret_cast<h16x8*>(smem + i * 16) = zero8;
    __syncthreads();
    if (ONES && tid < 8 * NSLOT)
        *reinterpret_cast<h16x8*>(smem + (tid >> 3) * BUF + K_BYTES + (DV - 1) * VSTR + (tid & 7) * 16) = one8;
    if constexpr (MPAD) {                                 // K[:, D] = 64 in every tile image (chunk DC - 1 of a row is never a DMA target)
        const h16x8 kpad = {(h16)64.f, 0, 0, 0, 0, 0, 0, 0};
        for (int i = tid; i < NSLOT * KT; i += 256)
            *reinterpret_cast<h16x8*>(smem + (i / KT) * BUF + (i % KT) * KSTR + (T::DC - 1) * 16) = kpad;
    }

    // ---- per-lane DMA sources: slot (j*4 + wave)*64 + lane of the tile image, advanced by a constant per tile ----
    const h16* src[NPW];
    int inc[NPW];
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
        const int slot = (j * 4 + wave) * 64 + lane;
        src[j] = nullptr; inc[j] = 0;
        if (slot < KSLOTS) {
            const int row = slot / KR, c = slot - row * KR;
            if (c * 8 < D) { src[j] = Kb + (long)attn_key_of_row(row) * p.k_rs + c * 8; inc[j] = KT * (int)p.k_rs; }
        } else if (slot < SLOTS) {
            const int sv = slot - KSLOTS, row = sv / VR, c = sv - row * VR;
            if (c < KT / 8 && row < D) { src[j] = Vb + (long)row * p.vt_rs + c * 8; inc[j] = KT; }
        }
    }
    const int nt = (p.Nk + KT - 1) / KT;
    const int rem = p.Nk - (nt - 1) * KT;                 // keys in the last tile (1..64)
    auto issue_tile = [&](int t, int buf) {
        unsigned char* dst = smem + buf * BUF + wave * 1024;
        if (t + 1 < nt || rem == KT) {                    // full tile: constant-stride pointers
#pragma unroll
            for (int j = 0; j < NPW; ++j) {
                if (src[j]) PBE_GLDS16(src[j], dst + j * 4096);
                src[j] += inc[j];
            }
        } else {                                          // ragged last tile: clamp K rows, skip V^T chunks past Nk
            const int kv0 = t * KT;
#pragma unroll
            for (int j = 0; j < NPW; ++j) {
                const int slot = (j * 4 + wave) * 64 + lane;
                const h16* s = nullptr;
                if (slot < KSLOTS) {
                    const int row = slot / KR, c = slot - row * KR;
                    const int key = min(kv0 + attn_key_of_row(row), p.Nk - 1);
                    if (c * 8 < D) s = Kb + (long)key * p.k_rs + c * 8;
                } else if (slot < SLOTS) {
                    const int sv = slot - KSLOTS, row = sv / VR, c = sv - row * VR;
                    if (c < KT / 8 && row < D && kv0 + c * 8 < p.Nk) s = Vb + (long)row * p.vt_rs + kv0 + c * 8;
                }
                if (s) PBE_GLDS16(s, dst + j * 4096);
            }
        }
    };
    auto fix_tail = [&](int buf) {                        // V^T columns past the last key -> 0 (P there is 0, 0 * garbage must stay 0)
        unsigned char* sv = smem + buf * BUF + K_BYTES;
        for (int i = tid; i < DV * KT; i += 256) {
            const int row = i >> 6, col = i & 63;
            if (col >= rem) *reinterpret_cast<h16*>(sv + row * VSTR + col * 2) = (h16)0.f;
        }
    };

    // Q fragments (B operand of S^T = K Q^T): lane = query column, 8 consecutive d per k-step half
    h16x8 qf[QW][NDS];
#pragma unroll
    for (int i = 0; i < QW; ++i)
#pragma unroll
        for (int ds = 0; ds < NDS; ++ds) {
            const int d0 = ds * 16 + 8 * h5;
            const bool ok = (q0 + 32 * i) < p.Nq && d0 < D;
            const h16* qsrc = ok ? Qb + (long)(q0 + 32 * i) * p.q_rs + d0 : p.Q;
            h16x8 v = *reinterpret_cast<const h16x8*>(qsrc);
            v = ok ? v : zero8;
            if constexpr (MPAD) {                            // scale log2e folded into Q: one extra fp16 rounding of q (relative 2^-11) unless
                if (!p.q_pre) {                              // the projection GEMM already applied it in fp32 (pbe_attn_desc.q_prescaled)
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (h16)((float)v[e] * p.scale_log2e);
                }
            }
            qf[i][ds] = v;
        }

    f32x16 o[QW][NDT];
    float m_run[QW], l_run[QW];
#pragma unroll
    for (int i = 0; i < QW; ++i) {
        m_run[i] = MPAD ? 0.f : -INFINITY; l_run[i] = 0.f;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][dt][r] = 0.f;
    }

    __syncthreads();                                      // init stores done before any DMA lands on them
#pragma unroll
    for (int j = 0; j < KH; ++j)
        if (j < nt) issue_tile(j, j);
    __syncthreads();                                      // (drains vmcnt) the first group landed for every wave
    if (nt <= KH && rem < KT) { fix_tail(nt - 1); __syncthreads(); }

    // ---- the phases of one (tile, 32-query sub-block) unit ----
    // K / V^T fragments are fetched from LDS into registers ONCE per tile (shared by the QW sub-blocks) and EARLY:
    // the V^T reads are issued before the softmax they hide under, the next tile's K reads right after the barrier.
    // (Fetching each fragment just before its MFMA costs an exposed LDS round trip per MFMA: 14 per unit.)
    constexpr bool PREF = DP <= 64, PREFV = DP <= 80;      // fragment registers: K 8 NDS, V^T 16 NDT VGPRs
    f32x16 sc[2];                                          // S^T accumulators of the two 32-key halves
    h16x8 pf[4];
    h16x8 kf[PREF ? 2 : 1][PREF ? NDS : 1];
    h16x8 vf[PREFV ? NDT : 1][4];
    auto load_k = [&](const unsigned char* sk) {
        if constexpr (PREF) {
#pragma unroll
            for (int ds = 0; ds < NDS; ++ds) {
                kf[0][ds] = *reinterpret_cast<const h16x8*>(sk + l31 * KSTR + (ds * 2 + h5) * 16);
                kf[1][ds] = *reinterpret_cast<const h16x8*>(sk + (32 + l31) * KSTR + (ds * 2 + h5) * 16);
            }
        }
    };
    auto load_v = [&](const unsigned char* sv) {
        if constexpr (PREFV) {
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    vf[dt][ks] = *reinterpret_cast<const h16x8*>(sv + (dt * 32 + l31) * VSTR + h5 * 16 + ks * 32);
        }
    };
    auto qk = [&](int i, const unsigned char* sk) {        // S^T = K Q^T : two 32-key sub-tiles
        f32x16& s0 = sc[0];
        f32x16& s1 = sc[1];
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
        if constexpr (PREF) {
#pragma unroll
            for (int ds = 0; ds < NDS; ++ds) {
                s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[0][ds], qf[i][ds], s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[1][ds], qf[i][ds], s1, 0, 0, 0);
            }
        } else {                                               // large d: fragments read one k-step ahead of their MFMAs
            const unsigned char* r0 = sk + l31 * KSTR + h5 * 16;
            const unsigned char* r1 = sk + (32 + l31) * KSTR + h5 * 16;
            h16x8 a0 = *reinterpret_cast<const h16x8*>(r0), a1 = *reinterpret_cast<const h16x8*>(r1);
#pragma unroll
            for (int ds = 0; ds < NDS; ++ds) {
                const h16x8 k0 = a0, k1 = a1;
                if (ds + 1 < NDS) {
                    a0 = *reinterpret_cast<const h16x8*>(r0 + (ds + 1) * 32);
                    a1 = *reinterpret_cast<const h16x8*>(r1 + (ds + 1) * 32);
                }
                s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0, qf[i][ds], s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k1, qf[i][ds], s1, 0, 0, 0);
            }
        }
    };
    // Online softmax over this lane's 32 keys (+ the other half-wave's 32).  The VALU budget matters (d = 40: 14 MFMAs per unit,
    // 448 matrix-pipe cycles, against what was ~130 VALU issues, ~600 cycles): the max runs on RAW scores, the reference maximum is
    // deferred (ATTN_THR), the cross-half exchange is one v_permlane32_swap, and in the MPAD form the scores arrive with the scale
    // and the reference already applied.  Leaves P^T in pf[]: pf[ks] = keys 16 ks + 8 h + 0..7.
    auto half_max = [&](float v) {                      // max with the other half-wave's value (same query, the other 32 keys)
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        const unsigned int u = __builtin_bit_cast(unsigned int, v);
        const u32x2 r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        unsigned int r0 = r[0], r1 = r[1];
        // hipcc (ROCm 7.2) treats the two results of a swap of one value with itself as EQUAL and folds max(r0, r1) to r0 (checked in
        // isolation: r0 + 2 r1 compiles to v1 + 2 v1): both results are made opaque before they are combined
        asm volatile("" : "+v"(r0), "+v"(r1));
        return fmaxf(__builtin_bit_cast(float, r0), __builtin_bit_cast(float, r1));
    };
    auto softmax = [&](int i, int kv0, bool first) {
        f32x16& s0 = sc[0];
        f32x16& s1 = sc[1];
        if (kv0 + KT > p.Nk) {                          // ragged last tile only (wave-uniform branch)
            int left = p.Nk - kv0 - 8 * h5;             // keys left from this half-wave's first key
            asm volatile("" : "+v"(left));              // pins the 32 compares inside the branch (else they are hoisted into every tile)
#pragma unroll
            for (int r = 0; r < 16; ++r) {              // register r of half-wave h: key 16 (r >> 3) + 8 h + (r & 7)
                const int k = 16 * (r >> 3) + (r & 7);
                s0[r] = (k >= left) ? -INFINITY : s0[r];
                s1[r] = (k + 32 >= left) ? -INFINITY : s1[r];
            }
        }
        float mx = fmaxf(s0[0], s1[0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, s0[r]), s1[r]);
        mx = half_max(mx);
        if constexpr (MPAD) {
            // s = scale log2e q.k - m_run already.  Raise the reference only when some query of the wave outgrew it by ATTN_THR (or at
            // the first tile, where the reference is still the arbitrary 0): every quantity at the old reference - O (with its ones
            // row), and THIS tile's scores - moves to the new one exactly once.
            if (__builtin_amdgcn_ballot_w64(first || mx > ATTN_THR) != 0) {
                float tgt = m_run[i] + (first ? mx : fmaxf(mx, 0.f));
                tgt = fminf(fmaxf(tgt, -4.0e6f), 4.0e6f);                       // -m / 64 must stay an fp16 number
                const float mref = 64.f * (float)(h16)(tgt * 0.015625f);
                const float d = mref - m_run[i];
                m_run[i] = mref;
                if (!first) {                                                     // (first tile: O is still 0, and 2^-d may overflow)
                    const float alpha = __builtin_amdgcn_exp2f(-d);
                    l_run[i] *= alpha;
#pragma unroll
                    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[i][dt][r] *= alpha;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) { s0[r] -= d; s1[r] -= d; }
                if (h5) qf[i][NDS - 1][0] = (h16)(-mref * 0.015625f);          // column D of this query's row: lanes 32-63, last k-step
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s0[r] = __builtin_amdgcn_exp2f(s0[r]);
                s1[r] = __builtin_amdgcn_exp2f(s1[r]);
            }
        } else {
            const float ms = mx * p.scale_log2e;
            if (__builtin_amdgcn_ballot_w64(ms - m_run[i] > ATTN_THR) != 0) {     // (first tile: m_run = -inf)
                const float m_new = fmaxf(m_run[i], ms);
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
        }
        if (!ONES) {
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) psum += s0[r] + s1[r];
            l_run[i] += psum;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            pf[0][j] = (h16)s0[j];
            pf[1][j] = (h16)s0[8 + j];
            pf[2][j] = (h16)s1[j];
            pf[3][j] = (h16)s1[8 + j];
        }
    };
    auto pv = [&](int i, const unsigned char* sv) {        // O^T += V^T P^T
        if constexpr (PREFV) {
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) o[i][dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[dt][ks], pf[ks], o[i][dt], 0, 0, 0);
        } else {                                               // large d: fragments read two MFMAs ahead
            const unsigned char* vb = sv + l31 * VSTR + h5 * 16;
            h16x8 nx[2] = {*reinterpret_cast<const h16x8*>(vb), *reinterpret_cast<const h16x8*>(vb + 32)};
#pragma unroll
            for (int x = 0; x < NDT * 4; ++x) {
                const h16x8 v = nx[x & 1];
                if (x + 2 < NDT * 4) nx[x & 1] = *reinterpret_cast<const h16x8*>(vb + ((x + 2) >> 2) * 32 * VSTR + ((x + 2) & 3) * 32);
                o[i][x >> 2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v, pf[x & 3], o[i][x >> 2], 0, 0, 0);
            }
        }
    };
    // one staged tile: every sub-block's S -> softmax -> PV; `next_k` = K image of the next tile when it is already in LDS
    auto tile = [&](int t, const unsigned char* next_k) {
        const unsigned char* sk = smem + (t % NSLOT) * BUF;
        const unsigned char* sv = sk + K_BYTES;
        const int kv0 = t * KT;
#pragma unroll
        for (int i = 0; i < QW; ++i) {
#if PBE_ATTN_PRIO
            __builtin_amdgcn_s_setprio(1);
#endif
            qk(i, sk);
#if PBE_ATTN_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            if (PREFV && i == 0) load_v(sv);                 // in flight under the first softmax
            if (PREF && i == QW - 1 && next_k) load_k(next_k);   // the S MFMAs above were the last readers of kf
            __builtin_amdgcn_sched_barrier(0);
            softmax(i, kv0, t == 0);
#if PBE_ATTN_PRIO
            __builtin_amdgcn_s_setprio(1);
#endif
            pv(i, sv);
#if PBE_ATTN_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            __builtin_amdgcn_sched_barrier(0);              // sub-blocks stay sequential: their S / P registers are shared
        }
    };

#ifdef PBE_ATTN_STAMPS
    if (p.stamps && tid == 0) { p.stamps[blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memtime(); p.stamps[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime(); }
#endif
    if (PREF) load_k(smem);
    for (int g = 0; g * KH < nt; ++g) {
        const int t0 = g * KH;
#pragma unroll
        for (int j = 0; j < KH; ++j)
            if (t0 + KH + j < nt) issue_tile(t0 + KH + j, (t0 + KH + j) % NSLOT);
#pragma unroll
        for (int j = 0; j < KH; ++j)
            if (t0 + j < nt) tile(t0 + j, (j + 1 < KH && t0 + j + 1 < nt) ? smem + ((t0 + j + 1) % NSLOT) * BUF : nullptr);
        __syncthreads();                                  // all waves done with this group; (vmcnt drained) the next group landed
        if (rem < KT && nt - 1 >= t0 + KH && nt - 1 < t0 + 2 * KH) { fix_tail((nt - 1) % NSLOT); __syncthreads(); }
        if (PREF && t0 + KH < nt) load_k(smem + ((t0 + KH) % NSLOT) * BUF);
        __builtin_amdgcn_sched_barrier(0);
    }

#ifdef PBE_ATTN_STAMPS
    if (p.stamps && tid == 0) { p.stamps[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memtime(); p.stamps[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime(); }
#endif
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

#ifdef PBE_ATTN_STAMPS
unsigned long long* g_pbe_attn_stamps = nullptr;
extern "C" int pbe_debug_set_attn_stamps(void* buf) { g_pbe_attn_stamps = (unsigned long long*)buf; return PBE_OK; }
#endif
int g_pbe_attn_pad_lds = 0;  // pbe_tune(8, bytes): extra dynamic LDS per workgroup (fewer resident workgroups per CU: occupancy experiments only)

template <int DP, int QW, int KH = 1, bool MPAD = false>
static void launch_attn(const AttnP& p, hipStream_t s) {
    constexpr size_t lds0 = 2 * KH * AttnTile<DP>::BUF;
    static_assert(lds0 <= 160 * 1024, "attention tile exceeds the LDS");
    const size_t lds = lds0 + (g_pbe_attn_pad_lds > 0 && lds0 + g_pbe_attn_pad_lds <= 160 * 1024 ? g_pbe_attn_pad_lds : 0);      // (occupancy experiments)
    static std::atomic<uint64_t> attr_done{0};
    pbe_raise_dynamic_lds(attr_done, reinterpret_cast<const void*>(&attn_kernel<DP, QW, KH, MPAD>), 160 * 1024);
    AttnP q = p;
    q.nqb = cdiv(p.Nq, 128 * QW);
#ifdef PBE_ATTN_STAMPS
    q.stamps = g_pbe_attn_stamps;
#endif
    dim3 grid((unsigned)(q.nqb * p.B * p.H));
    hipLaunchKernelGGL((attn_kernel<DP, QW, KH, MPAD>), grid, dim3(256), lds, s, q);
}

int g_pbe_attn_qw = 0;       // pbe_tune(3, v): 0 = heuristic, 1 / 2 = force queries-per-wave factor
int g_pbe_attn_mpad = 1;     // pbe_tune(6, 0/1): reference maximum in the head-dim padding (d = 40)

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
    PBE_REQUIRE((long)d->B * d->H * ((d->Nq + 127) / 128) < (1L << 31), "pbe_attention_f16: too many workgroups");
    AttnP p;
    p.Q = (const h16*)d->Q; p.K = (const h16*)d->K; p.VT = (const h16*)d->VT; p.O = (h16*)d->O;
    p.B = d->B; p.H = d->H; p.Nq = d->Nq; p.Nk = d->Nk; p.D = d->D;
    p.q_bs = d->q_bs; p.q_rs = d->q_rs; p.k_bs = d->k_bs; p.k_rs = d->k_rs;
    p.vt_bs = d->vt_bs; p.vt_rs = d->vt_rs; p.o_bs = d->o_bs; p.o_rs = d->o_rs;
    p.q_pre = d->q_prescaled ? 1 : 0;
    p.scale_log2e = p.q_pre ? 1.0f : d->scale * 1.4426950408889634f;
    hipStream_t s = (hipStream_t)stream;
    pbe_prof_begin(PBE_K_ATTN, s);
    const int D = d->D;
    // 64 queries per wave when the grid still covers the chip (256 CUs x ~3 resident workgroups)
    const long blocks2 = (long)cdiv(d->Nq, 256) * d->B * d->H;
    const bool two = g_pbe_attn_qw ? g_pbe_attn_qw == 2 : (blocks2 >= 512 && D <= 80);
    if (D <= 16) launch_attn<16, 1>(p, s);
    else if (D <= 32) launch_attn<32, 1>(p, s);
    else if (D == 40 && g_pbe_attn_mpad) {                                                                                             // the U-Net's 64x64 / CFG heads
        // (two K/V tiles per barrier where the grid covers the chip: 236 vs 240 us at 64x64, tools/attn_ab.py - the deeper prefetch alone is
        //  worth 1.5 %: the kernel is bound by its MFMA / VALU streams, not by the DMA latency)
        if (g_pbe_attn_qw == 2) launch_attn<48, 2, 1, true>(p, s);
        else if (g_pbe_attn_qw == 1 || !two) launch_attn<48, 1, 1, true>(p, s);
        else launch_attn<48, 2, 2, true>(p, s);
    }
    else if (D <= 48) { if (g_pbe_attn_qw == 3) launch_attn<48, 2, 2>(p, s); else if (two) launch_attn<48, 2>(p, s); else launch_attn<48, 1>(p, s); }
    else if (D <= 64) { if (two) launch_attn<64, 2>(p, s); else launch_attn<64, 1>(p, s); }
    else if (D <= 80) { if (two) launch_attn<80, 2>(p, s); else launch_attn<80, 1>(p, s); }
    else if (D <= 128) launch_attn<128, 1>(p, s);
    else launch_attn<160, 1>(p, s);
    pbe_prof_end(PBE_K_ATTN, s, 4.0 * d->B * d->H * (double)d->Nq * d->Nk * d->D,
                 2.0 * d->B * d->H * (double)d->D * (2.0 * d->Nq + 2.0 * d->Nk));                 // q, o, k, v^T once each
    PBE_LAUNCH_CHECK("pbe_attention_f16");
    return PBE_OK;
}
