// Implicit-GEMM on the gfx950 matrix cores: one kernel body, two activation loaders.
//
//   MODE 0 (dense)   C[m,n] = sum_k A[m,k] W[n,k]          Linear / 1x1 conv / bmm
//   MODE 1 (conv3x3) m = (b,oy,ox), k = (ci/cb, tap, ci%cb) NHWC 3x3 conv gather, zero padding,
//                                                          optional fused nearest-2x upsample and
//                                                          two-source channel concat
//
// Structure (CDNA4, wave64):
//   * workgroup = NWM x NWN waves, block tile BM x BN x 32, each wave owns (BM/NWM) x (BN/NWN) as
//     16x16 tiles of v_mfma_f32_16x16x32_f16.  The weights are the MFMA "A" operand and the
//     activations the "B" operand, so an accumulator register quad holds 4 consecutive n for one m.
//   * tiles stream global -> LDS with global_load_lds_dwordx4 (LDS-DMA, no staging registers)
//     into a 4-slot ring: 3 k-tiles in flight per workgroup across ONE raw s_barrier per k-tile,
//     retired with counted s_waitcnt vmcnt.  Measured: the kernel is bound by the per-CU LDS-DMA
//     fill rate (~30 GB/s/CU from the Infinity Cache), so the block tile is as large as the
//     problem allows (256x256 = 128 FLOP per staged byte) and small-M / deep-K problems are split
//     along K over gridDim.z with a deterministic fp32 slab reduction.
//   * LDS image: rows of 32 halfs (64 B = 4 chunks of 16 B).  An LDS-DMA instruction writes
//     64 lanes x 16 B linearly, so the bank swizzle is applied to the per-lane SOURCE chunk
//     (position p of row r holds global chunk p ^ f(r), f = {0,2,3,1}[(r>>2)&3]) and again on
//     the fragment read: every ds_read_b128 of a 16x16x32 fragment is bank-conflict free.
//     Out-of-range rows / taps / k read a 16-byte zero block (LDS-DMA cannot mask a lane).
//   * workgroup ids are remapped so each XCD owns a contiguous run of tiles, n fastest: the
//     weight panel and the activation rows a run touches stay in that XCD's L2.
//   * epilogue: alpha, bias, row-broadcast vector, activation in fp32 registers -> fp16 C tile in
//     LDS (one wave-row group at a time) -> whole 16-byte row segments to HBM, residual fused.
#include <type_traits>
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
    int H, Wd, C1, C2, Ho, Wo, cstride, pad, ups, cb;   // cb = channel block of the K order (multiple of 32)
    // split-K: gridDim.z slices of the k-tile range, fp32 partial slabs [splits][M][N]
    int splits; float* ws;
    int sv_ok;      // bias + row vector of a tile come from LDS (set per tile shape in launch_cfg)
};

__device__ __attribute__((aligned(16))) unsigned int g_pbe_zero16[4] = {0u, 0u, 0u, 0u};

#define PBE_GLDS16(gsrc, ldst)                                                                     \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc),         \
                                     (__attribute__((address_space(3))) void*)(ldst), 16, 0, 0)

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int BM, int BN, int NWM, int NWN, int MODE, int S = 4, bool PP = false>
__global__ void __launch_bounds__(NWM* NWN * 64) igemm_kernel(const IGemmP p, int tiles_n) {
    // PP (8-wave tiles only): "ping-pong" main loop.  Waves w and w + 4 share a SIMD; the two wave groups run the same
    // [R: fragment reads + DMA issue | M: MFMAs] sequence offset by ONE barrier, so while one group's MFMAs own the
    // SIMD's matrix pipe its partner's LDS reads, address arithmetic and LDS-DMA issue run in their shadow (without the
    // stagger both waves of a SIMD read at the same time and then contend for the pipe: the read latency is exposed
    // once per k-tile).  Two raw barriers per k-tile; see the hazard notes at the loop.
    // S = LDS ring depth (4: three k-tiles in flight, one workgroup per CU for the big tiles; 2: one in flight, the
    // smaller ring lets 2-4 workgroups share a CU so their prologues / epilogues overlap - shallow-K problems).
    constexpr int NW = NWM * NWN, NT = NW * 64;
    constexpr int D = S - 1;
    constexpr int WM = BM / NWM, WN = BN / NWN, TM = WM / 16, TN = WN / 16;
    constexpr int A_BYTES = BM * 64, W_BYTES = BN * 64, STAGE = A_BYTES + W_BYTES;
    constexpr int PW = BN / 16;                       // 16-row DMA pieces of the weight tile, dealt round-robin to the waves
    constexpr int LA = BM / 16 / NW, LW = (PW + NW - 1) / NW, LPT = LA + LW;
    constexpr int CLD = BN + 8;
    static_assert(LA >= 1 && BM % (16 * NW) == 0 && BN % 16 == 0, "activation pieces must divide over the waves");
    // When PW is not a multiple of NW (BN = 320, 8 waves) the waves left without a piece issue a dummy DMA of
    // the zero block into a 1-KiB dump slot, so every wave's vmcnt stays uniform (LPT loads per k-tile).
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % NWM, wn = wave / NWM;
    int tile;
    {   // XCD-aware tile order (bijective for any grid size)
        const int nwg = gridDim.x, id = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = id & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
    }
    const int tn_i = tile % tiles_n, tm_i = tile / tiles_n;
    const int m0 = tm_i * BM, n0 = tn_i * BN;
    const long bz = blockIdx.y;

    // ---- loader state: this lane's row inside a 16-row DMA piece and its source chunk ----
    const int lrow = lane >> 2;
    const int gch = (lane & 3) ^ ((0x78 >> (2 * ((lane >> 4) & 3))) & 3);
    const h16* zsrc = reinterpret_cast<const h16*>(g_pbe_zero16);

    bool a_ok[LA];
    const h16* a_row[LA];
    const h16* a_row2[LA];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int m = m0 + (wave * LA + i) * 16 + lrow;
        a_ok[i] = m < p.M;
        a_row[i] = p.A + bz * p.sA + (long)m * p.lda;
        a_row2[i] = p.A2 ? p.A2 + (long)m * p.lda2 : p.A;
    }
    // conv: K is ordered (channel block of 32, tap, channel) so the 9 taps of a pixel's 32 channels are
    // consecutive k-tiles — the shifted re-reads hit L2 instead of going back to the Infinity Cache / HBM
    // (measured: tap-major order re-fetched the input 9x beyond L2).  The tap -> input-pixel map of this
    // tile's BM rows is built once into LDS: tab[tap][row] = pixel index, or -1 outside the (virtual) image.
    int* tab = reinterpret_cast<int*>(smem + S * STAGE);
    unsigned char* dump = smem + S * STAGE + (MODE == 1 ? 9 * BM * 4 : 0);
    // Epilogue vectors of this tile, staged ONCE at kernel start (their global latency hides under the whole main loop;
    // fetched inside the epilogue they cost one exposed round trip per 16 output columns - measured 35 % of a K = 320 GEMM):
    // svec[s][c] = bias[n0 + c] + rowvec[first sample of the tile + s][n0 + c], up to 4 samples per tile.
    float* svec = reinterpret_cast<float*>(smem + (S * STAGE > WM * CLD * 2 ? S * STAGE : WM * CLD * 2) + (MODE == 1 ? 9 * BM * 4 : 0) +
                                           ((PW % NW) ? 1024 : 0));
    const int sv_ns = (p.rowvec && p.group_rows < BM) ? BM / p.group_rows : 1;       // samples per tile (tile is sample-aligned when sv_ok)
    if (p.sv_ok && p.splits <= 1) {
        const int s0 = p.rowvec ? m0 / p.group_rows : 0;
        for (int idx = tid; idx < sv_ns * BN; idx += NT) {
            const int si = idx / BN, c = idx - si * BN, n = n0 + c;
            float v = 0.f;
            if (n < p.N) {
                if (p.bias && !p.bias_row) v = p.bias[n];
                if (p.rowvec && (long)(s0 + si) * p.group_rows < p.M) v += (float)p.rowvec[(long)(s0 + si) * p.ldv + n];
            }
            svec[si * BN + c] = v;
        }
    }
    if (MODE == 1) {
        const int hw = p.Ho * p.Wo, Hv = p.H << p.ups, Wv = p.Wd << p.ups;
        for (int row = tid; row < BM; row += NT) {            // one thread per tile row: one (b, oy, ox) decode, 9 taps
            const int m = m0 + row;
            const bool rok = m < p.M;
            const int b = m / hw, rem = m - b * hw;
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            const int iy0 = oy * p.cstride - p.pad, ix0 = ox * p.cstride - p.pad;
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                const int iy = iy0 + tp / 3, ix = ix0 + tp % 3;
                const bool ok = rok && (unsigned)iy < (unsigned)Hv && (unsigned)ix < (unsigned)Wv;
                tab[tp * BM + row] = ok ? (b * p.H + (iy >> p.ups)) * p.Wd + (ix >> p.ups) : -1;
            }
        }
        __syncthreads();
    }
    bool w_ok[LW];
    const h16* w_row[LW];
#pragma unroll
    for (int i = 0; i < LW; ++i) {
        const int pc = wave + NW * i;
        const int n = n0 + pc * 16 + lrow;
        w_ok[i] = pc < PW && n < p.N;
        w_row[i] = p.W + bz * p.sW + (long)(w_ok[i] ? n : 0) * p.ldw;
    }
    const int nk_all = (p.K + 31) >> 5;
    int kt0 = 0, nk = nk_all;                        // this workgroup's k-tile range [kt0, nk)
    if (p.splits > 1) {
        const int per = (nk_all + p.splits - 1) / p.splits;
        kt0 = blockIdx.z * per;
        nk = min(nk_all, kt0 + per);
    }
    // conv K order: (channel block of cb, tap, channel): state of the NEXT k-tile to issue
    const int KB = MODE == 1 ? p.cb >> 5 : 1;        // k-tiles per (block, tap) visit
    int tap = 0, c0 = 0, kj = 0;
    if (MODE == 1 && kt0 > 0) {
        const int per_blk = 9 * KB, cblk = kt0 / per_blk, r = kt0 - cblk * per_blk;
        tap = r / KB; kj = r - tap * KB; c0 = cblk * p.cb + kj * 32;
    }
    const h16* a_src[LA];
    bool fresh = true;
#pragma unroll
    for (int i = 0; i < LA; ++i) a_src[i] = zsrc;

    auto issue = [&](int kt) {
        unsigned char* sa = smem + (kt & (S - 1)) * STAGE;
        unsigned char* sw = sa + A_BYTES;
        const int k = kt * 32 + gch * 8;
        const bool kok = k < p.K;
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < LA; ++i) {
                const h16* src = (k < p.K1) ? a_row[i] + k : a_row2[i] + (k - p.K1);
                src = (a_ok[i] && kok) ? src : zsrc;
                PBE_GLDS16(src, sa + (wave * LA + i) * 1024);
            }
        } else {
            if (kj == 0 || fresh) {                   // new (block, tap): look the pixels up; otherwise +32 channels
                fresh = false;
                const bool first = c0 < p.C1;         // which concat source this channel block lives in (uniform)
                const h16* base = first ? p.A + c0 + gch * 8 : p.A2 + (c0 - p.C1) + gch * 8;
                const long cs = first ? p.C1 : p.C2;
#pragma unroll
                for (int i = 0; i < LA; ++i) {
                    const int pix = tab[tap * BM + (wave * LA + i) * 16 + lrow];
                    a_src[i] = pix >= 0 ? base + (long)pix * cs : nullptr;
                }
            }
#pragma unroll
            for (int i = 0; i < LA; ++i) {
                PBE_GLDS16(a_src[i] ? a_src[i] : zsrc, sa + (wave * LA + i) * 1024);
                if (a_src[i]) a_src[i] += 32;
            }
            c0 += 32;
            if (++kj == KB) {
                kj = 0;
                if (++tap == 9) tap = 0; else c0 -= p.cb;      // next tap of the same block, or first tap of the next block
            }
        }
#pragma unroll
        for (int i = 0; i < LW; ++i) {
            const h16* src = (w_ok[i] && kok) ? w_row[i] + k : zsrc;
            const int pc = wave + NW * i;
            PBE_GLDS16(src, (PW % NW == 0 || pc < PW) ? sw + pc * 1024 : dump);
        }
    };

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    const int rsw = (fq ^ ((0x78 >> (2 * ((fr >> 2) & 3))) & 3)) << 4;
    const int a_rd = (wm * WM + fr) * 64 + rsw, w_rd = (wn * WN + fr) * 64 + rsw;

#pragma unroll
    for (int t = 0; t < D; ++t)
        if (kt0 + t < nk) issue(kt0 + t);

    if constexpr (!PP) {
    for (int kt = kt0; kt < nk; ++kt) {
        const int rem = nk - 1 - kt;                 // tiles issued after tile kt that may stay in flight
        if (rem >= D - 1) wait_vmcnt<(D - 1) * LPT>();
        else if (rem == 1) wait_vmcnt<LPT>();
        else wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                // tile kt landed for every wave; slot (kt-1)%S is free
        const unsigned char* sa = smem + (kt & (S - 1)) * STAGE;
        const unsigned char* sw = sa + A_BYTES;
        h16x8 fa[TM], fw[TN];
#pragma unroll
        for (int j = 0; j < TM; ++j) fa[j] = *reinterpret_cast<const h16x8*>(sa + a_rd + j * 16 * 64);
#pragma unroll
        for (int i = 0; i < TN; ++i) fw[i] = *reinterpret_cast<const h16x8*>(sw + w_rd + i * 16 * 64);
        __builtin_amdgcn_sched_barrier(0);           // fragment reads go out first; the DMA address math runs in their shadow
        if (kt + D < nk) issue(kt + D);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[i], fa[j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    }
    } else {
        static_assert(!PP || (NW == 8 && S == 4), "ping-pong needs two waves per SIMD and the 4-slot ring");
        // Barriers b0, b1, ... as group 0 (waves 0-3) counts them: group 0 runs R(kt) | b(2i) | M(kt) | b(2i+1), group 1
        // (waves 4-7) passes one extra barrier first and therefore runs R(kt) between b(2i) and b(2i+1).
        //   RAW: a wave retires its own DMAs of tile kt+1 (counted vmcnt) at the END of R(kt), i.e. before b(2i) / b(2i+1);
        //        tile kt+1 is first read by group 0 after b(2i+1) - one barrier after every wave's wait.
        //   WAR: DMA(kt+3) overwrites slot (kt-1) % 4.  Its last readers (group 1, R(kt-1)) complete their reads
        //        (lgkmcnt(0)) before b(2i-1); the earliest issue (group 0, R(kt)) comes after b(2i-1).
        const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);
        if (kt0 < nk) {
            const int rem0 = nk - 1 - kt0;
            if (rem0 >= D - 1) wait_vmcnt<(D - 1) * LPT>();
            else if (rem0 == 1) wait_vmcnt<LPT>();
            else wait_vmcnt<0>();
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                // tile kt0 landed for every wave
        if (grp == 1) __builtin_amdgcn_s_barrier();  // the stagger
        for (int kt = kt0; kt < nk; ++kt) {
            const unsigned char* sa = smem + (kt & (S - 1)) * STAGE;
            const unsigned char* sw = sa + A_BYTES;
            h16x8 fa[TM], fw[TN];
#pragma unroll
            for (int j = 0; j < TM; ++j) fa[j] = *reinterpret_cast<const h16x8*>(sa + a_rd + j * 16 * 64);
#pragma unroll
            for (int i = 0; i < TN; ++i) fw[i] = *reinterpret_cast<const h16x8*>(sw + w_rd + i * 16 * 64);
            __builtin_amdgcn_sched_barrier(0);
            if (kt + D < nk) issue(kt + D);
            const int rem = nk - 2 - kt;             // tiles issued after tile kt+1 that may stay in flight
            if (rem >= D - 1) wait_vmcnt<(D - 1) * LPT>();
            else if (rem == 1) wait_vmcnt<LPT>();
            else wait_vmcnt<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();            // end of R: my fragments are in registers, my share of tile kt+1 landed
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[i], fa[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();            // end of M
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();  // group 1's last M phase
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    if (p.splits > 1) {
        // raw fp32 partial sums -> slab blockIdx.z; splitk_reduce_kernel applies the epilogue
        float* slab = p.ws + (long)blockIdx.z * p.M * p.N;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int n = n0 + wn * WN + i * 16 + fq * 4;
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int m = m0 + wm * WM + j * 16 + fr;
                if (m < p.M && n < p.N) *reinterpret_cast<f32x4*>(slab + (long)m * p.N + n) = acc[i][j];
            }
        }
        return;
    }

    // ---- epilogue: one wave-row group (WM rows of the tile) at a time through LDS ----
    h16* sC = reinterpret_cast<h16*>(smem);
    h16* Cb = p.C + bz * p.sC;
    const h16* Rb = p.resid ? p.resid + bz * p.sR : nullptr;
    constexpr int CPR = BN / 8;
    // whole C tile at once when it fits the ring's LDS, else one wave-row group per pass
    constexpr bool ONE_PASS = (size_t)BM * CLD * 2 <= (size_t)S * STAGE;
    constexpr int NG = ONE_PASS ? 1 : NWM;            // passes
    constexpr int GR = ONE_PASS ? BM : WM;            // rows per pass
    // The register -> LDS half has a compile-time FAST path (bias / row vector staged in svec, one sample per tile, no
    // per-row bias): the generic form (per-element bounds checks, global bias loads and row-vector gathers with an
    // integer division per quad, all in one unrolled body) ran ~7 000 instructions per thread and cost 35 % of a
    // K = 320 GEMM.
    // Tiles with <= 96 accumulator registers also specialise the activation at compile time (5 copies of the unrolled body);
    // for the 256-row tiles (128-160 accumulator registers) that many copies push the allocator into scratch, so there the
    // activation stays a uniform run-time branch per quad.
    constexpr bool ARMS = TM * TN * 4 <= 96;
    auto stage = [&](int g, auto FAST, auto ACT) {
        constexpr bool F = decltype(FAST)::value;
        const int A = decltype(ACT)::value >= 0 ? decltype(ACT)::value : p.act;
        // alpha is re-materialised opaquely per pass: as a plain loop invariant, acc * alpha is hoisted out of the pass loop
        // into a second full set of accumulator registers and the 256-row tiles spill (measured: 30 us of epilogue per tile).
        float al = p.alpha;
        asm volatile("" : "+s"(al));
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int nl = wn * WN + i * 16 + fq * 4;
            const int n = n0 + nl;
            float bn[4] = {0.f, 0.f, 0.f, 0.f};
            if (F) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(svec + nl);
#pragma unroll
                for (int r = 0; r < 4; ++r) bn[r] = t[r];
            } else if (!p.sv_ok && p.bias && !p.bias_row) {
#pragma unroll
                for (int r = 0; r < 4; ++r) bn[r] = (n + r < p.N) ? p.bias[n + r] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int ml = (ONE_PASS ? wm * WM : 0) + j * 16 + fr;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf(acc[i][j][r], al, bn[r]);
                if (!F) {
                    const int m = m0 + g * GR + ml;
                    if (p.sv_ok) {                           // several samples per tile (8x8 level), or a per-row bias
                        const f32x4 t = *reinterpret_cast<const f32x4*>(svec + (sv_ns > 1 ? (g * GR + ml) / p.group_rows : 0) * BN + nl);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += t[r];
                    } else if (p.rowvec && m < p.M) {
                        const h16* rv = p.rowvec + (long)(m / p.group_rows) * p.ldv + n;
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (n + r < p.N) v[r] += (float)rv[r];
                    }
                    if (p.bias && p.bias_row) {
                        const float bm = (m < p.M) ? p.bias[m] : 0.f;
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += bm;
                    }
                }
                if (A == PBE_ACT_GEGLU) {                    // columns interleaved (x_j, gate_j): out_j = x_j * gelu(gate_j)
                    h16x2 o2 = {(h16)(v[0] * gelu_erf_f(v[1])), (h16)(v[2] * gelu_erf_f(v[3]))};
                    *reinterpret_cast<h16x2*>(sC + ml * CLD + (nl >> 1)) = o2;
                } else {
                    apply_act4(v, A);
                    h16x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (h16)v[r];
                    *reinterpret_cast<h16x4*>(sC + ml * CLD + nl) = o;
                }
            }
        }
    };
    // LDS -> global half: whole 16-byte row segments, residual added on the way out
    auto copy_out = [&](int g, auto GG) {
        constexpr bool gg = decltype(GG)::value;             // GEGLU halves the output width
        constexpr int cpr = gg ? CPR / 2 : CPR;
        const int Nout = gg ? p.N >> 1 : p.N, nb = gg ? n0 >> 1 : n0;
        for (int idx = tid; idx < GR * cpr; idx += NT) {
            const int row = idx / cpr, ch = idx - row * cpr;
            const int m = m0 + g * GR + row, n = nb + ch * 8;
            if (m >= p.M || n >= Nout) continue;
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
                    if (n + e < Nout) {
                        float f = (float)v[e];
                        if (Rb) f += (float)Rb[(long)m * p.ldr + n + e];
                        Cb[(long)m * p.ldc + n + e] = (h16)f;
                    }
                }
            }
        }
    };
    const bool fast = p.sv_ok && sv_ns == 1 && !(p.bias && p.bias_row);
#pragma unroll 1
    for (int g = 0; g < NG; ++g) {
        __syncthreads();                              // ring reads (g == 0) / previous group's copy-out done
        if (ONE_PASS || wm == g) {
            auto run = [&](auto FAST) {
                if constexpr (ARMS) {
                    switch (p.act) {
                        case 1: stage(g, FAST, std::integral_constant<int, 1>{}); break;
                        case 2: stage(g, FAST, std::integral_constant<int, 2>{}); break;
                        case 3: stage(g, FAST, std::integral_constant<int, 3>{}); break;
                        case PBE_ACT_GEGLU: stage(g, FAST, std::integral_constant<int, PBE_ACT_GEGLU>{}); break;
                        default: stage(g, FAST, std::integral_constant<int, 0>{}); break;
                    }
                } else {
                    stage(g, FAST, std::integral_constant<int, -1>{});
                }
            };
            if (fast) run(std::true_type{});
            else run(std::false_type{});
        }
        __syncthreads();
        if (p.act == PBE_ACT_GEGLU) copy_out(g, std::true_type{});
        else copy_out(g, std::false_type{});
    }
}

// Sum the split-K slabs in a fixed order (deterministic) and apply the epilogue: 4 columns per thread.
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const IGemmP p) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int n4 = p.N >> 2;
    if (i >= (long)p.M * n4) return;
    const int m = (int)(i / n4), n = (int)(i - (long)m * n4) * 4;
    // every global load of this thread is issued up front (epilogue operands, then the slabs four at a time): with one
    // load per loop trip the kernel is a chain of `splits` memory round trips.  The ADD order stays slab 0, 1, 2, ... (deterministic).
    float ev[4] = {0.f, 0.f, 0.f, 0.f}, rs[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
#pragma unroll
        for (int r = 0; r < 4; ++r) ev[r] = p.bias_row ? p.bias[m] : p.bias[n + r];
    }
    if (p.rowvec) {
        const h16* rv = p.rowvec + (long)(m / p.group_rows) * p.ldv + n;
#pragma unroll
        for (int r = 0; r < 4; ++r) ev[r] += (float)rv[r];
    }
    if (p.resid && p.act != PBE_ACT_GEGLU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) rs[r] = (float)p.resid[(long)m * p.ldr + n + r];
    }
    const float* sl = p.ws + (long)m * p.N + n;
    const long zs = (long)p.M * p.N;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    int z = 0;
    for (; z + 4 <= p.splits; z += 4) {
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(sl + (z + 0) * zs), t1 = *reinterpret_cast<const f32x4*>(sl + (z + 1) * zs);
        const f32x4 t2 = *reinterpret_cast<const f32x4*>(sl + (z + 2) * zs), t3 = *reinterpret_cast<const f32x4*>(sl + (z + 3) * zs);
        a += t0; a += t1; a += t2; a += t3;
    }
    for (; z < p.splits; ++z) a += *reinterpret_cast<const f32x4*>(sl + z * zs);
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = a[r] * p.alpha + ev[r];
    if (p.act == PBE_ACT_GEGLU) {
        h16x2 o2 = {(h16)(v[0] * gelu_erf_f(v[1])), (h16)(v[2] * gelu_erf_f(v[3]))};
        *reinterpret_cast<h16x2*>(p.C + (long)m * p.ldc + (n >> 1)) = o2;
        return;
    }
    h16x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float f = (float)(h16)apply_act(v[r], p.act) + rs[r];          // same rounding point as the fused epilogue
        o[r] = (h16)f;
    }
    *reinterpret_cast<h16x4*>(p.C + (long)m * p.ldc + n) = o;
}

// ---- host side --------------------------------------------------------------------------------
static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
struct Plan { int cfg; int splits; };
struct TileCfg { int bm, bn, nwm, nwn, slots_per_cu; double eff; };
// eff = relative per-FLOP efficiency of the tile when the chip is full (ordered by staged bytes per FLOP)
static const TileCfg kCfg[] = {
    {256, 256, 2, 4, 1, 1.00}, {256, 128, 4, 2, 1, 0.82}, {128, 256, 2, 4, 1, 0.75},
    {128, 128, 2, 2, 2, 0.88}, {128, 64, 2, 2, 2, 0.65}, {64, 128, 2, 2, 2, 0.65}, {64, 64, 2, 2, 2, 0.60},
    {256, 320, 2, 4, 1, 1.05},           // N = 320 / 640 / 960 / 1280 without column padding (142 FLOP per staged byte)
    {128, 320, 2, 4, 1, 0.96}};          // same, half the rows: fills the chip when M / 256 < 256 tiles
// (Ring depth 2 variants of the 128-row tiles - 2-4 workgroups per CU - were measured for every shape of the path,
//  profiles/r01_autotune_report.txt round "S2": 20-30 % slower than depth 4 on the shallow-K GEMMs they were meant for.)
static const int kNCfg = sizeof(kCfg) / sizeof(kCfg[0]);

int g_pbe_force_cfg = -1;        // pbe_tune(1, cfg index [| splits << 8]) forces a tile config (and split-K factor); -1 = heuristic
int g_pbe_allow_splitk = 1;      // pbe_tune(2, 0/1)

static int splits_for(const IGemmP& p, const TileCfg& c, int batch, size_t ws_bytes, long tiles) {
    if (!g_pbe_allow_splitk || batch != 1 || !p.ws || (p.N & 3) || (p.ldc & 3) || (p.resid && (p.ldr & 3))) return 1;
    const int nk = (p.K + 31) >> 5;
    const long slots = 256L * c.slots_per_cu;
    if (tiles * 4 > slots * 3 || nk < 16) return 1;          // grid already fills >= 75 % of the chip
    int s = (int)((slots * 5 / 4 + tiles - 1) / tiles);
    if (s > nk / 8) s = nk / 8;
    if (s > 32) s = 32;
    while (s > 1 && (size_t)s * p.M * p.N * sizeof(float) > ws_bytes) --s;
    if (s < 2) return 1;
    const int per = (nk + s - 1) / s;
    return (nk + per - 1) / per;                      // no empty slice
}

// Shallow-K problems (< 20 k-tiles) are dominated by the prologue / epilogue, deep-K problems by staged bytes per
// FLOP.  Both efficiency rows are fitted to the 472 measured shapes of profiles/r01_autotune_report.txt (the
// heuristic then costs 4 % over the best tile per shape, 12 % before the fit).  pbe_amd/tuned_mi355x.json overrides
// this per shape (desc.tile_cfg), so these rows only decide shapes outside the table.
static const double kEffShallow[] = {0.82, 0.72, 0.66, 1.00, 0.84, 0.85, 0.80, 0.80, 0.95};

// want_cfg: -1 = heuristic; else (tile config index) | (split-K factor << 8), factor 0 = heuristic factor for that tile.
// A requested factor is clamped to what the problem allows (batch 1, >= 8 k-tiles per slice, slabs fit the workspace).
static Plan plan_igemm(const IGemmP& p, int batch, size_t ws_bytes, int want_cfg) {
    Plan best{3, 1};
    double best_score = -1.0;
    const int want = g_pbe_force_cfg >= 0 ? g_pbe_force_cfg : want_cfg;
    const int forced = (want >= 0 && (want & 255) < kNCfg) ? (want & 255) : -1;
    const int want_splits = want >= 0 ? (want >> 8) & 255 : 0;
    const bool shallow = ((p.K + 31) >> 5) < 20;
    for (int c = 0; c < kNCfg; ++c) {
        if (forced >= 0 && c != forced) continue;
        TileCfg t = kCfg[c];
        if (shallow) t.eff = kEffShallow[c];
        const long tm = (p.M + t.bm - 1) / t.bm, tn = (p.N + t.bn - 1) / t.bn;
        const long tiles = tm * tn * batch;
        int sp = splits_for(p, t, batch, ws_bytes, tiles);
        if (forced >= 0 && want_splits > 0) {
            const int nk = (p.K + 31) >> 5;
            sp = want_splits;
            if (!g_pbe_allow_splitk || batch != 1 || !p.ws || (p.N & 3) || (p.ldc & 3) || (p.resid && (p.ldr & 3))) sp = 1;
            if (sp > nk / 8) sp = nk / 8;
            while (sp > 1 && (size_t)sp * p.M * p.N * sizeof(float) > ws_bytes) --sp;
            if (sp < 2) sp = 1;
            else { const int per = (nk + sp - 1) / sp; sp = (nk + per - 1) / per; }
        }
        const double useful = (double)p.M * p.N * batch / ((double)tiles * t.bm * t.bn);
        const double blocks = (double)tiles * sp, slots = 256.0 * t.slots_per_cu;
        const double rounds = (double)((long)((blocks + slots - 1) / slots));
        const double quant = blocks / (rounds * slots);
        const double split_cost = sp > 1 ? 0.93 : 1.0;        // slab write + reduce launch
        const double score = t.eff * useful * (0.30 + 0.70 * quant) * split_cost;
        if (score > best_score) { best_score = score; best = Plan{c, sp}; }
    }
    return best;
}

int g_pbe_pingpong = 1;          // pbe_tune(4, 0/1): staggered two-group main loop for the 8-wave tiles

template <int BM, int BN, int NWM, int NWN, int MODE, int S = 4, bool PP = false>
static void launch_cfg(IGemmP p, int batch, hipStream_t s) {
    if constexpr (!PP && NWM * NWN == 8 && S == 4) {
        if (g_pbe_pingpong) { launch_cfg<BM, BN, NWM, NWN, MODE, S, true>(p, batch, s); return; }
    }
    constexpr size_t ring = S * (BM + BN) * 64;
    constexpr size_t c_bytes = (size_t)(BM / NWM) * (BN + 8) * 2;
    constexpr size_t lds = (ring > c_bytes ? ring : c_bytes) + (MODE == 1 ? 9 * BM * sizeof(int) : 0) + ((BN / 16) % (NWM * NWN) ? 1024 : 0) +
                           4 * BN * sizeof(float);                                      // + svec[NSV = 4][BN]
    p.sv_ok = !p.rowvec || p.group_rows % BM == 0 || (BM % p.group_rows == 0 && BM / p.group_rows <= 4);
    static_assert(lds <= 160 * 1024, "tile does not fit the 160 KiB LDS");
    static std::atomic<uint64_t> attr_done{0};
    pbe_raise_dynamic_lds(attr_done, reinterpret_cast<const void*>(&igemm_kernel<BM, BN, NWM, NWN, MODE, S, PP>), (int)lds);
    const int tiles_m = cdiv(p.M, BM), tiles_n = cdiv(p.N, BN);
    dim3 grid((unsigned)(tiles_m * tiles_n), batch, p.splits > 1 ? p.splits : 1);
    // profiling brackets exactly ONE kernel each, so the event averages agree with rocprofv3's per-kernel averages
    pbe_prof_begin(MODE == 1 ? PBE_K_CONV3 : PBE_K_GEMM, s);
    hipLaunchKernelGGL((igemm_kernel<BM, BN, NWM, NWN, MODE, S, PP>), grid, dim3(NWM * NWN * 64), lds, s, p, tiles_n);
    pbe_prof_end(MODE == 1 ? PBE_K_CONV3 : PBE_K_GEMM, s, 2.0 * p.M * (double)p.N * p.K * batch);
    if (p.splits > 1) {
        const long work = (long)p.M * (p.N >> 2);
        pbe_prof_begin(PBE_K_SPLITK, s);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, s, p);
        pbe_prof_end(PBE_K_SPLITK, s, (double)p.M * p.N * (4.0 * p.splits + 2.0));       // bytes: fp32 slabs in, fp16 out
    }
}

template <int MODE>
static void dispatch_igemm(IGemmP p, int batch, hipStream_t s, size_t ws_bytes, int want_cfg) {
    const Plan pl = plan_igemm(p, batch, ws_bytes, want_cfg);
    p.splits = pl.splits;
    switch (pl.cfg) {
        case 0: launch_cfg<256, 256, 2, 4, MODE>(p, batch, s); break;
        case 1: launch_cfg<256, 128, 4, 2, MODE>(p, batch, s); break;
        case 2: launch_cfg<128, 256, 2, 4, MODE>(p, batch, s); break;
        case 3: launch_cfg<128, 128, 2, 2, MODE>(p, batch, s); break;
        case 4: launch_cfg<128, 64, 2, 2, MODE>(p, batch, s); break;
        case 5: launch_cfg<64, 128, 2, 2, MODE>(p, batch, s); break;
        case 7: launch_cfg<256, 320, 2, 4, MODE>(p, batch, s); break;
        case 8: launch_cfg<128, 320, 2, 4, MODE>(p, batch, s); break;
        default: launch_cfg<64, 64, 2, 2, MODE>(p, batch, s); break;
    }
}

extern "C" int pbe_tune(int32_t key, int32_t value) {
    if (key == 1) { g_pbe_force_cfg = (value >= 0 && (value & 255) < kNCfg) ? value : -1; return PBE_OK; }
    if (key == 2) { g_pbe_allow_splitk = value ? 1 : 0; return PBE_OK; }
    if (key == 3) { extern int g_pbe_attn_qw; g_pbe_attn_qw = value; return PBE_OK; }
    if (key == 4) { g_pbe_pingpong = value ? 1 : 0; return PBE_OK; }
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
    PBE_REQUIRE(d->ldc >= Nout, "%s: ldc too small", who);
    p.vec = (Nout % 8 == 0) && (d->ldc % 8 == 0) && al16(d->C) && (d->strideC % 8 == 0) &&
            (!d->resid || ((d->ldr % 8 == 0) && al16(d->resid) && (d->strideR % 8 == 0)));
    p.ws = (float*)d->workspace;
    return PBE_OK;
}

extern "C" int pbe_gemm_f16(const pbe_gemm_desc* d, pbe_stream_t stream) {
    IGemmP p;
    const int rc = fill_gemm(d, p, "pbe_gemm_f16");
    if (rc != PBE_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    dispatch_igemm<0>(p, d->batch, s, d->workspace ? d->workspace_bytes : 0, d->tile_cfg);
    PBE_LAUNCH_CHECK("pbe_gemm_f16");
    return PBE_OK;
}

static void report_plan(const IGemmP& p, int batch, size_t ws_bytes, int want_cfg, int32_t* out) {
    const Plan pl = plan_igemm(p, batch, ws_bytes, want_cfg);
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
    report_plan(p, d->batch, d->workspace ? d->workspace_bytes : 0, d->tile_cfg, out6);
    *workspace_needed = out6[1] > 1 ? (size_t)out6[1] * p.M * p.N * sizeof(float) : 0;
    return PBE_OK;
}

static int fill_conv(const pbe_conv3x3_desc* d, IGemmP& p, const char* who) {
    PBE_REQUIRE(d && d->X && d->Wp && d->Y, "%s: null operand", who);
    const int Cin = d->C1 + d->C2;
    PBE_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Cout > 0, "%s: bad dims", who);
    PBE_REQUIRE(d->C1 > 0 && d->C1 % 32 == 0 && d->C2 >= 0 && d->C2 % 32 == 0, "%s: C1=%d C2=%d must be multiples of 32 (use pbe_im2col3x3_f16 + pbe_gemm_f16 for small Cin)", who, d->C1, d->C2);
    PBE_REQUIRE((d->C2 == 0) == (d->X2 == nullptr), "%s: X2 / C2 mismatch", who);
    PBE_REQUIRE(d->stride == 1 || d->stride == 2, "%s: stride must be 1 or 2", who);
    PBE_REQUIRE(d->pad == 0 || d->pad == 1, "%s: pad must be 0 or 1", who);
    PBE_REQUIRE(d->upsample == 0 || (d->upsample == 1 && d->stride == 1), "%s: upsample only with stride 1", who);
    PBE_REQUIRE(al16(d->X) && al16(d->Wp) && al16(d->Y) && (!d->X2 || al16(d->X2)), "%s: 16-byte alignment", who);
    const int Hv = d->H << d->upsample, Wv = d->W << d->upsample;
    // output size: pad=1 -> floor((Hv + 2 - 3)/s) + 1 ; pad=0 is the VAE (0,1,0,1) pad: floor((Hv + 1 - 3)/s) + 1
    const int extra = d->pad ? 2 : 1;
    const int Ho = (Hv + extra - 3) / d->stride + 1, Wo = (Wv + extra - 3) / d->stride + 1;
    PBE_REQUIRE(Ho > 0 && Wo > 0, "%s: empty output", who);
    const long M = (long)d->B * Ho * Wo;
    PBE_REQUIRE(M < (1L << 31), "%s: too many output pixels", who);
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
    p.cb = d->kblock > 0 ? d->kblock : 32;
    PBE_REQUIRE(p.cb % 32 == 0 && d->C1 % p.cb == 0 && d->C2 % p.cb == 0, "%s: kblock=%d must be a multiple of 32 dividing C1=%d and C2=%d", who, p.cb, d->C1, d->C2);
    p.ws = (float*)d->workspace;
    return PBE_OK;
}

extern "C" int pbe_conv3x3_f16(const pbe_conv3x3_desc* d, pbe_stream_t stream) {
    IGemmP p;
    const int rc = fill_conv(d, p, "pbe_conv3x3_f16");
    if (rc != PBE_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    dispatch_igemm<1>(p, 1, s, d->workspace ? d->workspace_bytes : 0, d->tile_cfg);
    PBE_LAUNCH_CHECK("pbe_conv3x3_f16");
    return PBE_OK;
}

extern "C" int pbe_conv3x3_plan(const pbe_conv3x3_desc* d, int32_t* out6, size_t* workspace_needed) {
    PBE_REQUIRE(out6 && workspace_needed, "pbe_conv3x3_plan: null output");
    IGemmP p;
    const int rc = fill_conv(d, p, "pbe_conv3x3_plan");
    if (rc != PBE_OK) return rc;
    report_plan(p, 1, d->workspace ? d->workspace_bytes : 0, d->tile_cfg, out6);
    *workspace_needed = out6[1] > 1 ? (size_t)out6[1] * p.M * p.N * sizeof(float) : 0;
    return PBE_OK;
}
