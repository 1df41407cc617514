// Implicit-GEMM on the gfx950 matrix cores: one kernel body, three activation loaders.
//
//   MODE 0 (dense)   C[m,n] = sum_k A[m,k] W[n,k]          Linear / 1x1 conv / bmm
//   MODE 1 (conv3x3) m = (b,oy,ox), k = (ci/cb, tap, ci%cb) NHWC 3x3 conv gather, zero padding, cb % 64 == 0,
//                                                          optional fused nearest-2x upsample and
//                                                          two-source channel concat
//   MODE 2 (conv3x3) the same K order with the activation HALO resident in LDS (stride 1, pad 1, image width 8 .. 128):
//                    only the weight tile streams per k-tile; the two large tiles run a ping-pong main loop (see there)
//
// Structure (CDNA4, wave64):
//   * workgroup = NWM x NWN waves, block tile BM x BN x 64, each wave owns (BM/NWM) x (BN/NWN) as
//     16x16 tiles of v_mfma_f32_16x16x32_f16 (two k-steps per k-tile).  The weights are the MFMA "A" operand and the
//     activations the "B" operand, so an accumulator register quad holds 4 consecutive n for one m.
//   * tiles stream global -> LDS with global_load_lds_dwordx4 (LDS-DMA, no staging registers) into an S-slot ring
//     (S = 2 .. 4 by tile: S - 1 k-tiles in flight across ONE raw s_barrier per k-tile, retired with counted s_waitcnt vmcnt).
//     A k-tile row is 64 halfs = 128 B = one whole cache line: a DMA instruction fetches 8 complete lines.  (Round 1 staged
//     32 halfs per row; the half-line pieces capped the per-CU fill at 37 GB/s and the MFMA pipe sat idle for 55 % of the
//     main loop - tools/phase_stamps.py, profiles/r02_phase_stamps_bk32.txt.)
//   * LDS image: rows of 128 B = 8 chunks of 16 B.  An LDS-DMA instruction writes 64 lanes x 16 B linearly, so the bank
//     swizzle is applied to the per-lane SOURCE chunk (position p of row r holds global chunk p ^ (r & 7)) and again on
//     the fragment read: every ds_read_b128 of a 16x16x32 fragment is bank-conflict free in both k-steps.
//   * a DMA piece is LEAN (see the loader state): rows beyond M / N are clamped to the last row, the source is a pointer register
//     plus a scalar k offset, the LDS address an SGPR; padding pixels of the convs, a ragged last k-tile and a k-tile that
//     straddles the two concatenated sources read a 16-byte zero block through a per-lane select (LDS-DMA cannot mask a lane).
//   * workgroup ids are remapped so each XCD owns a contiguous run of tiles; the run walks n fastest or m fastest, whichever
//     fetches fewer bytes into the eight L2s for the launch's operand sizes (launch_cfg).
//   * small-M / deep-K problems are split along K over gridDim.z with a deterministic fp32 slab reduction.
//   * epilogue: alpha, bias, row-broadcast vector, activation in fp32 registers -> fp16 C tile in
//     LDS (one wave-row group at a time) -> whole 16-byte row segments to HBM, residual fused.
//   * what bounds the big conv tiles on real data is the power limit: cycles removed from the schedule come back
//     as a lower clock.
#pragma once
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
    int H, Wd, C1, C2, Ho, Wo, cstride, pad, ups, cb;   // cb = channel block of the K order (multiple of 64)
    int phase;                                           // MODE 1: nearest-2x upsample + 3x3 conv as FOUR 2x2 convs on the source grid (see the tap table); blockIdx.y = output phase
    int th;                                              // MODE 2: image rows per tile (tile = th full rows, or whole images)
    // fp8 (OCP e4m3) operands: rows are bytes (the loader sees them as K/2 halfs); C = acc * sa[m] * sw[n] (* alpha) + ...
    const float* sa; const float* sw; long ssa, ssw;
    // split-K: gridDim.z slices of the k-tile range, fp32 partial slabs [splits][M][N]
    int splits; float* ws;
    // extended epilogue of the dense tiles (EX instantiations, the transformer block's GEMM chain; ldm/modules/attention.py:198-252):
    int alpha_cols;                  // > 0: alpha multiplies columns n < alpha_cols only (q of a fused q | k | v launch, pre-scaled for the attention kernel)
    const float* ln_stat;            // LayerNorm folded into THIS GEMM: A holds the raw rows x, W = W * gamma, bias = W beta (+ bias), and
    int ln_parts; long ln_ld;        //   the epilogue forms rstd[m] * (acc - mean[m] * colsum[n]); (sum, sumsq) of row m = sum over ln_parts float2 partials
    const float* ln_c1; float ln_eps; double ln_inv_k;
    long rstat_ld;
    float* rstat;                    // this launch's OUTPUT rows feed a LayerNorm: per (column tile, row) partial (sum, sumsq) of the stored fp16 values
    h16* vt; int vt_col0, vt_tok;    // columns n >= vt_col0 are stored TRANSPOSED: vt[b * vt_bs + (n - vt_col0) * vt_rs + tok], m = b * vt_tok + tok
    long vt_bs, vt_rs;               //   (V^T for the attention kernel from the same launch as q | k)
    // conv output feeds a GroupNorm: per (sample, row block of the tile grid, group) partial (sum, sumsq) of the STORED fp16 values (after the
    // residual add), in pbe_groupnorm_f16's partial layout [B][blocks][groups][2] - the norm's statistics pass is then not launched
    float* gstat; int gs_cg, gs_groups, gs_hw;      // channels per group, groups of the whole tensor, output pixels per sample
    int* gs_report;                                 // HOST pointer (launch_cfg): row blocks per sample the launch writes partials for, 0 = this tile cannot
    int sv_ok;      // bias + row vector of a tile come from LDS (set per tile shape in launch_cfg)
    int m_fast;     // an XCD's run of tiles walks m fastest (one weight panel, many activation rows) instead of n fastest (launch_cfg)
#ifdef PBE_STAMPS
    unsigned long long* stamps;     // diagnostic build only (tools/phase_stamps.py): 16 words per workgroup
#endif
};

// Diagnostic build (-DPBE_STAMPS, never the shipped library): wave 0 of every workgroup records s_memtime at its phase
// boundaries into a buffer of its own; no output value depends on a stamp (cdna_hip_programming.md section 7, in-kernel stamps).
#ifdef PBE_STAMPS
#define PBE_STAMP(i)                                                                                                   \
    do {                                                                                                               \
        if (p.stamps && threadIdx.x == 0) {                                                                            \
            const long wg_ = ((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;                     \
            p.stamps[wg_ * 16 + (i)] = (i) >= 7 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime();    \
        }                                                                                                              \
    } while (0)
// main-loop accounting of wave 0: cycles inside the counted vmcnt waits (words 9), the barrier after the fragment reads (10) and
// the barrier that ends a k-tile (11), DMA issue + fragment reads up to lgkmcnt(0) (12), the MFMA block (13)
#define PBE_ACC_DECL unsigned long long acc_w_ = 0, acc_b1_ = 0, acc_b2_ = 0, acc_r_ = 0, acc_m_ = 0, acc_t_ = 0
#define PBE_ACC_T0() acc_t_ = __builtin_amdgcn_s_memtime()
#define PBE_ACC(v) v += __builtin_amdgcn_s_memtime() - acc_t_
#define PBE_ACC_STORE()                                                                                                \
    do {                                                                                                               \
        if (p.stamps && threadIdx.x == 0) {                                                                            \
            const long wg_ = ((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;                     \
            p.stamps[wg_ * 16 + 9] = acc_w_; p.stamps[wg_ * 16 + 10] = acc_b1_; p.stamps[wg_ * 16 + 11] = acc_b2_;    \
            p.stamps[wg_ * 16 + 12] = acc_r_; p.stamps[wg_ * 16 + 13] = acc_m_;                                       \
        }                                                                                                              \
    } while (0)
#else
#define PBE_STAMP(i) do { } while (0)
#define PBE_ACC_DECL do { } while (0)
#define PBE_ACC_T0() do { } while (0)
#define PBE_ACC(v) do { } while (0)
#define PBE_ACC_STORE() do { } while (0)
#endif

// Priority of the MFMA block.  PBE_PRIO_MODE: 0 = s_setprio 1 around every MFMA block, 1 = none (shipped: same-device A/B over the
// whole pipeline, conv class 193.2 -> 191.9 ms, GEMM class unchanged), 2 = static: waves 4-7 of an 8-wave tile at priority 1 for
// the whole main loop (MI355X_MICROARCH.md, Two waves per SIMD, item 4: 192.1 ms)
#ifndef PBE_PRIO_MODE
#define PBE_PRIO_MODE 1
#endif
#if PBE_PRIO_MODE == 0
#define PBE_SETPRIO(x) __builtin_amdgcn_s_setprio(x)
#else
#define PBE_SETPRIO(x) do { } while (0)
#endif

static __device__ __attribute__((aligned(16))) unsigned int g_pbe_zero16[4] = {0u, 0u, 0u, 0u};   // (one copy per translation unit)

#define EX_LN 1
#define EX_ST 2
#define EX_VT 4
#define PBE_GLDS16(gsrc, ldst)                                                                     \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc),         \
                                     (__attribute__((address_space(3))) void*)(ldst), 16, 0, 0)

// a / b for 0 <= a < 2^20, 1 <= b < 2^20, rb = 1.0f / b: the float quotient is off by at most one, two fix-ups make it exact
// (9 instructions against ~40 of the generic 32-bit division: the halo tile's setup does ~50 of them per thread)
__device__ __forceinline__ int small_div(int a, int b, float rb) {
    int q = (int)((float)a * rb);
    const int r = a - q * b;
    q += (r >= b) ? 1 : 0;
    q -= (r < 0) ? 1 : 0;
    return q;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

typedef long i64;
// one 16-byte fragment = two 8-byte fp8 operands: both MFMAs take the SAME byte positions from the two matrices, so together
// they contract the 64 k-values of a (4 lane-quads x 16 bytes) k-step whatever order the hardware assigns inside an operand
__device__ __forceinline__ f32x4 mfma_pair_f8(const h16x8& w, const h16x8& a, f32x4 acc) {
    typedef i64 i64x2 __attribute__((ext_vector_type(2)));
    const i64x2 wv = __builtin_bit_cast(i64x2, w), av = __builtin_bit_cast(i64x2, a);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(wv[0], av[0], acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(wv[1], av[1], acc, 0, 0, 0);
}

// (the 4-wave tiles live two workgroups per CU - two waves per SIMD, 256 registers each.  The V^T form of the 128x160 tile would take 268
//  and halve the occupancy: its bound is declared; the LayerNorm-only (252) and statistics-only (228) forms fit unconstrained and
//  keep their natural allocation - capped to 192 registers the GEGLU projection ran 11 % slower)
template <int BM, int BN, int NWM, int NWN, int MODE, int S, int HPA = 0, bool PP = false, bool F8 = false, int EX = 0>
#ifndef PBE_EX_MINWAVES
#define PBE_EX_MINWAVES 2
#endif
__global__ void __launch_bounds__(NWM* NWN * 64, ((EX & 4) != 0 && NWM * NWN == 4 && S == 2) ? PBE_EX_MINWAVES : 1) igemm_kernel(const IGemmP p, int tiles_n) {
    // EX (MODE 0, fp16 operands, one-pass epilogue tiles): the extended epilogue, a bit mask of what THIS instantiation carries -
    // EX_LN LayerNorm folded in, EX_ST row statistics out, EX_VT column-range alpha + transposed V^T tiles (see IGemmP).  Separate
    // instantiations per combination in use (GEGLU: EX_LN; proj_in / to_out: EX_ST; q|k|v^T: EX_LN | EX_VT): a kernel's registers are
    // the maximum over its paths, and the all-in-one form ran the GEGLU projection 11 % slower than the plain tile (tools/gemm_ex_ab.py).
    constexpr bool EXL = (EX & 1) != 0, EXS = (EX & 2) != 0, EXV = (EX & 4) != 0;
    // F8 (MODE 0 only): A and W hold OCP e4m3 bytes, a k-tile row of 128 B is 128 k-values; v_mfma_f32_16x16x32_fp8_fp8 runs at
    // the fp16 MFMA's rate, the gain is half the bytes through the fill path that bounds these GEMMs.  The epilogue multiplies
    // by the per-row scale of A and the per-row scale of W (per output column) before bias / activation / residual.
    // MODE 2 = 3x3 conv (stride 1, pad 1) with the activation HALO resident in LDS: a tile is BM pixels = whole image rows (or whole
    // images); for every 64-channel block its (rows + 2) x (width + 2) halo (HPA rows of 128 B, zero outside the image) is staged
    // ONCE and all 9 taps read their shifted windows from it, so only the weight tile streams per k-tile.  MODE 1 re-stages the
    // activation tile for every tap: with two 128x160 workgroups per CU that is 74 KB of LDS-DMA per 1 280 MFMA pipe cycles -
    // more than a CU's fill path delivers (phase stamps: 3 190 cycles per k-tile inside the sampler).  Here a 256x160 tile moves
    // 20.5 KB + 5.7 KB for the same MFMA work.  S = weight ring depth.
    // S = LDS ring depth (S - 1 k-tiles of 64 in flight).  Sized per tile so the ring fills the LDS one workgroup (8-wave tiles)
    // or two to three workgroups (4-wave tiles) can own on a CU.
    // (A "ping-pong" form of the main loop for the 8-wave tiles - wave groups 0-3 / 4-7 offset by one barrier, [fragment reads +
    //  DMA issue | MFMAs] with two barriers per k-tile - was built and measured on every conv / GEMM shape of the path: bit-identical
    //  output, 7 % slower on the conv class (780.9 -> 724.0 TFLOP/s), 2 % slower on the GEMM class.  profiles/r02_ab_pingpong_rejected.txt.)
    constexpr int NW = NWM * NWN, NT = NW * 64;
    constexpr int D = S - 1;
    constexpr int WM = BM / NWM, WN = BN / NWN, TM = WM / 16, TN = WN / 16;
    constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, STAGE = A_BYTES + W_BYTES;
    constexpr int PA = BM / 8, PW = BN / 8;           // 8-row x 128-byte DMA pieces (1 KiB = one wave instruction)
    constexpr int LA = PA / NW, LW = (PW + NW - 1) / NW, LPT = LA + LW;
    constexpr int CLD = BN + 8;
    // MODE 2 LDS: [halo image 0][halo image 1][weight ring S x BN rows]
    constexpr int RING = MODE == 2 ? 2 * HPA * 128 + S * W_BYTES : S * STAGE;
    static_assert(S >= 2 && S <= 4 && PA % NW == 0 && BM % 16 == 0 && BN % 16 == 0, "A pieces must divide over the waves (weight pieces may be padded)");
    static_assert(MODE != 2 || (HPA % 8 == 0 && HPA >= BM), "halo image must hold the tile");
    static_assert(EX == 0 || (MODE == 0 && !F8), "extended epilogue: dense fp16 tiles only");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: every LDS-DMA destination below is an SGPR expression
    const int wm = wave % NWM, wn = wave / NWM;
    PBE_ACC_DECL;
#if PBE_PRIO_MODE == 2
    if (NW == 8 && __builtin_amdgcn_readfirstlane(wave) >= 4) __builtin_amdgcn_s_setprio(1);
#endif
    PBE_STAMP(0);                                    // workgroup start
    PBE_STAMP(7);                                    // wall clock (100 MHz) of the start
    int tile;
    {   // XCD-aware tile order (bijective for any grid size)
        const int nwg = gridDim.x, id = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = id & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
    }
    // Each XCD's L2 fetches what ITS tiles touch.  n fastest: an XCD's run covers few row blocks of A and every column block of
    // W - right where A is the big operand (64x64 maps).  m fastest: few column blocks of W and every row block of A - right where
    // W is (16x16 / 8x8 maps: 29 MB of weights against 5 MB of activations; n fastest made all 8 L2s fetch all 29).
    const int tiles_m = gridDim.x / tiles_n;
    const int tn_i = p.m_fast ? tile / tiles_m : tile % tiles_n, tm_i = p.m_fast ? tile - tn_i * tiles_m : tile / tiles_n;
    const int m0 = tm_i * BM, n0 = tn_i * BN;
    const long bz = blockIdx.y;

    // ---- loader state: this lane's row inside an 8-row DMA piece and its source chunk ----
    // LDS rows are 128 B (64 halfs = 8 chunks of 16 B) = one whole cache line per row: a wave's DMA instruction fetches 8
    // complete lines (64-byte row pieces - the BK = 32 layout of round 1 - filled at 37 GB/s per CU in the real kernel,
    // tools/phase_stamps.py: 1 440 cycles per k-tile of a 128x320 tile against 640 cycles of MFMA work).
    // LDS-DMA writes lane l at base + 16 l, so the bank swizzle goes on the SOURCE chunk: position p of row r holds global
    // chunk p ^ (r & 7); the fragment read applies the same XOR (conflict-free ds_read_b128 for both k-steps, checked by
    // enumeration over the hardware's 16-lane service groups).
    const int lrow = lane >> 3;
    const int gch = (lane & 7) ^ lrow;
    const h16* zsrc = reinterpret_cast<const h16*>(g_pbe_zero16);

    // A DMA piece must stay LEAN: tools/ubench_loop.hip (this loop's instruction mix, operands from L2) runs a k-tile of the
    // 256x160 tile in 1 550-1 630 cycles with "pointer + scalar k offset" pieces and in 1 850-2 500 with the form this kernel used to
    // compile to (per-lane zero-block select, LDS address through a VGPR, a branch per piece): next to MFMAs every VALU instruction
    // of a piece waits for an issue slot.  So: rows beyond M / N are CLAMPED to the last row (their products land in accumulators
    // that are never stored), the source pointer of a piece is one 64-bit add, its LDS address an SGPR; only a k-tile that needs
    // per-lane zeros (K % 64 != 0: the last one) or straddles the two concatenated sources takes the select form.
    const h16* a_cur[LA];                             // source row of each piece at k = 0, + this lane's chunk (current concat source)
    const h16* a_alt[LA];                             // the same for the second source, pre-offset by -K1
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int m = min(m0 + (wave * LA + i) * 8 + lrow, p.M - 1);
        a_cur[i] = p.A + bz * p.sA + (long)m * p.lda + gch * 8;
        a_alt[i] = p.A2 ? p.A2 + (long)m * p.lda2 + gch * 8 - p.K1 : a_cur[i];
    }
    // conv: K is ordered (channel block cb, tap, channel) so the 9 taps of a pixel's cb channels are consecutive k-tiles -
    // the shifted re-reads hit L2 instead of going back to the Infinity Cache / HBM (measured: tap-major order re-fetched the
    // input 9x beyond L2).  The tap -> input-pixel map of this tile's BM rows is built once into LDS:
    // tab[tap][row] = pixel index, or -1 outside the (virtual) image.
    int* tab = reinterpret_cast<int*>(smem + RING);
    // Epilogue vectors of this tile, staged ONCE (their global latency hides under the first DMA tile):
    // svec[s][c] = bias[n0 + c] + rowvec[first sample of the tile + s][n0 + c], up to 4 samples per tile.
    float* svec = reinterpret_cast<float*>(smem + (RING > WM * CLD * 2 ? RING : WM * CLD * 2) + (MODE == 1 ? 9 * BM * 4 : 0));
    const int sv_ns = (p.rowvec && p.group_rows < BM) ? BM / p.group_rows : 1;       // samples per tile (tile is sample-aligned when sv_ok)
    if (MODE == 1) {
        const int hw = p.Ho * p.Wo, Hv = p.H << p.ups, Wv = p.Wd << p.ups;
        // Phase form (p.phase): a 3x3 conv over a nearest-2x upsampled image reads, for the output pixel (2y + py, 2x + px), only a 2x2
        // block of SOURCE pixels - rows {y - 1 + py, y + py}, columns {x - 1 + px, x + px} - because the 9 taps fall on 4 distinct
        // sources (py = 0: taps ky = 0 | 1, 2 -> rows y - 1 | y; py = 1: ky = 0, 1 | 2 -> y | y + 1).  The host sums the weights that
        // share a source (4 weight sets of 4 taps): 4 / 9 of the MACs, rows = source pixels, blockIdx.y = phase, and the epilogue
        // scatters row m = (b, y, x) to output pixel (2y + py, 2x + px).  Zero padding agrees: source row -1 <=> upsampled row -1 / -2.
        const int py = p.phase ? (int)(blockIdx.y >> 1) : 0, px = p.phase ? (int)(blockIdx.y & 1) : 0;
        for (int row = tid; row < BM; row += NT) {            // one thread per tile row: one (b, oy, ox) decode, 9 (or 4) taps
            const int m = m0 + row;
            const bool rok = m < p.M;
            const int b = m / hw, rem = m - b * hw;
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            const int iy0 = oy * p.cstride - p.pad, ix0 = ox * p.cstride - p.pad;
            if (p.phase) {
#pragma unroll
                for (int tp = 0; tp < 4; ++tp) {
                    const int iy = oy - 1 + py + (tp >> 1), ix = ox - 1 + px + (tp & 1);
                    const bool ok = rok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.Wd;
                    tab[tp * BM + row] = ok ? (b * p.H + iy) * p.Wd + ix : -1;
                }
                continue;
            }
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                const int iy = iy0 + tp / 3, ix = ix0 + tp % 3;
                const bool ok = rok && (unsigned)iy < (unsigned)Hv && (unsigned)ix < (unsigned)Wv;
                tab[tp * BM + row] = ok ? (b * p.H + (iy >> p.ups)) * p.Wd + (ix >> p.ups) : -1;
            }
        }
        __syncthreads();
    }
    const h16* w_row[LW];
#pragma unroll
    for (int i = 0; i < LW; ++i) {
        // (a padding piece - PW not a multiple of NW, halo tiles only - repeats the last real piece: same source, same LDS bytes)
        const int n = min(n0 + min(wave + NW * i, PW - 1) * 8 + lrow, p.N - 1);
        w_row[i] = p.W + bz * p.sW + (long)n * p.ldw + gch * 8;
    }
    const int nk_all = (p.K + 63) >> 6;
    int kt0 = 0, nk = nk_all;                        // this workgroup's k-tile range [kt0, nk)
    if (p.splits > 1) {
        const int per = (nk_all + p.splits - 1) / p.splits;
        kt0 = blockIdx.z * per;
        nk = min(nk_all, kt0 + per);
    }
    // conv K order: (channel block of cb, tap, channel): state of the NEXT k-tile to issue
    const int KB = MODE == 1 ? p.cb >> 6 : 1;        // k-tiles per (block, tap) visit
    const int ntap = (MODE == 1 && p.phase) ? 4 : 9;
    int tap = 0, c0 = 0, kj = 0;
    if (MODE == 1 && kt0 > 0) {
        const int per_blk = ntap * KB, cblk = kt0 / per_blk, r = kt0 - cblk * per_blk;
        tap = r / KB; kj = r - tap * KB; c0 = cblk * p.cb + kj * 64;
    }
    // MODE 1: per piece the running source pointer and its advance per k-tile (a padding pixel keeps reading the zero block: advance 0)
    const h16* a_src[LA];
    int a_inc[LA];
    bool fresh = true;
#pragma unroll
    for (int i = 0; i < LA; ++i) { a_src[i] = zsrc; a_inc[i] = 0; }
    // MODE 0: the k-tile where the second concat source takes over, and the two rare k-tiles that need per-lane selects
    const bool ktail = (p.K & 63) != 0, straddle = p.A2 && (p.K1 & 63);
    const int kt_sw = p.A2 ? p.K1 >> 6 : 0x7fffffff;
    if (MODE == 0 && kt0 > kt_sw) {
#pragma unroll
        for (int i = 0; i < LA; ++i) a_cur[i] = a_alt[i];
    }

    auto issue = [&](int kt, int slot) {              // k-tiles are issued in order: kt0, kt0 + 1, ...
        unsigned char* sa = smem + slot * STAGE;
        unsigned char* sw = sa + A_BYTES;
        const long k = (long)kt * 64;                 // scalar
        const bool slow = (ktail && kt == nk_all - 1) || (straddle && kt == kt_sw);      // uniform
        const int kl = kt * 64 + gch * 8;             // this lane's k (slow form only)
        if (MODE == 0) {
            if (kt == kt_sw && !straddle) {
#pragma unroll
                for (int i = 0; i < LA; ++i) a_cur[i] = a_alt[i];
            }
            if (!slow) {
#pragma unroll
                for (int i = 0; i < LA; ++i) PBE_GLDS16(a_cur[i] + k, sa + (wave * LA + i) * 1024);
            } else {
#pragma unroll
                for (int i = 0; i < LA; ++i) {
                    const h16* src = (kl < p.K1 ? a_cur[i] : a_alt[i]) + k;
                    PBE_GLDS16(kl < p.K ? src : zsrc, sa + (wave * LA + i) * 1024);
                }
                if (straddle && kt == kt_sw) {
#pragma unroll
                    for (int i = 0; i < LA; ++i) a_cur[i] = a_alt[i];
                }
            }
        } else {
            if (kj == 0 || fresh) {                   // new (block, tap): look the pixels up; otherwise +64 channels
                fresh = false;
                const bool first = c0 < p.C1;         // which concat source this channel block lives in (uniform)
                const h16* base = first ? p.A + c0 + gch * 8 : p.A2 + (c0 - p.C1) + gch * 8;
                const long cs = first ? p.C1 : p.C2;
#pragma unroll
                for (int i = 0; i < LA; ++i) {
                    const int pix = tab[tap * BM + (wave * LA + i) * 8 + lrow];
                    a_src[i] = pix >= 0 ? base + (long)pix * cs : zsrc;
                    a_inc[i] = pix >= 0 ? 64 : 0;
                }
            }
#pragma unroll
            for (int i = 0; i < LA; ++i) {
                PBE_GLDS16(a_src[i], sa + (wave * LA + i) * 1024);
                a_src[i] += a_inc[i];
            }
            c0 += 64;
            if (++kj == KB) {
                kj = 0;
                if (++tap == ntap) tap = 0; else c0 -= p.cb;   // next tap of the same block, or first tap of the next block
            }
        }
        if (!slow) {
#pragma unroll
            for (int i = 0; i < LW; ++i) PBE_GLDS16(w_row[i] + k, sw + min(wave + NW * i, PW - 1) * 1024);      // (a padding piece repeats the last real one)
        } else {
#pragma unroll
            for (int i = 0; i < LW; ++i) PBE_GLDS16(kl < p.K ? w_row[i] + k : zsrc, sw + min(wave + NW * i, PW - 1) * 1024);
        }
    };

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    const int rsw = (fq ^ (fr & 7)) << 4;             // byte offset of k-step 0's chunk; k-step 1 is rsw ^ 64
    const int a_rd = (wm * WM + fr) * 128, w_rd = (wn * WN + fr) * 128;

    float* sca = svec + 4 * BN;                       // F8: per-row scale of A for this tile's rows, then per-column scale (W's rows)
    float* scw = sca + BM;                            // EX + LayerNorm fold: sca[row] = rstd, scw[row] = -mean * rstd, lnc[col] = colsum of W * gamma
    float* lnc = scw + BM;
    auto stage_svec = [&]() {
        if constexpr (F8) {
            for (int idx = tid; idx < BM + BN; idx += NT) {
                float v = 1.f;
                if (idx < BM) { if (p.sa && m0 + idx < p.M) v = p.sa[bz * p.ssa + m0 + idx]; sca[idx] = v; }
                else { const int c = idx - BM; if (p.sw && n0 + c < p.N) v = p.sw[bz * p.ssw + n0 + c]; scw[c] = v; }
            }
        }
        if constexpr (EXL) {
            if (p.ln_stat) {
                for (int idx = tid; idx < BM + BN; idx += NT) {
                    if (idx < BM) {
                        const int m = min(m0 + idx, p.M - 1);
                        float a = 0.f, q = 0.f;
                        for (int z = 0; z < p.ln_parts; ++z) {           // fixed order: deterministic
                            const float2 t = *reinterpret_cast<const float2*>(p.ln_stat + 2 * ((long)z * p.ln_ld + m));
                            a += t.x; q += t.y;
                        }
                        // (only the cancelling subtraction runs in fp64 - three full-rate operations; an fp64 divide + square root per row cost
                        //  the workgroup ~1 500 cycles of a 8 000-cycle K = 320 tile)
                        const double mean = (double)a * p.ln_inv_k;
                        const float var = (float)((double)q * p.ln_inv_k - mean * mean);
                        const float rstd = __builtin_amdgcn_rsqf(fmaxf(var, 0.f) + p.ln_eps);
                        sca[idx] = rstd; scw[idx] = -(float)mean * rstd;
                    } else {
                        const int c = idx - BM;
                        lnc[c] = n0 + c < p.N ? p.ln_c1[n0 + c] : 0.f;
                    }
                }
            }
        }
        if (p.sv_ok && p.splits <= 1) {                  // staged AFTER the first DMAs are in flight: both latencies overlap
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
    };
    // Tiles with few accumulators fetch BOTH k-steps' fragments before the first MFMA (the second set's LDS latency hides
    // under the first set's MFMAs); the 256-row tiles have no registers for that and read k-step 1 after issuing k-step 0.
    constexpr bool BOTH = TM * TN * 4 + 2 * (TM + TN) * 4 <= 176;

    if constexpr (MODE == 2) {
        constexpr int PAH = HPA / 8, LAH = (PAH + NW - 1) / NW;
        unsigned char* abuf = smem;
        unsigned char* wring = smem + 2 * HPA * 128;
        // Halo image of one (sub-)image of the tile: TH + 2 rows of TW + 1 pixels + 1.  A tile spans the image's whole width, so
        // x = -1 and x = TW are always padding: the zero row right of image row y IS the zero row left of image row y + 1
        // (row stride TW + 1 instead of TW + 2; 4 rows of 64 pixels: 391 halo rows instead of 396 - what lets a third weight slot
        // fit beside two halo images of a 256-pixel tile).
        const int TW = p.Wd, TH = p.th, HW2 = TW + 1, HPS = (TH + 2) * HW2 + 1;
        const int img_px = TH * TW, nsub = BM / img_px;                             // nsub > 1: the tile holds nsub whole images
        const int tiles_per_img = p.H / TH;
        const int b0 = nsub > 1 ? tm_i * nsub : tm_i / tiles_per_img;
        const int y0 = nsub > 1 ? 0 : (tm_i - b0 * tiles_per_img) * TH;
        const float r_hps = 1.0f / (float)HPS, r_hw2 = 1.0f / (float)HW2, r_img = 1.0f / (float)img_px, r_tw = 1.0f / (float)TW;
        int hpix[LAH];                                // source pixel of this lane's row in each of its halo pieces (-1: zero)
#pragma unroll
        for (int i = 0; i < LAH; ++i) {
            const int piece = min(wave + NW * i, PAH - 1), hp = piece * 8 + lrow;      // (padding pieces repeat the last real one)
            int pix = -1;
            if (hp < nsub * HPS) {
                const int sub = small_div(hp, HPS, r_hps), r = hp - sub * HPS, hy = small_div(r, HW2, r_hw2), hx = r - hy * HW2;
                const int y = y0 + hy - 1, x = hx - 1;
                if ((unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)TW) pix = ((b0 + sub) * p.H + y) * TW + x;
            }
            hpix[i] = pix;
        }
        int hc[TM];                                   // halo row of this lane's pixel in each of its 16-pixel groups (tap (1,1))
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int ml = wm * WM + j * 16 + fr, sub = small_div(ml, img_px, r_img), rr = ml - sub * img_px, ty = small_div(rr, TW, r_tw), tx = rr - ty * TW;
            hc[j] = sub * HPS + (ty + 1) * HW2 + tx + 1;
        }
        // running source pointer of each halo piece (channel block by channel block, +64 channels; a padding row keeps reading the
        // zero block) - rebuilt only where the channel blocks cross from the first concat source into the second
        const h16* hptr[LAH];
        auto halo_base = [&](int blk) {
            const int c0b = blk * 64;
            const bool first = c0b < p.C1;
            const h16* base = (first ? p.A + c0b : p.A2 + (c0b - p.C1)) + gch * 8;
            const long cs = first ? p.C1 : p.C2;
#pragma unroll
            for (int i = 0; i < LAH; ++i) {
                hptr[i] = hpix[i] >= 0 ? base + (long)hpix[i] * cs : zsrc;
            }
        };
        auto issue_a1 = [&](int buf_off, int i) {         // piece i (compile-time) of the next block in line, into the halo buffer at buf_off
            PBE_GLDS16(hptr[i], abuf + buf_off + min(wave + NW * i, PAH - 1) * 1024);
            hptr[i] += hpix[i] >= 0 ? 64 : 0;
        };
        auto issue_a = [&](int blk, int buf) {
            if (p.C2 && blk * 64 == p.C1) halo_base(blk);
#pragma unroll
            for (int i = 0; i < LAH; ++i) issue_a1(buf * (HPA * 128), i);
        };
        auto issue_w1 = [&](int kt, int slot, int i) {
            PBE_GLDS16(w_row[i] + (long)kt * 64, wring + slot * W_BYTES + min(wave + NW * i, PW - 1) * 1024);
        };
        auto issue_w = [&](int kt, int slot) {
#pragma unroll
            for (int i = 0; i < LW; ++i) issue_w1(kt, slot, i);
        };
        const int nblk_all = (p.C1 + p.C2) >> 6;
        int blk0 = 0, blk1 = nblk_all;                // this workgroup's channel blocks (split-K at block granularity)
        if (p.splits > 1) {
            const int per = (nblk_all + p.splits - 1) / p.splits;
            blk0 = blockIdx.z * per;
            blk1 = min(nblk_all, blk0 + per);
        }
        const int nk2 = blk1 * 9;
        PBE_STAMP(1);
        if (blk0 < blk1) {
            halo_base(blk0);
            issue_a(blk0, 0);
#pragma unroll
            for (int t = 0; t < (PP ? 2 : D); ++t) issue_w(blk0 * 9 + t, t);
        }
        PBE_STAMP(2);
        stage_svec();
        if constexpr (PP) {
            // Ping-pong form (8 waves, two per SIMD; weight ring of 3).  Waves 0-3 and 4-7 alternate: in every half period one group
            // reads its 18 fragments of a k-tile while the other group's MFMAs own the matrix pipe.
            //   period P (tile P), barriers b(2P-1) | b(2P) | b(2P+1):
            //     first half:   group 0 reads tile P's fragments        | group 1 runs the MFMAs of tile P-1
            //     second half:  group 0 runs the MFMAs of tile P        | group 1 reads tile P's fragments
            //   The DMA pieces of period P - W(P + 2) and a share of the next block's halo - go out right behind the fragment reads of
            //   a wave's read half.  The first form of this loop (weight ring of 2, all 7 halo pieces at tap 0, pieces with per-lane
            //   selects at the HEAD of the read half) spent 1 150 cycles in the read half against 680 of MFMAs
            //   (tools/phase_stamps.py); spreading the pieces between the MFMAs was worse still (tools/ubench_loop.hip).
            //   RAW: tile P+1 (issued in period P-1, retired by its issuers at the end of period P) and a halo retired in the second
            //   half of period 9b+8 have landed at b(2P+1), before their first read.  WAR: W(P+2) lands in tile P-1's slot and block
            //   b+1's halo in block b-1's buffer; their last reads (group 1, second half of period P-1 / 9b-1) precede b(2P-1).
            static_assert(NW == 8 && BOTH && S == 3 && LAH <= 8, "ping-pong: two waves per SIMD, both k-steps' fragments in registers, weight ring of 3");
            const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);
            h16x8 fa[2][TM], fw[2][TN];
            // The read half is the long pole (its partner group is computing: every VALU instruction here waits for an issue slot
            // next to the partner's MFMAs - 930 cycles with the fragment addresses computed in place against 680 of MFMAs), so it
            // holds NOTHING but the 18 ds_read_b128 and the DMA pieces: the LDS byte addresses of the next tile's fragments (tap
            // shift, swizzle, halo buffer, ring slot) and the bumped weight pointers are computed inside the wave's OWN MFMA block,
            // a few VALU instructions after every other MFMA, where the matrix pipe covers them.
            int aaddr[2][TM], waddr[2];
            auto tile_shift = [&](int tap) {
                const int trow = tap >= 6 ? 2 : (tap >= 3 ? 1 : 0);
                return (trow - 1) * HW2 + (tap - 3 * trow - 1);
            };
            auto frag_addrs = [&](int blk_n, int tap_n, int slot_n) {      // in one go (prologue only)
                const int shift = tile_shift(tap_n), abase = ((blk_n - blk0) & 1) * (HPA * 128), wb = 2 * HPA * 128 + slot_n * W_BYTES;
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    const int ar = hc[j] + shift;
                    aaddr[0][j] = abase + ar * 128 + ((fq ^ (ar & 7)) << 4);
                    aaddr[1][j] = aaddr[0][j] ^ 64;
                }
                waddr[0] = wb + w_rd + rsw;
                waddr[1] = wb + w_rd + (rsw ^ 64);
            };
            auto read_frags = [&]() {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                    for (int j = 0; j < TM; ++j) fa[ks][j] = *reinterpret_cast<const h16x8*>(smem + aaddr[ks][j]);
#pragma unroll
                    for (int i = 0; i < TN; ++i) fw[ks][i] = *reinterpret_cast<const h16x8*>(smem + waddr[ks] + i * 16 * 128);
                }
            };
            const int kt_first = blk0 * 9;
            // The two streams go out through DIFFERENT waves: vmcnt retires in order, so a halo piece (activations no other workgroup
            // shares: full HBM / Infinity Cache latency, ~2 500 cycles under load) ahead of a weight piece (L2, ~600) in ONE wave's
            // queue makes the weight tile as late as the halo - with both in every wave the end-of-period wait stalled for ~500
            // cycles per period.  Group 0 (waves 0-3) issues the weight tiles and retires them every period; group 1 (waves 4-7)
            // issues the next block's halo, two pieces per period from tap 0 on, and retires it once, at tap 8.
            constexpr int PWG = PW / 4, LAHG = (PAH + 3) / 4;
            static_assert(PW % 4 == 0 && LAHG <= 14, "weight pieces split over 4 waves; at most two halo pieces per period over taps 0 .. 6");
            const int gw = wave & 3;
            const h16* wp[PWG];                            // group 0: weight pointers of the NEXT tile to issue
            int hq[LAHG];                                  // group 1: source pixel of this lane's row in halo pieces gw, gw + 4, ...
#pragma unroll
            for (int i = 0; i < PWG; ++i)
                wp[i] = p.W + bz * p.sW + (long)min(n0 + (gw * PWG + i) * 8 + lrow, p.N - 1) * p.ldw + gch * 8 + (long)(kt_first + 2) * 64;
#pragma unroll
            for (int i = 0; i < LAHG; ++i) {
                const int hp = min(gw + 4 * i, PAH - 1) * 8 + lrow;
                int pix = -1;
                if (hp < nsub * HPS) {
                    const int sub = small_div(hp, HPS, r_hps), r = hp - sub * HPS, hy = small_div(r, HW2, r_hw2), hx = r - hy * HW2;
                    const int y = y0 + hy - 1, x = hx - 1;
                    if ((unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)TW) pix = ((b0 + sub) * p.H + y) * TW + x;
                }
                hq[i] = pix;
            }
            // One MFMA block = the wave's 2 TM TN MFMAs + (a few VALU instructions after every other one) the LDS byte addresses of
            // the fragments of the tile this wave reads NEXT and the bumped weight pointers.
            auto mfmas = [&](int shift_n, int abase_n, int wb_n) {
                static_assert(TM * TN >= 2 * TM + 1 + PWG, "not enough MFMAs to spread the address steps over");
                PBE_SETPRIO(1);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < TN; ++i)
#pragma unroll
                        for (int j = 0; j < TM; ++j) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[ks][i], fa[ks][j], acc[i][j], 0, 0, 0);
                            const int q = (ks * TN + i) * TM + j;             // compile-time after unrolling
                            if (q & 1) {
                                const int st = q >> 1;
                                __builtin_amdgcn_sched_barrier(0);       // (each step pinned between its MFMAs: left free, the unrolled taps' steps
                                if (st < 2 * TM) {                       //  are hoisted together and the block spills)
                                    const int jj = st >> 1;
                                    if ((st & 1) == 0) {
                                        const int ar = hc[jj] + shift_n;
                                        aaddr[0][jj] = abase_n + ar * 128 + ((fq ^ (ar & 7)) << 4);
                                    } else aaddr[1][jj] = aaddr[0][jj] ^ 64;
                                } else if (st == 2 * TM) { waddr[0] = wb_n + w_rd + rsw; waddr[1] = wb_n + w_rd + (rsw ^ 64); }
                                else if (st < 2 * TM + 1 + PWG) wp[st - 2 * TM - 1] += 64;
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                PBE_SETPRIO(0);
            };
            // One channel block = 9 periods, UNROLLED: the tap, the ring slot (9 % 3 == 0: a block starts at slot 0), which DMA pieces
            // a period issues and how many may stay in flight at its end are compile-time constants - the run-time form of this
            // loop spent ~25 scalar branches per period on them (and copied the halo pointer arrays around a switch).
            //   period (blk, tap), right behind the fragment reads: group 0 issues W(kt + 2) into tile kt - 1's slot (unless the block
            //   is the workgroup's last and tap >= 7) and retires W(kt + 1) at the period's end; group 1 issues halo pieces 2 tap and
            //   2 tap + 1 of block blk + 1 and retires the whole halo in its read half of tap 8, before the barrier that opens the
            //   next block.
            auto block = [&](auto LASTC, int blk) {
                constexpr bool LAST = decltype(LASTC)::value;
                const int abase = ((blk - blk0) & 1) * (HPA * 128), abase_o = (HPA * 128) - abase;
                const h16* hbase = zsrc;                   // source of halo row 0's chunk for block blk + 1
                long hcs = 0;
                if (!LAST) {
                    const int c0b = (blk + 1) * 64;
                    const bool first = c0b < p.C1;
                    hbase = (first ? p.A + c0b : p.A2 + (c0b - p.C1)) + gch * 8;
                    hcs = first ? p.C1 : p.C2;
                }
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
#ifdef PBE_STAMPS
                    if (tap == 1 && blk == blk0) PBE_STAMP(3);
#endif
                    constexpr int kPWG = PWG;
                    const bool w = !LAST || tap < 7;
                    PBE_ACC_T0();
                    __builtin_amdgcn_sched_barrier(0);
                    read_frags();
                    __builtin_amdgcn_sched_barrier(0);
                    if (grp == 0) {
                        if (w) {
#pragma unroll
                            for (int i = 0; i < PWG; ++i) PBE_GLDS16(wp[i], wring + ((tap + 2) % 3) * W_BYTES + (gw * PWG + i) * 1024);
                        }
                    } else if (!LAST) {
#pragma unroll
                        for (int i = 2 * tap; i < 2 * tap + 2; ++i)
                            if (i < LAHG) PBE_GLDS16(hq[i] >= 0 ? hbase + (long)hq[i] * hcs : zsrc, abuf + abase_o + min(gw + 4 * i, PAH - 1) * 1024);
                        if (tap == 8) wait_vmcnt<0>();     // the next block's halo has landed (this group has nothing else in flight)
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    PBE_ACC(acc_r_);
                    PBE_ACC_T0();
                    __builtin_amdgcn_s_barrier();
                    PBE_ACC(acc_b1_);
                    __builtin_amdgcn_sched_barrier(0);
                    PBE_ACC_T0();
                    {   // this wave reads tile kt + 1 next: tap + 1 of this block, or tap 0 of the next (other halo buffer)
                        const int tn = tap == 8 ? 0 : tap + 1, trow = tn / 3;
                        // (the row stride re-materialised opaquely per tap: as a loop invariant, all 9 taps' fragment addresses are
                        //  hoisted out of the block loop into 70 more registers and the kernel spills)
                        int hw2 = HW2;
                        asm volatile("" : "+s"(hw2));
                        mfmas((trow - 1) * hw2 + (tn - 3 * trow - 1), tap == 8 ? abase_o : abase, 2 * HPA * 128 + ((tap + 1) % 3) * W_BYTES);
                    }
                    PBE_ACC(acc_m_);
                    PBE_ACC_T0();
                    if (grp == 0) { if (w) wait_vmcnt<kPWG>(); else wait_vmcnt<0>(); }      // W(kt + 1) landed; this period's tile may fly
                    PBE_ACC(acc_w_);
                    __builtin_amdgcn_sched_barrier(0);
                    PBE_ACC_T0();
                    __builtin_amdgcn_s_barrier();
                    PBE_ACC(acc_b2_);
                }
            };
            if (blk0 < blk1) {
                frag_addrs(blk0, 0, 0);
                wait_vmcnt<0>();                            // halo of block blk0, W(kt_first) and W(kt_first + 1) landed: every queue starts empty
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (grp == 1) __builtin_amdgcn_s_barrier();  // the stagger: group 1 runs one barrier late
#pragma unroll 1
                for (int blk = blk0; blk + 1 < blk1; ++blk) block(std::false_type{}, blk);
                block(std::true_type{}, blk1 - 1);
                if (grp == 0) __builtin_amdgcn_s_barrier();  // group 1's last M phase
            }
        } else {
            int slot_rd = 0, slot_wr = D % S;
            for (int blk = blk0; blk < blk1; ++blk) {
                const unsigned char* ab = abuf + ((blk - blk0) & 1) * (HPA * 128);
                const bool next_a = blk + 1 < blk1;
#pragma unroll 1
                for (int tap = 0; tap < 9; ++tap) {          // (not unrolled: 9 copies of the body cost registers and 30 000 lines of ISA)
                    const int kt = blk * 9 + tap;
#ifdef PBE_STAMPS
                    if (kt == blk0 * 9 + 1) PBE_STAMP(3);
#endif
                    // W(kt) must have landed.  Younger DMAs that may stay in flight: the later W tiles (D - 1, fewer at the end) and,
                    // at taps 1 .. D, the halo pieces of block blk + 1 (issued at tap 0 right after W(kt0 + D)).  The halo of THIS
                    // block is older than W(kt) (in-order vmcnt), so it has landed too.
                    const int wleft = min(nk2 - 1 - kt, D - 1);
                    const bool a_young = next_a && tap >= 1 && tap <= D;
                    PBE_ACC_T0();
                    if (a_young) {
                        if (D >= 3 && wleft >= 2) wait_vmcnt<2 * LW + LAH>();
                        else if (D >= 2 && wleft >= 1) wait_vmcnt<LW + LAH>();
                        else wait_vmcnt<LAH>();
                    } else {
                        if (D >= 3 && wleft >= 2) wait_vmcnt<2 * LW>();
                        else if (D >= 2 && wleft >= 1) wait_vmcnt<LW>();
                        else wait_vmcnt<0>();
                    }
                    PBE_ACC(acc_w_);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    PBE_ACC_T0();
                    __builtin_amdgcn_s_barrier();
                    PBE_ACC(acc_b2_);
                    const unsigned char* sw = wring + slot_rd * W_BYTES;
                    const int trow = tap >= 6 ? 2 : (tap >= 3 ? 1 : 0);
                    const int shift = (trow - 1) * HW2 + (tap - 3 * trow - 1);
                    h16x8 fa[BOTH ? 2 : 1][TM], fw[BOTH ? 2 : 1][TN];
                    int arow[TM];
#pragma unroll
                    for (int j = 0; j < TM; ++j) arow[j] = hc[j] + shift;
#pragma unroll
                    for (int ks = 0; ks < (BOTH ? 2 : 1); ++ks) {
#pragma unroll
                        for (int j = 0; j < TM; ++j)
                            fa[ks][j] = *reinterpret_cast<const h16x8*>(ab + arow[j] * 128 + (((ks * 4 + fq) ^ (arow[j] & 7)) << 4));
#pragma unroll
                        for (int i = 0; i < TN; ++i) fw[ks][i] = *reinterpret_cast<const h16x8*>(sw + w_rd + (rsw ^ (ks * 64)) + i * 16 * 128);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (kt + D < nk2) issue_w(kt + D, slot_wr);
                    if (tap == 0 && next_a) issue_a(blk + 1, (blk + 1 - blk0) & 1);
                    __builtin_amdgcn_sched_barrier(0);
                    PBE_SETPRIO(1);
#pragma unroll
                    for (int i = 0; i < TN; ++i)
#pragma unroll
                        for (int j = 0; j < TM; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[0][i], fa[0][j], acc[i][j], 0, 0, 0);
                    if constexpr (!BOTH) {
#pragma unroll
                        for (int j = 0; j < TM; ++j)
                            fa[0][j] = *reinterpret_cast<const h16x8*>(ab + arow[j] * 128 + (((4 + fq) ^ (arow[j] & 7)) << 4));
#pragma unroll
                        for (int i = 0; i < TN; ++i) fw[0][i] = *reinterpret_cast<const h16x8*>(sw + w_rd + (rsw ^ 64) + i * 16 * 128);
                    }
#pragma unroll
                    for (int i = 0; i < TN; ++i)
#pragma unroll
                        for (int j = 0; j < TM; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[BOTH ? 1 : 0][i], fa[BOTH ? 1 : 0][j], acc[i][j], 0, 0, 0);
                    PBE_SETPRIO(0);
                    slot_rd = slot_rd + 1 == S ? 0 : slot_rd + 1;
                    slot_wr = slot_wr + 1 == S ? 0 : slot_wr + 1;
                }
            }
        }
    } else {
    PBE_STAMP(1);                                    // loader state (+ tap table) ready
#pragma unroll
    for (int t = 0; t < D; ++t)
        if (kt0 + t < nk) issue(kt0 + t, t);
    PBE_STAMP(2);                                    // ring primed (issue only)
    stage_svec();
    int slot_rd = 0, slot_wr = D % S;
    for (int kt = kt0; kt < nk; ++kt) {
#ifdef PBE_STAMPS
        if (kt == kt0 + 1) PBE_STAMP(3);             // first k-tile consumed: prologue latency ends
#endif
        const int rem = min(nk - 1 - kt, D - 1);     // later tiles that may stay in flight
        PBE_ACC_T0();
        if (D >= 3 && rem >= 2) wait_vmcnt<2 * LPT>();
        else if (D >= 2 && rem >= 1) wait_vmcnt<LPT>();
        else wait_vmcnt<0>();
        PBE_ACC(acc_w_);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PBE_ACC_T0();
        __builtin_amdgcn_s_barrier();                // tile kt landed for every wave; slot_wr (tile kt-1's) is free
        PBE_ACC(acc_b2_);
        PBE_ACC_T0();
        const unsigned char* sa = smem + slot_rd * STAGE;
        const unsigned char* sw = sa + A_BYTES;
        h16x8 fa[BOTH ? 2 : 1][TM], fw[BOTH ? 2 : 1][TN];
#pragma unroll
        for (int ks = 0; ks < (BOTH ? 2 : 1); ++ks) {
#pragma unroll
            for (int j = 0; j < TM; ++j) fa[ks][j] = *reinterpret_cast<const h16x8*>(sa + a_rd + (rsw ^ (ks * 64)) + j * 16 * 128);
#pragma unroll
            for (int i = 0; i < TN; ++i) fw[ks][i] = *reinterpret_cast<const h16x8*>(sw + w_rd + (rsw ^ (ks * 64)) + i * 16 * 128);
        }
        __builtin_amdgcn_sched_barrier(0);           // fragment reads go out first, the DMA pieces behind them (tools/ubench_loop.hip:
        if (kt + D < nk) issue(kt + D, slot_wr);     //  cheaper there than before the reads, between the MFMAs or after them)
        __builtin_amdgcn_sched_barrier(0);
        PBE_ACC(acc_r_);
        PBE_ACC_T0();
        PBE_SETPRIO(1);
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j)
                acc[i][j] = F8 ? mfma_pair_f8(fw[0][i], fa[0][j], acc[i][j]) : __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[0][i], fa[0][j], acc[i][j], 0, 0, 0);
        if constexpr (!BOTH) {
#pragma unroll
            for (int j = 0; j < TM; ++j) fa[0][j] = *reinterpret_cast<const h16x8*>(sa + a_rd + (rsw ^ 64) + j * 16 * 128);
#pragma unroll
            for (int i = 0; i < TN; ++i) fw[0][i] = *reinterpret_cast<const h16x8*>(sw + w_rd + (rsw ^ 64) + i * 16 * 128);
        }
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j)
                acc[i][j] = F8 ? mfma_pair_f8(fw[BOTH ? 1 : 0][i], fa[BOTH ? 1 : 0][j], acc[i][j])
                               : __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[BOTH ? 1 : 0][i], fa[BOTH ? 1 : 0][j], acc[i][j], 0, 0, 0);
        PBE_SETPRIO(0);
        PBE_ACC(acc_m_);
        slot_rd = slot_rd + 1 == S ? 0 : slot_rd + 1;
        slot_wr = slot_wr + 1 == S ? 0 : slot_wr + 1;
    }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    PBE_STAMP(4);                                    // main loop issued
    PBE_ACC_STORE();

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
        PBE_STAMP(6);
        PBE_STAMP(8);
        return;
    }

    // ---- epilogue: one wave-row group (WM rows of the tile) at a time through LDS ----
    h16* sC = reinterpret_cast<h16*>(smem);
    h16* Cb = p.C + bz * p.sC;
    // phase form: row m = (b, y, x) of phase (py, px) lands on output pixel (2y + py, 2x + px) of the [B, 2H, 2W, N] tensor
    const int ph_off = (MODE == 1 && p.phase) ? (int)(blockIdx.y >> 1) * 2 * p.Wd + (int)(blockIdx.y & 1) : 0;
    auto out_row = [&](int m) -> long {
        if (MODE == 1 && p.phase) { const int q = m / p.Wd; return 4L * p.Wd * q + 2 * (m - q * p.Wd) + ph_off; }
        return m;
    };
    const h16* Rb = p.resid ? p.resid + bz * p.sR : nullptr;
    constexpr int CPR = BN / 8;
    // whole C tile at once when it fits the ring's LDS, else one wave-row group per pass
    constexpr bool ONE_PASS = (size_t)BM * CLD * 2 <= (size_t)S * STAGE;
    constexpr int NG = ONE_PASS ? 1 : NWM;            // passes
    // GroupNorm statistics from the copy-out (convs, whole-tile epilogues): scratch behind the C tile, inside the ring's LDS
    constexpr int GS_OFF = (BM * CLD * 2 + 15) & ~15;
    constexpr bool GS_OK = MODE != 0 && ONE_PASS && !F8 && (NT / (BN / 8)) >= 1 && (size_t)GS_OFF + ((size_t)(NT / (BN / 8)) * BN + BN) * 8 <= (size_t)RING;
    constexpr int GR = ONE_PASS ? BM : WM;            // rows per pass
    // The register -> LDS half has a compile-time FAST path (bias / row vector staged in svec, one sample per tile, no
    // per-row bias): the generic form (per-element bounds checks, global bias loads and row-vector gathers with an
    // integer division per quad, all in one unrolled body) ran ~7 000 instructions per thread and cost 35 % of a
    // K = 320 GEMM.
    // Tiles with <= 96 accumulator registers also specialise the activation at compile time (5 copies of the unrolled body);
    // for the 256-row tiles (128-160 accumulator registers) that many copies push the allocator into scratch, so there the
    // activation stays a uniform run-time branch per quad.
    constexpr bool ARMS = TM * TN * 4 <= 96;
    constexpr int CLDT = GR + 8;                      // EX: row stride of the transposed C tile (V^T tiles)
    static_assert(EX == 0 || (ONE_PASS && (size_t)BN * CLDT * 2 <= (size_t)S * STAGE), "extended epilogue: whole C tile (either orientation) in the ring's LDS");
    const bool vtile = EXV && p.vt && n0 >= p.vt_col0;     // uniform: this tile's columns belong to the transposed output
    auto stage = [&](int g, auto FAST, auto ACT, auto VTT) {
        constexpr bool F = decltype(FAST)::value, VT = decltype(VTT)::value;
        const int A = decltype(ACT)::value >= 0 ? decltype(ACT)::value : p.act;
        // alpha is re-materialised opaquely per pass: as a plain loop invariant, acc * alpha is hoisted out of the pass loop
        // into a second full set of accumulator registers and the 256-row tiles spill (measured: 30 us of epilogue per tile).
        float al = p.alpha;
        asm volatile("" : "+s"(al));
        float ln_rs[EXL ? TM : 1], ln_nm[EXL ? TM : 1];    // LayerNorm fold: this lane's rows' rstd and -mean rstd (LDS, read once)
        if constexpr (EXL) {
            if (EX != 7 || p.ln_stat) {
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    const int ml = (ONE_PASS ? wm * WM : 0) + j * 16 + fr;
                    ln_rs[j] = sca[g * GR + ml]; ln_nm[j] = scw[g * GR + ml];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int nl = wn * WN + i * 16 + fq * 4;
            const int n = n0 + nl;
            float ali = al, c1a[4] = {0.f, 0.f, 0.f, 0.f};
            if constexpr (EXV) ali = (p.alpha_cols > 0 && n >= p.alpha_cols) ? 1.f : al;
            if constexpr (EXL) {
                if (EX != 7 || p.ln_stat) {
                    const f32x4 c1 = *reinterpret_cast<const f32x4*>(lnc + nl);
#pragma unroll
                    for (int r = 0; r < 4; ++r) c1a[r] = ali * c1[r];
                }
            }
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
                // every path forms  acc * alpha + (bias + row vector)  with the SAME association (the vector sum first, one fused
                // multiply-add): which path a tile takes depends on the tile shape, and the tile shape must not change the bits
                float add[4] = {bn[0], bn[1], bn[2], bn[3]};
                if (!F) {
                    const int m = m0 + g * GR + ml;
                    if (p.sv_ok) {                           // several samples per tile (8x8 level), or a per-row bias
                        const f32x4 t = *reinterpret_cast<const f32x4*>(svec + (sv_ns > 1 ? (g * GR + ml) / p.group_rows : 0) * BN + nl);
#pragma unroll
                        for (int r = 0; r < 4; ++r) add[r] = t[r];
                    } else if (p.rowvec && m < p.M) {
                        const h16* rv = p.rowvec + (long)(m / p.group_rows) * p.ldv + n;
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (n + r < p.N) add[r] = bn[r] + (float)rv[r];
                    }
                }
                if constexpr (F8) {
                    const float sm = sca[g * GR + ml] * al;
                    const f32x4 sn = *reinterpret_cast<const f32x4*>(scw + nl);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf(acc[i][j][r], sm * sn[r], add[r]);
                } else if constexpr (EX != 0) {
                    // NO run-time branch inside the unrolled body: a uniform branch per quad splits it into 40 basic blocks and the
                    // GELU chains (rcp -> 4 fma -> exp) of different quads can no longer be interleaved (the GEGLU projection ran 12 % slower).
                    // The LayerNorm-only / q|k|v forms therefore ALWAYS fold (their dispatch guarantees ln_stat), the V^T orientation is a
                    // compile-time argument of stage(); only the catch-all form (EX = 7) decides at run time.
                    if constexpr (EXL && EX != 7) {          // LN(x) W^T = rstd (x W'^T - mean colsum(W')) with W' = W gamma (bias holds W beta + b)
                        const float ars = ali * ln_rs[j];    //   alpha (rstd acc - mean rstd colsum) + bias  =  acc (alpha rstd) + ((-mean rstd) (alpha colsum) + bias)
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf(acc[i][j][r], ars, __builtin_fmaf(ln_nm[j], c1a[r], add[r]));
                    } else {
                        const float ars = (EXL && p.ln_stat) ? ali * ln_rs[j] : ali, nmj = (EXL && p.ln_stat) ? ln_nm[j] : 0.f;      // (c1a = 0 without ln_stat)
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf(acc[i][j][r], ars, __builtin_fmaf(nmj, c1a[r], add[r]));
                    }
                    if constexpr (VT) {                      // V^T tile: the C tile goes to LDS transposed, [column][row]
#pragma unroll
                        for (int r = 0; r < 4; ++r) sC[(nl + r) * CLDT + ml] = (h16)v[r];
                        continue;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf(acc[i][j][r], al, add[r]);
                }
                if (!F) {
                    const int m = m0 + g * GR + ml;
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
    // LDS -> global half: whole 16-byte row segments, residual added on the way out.  A thread's chunks are handled in batches
    // of UB: all residual loads of a batch are issued first, then the LDS reads, then the adds and stores - one exposed memory
    // round trip per batch instead of one per chunk (tools/phase_stamps.py: the chunk-at-a-time form spent 11 400 of a
    // K = 320 GEMM workgroup's 33 800 cycles here when a residual is fused, 4 000 without).
    auto copy_out = [&](int g, auto GG) {
        constexpr bool gg = decltype(GG)::value;             // GEGLU halves the output width
        constexpr int cpr = gg ? CPR / 2 : CPR;
        constexpr int TOT = GR * cpr, IT = (TOT + NT - 1) / NT;
        // (tiles whose accumulators stay live across the passes - NG > 1 - have no registers for a deep batch)
        constexpr int UB = !ONE_PASS ? (IT % 2 == 0 ? 2 : 1) : (IT % 5 == 0 ? 5 : (IT % 4 == 0 ? 4 : (IT % 3 == 0 ? 3 : (IT % 2 == 0 ? 2 : 1))));
        const int Nout = gg ? p.N >> 1 : p.N, nb = gg ? n0 >> 1 : n0;
        if constexpr (EXV && !gg) {
            if (vtile) {
                // transposed tile: LDS row = output channel, 16-byte chunks of 8 consecutive tokens -> vt[b][channel][token] rows
                constexpr int tpr = GR / 8;
                for (int idx = tid; idx < BN * tpr; idx += NT) {
                    const int crow = idx / tpr, tch = idx - crow * tpr;
                    const int m = m0 + tch * 8, c = n0 - p.vt_col0 + crow;
                    if (m >= p.M || n0 + crow >= p.N) continue;
                    const int b = m / p.vt_tok, tok = m - b * p.vt_tok;
                    *reinterpret_cast<h16x8*>(p.vt + (long)b * p.vt_bs + (long)c * p.vt_rs + tok) = *reinterpret_cast<const h16x8*>(sC + crow * CLDT + tch * 8);
                }
                return;
            }
        }
        if constexpr (EXS && !gg) {
            if (p.rstat) {
                // Row statistics for the LayerNorm that reads this output: 8 lanes per row (full 128-byte lines per row and
                // instruction), each lane sums the STORED fp16 values of its chunks, three xor-shuffles finish the row -
                // fixed order, no atomics.  Partial (sum, sumsq) of (column tile tn_i, row m) -> rstat[tn_i * M + m].
                constexpr int RPP = NT / 8, CPL = (CPR + 7) / 8, NIT = (GR + RPP - 1) / RPP;
                const int l8 = tid & 7, rr = tid >> 3;
                h16x8 r[NIT][CPL];
                if (Rb) {                                    // every residual load of the thread first: one exposed round trip, not NIT
#pragma unroll
                    for (int it = 0; it < NIT; ++it) {
                        const int row = it * RPP + rr, m = m0 + g * GR + row;
#pragma unroll
                        for (int c = 0; c < CPL; ++c) {
                            const int ch = l8 + 8 * c, n = n0 + ch * 8;
                            if (row < GR && m < p.M && ch < CPR && n < p.N) r[it][c] = *reinterpret_cast<const h16x8*>(Rb + (long)m * p.ldr + n);
                        }
                    }
                }
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int row = it * RPP + rr, m = m0 + g * GR + row;
                    const bool rok = row < GR && m < p.M;
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int c = 0; c < CPL; ++c) {
                        const int ch = l8 + 8 * c, n = n0 + ch * 8;
                        if (!(rok && ch < CPR && n < p.N)) continue;
                        h16x8 v = *reinterpret_cast<const h16x8*>(sC + row * CLD + ch * 8);
                        if (Rb) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = (h16)((float)v[e] + (float)r[it][c][e]);
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; s1 += f; s2 = __builtin_fmaf(f, f, s2); }
                        *reinterpret_cast<h16x8*>(Cb + (long)m * p.ldc + n) = v;
                    }
#pragma unroll
                    for (int o = 1; o < 8; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
                    if (rok && l8 == 0) *reinterpret_cast<float2*>(p.rstat + 2 * ((long)tn_i * p.rstat_ld + m)) = make_float2(s1, s2);
                }
                return;
            }
        }
        if constexpr (GS_OK && !gg) {
            if (p.gstat) {
                // GroupNorm statistics of this tile's stored values.  A thread keeps ONE 16-byte column chunk and walks rows (slot, slot + RSL,
                // ...): 8 column sums + 8 sums of squares in registers, then a fixed-order fold over the row slots and over a group's columns
                // through LDS (no atomics: bit-reproducible).  Row batches of 4: the residual loads of a batch are issued together.
                constexpr int RSL = NT / CPR, GB = 4;
                float* red = reinterpret_cast<float*>(smem + GS_OFF);
                const int slot = tid / CPR, ch = tid - slot * CPR, n = n0 + ch * 8;
                float ca[8], cq[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { ca[e] = 0.f; cq[e] = 0.f; }
                if (slot < RSL && n < p.N) {
                    for (int r0 = slot; r0 < GR; r0 += RSL * GB) {
                        h16x8 v[GB], r[GB];
                        bool ok[GB];
#pragma unroll
                        for (int u = 0; u < GB; ++u) {
                            const int row = r0 + u * RSL, m = m0 + row;
                            ok[u] = row < GR && m < p.M;
                            if (ok[u] && Rb) r[u] = *reinterpret_cast<const h16x8*>(Rb + (long)m * p.ldr + n);
                        }
#pragma unroll
                        for (int u = 0; u < GB; ++u)
                            if (ok[u]) v[u] = *reinterpret_cast<const h16x8*>(sC + (r0 + u * RSL) * CLD + ch * 8);
#pragma unroll
                        for (int u = 0; u < GB; ++u) {
                            if (!ok[u]) continue;
                            if (Rb) {
#pragma unroll
                                for (int e = 0; e < 8; ++e) v[u][e] = (h16)((float)v[u][e] + (float)r[u][e]);
                            }
#pragma unroll
                            for (int e = 0; e < 8; ++e) { const float f = (float)v[u][e]; ca[e] += f; cq[e] = __builtin_fmaf(f, f, cq[e]); }
                            *reinterpret_cast<h16x8*>(Cb + out_row(m0 + r0 + u * RSL) * p.ldc + n) = v[u];
                        }
                    }
                }
                if (slot < RSL) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) *reinterpret_cast<float2*>(red + ((slot * BN) + ch * 8 + e) * 2) = make_float2(ca[e], cq[e]);
                }
                __syncthreads();
                float* col = red + RSL * BN * 2;               // [BN][2] column totals
                if (tid < BN) {
                    float a = 0.f, q = 0.f;
                    for (int sl = 0; sl < RSL; ++sl) { const float2 t = *reinterpret_cast<const float2*>(red + (sl * BN + tid) * 2); a += t.x; q += t.y; }
                    *reinterpret_cast<float2*>(col + tid * 2) = make_float2(a, q);
                }
                __syncthreads();
                const int cg = p.gs_cg, gl = tid;               // group gl of this tile's BN / cg
                if (gl * cg < BN && n0 + gl * cg < p.N) {
                    float a = 0.f, q = 0.f;
                    for (int c = gl * cg; c < (gl + 1) * cg; ++c) { const float2 t = *reinterpret_cast<const float2*>(col + c * 2); a += t.x; q += t.y; }
                    const int b = m0 / p.gs_hw, blk = (m0 - b * p.gs_hw) / BM, nblk = p.gs_hw / BM;
                    *reinterpret_cast<float2*>(p.gstat + (((long)b * nblk + blk) * p.gs_groups + n0 / cg + gl) * 2) = make_float2(a, q);
                }
                return;
            }
        }
        if (p.vec) {
            for (int it0 = 0; it0 < IT; it0 += UB) {
                h16x8 v[UB], r[UB];
                long go[UB];
                int lo[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int idx = tid + (it0 + u) * NT;
                    const int row = idx / cpr, ch = idx - row * cpr;
                    const int m = m0 + g * GR + row, n = nb + ch * 8;
                    const bool ok = idx < TOT && m < p.M && n < Nout;
                    go[u] = ok ? out_row(m) * p.ldc + n : -1;
                    lo[u] = row * CLD + ch * 8;
                    if (ok && Rb) r[u] = *reinterpret_cast<const h16x8*>(Rb + (long)m * p.ldr + n);
                }
#pragma unroll
                for (int u = 0; u < UB; ++u)
                    if (go[u] >= 0) v[u] = *reinterpret_cast<const h16x8*>(sC + lo[u]);
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    if (go[u] < 0) continue;
                    if (Rb) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[u][e] = (h16)((float)v[u][e] + (float)r[u][e]);
                    }
                    *reinterpret_cast<h16x8*>(Cb + go[u]) = v[u];
                }
            }
            return;
        }
        for (int idx = tid; idx < TOT; idx += NT) {
            const int row = idx / cpr, ch = idx - row * cpr;
            const int m = m0 + g * GR + row, n = nb + ch * 8;
            if (m >= p.M || n >= Nout) continue;
            const h16x8 v = *reinterpret_cast<const h16x8*>(sC + row * CLD + ch * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (n + e < Nout) {
                    float f = (float)v[e];
                    if (Rb) f += (float)Rb[(long)m * p.ldr + n + e];
                    Cb[out_row(m) * p.ldc + n + e] = (h16)f;
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
                if constexpr (EXV) {
                    if (vtile) { stage(g, FAST, std::integral_constant<int, 0>{}, std::true_type{}); return; }      // V^T tiles: no activation
                }
                if constexpr (ARMS) {
                    switch (p.act) {
                        case 1: stage(g, FAST, std::integral_constant<int, 1>{}, std::false_type{}); break;
                        case 2: stage(g, FAST, std::integral_constant<int, 2>{}, std::false_type{}); break;
                        case 3: stage(g, FAST, std::integral_constant<int, 3>{}, std::false_type{}); break;
                        case PBE_ACT_GEGLU: stage(g, FAST, std::integral_constant<int, PBE_ACT_GEGLU>{}, std::false_type{}); break;
                        default: stage(g, FAST, std::integral_constant<int, 0>{}, std::false_type{}); break;
                    }
                } else {
                    stage(g, FAST, std::integral_constant<int, -1>{}, std::false_type{});
                }
            };
            if (fast) run(std::true_type{});
            else run(std::false_type{});
        }
        __syncthreads();
#ifdef PBE_STAMPS
        if (g == NG - 1) PBE_STAMP(5);               // last register -> LDS pass done (includes waiting for the MFMAs to drain)
#endif
        if (p.act == PBE_ACT_GEGLU) copy_out(g, std::true_type{});
        else copy_out(g, std::false_type{});
    }
    PBE_STAMP(6);                                    // stores issued
    PBE_STAMP(8);                                    // wall clock (100 MHz) of the end
}

// Sum the split-K slabs in a fixed order (deterministic) and apply the epilogue: 4 columns per thread.
static __global__ void __launch_bounds__(256) splitk_reduce_kernel(const IGemmP p) {
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
struct TileCfg { int bm, bn, nwm, nwn, slots_per_cu; double eff; int hpa; };      // hpa > 0: halo-resident conv tile (MODE 2), rows of its halo image
// eff = relative per-FLOP efficiency of the tile when the chip is full (ordered by staged bytes per FLOP).
// Ring depth per tile (k-tiles of 64): the deepest that fits the LDS share of the tile's workgroups per CU.
static const TileCfg kCfg[] = {
    {256, 256, 2, 4, 1, 1.00},           // 0: S = 2, 128 KiB
    {256, 128, 4, 2, 1, 0.82},           // 1: S = 3, 144 KiB
    {128, 256, 2, 4, 1, 0.75},           // 2: S = 3, 144 KiB
    {128, 128, 2, 2, 2, 0.88},           // 3: S = 2,  64 KiB, 4 waves: two workgroups per CU
    {128, 64, 2, 2, 3, 0.65},            // 4: S = 2,  48 KiB
    {64, 128, 2, 2, 3, 0.65},            // 5: S = 2,  48 KiB
    {64, 64, 2, 2, 4, 0.60},             // 6: S = 2,  32 KiB
    {256, 320, 2, 4, 1, 1.05},           // 7: S = 2, 144 KiB: N = 320 / 640 / 960 / 1280 without column padding (142 FLOP per staged byte)
    {128, 320, 2, 4, 1, 0.96},           // 8: S = 2, 112 KiB: same, half the rows: fills the chip when M / 256 < 256 tiles
    {128, 160, 2, 2, 2, 0.90},           // 9: S = 2,  72 KiB, 4 waves: two workgroups per CU overlap each other's prologue / epilogue
    // halo-resident 3x3 conv tiles (stride 1, pad 1, image width 8 .. 128 = tile width): only the weights stream per k-tile
    {256, 160, 4, 2, 1, 1.40, 392},      // 10: weight ring 3, 160 KiB: 256 pixels (4 rows at 64x64) x 160 channels, 8 waves, ping-pong
    {128, 160, 4, 2, 1, 1.20, 264},      // 11: weight ring 3, 128 KiB: 128 pixels x 160 channels
    {128, 320, 2, 4, 1, 1.25, 264},      // 12: weight ring 2, 149 KiB: 128 pixels x 320 channels
    {256, 128, 4, 2, 1, 1.30, 392},      // 13: weight ring 3, 148 KiB: channel counts that are multiples of 128 only, ping-pong
    {128, 128, 4, 2, 1, 1.10, 392},      // 14: weight ring 3, 148 KiB: one 128-pixel row of a 128-wide image (VAE)
    // deep-ring forms of the small dense tiles, ONE workgroup per CU: for grids of <= 256 workgroups (M <= 2 048 rows) the second
    // workgroup of a CU never arrives, and S = 2 then leaves one k-tile in flight per CU - a k-tile per DMA round trip
    {128, 128, 2, 2, 1, 0.50},           // 15: S = 4, 128 KiB
    {128, 64, 2, 2, 1, 0.40},            // 16: S = 4,  96 KiB
    {64, 64, 2, 2, 2, 0.38},             // 17: S = 4,  64 KiB
    {128, 160, 2, 2, 1, 0.52},           // 18: S = 4, 144 KiB
    // A-stationary persistent tiles (igemm_astat.hip): K = 320 GEGLU projection with the LayerNorm fold, A block in registers, 4 waves,
    // two workgroups per CU; never picked by the heuristic (eff 0), only through desc.tile_cfg / the tuned table where pbe_astat_ok() holds
    {128, 128, 4, 1, 2, 0.0},            // 19: weight ring 48 KiB
    {128, 160, 4, 1, 2, 0.0},            // 20: weight ring 60 KiB (256 registers, 14 spilled)
    // dense 8-wave tile with 160 columns: 6.5 LDS-DMA pieces per wave and k-tile against 40 MFMAs (the 4-wave 128x160 tile: 9)
    {256, 160, 4, 2, 1, 0.0}};           // 21: S = 3, 156 KiB (MODE 0 only; through the tuned table)
static const int kNCfg = sizeof(kCfg) / sizeof(kCfg[0]);

#ifdef PBE_STAMPS
extern unsigned long long* g_pbe_stamps;
#endif
extern int g_pbe_force_cfg;     // (defined in igemm.hip) pbe_tune(1, cfg index [| splits << 8]) forces a tile config (and split-K factor); -1 = heuristic
extern int g_pbe_allow_splitk;  // pbe_tune(2, 0/1)

// k-tiles of 64 per split-K granule: a halo tile splits at channel-block boundaries (9 taps)
static inline int no_empty_slices(int nk, int granule, int s) {
    const int units = (nk + granule - 1) / granule;
    if (s > units) s = units;
    if (s < 2) return 1;
    const int per = (units + s - 1) / s;
    return (units + per - 1) / per;
}

static int splits_for(const IGemmP& p, const TileCfg& c, int batch, size_t ws_bytes, long tiles) {
    if (!g_pbe_allow_splitk || batch != 1 || !p.ws || (p.N & 3) || (p.ldc & 3) || (p.resid && (p.ldr & 3))) return 1;
    const int nk = (p.K + 63) >> 6;                          // k-tiles of 64
    const long slots = 256L * c.slots_per_cu;
    if (tiles * 4 > slots * 3 || nk < 8) return 1;           // grid already fills >= 75 % of the chip
    int s = (int)((slots * 5 / 4 + tiles - 1) / tiles);
    if (s > nk / 4) s = nk / 4;
    if (s > 32) s = 32;
    while (s > 1 && (size_t)s * p.M * p.N * sizeof(float) > ws_bytes) --s;
    return no_empty_slices(nk, c.hpa ? 9 : 1, s);
}

// Shallow-K problems (< 20 k-tiles) are dominated by the prologue / epilogue, deep-K problems by staged bytes per
// FLOP.  Both efficiency rows are fitted to the 472 measured shapes of profiles/r01_autotune_report.txt (the
// heuristic then costs 4 % over the best tile per shape, 12 % before the fit).  pbe_amd/tuned_mi355x.json overrides
// this per shape (desc.tile_cfg), so these rows only decide shapes outside the table.
static const double kEffShallow[] = {0.82, 0.72, 0.66, 1.00, 0.84, 0.85, 0.80, 0.80, 0.95, 0.97, 1.4, 1.2, 1.25, 1.3, 1.1, 0.5, 0.4, 0.38, 0.52, 0.0, 0.0, 0.0};

// Can this conv run as a halo-resident tile of bm pixels with a halo image of hpa rows?  Returns the image rows per tile (0: no).
static int halo_rows(const IGemmP& p, int mode, int bm, int hpa) {
    if (mode != 1 || p.cstride != 1 || p.pad != 1 || p.ups || p.phase || p.cb != 64) return 0;
    const int W = p.Wd, H = p.H;
    if (W < 8 || W > 128 || (W & (W - 1)) || bm % W || (p.M % bm)) return 0;
    const int th = bm / W < H ? bm / W : H;
    if (H % th) return 0;
    const int nsub = bm / (th * W);                               // whole images per tile when the image is smaller than the tile
    if (nsub > 1 && th != H) return 0;
    if (nsub * ((th + 2) * (W + 1) + 1) > hpa) return 0;       // halo rows: row stride W + 1 (shared zero column) + 1
    return th;
}

// dense tiles that are instantiated with the extended epilogue (EX): the one-pass 4-wave tiles and 128x320
static const unsigned kExCfgs = (1u << 3) | (1u << 4) | (1u << 5) | (1u << 6) | (1u << 8) | (1u << 9) | (1u << 15) | (1u << 16) | (1u << 17) | (1u << 18) | (3u << 19);
bool pbe_astat_ok(const IGemmP& p, int batch, int cfg);      // igemm_astat.hip: can tile 19 run this problem?
static inline bool ex_needed(const IGemmP& p) { return p.alpha_cols > 0 || p.ln_stat || p.rstat || p.vt; }

// want_cfg: -1 = heuristic; else (tile config index) | (split-K factor << 8), factor 0 = heuristic factor for that tile.
// A requested factor is clamped to what the problem allows (batch 1, >= 4 k-tiles of 64 per slice, slabs fit the workspace).
// (the developer knobs g_pbe_force_cfg / g_pbe_allow_splitk are READ here and never written outside pbe_tune: a caller that must
//  not split passes ws_bytes = 0, the fallback below passes use_force = false)
static Plan plan_igemm(const IGemmP& p, int batch, size_t ws_bytes, int want_cfg, int mode, bool use_force = true) {
    Plan best{3, 1};
    double best_score = -1.0;
    const int want = (use_force && g_pbe_force_cfg >= 0) ? g_pbe_force_cfg : want_cfg;
    const int forced = (want >= 0 && (want & 255) < kNCfg) ? (want & 255) : -1;
    const int want_splits = want >= 0 ? (want >> 8) & 255 : 0;
    const bool shallow = ((p.K + 63) >> 6) < 10;
    for (int c = 0; c < kNCfg; ++c) {
        if (forced >= 0 && c != forced) continue;
        TileCfg t = kCfg[c];
        if (c >= 19 && (forced != c || mode != 0 || (c < 21 && !pbe_astat_ok(p, batch, c)))) continue;
        if (ex_needed(p) && (!(kExCfgs >> c & 1) || (p.vt && p.vt_col0 % t.bn))) continue;   // extended epilogue: its tiles only, V^T columns start on a tile
        if (t.hpa && !halo_rows(p, mode, t.bm, t.hpa)) continue;              // a forced halo tile that does not apply falls back below
        if (t.hpa && p.N % 8) continue;
        if (shallow) t.eff = kEffShallow[c];
        const long tm = (p.M + t.bm - 1) / t.bm, tn = (p.N + t.bn - 1) / t.bn;
        const long tiles = tm * tn * batch;
        int sp = splits_for(p, t, batch, ws_bytes, tiles);
        if (forced >= 0 && want_splits > 0) {
            const int nk = (p.K + 63) >> 6;
            sp = want_splits;
            if (!g_pbe_allow_splitk || batch != 1 || !p.ws || (p.N & 3) || (p.ldc & 3) || (p.resid && (p.ldr & 3))) sp = 1;
            if (sp > nk / 4) sp = nk / 4;
            while (sp > 1 && (size_t)sp * p.M * p.N * sizeof(float) > ws_bytes) --sp;
            sp = no_empty_slices(nk, t.hpa ? 9 : 1, sp);
        }
        const double useful = (double)p.M * p.N * batch / ((double)tiles * t.bm * t.bn);
        const double blocks = (double)tiles * sp, slots = 256.0 * t.slots_per_cu;
        const double rounds = (double)((long)((blocks + slots - 1) / slots));
        const double quant = blocks / (rounds * slots);
        const double split_cost = sp > 1 ? 0.93 : 1.0;        // slab write + reduce launch
        const double score = t.eff * useful * (0.30 + 0.70 * quant) * split_cost;
        if (score > best_score) { best_score = score; best = Plan{c, sp}; }
    }
    if (best_score < 0.0 && forced >= 0)              // the requested tile cannot run this problem: let the heuristic choose
        best = plan_igemm(p, batch, ws_bytes, -1, mode, false);
    return best;
}

extern int g_pbe_pingpong;      // pbe_tune(4, 0/1): ping-pong main loop of the halo-resident conv tiles
extern int g_pbe_mfast;         // pbe_tune(5, 0/1): let a launch walk its tiles m fastest per XCD when that fetches fewer bytes

template <int BM, int BN, int NWM, int NWN, int S, int MODE, int HPA = 0, bool PP = false, bool F8 = false, int EX = 0>
static void launch_cfg(IGemmP p, int batch, hipStream_t s) {
    // (ping-pong only where a wave's MFMA phase - (BM/NWM/16) x (BN/NWN/16) x 2 MFMAs - is as long as its read phase: measured
    //  25 % SLOWER on the 128x160 halo tile, whose 20 MFMAs cannot cover 14 fragment reads + 3 DMA issues)
    if constexpr (MODE == 2 && !PP && S == 3 && (BM / NWM / 16) * (BN / NWN / 16) >= 16) {
        if (g_pbe_pingpong) { launch_cfg<BM, BN, NWM, NWN, S, MODE, HPA, true>(p, batch, s); return; }
    }
    constexpr size_t ring = MODE == 2 ? (size_t)2 * HPA * 128 + (size_t)S * BN * 128 : (size_t)S * (BM + BN) * 128;
    constexpr size_t c_bytes = (size_t)(BM / NWM) * (BN + 8) * 2;
    constexpr int SVR = (ring > c_bytes ? ring : c_bytes) + 4 * BN * sizeof(float) <= 160 * 1024 ? 4 : 3;                              // svec rows (samples per tile)
    constexpr size_t lds = (ring > c_bytes ? ring : c_bytes) + (MODE == 1 ? 9 * BM * sizeof(int) : 0) + (EX != 0 ? 4 : SVR) * BN * sizeof(float) +     // + svec[SVR][BN]
                           (F8 ? (BM + BN) * sizeof(float) : 0) + (EX != 0 ? (2 * BM + BN) * sizeof(float) : 0);                            // + operand scales / LayerNorm rows + colsum
    p.sv_ok = !p.rowvec || p.group_rows % BM == 0 || (BM % p.group_rows == 0 && BM / p.group_rows <= SVR);
    static_assert(lds <= 160 * 1024, "tile does not fit the 160 KiB LDS");
    static std::atomic<uint64_t> attr_done{0};
    pbe_raise_dynamic_lds(attr_done, reinterpret_cast<const void*>(&igemm_kernel<BM, BN, NWM, NWN, MODE, S, HPA, PP, F8, EX>), (int)lds);
    const int tiles_m = cdiv(p.M, BM), tiles_n = cdiv(p.N, BN);
    dim3 grid((unsigned)(tiles_m * tiles_n), batch, p.splits > 1 ? p.splits : 1);
    {   // bytes the 8 L2s fetch under either tile order (an XCD owns a contiguous run of tiles_m * tiles_n / 8 tiles)
        const double a_bytes = MODE != 0 ? 2.0 * (double)(p.M / (p.Ho * p.Wo)) * p.H * p.Wd * (p.C1 + p.C2) : 2.0 * p.M * (double)p.K;
        const double w_bytes = 2.0 * p.N * (double)p.K;
        const double run = tiles_m * (double)tiles_n / 8.0;
        auto frac = [](double blocks_touched, int blocks) { const double f = blocks_touched / blocks; return f < 1.0 ? f : 1.0; };
        const double n_fast = a_bytes * frac(run / tiles_n + 1.0, tiles_m) + w_bytes * frac(run, tiles_n);
        const double m_fast = w_bytes * frac(run / tiles_m + 1.0, tiles_n) + a_bytes * frac(run, tiles_m);
        p.m_fast = (g_pbe_mfast && batch == 1 && m_fast < 0.9 * n_fast) ? 1 : 0;
    }
    {   // GroupNorm statistics from the copy-out: only where the kernel's GS_OK holds and a tile is a whole number of groups inside one sample
        constexpr int NTH = NWM * NWN * 64;
        constexpr bool one_pass = (size_t)BM * (BN + 8) * 2 <= (size_t)S * (BM + BN) * 128;
        constexpr size_t gs_off = ((size_t)BM * (BN + 8) * 2 + 15) & ~(size_t)15;
        constexpr bool gs_ok = MODE != 0 && one_pass && !F8 && gs_off + ((size_t)(NTH / (BN / 8)) * BN + BN) * 8 <= ring;
        if (p.gstat && !(gs_ok && p.splits <= 1 && batch == 1 && p.gs_cg > 0 && p.gs_hw % BM == 0 && BN % p.gs_cg == 0 && !p.phase && p.vec && p.act != PBE_ACT_GEGLU))
            p.gstat = nullptr;
        if (p.gs_report) *p.gs_report = p.gstat ? p.gs_hw / BM : 0;
    }
    // profiling brackets exactly ONE kernel each, so the event averages agree with rocprofv3's per-kernel averages
    if (MODE == 2) p.th = BM / p.Wd < p.H ? BM / p.Wd : p.H;
    pbe_prof_begin(MODE != 0 ? PBE_K_CONV3 : PBE_K_GEMM, s);
    hipLaunchKernelGGL((igemm_kernel<BM, BN, NWM, NWN, MODE, S, HPA, PP, F8, EX>), grid, dim3(NWM * NWN * 64), lds, s, p, tiles_n);
    {   // algorithmic bytes: every operand once (fp16): activations, weights, output, fused residual
        const double nout = p.act == PBE_ACT_GEGLU ? p.N * 0.5 : (double)p.N;
        const double a_el = MODE != 0 ? (double)(p.M / (p.Ho * p.Wo)) * p.H * p.Wd * (p.C1 + p.C2) : (double)p.M * p.K * batch;
        const double w_el = (double)p.N * p.K * ((MODE == 0 && p.sW) ? batch : 1);
        const double c_el = (double)p.M * nout * batch * (p.resid ? 2.0 : 1.0);
        pbe_prof_end(MODE != 0 ? PBE_K_CONV3 : PBE_K_GEMM, s, 2.0 * p.M * (double)p.N * p.K * batch * (F8 ? 2.0 : 1.0), 2.0 * (a_el + w_el + c_el));
    }
    if (p.splits > 1) {
        const long work = (long)p.M * (p.N >> 2);
        pbe_prof_begin(PBE_K_SPLITK, s);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, s, p);
        pbe_prof_end(PBE_K_SPLITK, s, (double)p.M * p.N * (4.0 * p.splits + 2.0));       // bytes: fp32 slabs in, fp16 out
    }
}


// ---- the instantiations live in five translation units (igemm_dense / _conv / _halo / _f8 / _ex .hip), compiled in parallel ----
void pbe_dispatch_dense(IGemmP p, int batch, hipStream_t s, size_t ws_bytes, int want_cfg);     // MODE 0
void pbe_dispatch_conv(IGemmP p, int batch, hipStream_t s, size_t ws_bytes, int want_cfg);      // MODE 1 gather tiles; forwards tiles 10-14 to
void pbe_launch_halo(int cfg, IGemmP p, int batch, hipStream_t s);                              // MODE 2 halo-resident tiles
void pbe_dispatch_f8(IGemmP p, int batch, hipStream_t s, int want_cfg);
void pbe_dispatch_ex(IGemmP p, int batch, hipStream_t s, int want_cfg);          // picks the feature combination: igemm_ex_{ln,st,qkv,all}.hip
void pbe_launch_ex_ln(int cfg, IGemmP p, int batch, hipStream_t s);
void pbe_launch_ex_st(int cfg, IGemmP p, int batch, hipStream_t s);
void pbe_launch_ex_qkv(int cfg, IGemmP p, int batch, hipStream_t s);
void pbe_launch_ex_all(int cfg, IGemmP p, int batch, hipStream_t s);
void pbe_launch_astat(int cfg, IGemmP p, hipStream_t s);                                                  // tile 19 (igemm_astat.hip)
