// Micro-benchmark: main-loop structure of the implicit-GEMM tile on gfx950 (synthetic data, no epilogue).
// Build + run (GPU box):  hipcc -O3 --offload-arch=gfx950 tools/ubench_tile.hip -o tools/ubench_tile.bin && tools/ubench_tile.bin
// Varies: waves per block (8 = 2x4 of 128x64 wave tiles, 4 = 2x2 of 128x128), bytes per operand row per
// stage (64 = BK 32, 128 = BK 64), ring depth, whether the fragments of the NEXT k-substep are read
// from LDS while the MFMAs of the current one run (PIPE), and whether global loads happen at all (LOAD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int BM, int BN, int NWM, int NWN, int ROWB, int S, int PIPE, int LOAD>
__global__ __launch_bounds__(NWM * NWN * 64) void tile_kernel(const char* __restrict__ A, const char* __restrict__ W, long rowstride, int ksteps,
                                                              int a_tiles, float* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = NWM * NWN * 64;
    constexpr int STAGE = (BM + BN) * ROWB;
    constexpr int LPR = ROWB / 16;
    constexpr int PIECES = STAGE / (NT * 16);
    constexpr int TM = BM / NWM / 16, TN = BN / NWN / 16;
    constexpr int SUB = ROWB / 64;                    // 32-k substeps per stage
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave / NWN, wn = wave % NWN;
    const long arow0 = (long)(blockIdx.x % a_tiles) * BM;
    const char* src[PIECES];
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
        const int piece = (wave * PIECES + i) * 64 + lane;
        const int row = piece / LPR, c = piece % LPR;
        src[i] = (row < BM ? A + (arow0 + row) * rowstride : W + (long)(row - BM) * rowstride) + c * 16;
    }
    f4 acc[TM * TN];
#pragma unroll
    for (int i = 0; i < TM * TN; ++i) acc[i] = f4{0, 0, 0, 0};
    auto issue = [&](int kt) {
        if (!LOAD) return;
        char* st = smem + (kt % S) * STAGE;
#pragma unroll
        for (int i = 0; i < PIECES; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void*)(src[i] + (long)kt * ROWB), (lds_void*)(st + (wave * PIECES + i) * 1024), 16, 0, 0);
    };
    // fragment addresses inside a stage (bytes), swizzled like igemm.hip: 16-B unit u of row r lives at unit u ^ f(r)
    int aoff[TM], boff[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) aoff[i] = (wm * (BM / NWM) + i * 16 + (lane & 15)) * ROWB;
#pragma unroll
    for (int j = 0; j < TN; ++j) boff[j] = (BM + wn * (BN / NWN) + j * 16 + (lane & 15)) * ROWB;
    const int usw = (lane >> 2) & 3;
    auto unit = [&](int sub) { return (((lane >> 4) + sub * 4) ^ usw) * 16; };
    auto read_frags = [&](int kt, int sub, h8 (&a)[TM], h8 (&b)[TN]) {
        const char* st = smem + (kt % S) * STAGE;
        const int u = unit(sub);
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = *(const h8*)(st + aoff[i] + u);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = *(const h8*)(st + boff[j] + u);
    };
    auto mfmas = [&](h8 (&a)[TM], h8 (&b)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i * TN + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i * TN + j], 0, 0, 0);
    };
    auto wait_stage = [&](int outstanding_stages) {       // counted wait: allow that many later stages in flight
        if (!LOAD) return;
        if (outstanding_stages >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PIECES > 63 ? 63 : 3 * PIECES) : "memory");
        else if (outstanding_stages == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PIECES > 63 ? 63 : 2 * PIECES) : "memory");
        else if (outstanding_stages == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES > 63 ? 63 : PIECES) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    if (!PIPE) {
        for (int s = 0; s < S - 1; ++s) issue(s);
        for (int kt = 0; kt < ksteps; ++kt) {
            wait_stage(S - 2);
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int sub = 0; sub < SUB; ++sub) {
                h8 a[TM], b[TN];
                read_frags(kt, sub, a, b);
                if (sub == SUB - 1) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (kt + S - 1 < ksteps) issue(kt + S - 1);
                    else if (LOAD) { for (int i = 0; i < PIECES; ++i) asm volatile("s_nop 0"); }
                    __builtin_amdgcn_sched_barrier(0);
                }
                mfmas(a, b);
            }
        }
    } else {
        // fragments of substep n+1 are fetched from LDS while substep n's MFMAs run (one wave can cover its own LDS latency)
        static_assert(!PIPE || SUB == 1, "PIPE variant written for ROWB 64");
        for (int s = 0; s < S; ++s) issue(s);
        h8 a0[TM], b0[TN], a1[TM], b1[TN];
        wait_stage(S - 1);
        __builtin_amdgcn_s_barrier();
        read_frags(0, 0, a0, b0);
        for (int kt = 0; kt < ksteps; kt += 2) {
            // even step: compute set 0, fetch set 1 from stage kt+1
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            wait_stage(S - 2);
            __builtin_amdgcn_s_barrier();
            if (kt + S < ksteps) issue(kt + S);           // slot kt % S: every wave has its stage-kt fragments in registers
            read_frags(kt + 1, 0, a1, b1);
            mfmas(a0, b0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            wait_stage(S - 2);
            __builtin_amdgcn_s_barrier();
            if (kt + 1 + S < ksteps) issue(kt + 1 + S);
            read_frags(kt + 2, 0, a0, b0);
            mfmas(a1, b1);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float keep = 0.f;
    for (int i = 0; i < TM * TN; ++i) keep += acc[i][0] + acc[i][3];
    if (keep == 123.456f) sink[blockIdx.x] = keep;
}

template <int BM, int BN, int NWM, int NWN, int ROWB, int S, int PIPE, int LOAD>
static void run(const char* name, const char* A, const char* W, long rowstride, int a_tiles, int blocks, float* sink) {
    const int ksteps = (int)(rowstride / ROWB) & ~1;
    const int lds = S * (BM + BN) * ROWB;
    auto kern = tile_kernel<BM, BN, NWM, NWN, ROWB, S, PIPE, LOAD>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) kern<<<blocks, NWM * NWN * 64, lds>>>(A, W, rowstride, ksteps, a_tiles, sink);
    CK(hipEventRecord(e0));
    const int it = 5;
    for (int i = 0; i < it; ++i) kern<<<blocks, NWM * NWN * 64, lds>>>(A, W, rowstride, ksteps, a_tiles, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
    const double flops = (double)blocks * 2.0 * BM * BN * ((double)ksteps * ROWB / 2);
    hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, (const void*)kern));
    printf("%-58s blocks %5d %8.1f us %7.0f TF/s   (regs %d, spill %d B, lds %d)\n", name, blocks, ms * 1e3, flops / ms / 1e9, fa.numRegs,
           (int)fa.localSizeBytes, lds);
    fflush(stdout);
}

int main() {
    const long K = 2880, rowstride = K * 2;
    const int a_tiles_big = 128;
    char *A, *W; float* sink;
    CK(hipMalloc(&A, (size_t)a_tiles_big * 256 * rowstride));
    CK(hipMalloc(&W, (size_t)320 * rowstride));
    CK(hipMalloc(&sink, 1 << 20));
    {   // realistic operand bits: MFMA power (and so the clock) depends on the data, all-zero tiles flatter the result
        const size_t na = (size_t)a_tiles_big * 256 * K, nw = (size_t)320 * K;
        std::vector<_Float16> h(na);
        unsigned s = 12345u;
        for (size_t i = 0; i < na; ++i) { s = s * 1664525u + 1013904223u; h[i] = (_Float16)(((int)(s >> 16) % 2001 - 1000) * 1e-3f); }
        CK(hipMemcpy(A, h.data(), na * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(W, h.data() + 777, nw * 2, hipMemcpyHostToDevice));
    }
    const int blocks = 1024;
    for (int at : {a_tiles_big, 1}) {
        printf("---- A %s\n", at == 1 ? "shared by all blocks (L2 hits)" : "streamed (128 distinct row tiles)");
        run<256, 256, 2, 4, 64, 4, 0, 1>("8 waves 2x4, BK32, S4 (current igemm structure)", A, W, rowstride, at, blocks, sink);
        run<256, 256, 2, 4, 64, 4, 0, 0>("8 waves 2x4, BK32, no global loads (LDS reads + MFMA only)", A, W, rowstride, at, blocks, sink);
        run<256, 256, 2, 4, 64, 4, 1, 1>("8 waves 2x4, BK32, S4, fragment prefetch", A, W, rowstride, at, blocks, sink);
        run<256, 256, 2, 4, 128, 2, 0, 1>("8 waves 2x4, BK64, S2", A, W, rowstride, at, blocks, sink);
        run<256, 128, 4, 2, 128, 3, 0, 1>("8 waves 4x2 of 64x64, 256x128 tile, BK64, S3", A, W, rowstride, at, blocks, sink);
    }
    return 0;
}
