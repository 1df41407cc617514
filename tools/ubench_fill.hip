// Micro-benchmark: how fast can one CU stage K-major fp16 operand tiles into LDS on gfx950?
// Build + run (GPU box):  hipcc -O3 --offload-arch=gfx950 tools/ubench_fill.hip -o gpurun_out/ubench_fill && gpurun_out/ubench_fill
// Answers the design question behind igemm.hip's tile shapes: is the main loop bound by the
// L2 -> LDS fill rate (bytes per CU per second) or by the MFMA pipe?
//   mode 0: global_load_lds (LDS-DMA) 16 B per lane         mode 1: global_load_dwordx4 -> VGPR -> ds_write_b128
//   rowb  : contiguous bytes taken from each operand row per k-step (64 = BK 32, 128 = BK 64)
//   amode : 0 = every block streams its own 256 A rows (L2 misses to MALL/HBM), 1 = all blocks share A (L2 hits)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

template <int ROWB, int MODE, int MFMA>
__global__ __launch_bounds__(512) void fill_kernel(const char* __restrict__ A, const char* __restrict__ W, long rowstride, int ksteps,
                                                   int a_tiles, float* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROWS = 512;                         // 256 A rows + 256 W rows
    constexpr int STAGE = ROWS * ROWB;                // bytes per stage
    constexpr int S = (ROWB == 64) ? 4 : 2;           // 128 KiB ring either way
    constexpr int LPR = ROWB / 16;                    // lanes per row
    constexpr int PIECES = STAGE / (512 * 16);        // 16-B pieces per thread per stage
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const long arow0 = (long)(blockIdx.x % a_tiles) * 256;
    const char* src[PIECES];
    int dst[PIECES];
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
        const int piece = (wave * PIECES + i) * 64 + lane;     // 0 .. STAGE/16
        const int row = piece / LPR, c = piece % LPR;
        src[i] = (row < 256 ? A + (arow0 + row) * rowstride : W + (long)(row - 256) * rowstride) + c * 16;
        dst[i] = (wave * PIECES + i) * 1024;                     // wave-uniform LDS base; lanes land at +lane*16
    }
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 acc[MFMA ? 32 : 1];
#pragma unroll
    for (int i = 0; i < (MFMA ? 32 : 1); ++i) acc[i] = f4{0, 0, 0, 0};
    float keep = 0.f;
    auto issue = [&](int kt) {
        char* st = smem + (kt % S) * STAGE;
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            if (MODE == 0) {
                __builtin_amdgcn_global_load_lds((gbl_void*)(src[i] + (long)kt * ROWB), (lds_void*)(st + dst[i]), 16, 0, 0);
            } else {
                const uint4 v = *(const uint4*)(src[i] + (long)kt * ROWB);
                *(uint4*)(st + dst[i] + lane * 16) = v;
            }
        }
    };
    for (int s = 0; s < S - 1; ++s) issue(s);
    for (int kt = 0; kt < ksteps; ++kt) {
        if (MODE == 0) {
            if (S == 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PIECES) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (MFMA) {                                       // the 256x256 tile's fragment reads + MFMAs of one k-step (BK = 32 per 64 B)
            const char* st = smem + (kt % S) * STAGE;
#pragma unroll
            for (int kk = 0; kk < ROWB / 64; ++kk) {
                h8 a[8], b[4];
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = *(const h8*)(st + (((wave >> 2) * 128 + i * 16 + (lane & 15)) * LPR + (((lane >> 4) + kk * 4 + ((lane >> 2) & 3)) % LPR)) * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j] = *(const h8*)(st + ((256 + (wave & 3) * 64 + j * 16 + (lane & 15)) * LPR + (((lane >> 4) + kk * 4 + ((lane >> 2) & 3)) % LPR)) * 16);
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i * 4 + j], 0, 0, 0);
            }
        } else {
            keep += *(const float*)(smem + (kt % S) * STAGE + tid * 4);
        }
        if (kt + S - 1 < ksteps) issue(kt + S - 1);
        else if (MODE == 0) asm volatile("s_nop 0");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MFMA) for (int i = 0; i < 32; ++i) keep += acc[i][0] + acc[i][3];
    if (keep == 123.456f) sink[blockIdx.x] = keep;
}

template <int ROWB, int MODE, int MFMA>
static void run(const char* name, const char* A, const char* W, long rowstride, long kbytes, int a_tiles, int blocks, float* sink) {
    const int ksteps = (int)(kbytes / ROWB);
    const int lds = 131072;
    CK(hipFuncSetAttribute((const void*)fill_kernel<ROWB, MODE, MFMA>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) fill_kernel<ROWB, MODE, MFMA><<<blocks, 512, lds>>>(A, W, rowstride, ksteps, a_tiles, sink);
    CK(hipEventRecord(e0));
    const int it = 5;
    for (int i = 0; i < it; ++i) fill_kernel<ROWB, MODE, MFMA><<<blocks, 512, lds>>>(A, W, rowstride, ksteps, a_tiles, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
    const double bytes = (double)blocks * ksteps * 512 * ROWB;
    const double flops = MFMA ? (double)blocks * 2.0 * 256 * 256 * (kbytes / 2) : 0;
    printf("%-44s blocks %5d  %8.1f us  fill %7.2f TB/s  = %6.1f GB/s per CU   %s%.0f TF/s\n", name, blocks, ms * 1e3, bytes / ms / 1e9,
           bytes / ms / 1e6 / 256, MFMA ? "mfma " : "", flops / ms / 1e9);
}

int main() {
    const long K = 2880, rowstride = K * 2;                // a 320-channel 3x3 conv's K
    const int a_tiles_big = 128;                           // 32768 rows
    char *A, *W; float* sink;
    CK(hipMalloc(&A, (size_t)a_tiles_big * 256 * rowstride));
    CK(hipMalloc(&W, (size_t)256 * rowstride));
    CK(hipMalloc(&sink, 1 << 20));
    CK(hipMemset(A, 0, (size_t)a_tiles_big * 256 * rowstride));
    CK(hipMemset(W, 0, (size_t)256 * rowstride));
    for (int blocks : {256, 1024}) {
        run<64, 0, 0>("lds-dma  rowb 64  A streamed", A, W, rowstride, rowstride, a_tiles_big, blocks, sink);
        run<64, 0, 0>("lds-dma  rowb 64  A shared (L2 hits)", A, W, rowstride, rowstride, 1, blocks, sink);
        run<128, 0, 0>("lds-dma  rowb 128 A streamed", A, W, rowstride, rowstride, a_tiles_big, blocks, sink);
        run<128, 0, 0>("lds-dma  rowb 128 A shared (L2 hits)", A, W, rowstride, rowstride, 1, blocks, sink);
        run<64, 1, 0>("vgpr+ds_write rowb 64  A streamed", A, W, rowstride, rowstride, a_tiles_big, blocks, sink);
        run<64, 1, 0>("vgpr+ds_write rowb 64  A shared", A, W, rowstride, rowstride, 1, blocks, sink);
        run<128, 1, 0>("vgpr+ds_write rowb 128 A streamed", A, W, rowstride, rowstride, a_tiles_big, blocks, sink);
        run<64, 0, 1>("lds-dma rowb 64 + frag reads + MFMA, streamed", A, W, rowstride, rowstride, a_tiles_big, blocks, sink);
        run<64, 0, 1>("lds-dma rowb 64 + frag reads + MFMA, shared", A, W, rowstride, rowstride, 1, blocks, sink);
        run<128, 0, 1>("lds-dma rowb 128 + frag reads + MFMA, streamed", A, W, rowstride, rowstride, a_tiles_big, blocks, sink);
        run<128, 0, 1>("lds-dma rowb 128 + frag reads + MFMA, shared", A, W, rowstride, rowstride, 1, blocks, sink);
    }
    return 0;
}
