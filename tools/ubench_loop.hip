// Sandbox for the main loop of the halo-resident conv tile (256 pixels x 160 channels, 8 waves, wave tile 64 x 80):
// per k-tile every wave does 18 ds_read_b128 (2 k-steps x (4 activation + 5 weight fragments)), 40 MFMAs and issues its
// share of the weight tile's LDS-DMA pieces (3 of 20; the halo's 7 pieces per 9 k-tiles are modelled as a 4th piece).
// Same instruction mix and LDS / L2 traffic as igemm_kernel<256,160,4,2,2,3,392>, no arithmetic meaning: it times LOOP STRUCTURES
// in seconds of compile time instead of the library's two minutes.
// Build + run (GPU box):  hipcc -O3 --offload-arch=gfx950 tools/ubench_loop.hip -o gpurun_out/ubench_loop && gpurun_out/ubench_loop
//   STRUCT 0: ping-pong (waves 0-3 / 4-7 one barrier apart, [reads | MFMAs], two barriers per k-tile)
//   STRUCT 1: plain (one barrier per k-tile, every wave: reads, then MFMAs)
//   STRUCT 2: plain, software-pipelined inside the wave: k-step 1's fragments are read under k-step 0's MFMAs, the next tile's
//             k-step 0 fragments under k-step 1's MFMAs (barrier between the two halves)
//   PLACE  0: DMA pieces at the head of the read phase   1: after the reads are issued   2: spread between the MFMAs
//          3: after the MFMAs                            4: none (upper bound: operands resident)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __attribute__((aligned(16))) unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};
constexpr int TM = 4, TN = 5, NP = 4, S = 3, WB = 160 * 128, HALO = 392 * 128;

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int STRUCT, int PLACE, int SRC = 0, int FORM = 0>
__global__ __launch_bounds__(512) void loop_kernel(const char* __restrict__ src, int iters, unsigned long long* __restrict__ out, float* sink, int flag) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;
    const int lrow = lane >> 3, gch = (lane & 7) ^ lrow, fr = lane & 15, fq = lane >> 4;
    char* abuf = smem;
    char* wring = smem + 2 * HALO;
    const char* ptr[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        // SRC 0: every piece from a 1 MiB panel shared by the XCD's workgroups (L2 hits).  SRC 1: the 4th piece (the halo's) from a
        // stream of the workgroup's own that is never re-read (beyond L2).  SRC 2: all four pieces from such streams.
        // SRC 3: the weight pieces from a panel the XCD's workgroups share but walk ONCE (every k-tile a first touch of its lines in
        // that L2, all workgroups asking together - the real weight stream), the 4th piece from a stream of the workgroup's own
        const bool own = SRC == 2 || ((SRC == 1 || SRC == 3) && i == 3);
        if (SRC == 3 && i < 3) { ptr[i] = src + (16L << 20) + (256L * 32 * 8 * 65536) + ((long)((blockIdx.x & 7) * 32 + wave * NP + i) * 8 + lrow) * 65536 + gch * 16; continue; }
        ptr[i] = own ? src + (16L << 20) + ((long)(blockIdx.x * 32 + wave * NP + i) * 8 + lrow) * 65536 + gch * 16
                     : src + ((long)((blockIdx.x & 7) * 32 + wave * NP + i) * 8 + lrow) * 4096 + gch * 16;
    }
    const int kmask = SRC == 0 ? 31 : 511;
    f4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f4{0, 0, 0, 0};
    h8 fa[2][TM], fw[2][TN];
    const int rsw = (fq ^ (fr & 7)) << 4;
    const int a_rd = (wm * 64 + fr) * 128, w_rd = (wn * 80 + fr) * 128;
    // FORM 1: the address form igemm.hip compiles to: row base + k (64-bit shift-add), run-time select against a zero block,
    // LDS address derived from threadIdx (VGPR -> v_readfirstlane -> m0), a uniform run-time branch around the piece
    // FORM 2: FORM 1 without the branch          FORM 3: FORM 1 without the zero-block select
    const int vwave = tid >> 6;
    const bool okf = flag != 0;
    auto piece = [&](int it, int i) {                       // piece i of the k-tile issued in iteration it (ring slot it % S)
        if (FORM != 0) {
            if (FORM != 2 && flag == 7) return;
            char* dst = (i < 3) ? wring + (it % S) * WB + min(vwave + 8 * i, 19) * 1024 : abuf + ((it / 9) & 1) * HALO + min(vwave + 8 * (it % 7), 48) * 1024;
            const _Float16* sp = (const _Float16*)(ptr[i] - gch * 16) + ((it & 31) * 64 + gch * 8);
            if (FORM != 3) sp = okf ? sp : (const _Float16*)g_zero16;
            __builtin_amdgcn_global_load_lds((gbl_void*)sp, (lds_void*)dst, 16, 0, 0);
            return;
        }
        char* dst = (i < 3) ? wring + (it % S) * WB + min(wave + 8 * i, 19) * 1024 : abuf + ((it / 9) & 1) * HALO + min(wave + 8 * (it % 7), 48) * 1024;
        __builtin_amdgcn_global_load_lds((gbl_void*)(ptr[i] + (it & (i == 3 || SRC >= 2 ? kmask : 31)) * 128), (lds_void*)dst, 16, 0, 0);
    };
    auto reads = [&](int it, int ks) {
        const char* ab = abuf + ((it / 9) & 1) * HALO + (it % 9) * 128;
        const char* sw = wring + (it % S) * WB;
#pragma unroll
        for (int j = 0; j < TM; ++j) fa[ks][j] = *reinterpret_cast<const h8*>(ab + a_rd + (rsw ^ (ks * 64)) + j * 16 * 128);
#pragma unroll
        for (int i = 0; i < TN; ++i) fw[ks][i] = *reinterpret_cast<const h8*>(sw + w_rd + (rsw ^ (ks * 64)) + i * 16 * 128);
    };
    auto mfma_ks = [&](int ks, int it, bool spread, int qbase = -1, int qtotal = 40) {
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[ks][i], fa[ks][j], acc[i][j], 0, 0, 0);
                if (spread) {
                    const int q = (qbase < 0 ? ks * TN * TM : qbase) + i * TM + j;
#pragma unroll
                    for (int pc = (q * NP) / qtotal; pc < ((q + 1) * NP) / qtotal; ++pc) {
                        __builtin_amdgcn_sched_barrier(0);
                        piece(it, pc);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
    };
    auto burst = [&](int it) {
#pragma unroll
        for (int i = 0; i < NP; ++i) piece(it, i);
    };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (STRUCT == 0) {
        const int grp = wave >> 2;
        if (grp == 1) __builtin_amdgcn_s_barrier();
        for (int it = 0; it < iters; ++it) {
            __builtin_amdgcn_sched_barrier(0);
            if (PLACE == 0) burst(it + 2);
            reads(it, 0);
            reads(it, 1);
            if (PLACE == 1) burst(it + 2);
            if (grp == 1 && PLACE != 4) wait_vm<NP>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            mfma_ks(0, it + 2, PLACE == 2);
            mfma_ks(1, it + 2, PLACE == 2);
            if (PLACE == 3) burst(it + 2);
            if (grp == 0 && PLACE != 4) wait_vm<NP>();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();
    } else if (STRUCT == 1) {
        for (int it = 0; it < iters; ++it) {
            if (PLACE != 4) wait_vm<NP>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (PLACE == 0) burst(it + 2);
            reads(it, 0);
            reads(it, 1);
            __builtin_amdgcn_sched_barrier(0);
            if (PLACE == 1) burst(it + 2);
            __builtin_amdgcn_sched_barrier(0);
            mfma_ks(0, it + 2, PLACE == 2);
            mfma_ks(1, it + 2, PLACE == 2);
            if (PLACE == 3) burst(it + 2);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        reads(0, 0);
        for (int it = 0; it < iters; ++it) {
            __builtin_amdgcn_sched_barrier(0);
            reads(it, 1);                                  // k-step 1's fragments land under k-step 0's MFMAs
            if (PLACE == 0 || PLACE == 1) burst(it + 2);
            __builtin_amdgcn_sched_barrier(0);
            mfma_ks(0, it + 2, false);
            __builtin_amdgcn_sched_barrier(0);
            if (PLACE == 0 || PLACE == 1) wait_vm<NP>();   // tile it + 1 landed (the burst of this iteration may stay in flight)
            else if (PLACE != 4) wait_vm<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            {   // the next tile's k-step 0 fragments under k-step 1's MFMAs: a second register set for them
                h8 na[TM], nw[TN];
                const char* ab = abuf + (((it + 1) / 9) & 1) * HALO + ((it + 1) % 9) * 128;
                const char* sw = wring + ((it + 1) % S) * WB;
#pragma unroll
                for (int j = 0; j < TM; ++j) na[j] = *reinterpret_cast<const h8*>(ab + a_rd + rsw + j * 16 * 128);
#pragma unroll
                for (int i = 0; i < TN; ++i) nw[i] = *reinterpret_cast<const h8*>(sw + w_rd + rsw + i * 16 * 128);
                if (PLACE == 3) burst(it + 2);
                __builtin_amdgcn_sched_barrier(0);
                mfma_ks(1, it + 2, PLACE == 2, 0, 20);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < TM; ++j) fa[0][j] = na[j];
#pragma unroll
                for (int i = 0; i < TN; ++i) fw[0][i] = nw[i];
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) s += acc[i][j][0] + acc[i][j][3];
    if (s == 12345.f) sink[0] = s;
    if (tid == 0) out[blockIdx.x] = t1 - t0;
}

// ---- which MFMA shape costs less energy per FLOP?  Same ping-pong loop on a 256 x 128 tile (wave tile 64 x 64: 16 ds_read_b128
// per k-tile either way), 2 weight pieces + 1 halo piece per wave and k-tile, operands from the shared L2-resident panel:
//   MF 0: 32 x v_mfma_f32_16x16x32_f16 per wave and k-tile      MF 1: 16 x v_mfma_f32_32x32x16_f16
// Under the power limit (random operands) the cycle counts are equal and the clock - TFLOP/s - tells the shapes apart.
typedef float f16v __attribute__((ext_vector_type(16)));
template <int MF>
__global__ __launch_bounds__(512) void shape_kernel(const char* __restrict__ src, int iters, unsigned long long* __restrict__ out, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int WBS = 128 * 128, NPS = 3;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2, grp = wave >> 2;
    const int lrow = lane >> 3, gch = (lane & 7) ^ lrow;
    char* abuf = smem;
    char* wring = smem + 2 * HALO;
    const char* ptr[NPS];
#pragma unroll
    for (int i = 0; i < NPS; ++i) ptr[i] = src + ((long)((blockIdx.x & 7) * 32 + wave * NPS + i) * 8 + lrow) * 4096 + gch * 16;
    auto burst = [&](int it) {
#pragma unroll
        for (int i = 0; i < NPS; ++i) {
            char* dst = (i < 2) ? wring + (it % S) * WBS + (wave * 2 + i) * 1024 : abuf + ((it / 9) & 1) * HALO + min(wave + 8 * (it % 7), 48) * 1024;
            __builtin_amdgcn_global_load_lds((gbl_void*)(ptr[i] + (it & 31) * 128), (lds_void*)dst, 16, 0, 0);
        }
    };
    h8 fa[8], fw[8];
    float s = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (grp == 1) __builtin_amdgcn_s_barrier();
    if (MF == 0) {
        f4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f4{0, 0, 0, 0};
        const int fr = lane & 15, fq = lane >> 4, rsw = (fq ^ (fr & 7)) << 4;
        const int a_rd = (wm * 64 + fr) * 128, w_rd = (wn * 64 + fr) * 128;
        for (int it = 0; it < iters; ++it) {
            const char* ab = abuf + ((it / 9) & 1) * HALO + (it % 9) * 128;
            const char* sw = wring + (it % S) * WBS;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    fa[ks * 4 + j] = *reinterpret_cast<const h8*>(ab + a_rd + (rsw ^ (ks * 64)) + j * 16 * 128);
                    fw[ks * 4 + j] = *reinterpret_cast<const h8*>(sw + w_rd + (rsw ^ (ks * 64)) + j * 16 * 128);
                }
            burst(it + 2);
            if (grp == 1) wait_vm<NPS>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[ks * 4 + i], fa[ks * 4 + j], acc[i][j], 0, 0, 0);
            if (grp == 0) wait_vm<NPS>();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
    } else {
        f16v acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const int fr = lane & 31, fq = lane >> 5;
        const int a_rd = (wm * 64 + fr) * 128, w_rd = (wn * 64 + fr) * 128;
        for (int it = 0; it < iters; ++it) {
            const char* ab = abuf + ((it / 9) & 1) * HALO + (it % 9) * 128;
            const char* sw = wring + (it % S) * WBS;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {            // k-step of 16: chunks 2 ks + (lane >> 5), swizzled by row & 7
                const int off = ((2 * ks + fq) ^ (fr & 7)) << 4;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    fa[ks * 2 + j] = *reinterpret_cast<const h8*>(ab + a_rd + off + j * 32 * 128);
                    fw[ks * 2 + j] = *reinterpret_cast<const h8*>(sw + w_rd + off + j * 32 * 128);
                }
            }
            burst(it + 2);
            if (grp == 1) wait_vm<NPS>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[ks * 2 + i], fa[ks * 2 + j], acc[i][j], 0, 0, 0);
            if (grp == 0) wait_vm<NPS>();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][15];
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (s == 12345.f) sink[0] = s;
    if (tid == 0) out[blockIdx.x] = t1 - t0;
}

template <int MF>
static void run_shape(const char* name, const char* src, unsigned long long* out, float* sink) {
    const int iters = 360, blocks = 256;
    auto k = shape_kernel<MF>;
    const int lds = 2 * HALO + S * 128 * 128;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    float ms = 0.f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(512), lds, 0, src, iters, out, sink);
        CK(hipEventRecord(b, 0));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, a, b));
    }
    std::vector<unsigned long long> h(blocks);
    CK(hipMemcpy(h.data(), out, blocks * 8, hipMemcpyDeviceToHost));
    double cyc = 0;
    for (int i = 0; i < blocks; ++i) cyc += h[i];
    cyc /= (double)blocks * iters;
    const double tf = 256.0 * iters * 2.0 * 256 * 128 * 64 / (ms * 1e-3) / 1e12;
    printf("%-86s %7.1f cycles per k-tile (MFMA work 1024), %6.1f us, %6.0f TFLOP/s, clock %.2f GHz\n", name, cyc, ms * 1e3, tf, cyc * iters / (ms * 1e3) / 1e3);
}

template <int STRUCT, int PLACE, int SRC = 0, int FORM = 0>
static void run(const char* name, const char* src, unsigned long long* out, float* sink) {
    const int iters = 360, blocks = 256;
    auto k = loop_kernel<STRUCT, PLACE, SRC, FORM>;
    const int lds = 2 * HALO + S * WB;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(512), lds, 0, src, iters, out, sink, 1);
        CK(hipEventRecord(b, 0));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, a, b));
    }
    std::vector<unsigned long long> h(blocks);
    CK(hipMemcpy(h.data(), out, blocks * 8, hipMemcpyDeviceToHost));
    double cyc = 0;
    for (int i = 0; i < blocks; ++i) cyc += h[i];
    cyc /= (double)blocks * iters;
    const double tf = 256.0 * iters * 2.0 * 256 * 160 * 64 / (ms * 1e-3) / 1e12;
    printf("%-86s %7.1f cycles per k-tile (MFMA work 1280), %6.1f us, %6.0f TFLOP/s\n", name, cyc, ms * 1e3, tf);
}

int main() {
    char* src;
    unsigned long long* out;
    float* sink;
    const size_t bytes = (16UL << 20) + 256UL * 32 * 8 * 65536 + 8UL * 32 * 8 * 65536;      // shared panels + one 64 KiB row stream per piece row
    CK(hipMalloc(&src, bytes));
    CK(hipMemset(src, 0, bytes));
    const bool random_data = getenv("UBENCH_RANDOM") != nullptr;
    if (random_data) {                                   // fp16 values in [-2, 2): the matrix pipe's power depends on the operands
        std::vector<unsigned short> h(bytes / 2);
        unsigned x = 12345u;
        for (size_t i = 0; i < h.size(); ++i) { x = x * 1664525u + 1013904223u; h[i] = (unsigned short)(((x >> 16) & 0x83ffu) | 0x3c00u); }
        CK(hipMemcpy(src, h.data(), bytes, hipMemcpyHostToDevice));
    }
    printf("operands: %s\n", random_data ? "random fp16" : "zeros");
    CK(hipMalloc(&out, 256 * 8));
    CK(hipMalloc(&sink, 16));
    run<0, 4>("ping-pong, no DMA (bound of the structure)", src, out, sink);
    run<0, 0>("ping-pong, DMA at the head of the read half", src, out, sink);
    run<0, 1>("ping-pong, DMA after the reads are issued", src, out, sink);
    run<0, 2>("ping-pong, DMA spread between the MFMAs", src, out, sink);
    run<0, 3>("ping-pong, DMA after the MFMAs", src, out, sink);
    run<1, 4>("plain, no DMA", src, out, sink);
    run<1, 0>("plain, DMA before the reads", src, out, sink);
    run<1, 1>("plain, DMA after the reads are issued", src, out, sink);
    run<1, 2>("plain, DMA spread between the MFMAs", src, out, sink);
    run<1, 3>("plain, DMA after the MFMAs", src, out, sink);
    run<2, 4>("pipelined in the wave, no DMA", src, out, sink);
    run<2, 0>("pipelined in the wave, DMA with k-step 1's reads", src, out, sink);
    run<2, 2>("pipelined in the wave, DMA spread between k-step 1's MFMAs", src, out, sink);
    run<2, 3>("pipelined in the wave, DMA with the next tile's reads", src, out, sink);
    printf("-- MFMA shape, 256 x 128 tile, wave tile 64 x 64\n");
    run_shape<0>("ping-pong, v_mfma_f32_16x16x32_f16", src, out, sink);
    run_shape<1>("ping-pong, v_mfma_f32_32x32x16_f16", src, out, sink);
    run_shape<0>("ping-pong, v_mfma_f32_16x16x32_f16", src, out, sink);
    run_shape<1>("ping-pong, v_mfma_f32_32x32x16_f16", src, out, sink);
    printf("-- address form of the pieces\n");
    run<0, 0, 0, 1>("ping-pong, DMA at the head of the read half, form as compiled", src, out, sink);
    run<0, 0, 0, 2>("ping-pong, DMA at the head of the read half, as compiled without the branch", src, out, sink);
    run<0, 0, 0, 3>("ping-pong, DMA at the head of the read half, as compiled without the zero select", src, out, sink);
    run<0, 1, 0, 1>("ping-pong, DMA after the reads are issued, form as compiled", src, out, sink);
    run<0, 2, 0, 1>("ping-pong, DMA spread between the MFMAs, form as compiled", src, out, sink);
    run<0, 2, 0, 3>("ping-pong, DMA spread between the MFMAs, as compiled without the zero select", src, out, sink);
    printf("-- 4th piece from a stream of the workgroup's own (beyond L2)\n");
    run<0, 0, 1>("ping-pong, DMA at the head of the read half", src, out, sink);
    run<0, 1, 1>("ping-pong, DMA after the reads are issued", src, out, sink);
    run<0, 2, 1>("ping-pong, DMA spread between the MFMAs", src, out, sink);
    run<0, 3, 1>("ping-pong, DMA after the MFMAs", src, out, sink);
    run<2, 0, 1>("pipelined in the wave, DMA with k-step 1's reads", src, out, sink);
    printf("-- weight pieces shared by the XCD but first-touch, 4th piece the workgroup's own\n");
    run<0, 0, 3>("ping-pong, DMA at the head of the read half", src, out, sink);
    run<0, 1, 3>("ping-pong, DMA after the reads are issued", src, out, sink);
    run<2, 0, 3>("pipelined in the wave, DMA with k-step 1's reads", src, out, sink);
    printf("-- all four pieces from streams of the workgroup's own\n");
    run<0, 0, 2>("ping-pong, DMA at the head of the read half", src, out, sink);
    run<0, 1, 2>("ping-pong, DMA after the reads are issued", src, out, sink);
    run<0, 2, 2>("ping-pong, DMA spread between the MFMAs", src, out, sink);
    run<2, 0, 2>("pipelined in the wave, DMA with k-step 1's reads", src, out, sink);
    return 0;
}
