// Micro-benchmark: what does ISSUING one LDS-DMA piece (global_load_lds_dwordx4, 64 lanes x 16 B = 8 rows x 128 B) cost the issuing wave
// on gfx950, by address form and by what the rest of the CU is doing?
// Build + run (GPU box):  hipcc -O3 --offload-arch=gfx950 tools/ubench_dma_issue.hip -o gpurun_out/ubench_dma_issue && gpurun_out/ubench_dma_issue
// Behind igemm.hip's main loop: tools/phase_stamps.py shows ~70-170 wave cycles per piece wherever the pieces are placed; this asks
// whether the address form (64-bit VGPR address + zero-source select, as compiled / lean pointer bump / SGPR base + 32-bit VGPR
// offset), the number of waves issuing together, or an MFMA stream in the SIMD partner sets that price.
//   V 0: as igemm.hip compiles it: pointer = row base + k (v_lshl_add_u64), select against a zero block, LDS address through a VGPR
//   V 1: lean: per-piece pointer registers bumped by 128 B, LDS address in an SGPR
//   V 2: SGPR base + 32-bit VGPR offset (inline asm), LDS address in an SGPR
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

template <int V, int P, int NLOAD, int NMFMA, bool SYNC, int DEPTH = 0>
__global__ __launch_bounds__((NLOAD + NMFMA) * 64) void dma_kernel(const char* __restrict__ src, int iters, int ok_flag, unsigned long long* __restrict__ out, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane >> 3, gch = (lane & 7) ^ lrow;
    if (wave >= NLOAD) {                                   // SIMD partners: a bare MFMA stream for the loaders' whole lifetime
        f4 acc[8];
        h8 a = {1, 1, 1, 1, 1, 1, 1, 1}, b = {1, 1, 1, 1, 1, 1, 1, 1};
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = f4{0, 0, 0, 0};
        for (int it = 0; it < iters * 12; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
            if (SYNC && it % 12 == 11) __builtin_amdgcn_s_barrier();
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += acc[i][0];
        if (s == 12345.f) sink[0] = s;
        return;
    }
    const char* row[P];
    const char* ptr[P];
    unsigned voff[P];
    bool ok[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const long r = ((long)(wave * P + i) * 8 + lrow);
        row[i] = src + r * 4096;
        ptr[i] = row[i] + gch * 16;
        voff[i] = (unsigned)(r * 4096 + gch * 16);
        ok[i] = ok_flag != 0;                             // run-time true: keeps the zero-source select in the code
    }
    const unsigned lds0 = (unsigned)(size_t)(lds_void*)smem;
    unsigned long long t_issue = 0, t_land = 0;
    for (int it = 0; it < iters; ++it) {
        const int kt = it & 31;                            // 32 k-tiles of 128 B = one 4 KiB row, then again (L2 hits)
        const int slot = it & 3;
        if (SYNC) __builtin_amdgcn_s_barrier();
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        if (V == 0) {
            const int k = kt * 64 + gch * 8;
#pragma unroll
            for (int i = 0; i < P; ++i) {
                const _Float16* s = (const _Float16*)row[i] + k;
                s = ok[i] ? s : (const _Float16*)g_zero16;
                __builtin_amdgcn_global_load_lds((gbl_void*)s, (lds_void*)(smem + slot * 32768 + ((tid >> 6) * P + i) * 1024), 16, 0, 0);
            }
        } else if (V == 1) {
#pragma unroll
            for (int i = 0; i < P; ++i) {
                __builtin_amdgcn_global_load_lds((gbl_void*)(ptr[i] + kt * 128), (lds_void*)(smem + slot * 32768 + (wave * P + i) * 1024), 16, 0, 0);
            }
        } else {
            const unsigned long long sb = (unsigned long long)src + kt * 128;
#pragma unroll
            for (int i = 0; i < P; ++i) {
                const unsigned l = lds0 + slot * 32768 + (wave * P + i) * 1024;
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(l), "v"(voff[i]), "s"(sb) : "memory");
            }
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH * P) : "memory");     // DEPTH older bursts stay in flight
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        t_issue += t1 - t0;
        t_land += t2 - t1;
    }
    if (lane == 0) {
        out[((long)blockIdx.x * NLOAD + wave) * 2 + 0] = t_issue;
        out[((long)blockIdx.x * NLOAD + wave) * 2 + 1] = t_land;
    }
    if (smem[tid * 16] == 77 && ok_flag == 2) sink[1] = 1.f;
}

template <int V, int P, int NLOAD, int NMFMA, bool SYNC, int DEPTH = 0>
static void run(const char* name, const char* src, unsigned long long* out, float* sink) {
    const int iters = 400, blocks = 256;
    auto k = dma_kernel<V, P, NLOAD, NMFMA, SYNC, DEPTH>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k, dim3(blocks), dim3((NLOAD + NMFMA) * 64), 128 * 1024, 0, src, iters, 1, out, sink);
        CK(hipDeviceSynchronize());
    }
    std::vector<unsigned long long> h(blocks * NLOAD * 2);
    CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
    double ti = 0, tl = 0;
    for (int i = 0; i < blocks * NLOAD; ++i) { ti += h[2 * i]; tl += h[2 * i + 1]; }
    ti /= (double)blocks * NLOAD * iters;
    tl /= (double)blocks * NLOAD * iters;
    printf("%-64s P=%d loaders=%d mfma-waves=%d %s in flight <= %2d per wave: issue %7.1f cycles per burst = %6.1f per piece; then %7.1f waiting\n", name, P, NLOAD, NMFMA,
           SYNC ? "lockstep" : "free    ", (DEPTH + 1) * P, ti, ti / P, tl);
}

int main() {
    char* src;
    unsigned long long* out;
    float* sink;
    CK(hipMalloc(&src, 8 << 20));
    CK(hipMemset(src, 1, 8 << 20));
    CK(hipMalloc(&out, 256 * 8 * 2 * 8));
    CK(hipMalloc(&sink, 16));
    run<0, 4, 4, 0, false>("as compiled (64-bit VGPR address, zero select, VGPR LDS address)", src, out, sink);
    run<1, 4, 4, 0, false>("lean (pointer registers, SGPR LDS address)", src, out, sink);
    run<2, 4, 4, 0, false>("SGPR base + 32-bit VGPR offset", src, out, sink);
    run<0, 8, 4, 0, false>("as compiled", src, out, sink);
    run<1, 8, 4, 0, false>("lean", src, out, sink);
    run<2, 8, 4, 0, false>("SGPR base + 32-bit VGPR offset", src, out, sink);
    run<0, 4, 8, 0, false>("as compiled", src, out, sink);
    run<1, 4, 8, 0, false>("lean", src, out, sink);
    run<2, 4, 8, 0, false>("SGPR base + 32-bit VGPR offset", src, out, sink);
    run<0, 4, 8, 0, true>("as compiled", src, out, sink);
    run<1, 4, 8, 0, true>("lean", src, out, sink);
    run<2, 4, 8, 0, true>("SGPR base + 32-bit VGPR offset", src, out, sink);
    run<0, 4, 4, 4, false>("as compiled, MFMA stream in the SIMD partner", src, out, sink);
    run<1, 4, 4, 4, false>("lean, MFMA stream in the SIMD partner", src, out, sink);
    run<2, 4, 4, 4, false>("SGPR base + 32-bit offset, MFMA stream in the SIMD partner", src, out, sink);
    run<0, 1, 4, 4, false>("as compiled, MFMA stream in the SIMD partner", src, out, sink);
    run<1, 1, 4, 4, false>("lean, MFMA stream in the SIMD partner", src, out, sink);
    run<1, 1, 4, 0, false>("lean", src, out, sink);
    run<1, 1, 8, 0, true>("lean", src, out, sink);
    printf("-- outstanding depth (lean form)\n");
    run<1, 4, 8, 0, false, 0>("lean", src, out, sink);
    run<1, 4, 8, 0, false, 1>("lean", src, out, sink);
    run<1, 4, 8, 0, false, 2>("lean", src, out, sink);
    run<1, 4, 8, 0, false, 3>("lean", src, out, sink);
    run<1, 8, 8, 0, false, 1>("lean", src, out, sink);
    run<1, 8, 8, 0, false, 2>("lean", src, out, sink);
    run<1, 8, 8, 0, false, 3>("lean", src, out, sink);
    run<1, 4, 4, 4, false, 2>("lean, MFMA partner", src, out, sink);
    run<1, 8, 4, 4, false, 2>("lean, MFMA partner", src, out, sink);
    run<1, 8, 4, 4, false, 3>("lean, MFMA partner", src, out, sink);
    run<0, 8, 8, 0, false, 2>("as compiled", src, out, sink);
    run<0, 8, 8, 0, true, 2>("as compiled", src, out, sink);
    return 0;
}
