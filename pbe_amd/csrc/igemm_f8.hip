// fp8 (OCP e4m3) operand tiles of the implicit-GEMM kernel (igemm_kernel.h).
#include "igemm_kernel.h"

// fp8 operands: a subset of the dense tiles (no split-K: the slab reduce does not carry the operand scales)
void pbe_dispatch_f8(IGemmP p, int batch, hipStream_t s, int want_cfg) {
    p.ws = nullptr;                                   // no workspace: splits_for() returns 1 for every tile
    const Plan pl = plan_igemm(p, batch, 0, want_cfg, 0);
    p.splits = 1;
    switch (pl.cfg) {
        case 3: case 0: case 1: case 2: case 15: launch_cfg<128, 128, 2, 2, 2, 0, 0, false, true>(p, batch, s); break;
        case 4: case 16: launch_cfg<128, 64, 2, 2, 2, 0, 0, false, true>(p, batch, s); break;
        case 5: case 6: case 17: launch_cfg<64, 64, 2, 2, 2, 0, 0, false, true>(p, batch, s); break;
        case 8: case 7: launch_cfg<128, 320, 2, 4, 2, 0, 0, false, true>(p, batch, s); break;
        default: launch_cfg<128, 160, 2, 2, 2, 0, 0, false, true>(p, batch, s); break;
    }
}

