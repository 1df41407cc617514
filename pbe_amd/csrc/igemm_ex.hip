// Extended-epilogue (EX) dense tiles of the implicit-GEMM kernel (igemm_kernel.h): the transformer block's GEMM chain.
#include "igemm_kernel.h"

// extended epilogue (LayerNorm fold / row statistics / column-range alpha / V^T tiles): the EX instantiations, never split-K
void pbe_dispatch_ex(IGemmP p, int batch, hipStream_t s, int want_cfg) {
    p.ws = nullptr;
    const Plan pl = plan_igemm(p, batch, 0, want_cfg, 0);
    p.splits = 1;
    switch (pl.cfg) {
        case 3: launch_cfg<128, 128, 2, 2, 2, 0, 0, false, false, true>(p, batch, s); break;
        case 4: launch_cfg<128, 64, 2, 2, 2, 0, 0, false, false, true>(p, batch, s); break;
        case 5: launch_cfg<64, 128, 2, 2, 2, 0, 0, false, false, true>(p, batch, s); break;
        case 8: launch_cfg<128, 320, 2, 4, 2, 0, 0, false, false, true>(p, batch, s); break;
        case 9: launch_cfg<128, 160, 2, 2, 2, 0, 0, false, false, true>(p, batch, s); break;
        case 15: launch_cfg<128, 128, 2, 2, 4, 0, 0, false, false, true>(p, batch, s); break;
        case 16: launch_cfg<128, 64, 2, 2, 4, 0, 0, false, false, true>(p, batch, s); break;
        case 17: launch_cfg<64, 64, 2, 2, 4, 0, 0, false, false, true>(p, batch, s); break;
        case 18: launch_cfg<128, 160, 2, 2, 4, 0, 0, false, false, true>(p, batch, s); break;
        default: launch_cfg<64, 64, 2, 2, 2, 0, 0, false, false, true>(p, batch, s); break;
    }
}

