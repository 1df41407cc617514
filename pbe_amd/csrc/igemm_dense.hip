// Dense (MODE 0) tiles of the implicit-GEMM kernel (igemm_kernel.h).
#include "igemm_kernel.h"

void pbe_dispatch_dense(IGemmP p, int batch, hipStream_t s, size_t ws_bytes, int want_cfg) {
    constexpr int MODE = 0;
    const Plan pl = plan_igemm(p, batch, ws_bytes, want_cfg, MODE);
    p.splits = pl.splits;
#ifdef PBE_STAMPS
    p.stamps = g_pbe_stamps;
#endif
    switch (pl.cfg) {
        case 0: launch_cfg<256, 256, 2, 4, 2, MODE>(p, batch, s); break;
        case 1: launch_cfg<256, 128, 4, 2, 3, MODE>(p, batch, s); break;
        case 2: launch_cfg<128, 256, 2, 4, 3, MODE>(p, batch, s); break;
        case 3: launch_cfg<128, 128, 2, 2, 2, MODE>(p, batch, s); break;
        case 4: launch_cfg<128, 64, 2, 2, 2, MODE>(p, batch, s); break;
        case 5: launch_cfg<64, 128, 2, 2, 2, MODE>(p, batch, s); break;
        case 7: launch_cfg<256, 320, 2, 4, 2, MODE>(p, batch, s); break;
        case 8: launch_cfg<128, 320, 2, 4, 2, MODE>(p, batch, s); break;
        case 9: launch_cfg<128, 160, 2, 2, 2, MODE>(p, batch, s); break;
        case 15: launch_cfg<128, 128, 2, 2, 4, MODE>(p, batch, s); break;
        case 16: launch_cfg<128, 64, 2, 2, 4, MODE>(p, batch, s); break;
        case 17: launch_cfg<64, 64, 2, 2, 4, MODE>(p, batch, s); break;
        case 18: launch_cfg<128, 160, 2, 2, 4, MODE>(p, batch, s); break;
        case 21: launch_cfg<256, 160, 4, 2, 3, MODE>(p, batch, s); break;
        default: launch_cfg<64, 64, 2, 2, 2, MODE>(p, batch, s); break;
    }
}
