// Halo-resident 3x3 conv (MODE 2) tiles of the implicit-GEMM kernel (igemm_kernel.h).
#include "igemm_kernel.h"

void pbe_launch_halo(int cfg, IGemmP p, int batch, hipStream_t s) {
    switch (cfg) {
        case 10: launch_cfg<256, 160, 4, 2, 3, 2, 392>(p, batch, s); break;
        case 11: launch_cfg<128, 160, 4, 2, 3, 2, 264>(p, batch, s); break;
        case 12: launch_cfg<128, 320, 2, 4, 2, 2, 264>(p, batch, s); break;
        case 13: launch_cfg<256, 128, 4, 2, 3, 2, 392>(p, batch, s); break;
        default: launch_cfg<128, 128, 4, 2, 3, 2, 392>(p, batch, s); break;
    }
}
