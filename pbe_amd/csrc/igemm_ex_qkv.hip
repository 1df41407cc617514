// Extended-epilogue dense tiles of the implicit-GEMM kernel (igemm_kernel.h): LayerNorm folded in + q pre-scale + V^T tiles (the fused q | k | v projection).
#include "igemm_kernel.h"

void pbe_launch_ex_qkv(int cfg, IGemmP p, int batch, hipStream_t s) {
    constexpr int EXF = EX_LN | EX_VT;
    switch (cfg) {
        case 3: launch_cfg<128, 128, 2, 2, 2, 0, 0, false, false, EXF>(p, batch, s); break;
        case 4: launch_cfg<128, 64, 2, 2, 2, 0, 0, false, false, EXF>(p, batch, s); break;
        case 5: launch_cfg<64, 128, 2, 2, 2, 0, 0, false, false, EXF>(p, batch, s); break;
        case 8: launch_cfg<128, 320, 2, 4, 2, 0, 0, false, false, EXF>(p, batch, s); break;
        case 9: launch_cfg<128, 160, 2, 2, 2, 0, 0, false, false, EXF>(p, batch, s); break;
        case 15: launch_cfg<128, 128, 2, 2, 4, 0, 0, false, false, EXF>(p, batch, s); break;
        case 16: launch_cfg<128, 64, 2, 2, 4, 0, 0, false, false, EXF>(p, batch, s); break;
        case 17: launch_cfg<64, 64, 2, 2, 4, 0, 0, false, false, EXF>(p, batch, s); break;
        case 18: launch_cfg<128, 160, 2, 2, 4, 0, 0, false, false, EXF>(p, batch, s); break;
        default: launch_cfg<64, 64, 2, 2, 2, 0, 0, false, false, EXF>(p, batch, s); break;
    }
}
