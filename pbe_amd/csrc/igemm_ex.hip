// Extended-epilogue GEMMs (the transformer block's chain): plan, then the instantiation that carries exactly the requested features.
#include "igemm_kernel.h"

// (LayerNorm fold / row statistics / column-range alpha / V^T tiles; never split-K)
void pbe_dispatch_ex(IGemmP p, int batch, hipStream_t s, int want_cfg) {
    p.ws = nullptr;
    const Plan pl = plan_igemm(p, batch, 0, want_cfg, 0);
    p.splits = 1;
#ifdef PBE_STAMPS
    p.stamps = g_pbe_stamps;
#endif
    if (pl.cfg == 19 || pl.cfg == 20) { pbe_launch_astat(pl.cfg, p, s); return; }      // (plan_igemm admits them only where pbe_astat_ok holds)
    // (the ln / qkv instantiations fold unconditionally: they are only ever launched with ln_stat)
    const bool ln = p.ln_stat != nullptr, st = p.rstat != nullptr, vt = p.vt != nullptr || p.alpha_cols > 0;
    if (ln && !st && !vt) pbe_launch_ex_ln(pl.cfg, p, batch, s);
    else if (st && !ln && !vt) pbe_launch_ex_st(pl.cfg, p, batch, s);
    else if (ln && vt && !st) pbe_launch_ex_qkv(pl.cfg, p, batch, s);
    else pbe_launch_ex_all(pl.cfg, p, batch, s);
}
