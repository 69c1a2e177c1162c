// conv_shared.h — declarations shared by the direct (conv.hip) and the Winograd (wino.hip) 3x3 kernels.
#pragma once
#include "common.h"

// The BatchNorm-backward epilogue of a data-gradient launch (see conv3x3_mfma_fwd2_k in conv.hip for the derivation).
struct ConvBnRed {
    const float* pooled;     // [B][T][F][Cout] forward output of the block whose BatchNorm is being differentiated
    const float* gamma;      // [Cout]
    const float* beta;       // [Cout]
    const float* ybelow;     // != NULL: the block below stores its conv output — channels whose xhat cannot be recovered from the pooled
    const float* mean;       // output (|gamma| < |beta| / 64) then contribute 0 here and are recomputed by sed_bn_bwd_finalize_small_gamma
    const float* rstd;
    float keep, inv_keep;    // 1 - p, 1 / (1 - p)
    int pf, pt, Fy, Ty;      // pool and the extents of ybelow
    // RG (the block below is the recomputed 1-channel first block, pool (1,2)): also its weight-gradient sums, see the kernel
    const float* x1;         // the network input [B][RGC][Fy][Ty]
    const unsigned char* bits;   // arg-max bits of the block below: [B][T][F][Cout/4] bytes, bit e = channel 4q+e took the second time row
    float* rgp;              // out: [rows][Cout][1 + 9 RGC] = (sum g, R_k) per workgroup
    float invXT;             // 1 / (2 TT + 2)
};

// wino.hip: F(2x2, 3x3) forward / data gradient of the 128-channel blocks (see the file header)
int sed_internal_wino_rows(int B, int Cin, int F, int T, int Cout);
int sed_internal_wino_launch(const float* x, const float* uq, const float* bias, float* y, float* stat, const ConvBnRed* br,
                             int rgc, int B, int Cin, int F, int T, int Cout, hipStream_t s);
