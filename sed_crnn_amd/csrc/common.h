// common.h — shared helpers for libsedcrnn (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/sedcrnn.h"

#define SED_EINVAL (-1)
#define SED_EUNSUPPORTED (-2)

void sed_set_error(const char* fmt, ...);

#define SED_REQUIRE(cond, ...)                     \
    do {                                           \
        if (!(cond)) {                             \
            sed_set_error(__VA_ARGS__);            \
            return SED_EINVAL;                     \
        }                                          \
    } while (0)

#define SED_LAUNCH_CHECK(name)                                                      \
    do {                                                                            \
        hipError_t e__ = hipGetLastError();                                         \
        if (e__ != hipSuccess) {                                                    \
            sed_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));   \
            return (int)e__;                                                        \
        }                                                                           \
    } while (0)

#define SED_TRY(expr)            \
    do {                         \
        int r__ = (expr);        \
        if (r__ != 0) return r__; \
    } while (0)

// measurement-only launch bracket (see sed_prof_enable); no-op unless the tag is enabled
extern unsigned g_sed_prof_mask;
void sed_prof_begin(int tag, hipStream_t s, double units);
void sed_prof_end(int tag, hipStream_t s);
struct SedProfScope {
    int tag; hipStream_t s; bool on;
    SedProfScope(int tag_, hipStream_t s_, double units) : tag(tag_), s(s_), on((g_sed_prof_mask >> tag_) & 1u) {
        if (on) sed_prof_begin(tag, s, units);
    }
    ~SedProfScope() { if (on) sed_prof_end(tag, s); }
};

// internal (conv.hip): the fp32 weight packing of up to SED_MAX_CONV layers in ONE launch (the forward packs every conv layer
// of a step up front instead of once per layer on the critical chain); same layouts as sed_conv3x3_pack_weights
// wino_f / wino_d (may be NULL): per layer, wf / wd receives the Winograd-transformed packing of wino.hip instead
int sed_internal_conv_pack_multi(int n, const float* const* w, float* const* wf, float* const* wd, const int* Cout, const int* Cin,
                                 const int* wino_f, const int* wino_d, void* stream);

// internal (conv.hip): the inference form of the packing launch — every layer's fragments, the BatchNorm coefficients on running
// statistics, BatchNorm folded into weights + bias where fold[l], and (perm_src) the GRU input weights re-ordered to
// channels-last feature columns — see conv_pack_w_multi_k
int sed_internal_conv_pack_eval(int n, const float* const* w, const float* const* bias, const float* const* gamma,
                                const float* const* beta, const float* const* rm, const float* const* rv, float eps,
                                float* const* wf, float* const* scale, float* const* shift, float* const* bias_folded, const int* fold,
                                const int* wino, const int* Cout, const int* Cin, const float* perm_src0, const float* perm_src1, float* perm_dst,
                                int perm_rows, int perm_C, int perm_Fp, void* stream);

// internal (conv.hip): sed_conv3x3_wgrad_ex whose exact-fp32 MFMA kernels add 1 to *arrive (agent scope) as each workgroup
// starts; _workgroups: how many that will be (0: this shape's kernel does not announce itself)
int sed_internal_conv3x3_wgrad(const float* x, int x_is_nchw, const float* dy, float* dw, void* workspace,
                               int B, int Cin, int F, int T, int Cout, int mode, unsigned* arrive, void* stream);
int sed_internal_conv3x3_wgrad_workgroups(int B, int Cin, int F, int T, int Cout, int x_is_nchw, int mode);

// internal (misc.hip): a one-wave kernel on `stream` that returns once *counter >= target — i.e. once that many workgroups of a
// kernel on ANOTHER stream are resident — or after timeout_us (a deadlock guard, not the mechanism).  What is enqueued behind
// it on `stream` then moves in BESIDE that kernel instead of taking its CUs (see sed_net_backward).
int sed_internal_stream_gate(const unsigned* counter, unsigned target, int timeout_us, void* stream);

static inline hipStream_t as_stream(void* s) { return (hipStream_t)s; }
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifdef __HIPCC__
// ── counter-based dropout hash: same (seed, idx) -> same keep decision in fwd and bwd ──
__device__ __forceinline__ uint32_t sed_fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
// uniform in [0,1) with 24 bits
__device__ __forceinline__ float sed_uniform(uint64_t seed, uint64_t idx) {
    uint32_t lo = (uint32_t)idx, hi = (uint32_t)(idx >> 32);
    uint32_t h = sed_fmix32(lo ^ (uint32_t)seed);
    h = sed_fmix32(h + (uint32_t)(seed >> 32) + hi * 0x9E3779B1u);
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}
// multiplier applied to a kept/dropped element (0 or 1/(1-p)); p==0 -> 1
__device__ __forceinline__ float sed_drop_mult(uint64_t seed, uint64_t idx, float p, float inv_keep) {
    return (sed_uniform(seed, idx) >= p) ? inv_keep : 0.0f;
}

// n / d for 0 <= n < 2^20, 1 <= d <= 4096 with inv = 1.0f / d: (n + 0.5) / d is at least 0.5/d away from an
// integer, far more than the two float roundings, so the truncation is exact (a v_cvt/v_fma/v_cvt instead of ~30
// instructions of integer division in per-element address math).
__device__ __forceinline__ int sed_fdiv(int n, float inv) { return (int)(((float)n + 0.5f) * inv); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
#endif
