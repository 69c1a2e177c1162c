// misc.hip — loss heads, sigmoid, fused Adam and the global grad-norm.
//
// Replaces aten::binary_cross_entropy_with_logits (reference sed.py:136,160), FocalBCELoss
// (crnn_lightning.py:27-35), aten::sigmoid (sed.py:139), torch.optim.Adam's step (sed.py:159;
// crnn_lightning.py:195-197) and clip_grad_norm_ (train_lightning.py:50).  All pure streaming passes.
#include "common.h"

// ───────────────────────── loss ─────────────────────────
// single block: n is B*T'*K (a few thousand), the reduction order is fixed.
__global__ __launch_bounds__(1024) void loss_fwd_bwd_k(const float* __restrict__ x, const float* __restrict__ t,
                                                       int n, int kind, float alpha, float gamma, float scale,
                                                       float* __restrict__ loss, float* __restrict__ dx,
                                                       float* __restrict__ probs) {
    __shared__ double red[16];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) {
        float xi = x[i], ti = t[i];
        float p = 1.f / (1.f + expf(-xi));
        float l, d;
        if (kind == 0) {
            l = fmaxf(xi, 0.f) - xi * ti + log1pf(expf(-fabsf(xi)));
            d = p - ti;
        } else {
            bool pos = (ti == 1.f);
            float pt = pos ? p : 1.f - p;
            float om = 1.f - pt;
            float lg = logf(pt + 1e-12f);
            float pw = powf(om, gamma);
            l = -alpha * pw * lg;
            // d l / d pt
            float dpw = (gamma == 0.f) ? 0.f : gamma * powf(om, gamma - 1.f);
            float dl_dpt = -alpha * (-dpw * lg + pw / (pt + 1e-12f));
            float dpt_dx = pos ? p * (1.f - p) : -p * (1.f - p);
            d = dl_dpt * dpt_dx;
        }
        acc += (double)l;
        if (dx) dx[i] = d * scale;
        if (probs) probs[i] = p;
    }
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0;
        for (int w = 0; w < 16; ++w) a += red[w];
        loss[0] = (float)(a * (double)scale);
    }
}

extern "C" int sed_loss_fwd_bwd(const float* logits, const float* targets, int n, int kind, float alpha, float gamma,
                                int reduction_mean, float* loss, float* dlogits, float* probs, void* stream) {
    SED_REQUIRE(logits && targets && loss && n > 0, "loss_fwd_bwd: bad arguments");
    SED_REQUIRE(kind == 0 || kind == 1, "loss_fwd_bwd: kind must be 0 (bce) or 1 (focal)");
    float scale = reduction_mean ? 1.f / (float)n : 1.f;
    loss_fwd_bwd_k<<<1, 1024, 0, as_stream(stream)>>>(logits, targets, n, kind, alpha, gamma, scale, loss, dlogits, probs);
    SED_LAUNCH_CHECK("loss_fwd_bwd");
    return 0;
}

__global__ void sigmoid_k(const float* __restrict__ x, float* __restrict__ y, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = 1.f / (1.f + expf(-x[i]));
}

extern "C" int sed_sigmoid(const float* x, float* y, int n, void* stream) {
    SED_REQUIRE(x && y && n > 0, "sigmoid: bad arguments");
    sigmoid_k<<<cdiv(n, 256), 256, 0, as_stream(stream)>>>(x, y, n);
    SED_LAUNCH_CHECK("sigmoid");
    return 0;
}

__global__ void scale_k(float* __restrict__ x, long n, float alpha) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] *= alpha;
}

extern "C" int sed_scale(float* x, long n, float alpha, void* stream) {
    SED_REQUIRE(x && n > 0, "scale: bad arguments");
    scale_k<<<cdiv(n, 256), 256, 0, as_stream(stream)>>>(x, n, alpha);
    SED_LAUNCH_CHECK("scale");
    return 0;
}

// ───────────────────────── grad norm + clip coefficient ─────────────────────────
#define SQN_BLOCKS 512
__global__ __launch_bounds__(256) void sqnorm_partial_k(const float* __restrict__ g, long n, double* __restrict__ part) {
    __shared__ double red[4];
    double a = 0.0;
    long n4 = n >> 2;
    const f32x4* g4 = (const f32x4*)g;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        f32x4 v = g4[i];
        a += (double)(v[0] * v[0] + v[1] * v[1]) + (double)(v[2] * v[2] + v[3] * v[3]);
    }
    if (blockIdx.x == 0)
        for (long i = (n4 << 2) + threadIdx.x; i < n; i += 256) a += (double)g[i] * g[i];
    a = wave_sum_d(a);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void sqnorm_final_k(const double* __restrict__ part, int nb, float max_norm, float* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double a = 0.0;
        for (int i = 0; i < nb; ++i) a += part[i];
        float nrm = (float)sqrt(a);
        out[0] = nrm;
        float c = 1.f;
        if (max_norm > 0.f) { c = max_norm / (nrm + 1e-6f); if (c > 1.f) c = 1.f; }
        out[1] = c;
    }
}

extern "C" size_t sed_sqnorm_workspace_bytes(long n) { (void)n; return SQN_BLOCKS * sizeof(double); }

extern "C" int sed_grad_norm_clip_coef(const float* g, long n, float max_norm, float* norm_out, void* workspace,
                                       void* stream) {
    SED_REQUIRE(g && norm_out && workspace && n > 0, "grad_norm_clip_coef: bad arguments");
    SED_REQUIRE(((uintptr_t)g & 15) == 0, "grad_norm_clip_coef: gradient arena must be 16-byte aligned");
    hipStream_t s = as_stream(stream);
    int nb = (int)((n / 4 + 255) / 256);
    if (nb > SQN_BLOCKS) nb = SQN_BLOCKS;
    if (nb < 1) nb = 1;
    sqnorm_partial_k<<<nb, 256, 0, s>>>(g, n, (double*)workspace);
    SED_LAUNCH_CHECK("sqnorm_partial");
    sqnorm_final_k<<<1, 64, 0, s>>>((const double*)workspace, nb, max_norm, norm_out);
    SED_LAUNCH_CHECK("sqnorm_final");
    return 0;
}

// ───────────────────────── Adam (coupled L2 weight decay) ─────────────────────────
__global__ __launch_bounds__(256) void adam_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                                              float wd, float bc1, float bc2_sqrt, const float* __restrict__ gscale,
                                              const uint64_t* __restrict__ step_dev) {
    if (step_dev) {                       // optimiser step kept on the device (graph replay): bias corrections in-kernel
        const double t = (double)step_dev[1];
        bc1 = (float)(1.0 - pow((double)b1, t));
        bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, t));
    }
    const float gs = gscale ? gscale[0] : 1.f;
    const float step = lr / bc1;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float pi = p[i];
        float gi = g[i] * gs + wd * pi;
        float mi = b1 * m[i] + (1.f - b1) * gi;
        float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - step * (mi / denom);
    }
}

extern "C" int sed_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                             float eps, float weight_decay, int step, const float* grad_scale, const uint64_t* step_state,
                             void* stream) {
    SED_REQUIRE(p && g && m && v && n > 0 && (step >= 1 || step_state), "adam_step: bad arguments");
    if (step < 1) step = 1;
    double bc1 = 1.0 - pow((double)beta1, (double)step);
    double bc2 = 1.0 - pow((double)beta2, (double)step);
    int nb = (int)((n + 255) / 256);
    if (nb > 4096) nb = 4096;
    SedProfScope prof(SED_K_ADAM, as_stream(stream), 28.0 * n);
    adam_k<<<nb, 256, 0, as_stream(stream)>>>(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, (float)bc1,
                                              (float)sqrt(bc2), grad_scale, step_state);
    SED_LAUNCH_CHECK("adam_step");
    return 0;
}

// device-resident step state {dropout salt, optimiser step}: advanced once per fit step by this one-thread kernel so that a
// captured hipGraph of the step draws fresh dropout masks and uses the right Adam bias correction on every replay
__global__ void step_advance_k(uint64_t* st) { st[0] += 1; st[1] += 1; }

// One lane polls an arrival counter that the workgroups of a kernel on another stream bump as they start (agent-scope atomic
// adds, conv.hip; an sc1 load here: the per-XCD L2s are not coherent).  Every path leaves the loop: the count is reached, or
// the s_memrealtime (100 MHz) deadline passes.
__global__ void stream_gate_k(const unsigned* __restrict__ counter, unsigned target, unsigned timeout_ticks) {
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (__builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)timeout_ticks) break;
        __builtin_amdgcn_s_sleep(8);
    }
}
int sed_internal_stream_gate(const unsigned* counter, unsigned target, int timeout_us, void* stream) {
    if (!counter || target == 0) return 0;
    if (timeout_us < 1) timeout_us = 1;
    if (timeout_us > 5000) timeout_us = 5000;
    stream_gate_k<<<1, 64, 0, as_stream(stream)>>>(counter, target, (unsigned)timeout_us * 100u);
    SED_LAUNCH_CHECK("stream_gate");
    return 0;
}

extern "C" int sed_step_advance(uint64_t* state2, void* stream) {
    SED_REQUIRE(state2, "step_advance: null pointer");
    step_advance_k<<<1, 1, 0, as_stream(stream)>>>(state2);
    SED_LAUNCH_CHECK("step_advance");
    return 0;
}
