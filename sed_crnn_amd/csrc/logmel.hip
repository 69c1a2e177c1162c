// logmel.hip — log-mel front end: framing + Hann + 2048-point FFT -> |.|^2 -> 40x1025 mel matmul -> log.
//
// Replaces librosa.stft / librosa.filters.mel / np.log as called by _mbe at reference feature.py:55-59
// (and, optionally fused, the StandardScaler of feature.py:127-129).  One workgroup per frame: the frame
// (8 KB of PCM, read twice across frames because of the 50 % hop) is staged in LDS, transformed in place
// with a radix-2 DIT FFT (11 stages, 4 butterflies per thread per stage), and only 40 floats per frame
// go back to HBM, so the kernel is bound by the PCM read.
#include "common.h"

#define LM_NFFT 2048
#define LM_LOG2 11

__device__ __forceinline__ int bitrev11(int i) { return (int)(__brev((unsigned)i) >> (32 - LM_LOG2)); }

__global__ __launch_bounds__(256) void logmel_k(const float* __restrict__ pcm, long n_samples,
                                                const float* __restrict__ window, const float* __restrict__ tw,
                                                const float* __restrict__ melfb, const float* __restrict__ mu,
                                                const float* __restrict__ inv_sigma, float* __restrict__ out,
                                                int hop, int n_mels, int pad_mode) {
    __shared__ float re[LM_NFFT], im[LM_NFFT];
    __shared__ float pw[LM_NFFT / 2 + 1];
    const int tid = threadIdx.x;
    const long frame = blockIdx.x;
    const long start = frame * hop - LM_NFFT / 2;
    for (int i = tid; i < LM_NFFT; i += 256) {
        long n = start + i;
        float v = 0.f;
        if (n >= 0 && n < n_samples) v = pcm[n];
        else if (pad_mode == 1 && n_samples > 1) {                    // numpy 'reflect' (no edge repeat)
            long period = 2 * (n_samples - 1);
            long r = n % period;
            if (r < 0) r += period;
            if (r >= n_samples) r = period - r;
            v = pcm[r];
        }
        int j = bitrev11(i);
        re[j] = v * window[i];
        im[j] = 0.f;
    }
    __syncthreads();
#pragma unroll 1
    for (int s = 0; s < LM_LOG2; ++s) {
        const int m = 1 << s;
        const int tstride = (LM_NFFT / 2) >> s;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int j = tid + u * 256;
            int pos = j & (m - 1);
            int i0 = ((j >> s) << (s + 1)) + pos, i1 = i0 + m;
            float wr = tw[2 * pos * tstride], wi = tw[2 * pos * tstride + 1];
            float xr = re[i1], xi = im[i1];
            float tr = xr * wr - xi * wi, ti = xr * wi + xi * wr;
            float ar = re[i0], ai = im[i0];
            re[i0] = ar + tr; im[i0] = ai + ti;
            re[i1] = ar - tr; im[i1] = ai - ti;
        }
        __syncthreads();
    }
    for (int k = tid; k <= LM_NFFT / 2; k += 256) pw[k] = re[k] * re[k] + im[k] * im[k];
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const int nb = LM_NFFT / 2 + 1;
    for (int mI = wave; mI < n_mels; mI += 4) {
        const float* fb = melfb + (size_t)mI * nb;
        float a = 0.f;
        for (int k = lane; k < nb; k += 64) a += fb[k] * pw[k];
        a = wave_sum(a);
        if (lane == 0) {
            float v = logf(a);
            if (mu) v = (v - mu[mI]) * inv_sigma[mI];
            out[frame * n_mels + mI] = v;
        }
    }
}

extern "C" int sed_logmel(const float* pcm, long n_samples, const float* window, const float* twiddle,
                          const float* melfb, const float* mu, const float* inv_sigma, float* out, int n_fft,
                          int hop, int n_mels, int pad_mode, void* stream) {
    SED_REQUIRE(pcm && window && twiddle && melfb && out, "logmel: null pointer");
    SED_REQUIRE(n_fft == LM_NFFT, "logmel: n_fft must be %d (got %d)", LM_NFFT, n_fft);
    SED_REQUIRE(n_samples > 0 && hop > 0 && n_mels > 0, "logmel: bad sizes");
    SED_REQUIRE((mu == nullptr) == (inv_sigma == nullptr), "logmel: mu and inv_sigma go together");
    SED_REQUIRE(pad_mode == 0 || pad_mode == 1, "logmel: pad_mode must be 0 (constant) or 1 (reflect)");
    long frames = 1 + n_samples / hop;
    logmel_k<<<(unsigned)frames, 256, 0, as_stream(stream)>>>(pcm, n_samples, window, twiddle, melfb, mu, inv_sigma, out,
                                                              hop, n_mels, pad_mode);
    SED_LAUNCH_CHECK("logmel");
    return 0;
}
