// logmel.hip — log-mel front end: framing + window + 2048-point real FFT -> |.|^2 -> mel filterbank -> log.
//
// Replaces librosa.stft / librosa.filters.mel / np.log as called by _mbe at reference feature.py:55-59
// (and, optionally fused, the StandardScaler of feature.py:127-129).
//
// Shape of the kernel (gfx950):
//   * a 2048-sample REAL frame is one 1024-point COMPLEX transform of z[n] = x[2n] + i x[2n+1] plus a pairing pass
//     (X[k] from Z[k] and conj Z[1024-k]) — half the butterflies of the complex transform of the frame;
//   * 1024 = 32 x 32: every lane does a 32-point FFT entirely in registers (compile-time twiddles), the 32 lanes of a
//     HALF wave exchange once through LDS (32x33-float transposes, conflict-free) and do a second 32-point FFT, so a wave
//     transforms TWO frames at a time and needs no workgroup barrier: waves are independent and loop over frame pairs;
//   * window, inter-pass twiddles W_1024^{rq}, pairing twiddles W_2048^k and the mel filterbank live in LDS, loaded once
//     per (persistent) workgroup; the filterbank is applied in its sparse form (a Slaney bank has 2 050 non-zeros of
//     41 000): band-major entry list, 1/32 of it per lane, partial sums combined in a fixed order (deterministic);
//   * HBM traffic is the PCM once (the 50 % overlap of neighbouring frames is re-read by the neighbouring half wave at
//     the same time and comes from L2) + 40 floats per frame.
// Algorithmic bytes per frame: hop*4 read + n_mels*4 written (4 256 B at hop 1024 / 40 mel).
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "common.h"

#define LM_NFFT 2048
#define LM_N 1024                     // complex points
#define LM_TSTRIDE 33                 // floats per row of the 32x32 transpose (conflict-free both ways)
#define LM_SCR (32 * LM_TSTRIDE)      // 1056 floats >= 1025 power bins
#define LM_MAX_MELS 128
#define LM_PART (32 + LM_MAX_MELS)    // partial-sum slots per frame (plan 1: 2 per stored pair + the zero slot)
#define LM_TRI_BINS 33
#define LM_FRAME_SCR (LM_SCR + LM_PART)
#define LM_HDR 8                      // table header words

// table blob (32-bit words):  [0] magic  [1] n_mels  [2] iters  [3] n_slots  [4] total words  [5] plan (0 list, 1 two-band)  [6..7] 0
//   win  [2048]            window, natural order
//   tw   [32][32] float2   tw[q][r] = exp(-2 pi i r q / 1024)
//   pw   [513]   float2    exp(-2 pi i k / 2048), k = 0..512 (+ 1 pad float2)
//  plan 0 (any sparse bank): band-major list of the non-zeros, 1/32 of it per lane
//   ent  [iters][32] {float weight*0.25, u32 meta}    meta = k | emit << 11 | slot << 12
//   band [n_mels] u32      first_slot | count << 16
//  plan 1 (every bin feeds at most two ADJACENT bands — triangular banks such as librosa's): lane r owns the 33
//  consecutive bins 33r .. 33r+32, keeps one accumulator for the lower and one for the upper band of the current bin and
//  stores the pair whenever the band pair changes (iters = 33)
//   ent  [33][32] {float w_lower*0.25, float w_upper*0.25, u32 pair_slot (float index into part, even) or ~0, 0}
//   band [n_mels][8] u16   float indices into part of the partial sums of the band (unused = the always-zero slot)
#define LM_MAGIC 0x4C4D3332u
#define LM_OFF_WIN LM_HDR
#define LM_OFF_TW (LM_OFF_WIN + 2048)
#define LM_OFF_PW (LM_OFF_TW + 2048)
#define LM_OFF_ENT (LM_OFF_PW + 1028)

namespace {

__device__ __forceinline__ constexpr int brev5(int k) {
    return ((k & 1) << 4) | ((k & 2) << 2) | (k & 4) | ((k & 8) >> 2) | ((k & 16) >> 4);
}

// A complex point lives in ONE aligned VGPR pair (re, im): the butterfly's u + v / u - v are one v_pk_add_f32 each and a twiddle
// multiply (dr C + di S, di C - dr S) = d * (C, C) + swap(d) * (S, -S) is v_pk_mul_f32 + v_pk_fma_f32 — hipcc folds the swap into the
// instruction's op_sel bits, so there is no register shuffling: 259 vector instructions per 32-point transform instead of 456
// with separate re[] / im[] arrays (round 2 found the SLP vectoriser's packing of those arrays useless: it paid for every packed
// add with moves that built the pairs; here the pairs are how the data is loaded — global_load_dwordx2 of (x[2n], x[2n+1])).
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 swp(f2 a) { return __builtin_shufflevector(a, a, 1, 0); }
// The forms hipcc does not find by itself (it negates and moves halves with v_xor / v_mov instead) are written with the VOP3P
// modifiers spelled out: op_sel[i] = 1 takes the HIGH half of source i for the low result, op_sel_hi[i] = 0 the LOW half for the
// high result; neg_lo / neg_hi negate source i for the low / high result.  Plain (non-volatile) asm: free to be scheduled.
__device__ __forceinline__ f2 cmul(f2 x, f2 w) {               // x * w (complex): (xr wr - xi wi, xr wi + xi wr)
    f2 t, d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(x), "v"(w));                                   // (xr wr, xi wr)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(d) : "v"(x), "v"(w), "v"(t));   // + (-xi wi, xr wi)
    return d;
}
__device__ __forceinline__ f2 sub_rot(f2 a, f2 b) {            // -i (a - b) = (a.y - b.y, b.x - a.x)
    f2 d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_lo:[0,1] neg_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f2 add_conj(f2 a, f2 b) {           // a + conj(b) = (a.x + b.x, a.y - b.y)
    f2 d;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f2 rot_sub_conj(f2 a, f2 b) {       // -i (a - conj(b)) = (a.y + b.y, b.x - a.x)
    f2 d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// 32-point DIF FFT in registers, forward sign; result X[k] = z[brev5(k)].  All indices are compile-time constants.
__device__ __forceinline__ void fft32(f2 (&z)[32]) {
    constexpr float C32[16] = {1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f,
                               0.70710678118654752440f, 0.55557023301960222474f, 0.38268343236508977173f,
                               0.19509032201612826785f, 0.0f, -0.19509032201612826785f, -0.38268343236508977173f,
                               -0.55557023301960222474f, -0.70710678118654752440f, -0.83146961230254523708f,
                               -0.92387953251128675613f, -0.98078528040323044913f};
    constexpr float S32[16] = {0.0f, 0.19509032201612826785f, 0.38268343236508977173f, 0.55557023301960222474f,
                               0.70710678118654752440f, 0.83146961230254523708f, 0.92387953251128675613f,
                               0.98078528040323044913f, 1.0f, 0.98078528040323044913f, 0.92387953251128675613f,
                               0.83146961230254523708f, 0.70710678118654752440f, 0.55557023301960222474f,
                               0.38268343236508977173f, 0.19509032201612826785f};
#pragma unroll
    for (int h = 16; h >= 1; h >>= 1) {
#pragma unroll
        for (int blk = 0; blk < 32; blk += 2 * h) {
#pragma unroll
            for (int j = 0; j < h; ++j) {
                const int i0 = blk + j, i1 = i0 + h;
                const int m = j * (16 / h);                 // twiddle W_32^m = C32[m] - i S32[m]
                const f2 u = z[i0], v = z[i1];
                z[i0] = u + v;
                if (m == 0) z[i1] = u - v;
                else if (m == 8) z[i1] = sub_rot(u, v);      // W_32^8 = -i
                else { const f2 d = u - v; z[i1] = d * (f2){C32[m], C32[m]} + swp(d) * (f2){S32[m], -S32[m]}; }
            }
        }
    }
}

__device__ __forceinline__ void wave_lds_fence() {
    // LDS operations of ONE wave are executed in order by the hardware; this only stops the compiler from moving
    // them across the point where another lane of the same wave takes over the data
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Eight rows of the pass-1 -> pass-2 exchange in one statement: ds_write_addtid_b32 stores 4 B per lane at
// M0 + offset + 4*lane without an address register and at twice the rate of ds_write_b32 (MI355X_MICROARCH.md, LDS).  Row k
// of the wave's 32 x 65-float exchange buffer starts at byte 260*k (the odd row stride makes the column reads of pass 2
// conflict-free).  M0 is compiler-reserved: saved and restored inside the statement; the s_nop covers the
// SALU-writes-M0 -> LDS-add-TID wait state.  No VGPR is written, so the statement needs no completion count of its own
// (LDS operations of one wave complete in order).
#define LM_ROW_BYTES 260
template <int K0>
__device__ __forceinline__ void addtid_store8(unsigned base, float a0, float a1, float a2, float a3, float a4, float a5,
                                              float a6, float a7) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %9\n\ts_nop 0\n\t"
                 "ds_write_addtid_b32 %1 offset:%10\n\tds_write_addtid_b32 %2 offset:%11\n\t"
                 "ds_write_addtid_b32 %3 offset:%12\n\tds_write_addtid_b32 %4 offset:%13\n\t"
                 "ds_write_addtid_b32 %5 offset:%14\n\tds_write_addtid_b32 %6 offset:%15\n\t"
                 "ds_write_addtid_b32 %7 offset:%16\n\tds_write_addtid_b32 %8 offset:%17\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "s"(base),
                   "i"((K0 + 0) * LM_ROW_BYTES), "i"((K0 + 1) * LM_ROW_BYTES), "i"((K0 + 2) * LM_ROW_BYTES),
                   "i"((K0 + 3) * LM_ROW_BYTES), "i"((K0 + 4) * LM_ROW_BYTES), "i"((K0 + 5) * LM_ROW_BYTES),
                   "i"((K0 + 6) * LM_ROW_BYTES), "i"((K0 + 7) * LM_ROW_BYTES)
                 : "memory");
}
// rows k1 = 0..31 of one component (IMAG = 0: real parts, 1: imaginary parts): row k1 holds X[k1] = z[brev5(k1)] of every lane
template <int IMAG>
__device__ __forceinline__ void exchange_store(unsigned base, const f2 (&z)[32]) {
#define LM_C(k) (IMAG ? z[brev5(k)].y : z[brev5(k)].x)
    addtid_store8<0>(base, LM_C(0), LM_C(1), LM_C(2), LM_C(3), LM_C(4), LM_C(5), LM_C(6), LM_C(7));
    addtid_store8<8>(base, LM_C(8), LM_C(9), LM_C(10), LM_C(11), LM_C(12), LM_C(13), LM_C(14), LM_C(15));
    addtid_store8<16>(base, LM_C(16), LM_C(17), LM_C(18), LM_C(19), LM_C(20), LM_C(21), LM_C(22), LM_C(23));
    addtid_store8<24>(base, LM_C(24), LM_C(25), LM_C(26), LM_C(27), LM_C(28), LM_C(29), LM_C(30), LM_C(31));
#undef LM_C
}

__device__ __forceinline__ float pcm_at(const float* __restrict__ pcm, long n, long n_samples, int pad_mode) {
    if (n >= 0 && n < n_samples) return pcm[n];
    if (pad_mode == 1 && n_samples > 1) {                    // numpy 'reflect' (no edge repeat)
        const long period = 2 * (n_samples - 1);
        long r = n % period;
        if (r < 0) r += period;
        if (r >= n_samples) r = period - r;
        return pcm[r];
    }
    return 0.f;
}

// DB: the next pair's PCM goes to a second register set one iteration ahead (needs the 256 registers of <= 8 waves per CU);
// otherwise it is loaded into the FFT registers before the mel pass of the current pair (they are dead by then)
template <int WPB, bool DB = (WPB <= 8)>
__global__ __launch_bounds__(WPB * 64) void logmel_fft_k(const float* __restrict__ pcm, long n_samples,
                                                         const uint32_t* __restrict__ tables, int table_words,
                                                         const float* __restrict__ mu, const float* __restrict__ inv_sigma,
                                                         float* __restrict__ out, long n_frames, int hop, int pad_mode, int n_mels_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    {   // tables: global -> LDS, once per workgroup (table_words is a multiple of 4)
        const f32x4* src = reinterpret_cast<const f32x4*>(tables);
        f32x4* dst = reinterpret_cast<f32x4*>(lds);
        for (int i = tid; i < table_words / 4; i += WPB * 64) dst[i] = src[i];
    }
    // the fused scaler's coefficients go to LDS too: a global load inside the loop would have to retire IN ORDER behind the
    // PCM prefetch of the next pair (vmcnt counts in order), i.e. wait for exactly the latency the prefetch is there to hide
    float* s_mu = lds + table_words + WPB * 2 * LM_FRAME_SCR;
    float* s_is = s_mu + LM_MAX_MELS;
    if (mu)
        for (int i = tid; i < n_mels_out && i < LM_MAX_MELS; i += WPB * 64) { s_mu[i] = mu[i]; s_is[i] = inv_sigma[i]; }
    __syncthreads();
    const uint32_t* hdr = reinterpret_cast<const uint32_t*>(lds);
    // the row length of `out` is the caller's n_mels; a blob built for more bands than that never writes past a row
    const int n_mels = (int)hdr[1] < n_mels_out ? (int)hdr[1] : n_mels_out, iters = (int)hdr[2];
    const bool two_band = hdr[5] != 0;                       // which mel plan the blob carries (wave-uniform)
    const f2* s_win = reinterpret_cast<const f2*>(lds + LM_OFF_WIN);
    const f2* s_tw = reinterpret_cast<const f2*>(lds + LM_OFF_TW);
    const f2* s_pw = reinterpret_cast<const f2*>(lds + LM_OFF_PW);
    const float2* s_ent = reinterpret_cast<const float2*>(lds + LM_OFF_ENT);
    const uint32_t* s_band = reinterpret_cast<const uint32_t*>(lds + LM_OFF_ENT + (size_t)iters * 64);

    const int lane = tid & 63, wave = tid >> 6, half = lane >> 5, r = lane & 31;
    float* scr = lds + table_words + (wave * 2 + half) * LM_FRAME_SCR;      // this half wave's transpose / power buffer
    float* part = scr + LM_SCR;
    float* wscr = lds + table_words + wave * 2 * LM_FRAME_SCR;               // the wave's whole scratch: the 32 x 65 exchange buffer
    const unsigned xbase = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(wscr - lds) * 4u +
                                                          (unsigned)__builtin_amdgcn_groupstaticsize());
    const int partner = (lane & 32) | ((32 - r) & 31);
    const bool even_hop = (hop & 1) == 0 && (reinterpret_cast<uintptr_t>(pcm) & 7) == 0;
    const long n_pairs = (n_frames + 1) >> 1;

    // interior frame pairs with an even hop are read by plain 8-byte loads; the loads of the NEXT pair are issued before
    // the mel pass of the current one (the FFT registers are dead by then), so HBM latency hides behind it
    auto fast_ok = [&](long pair) -> bool {
        const long first = pair * 2 * hop - LM_NFFT / 2, last = first + hop + LM_NFFT;      // span of both frames (wave-uniform)
        return even_hop && first >= 0 && last <= n_samples && pair * 2 + 1 < n_frames;
    };
    f2 z[32];                                                // z[n] = (x[2n], x[2n+1]): one complex point per VGPR pair
    // Where the time goes (round 4, ablation builds on one box, one hour of audio, 12 waves per CU; the full kernel 0.33-0.35 ms):
    // without the mel pass 0.246, without the pairing pass 0.271, without the two FFTs 0.262, without any of the compute phases
    // 0.175, of which the PCM loads are 0.11 (0.062 with the loads replaced by a register fill): the phases ADD UP — each is a
    // chain of LDS round trips (table reads, exchange, bpermute), and neither the vector ALU (36 %) nor the LDS pipe (40-55 %)
    // nor HBM (1.9 of 6.3 TB/s) is saturated.  What was tried on that in round 4: complex points in VGPR pairs (v_pk_* with op_sel:
    // 38 % fewer vector instructions — no change in run time, kept: it is the smaller kernel); table and power-bin reads of the
    // mel and pairing passes issued in batches ahead of the stores that hipcc has to order them behind (-5 %); a second
    // register set for the next pair's PCM, loaded a whole iteration ahead (DB, needs the 256 registers of 8 waves per CU:
    // 0.36 ms, slower than 12 waves with the loads issued before the mel pass; kept for the large-table plans that run with
    // <= 8 waves anyway); de-phasing the waves of a SIMD by a third of an iteration (no change).
    f2 zn[DB ? 32 : 1];
    const long stride = (long)gridDim.x * WPB;
    const long pair0 = (long)blockIdx.x * WPB + wave;
    bool nloaded = DB && pair0 < n_pairs && fast_ok(pair0);
    if (DB && nloaded) {
        const f2* src = reinterpret_cast<const f2*>(pcm + (pair0 * 2 + half) * hop - LM_NFFT / 2) + r;
#pragma unroll
        for (int n1 = 0; n1 < 32; ++n1) zn[DB ? n1 : 0] = src[32 * n1];
    }
    for (long pair = pair0; pair < n_pairs; pair += stride) {
        long frame = pair * 2 + half;
        const bool live = frame < n_frames;
        if (!live) frame = n_frames - 1;                     // odd tail: the upper half recomputes the last frame, stores nothing
        const long start = frame * hop - LM_NFFT / 2;
        // ── z[32 n1 + r] = (x[64 n1 + 2r], x[64 n1 + 2r + 1]) * window ──
        const bool loaded = nloaded;
        if (DB && loaded) {
#pragma unroll
            for (int n1 = 0; n1 < 32; ++n1) z[n1] = zn[DB ? n1 : 0];
        }
        if (DB) nloaded = pair + stride < n_pairs && fast_ok(pair + stride);
        if (DB && nloaded) {                                 // the next pair's PCM: first use is the copy above, one iteration from now
            const f2* src = reinterpret_cast<const f2*>(pcm + ((pair + stride) * 2 + half) * hop - LM_NFFT / 2) + r;
#pragma unroll
            for (int n1 = 0; n1 < 32; ++n1) zn[DB ? n1 : 0] = src[32 * n1];
        }
        if (!loaded) {
            if (!DB && fast_ok(pair)) {
                const f2* src = reinterpret_cast<const f2*>(pcm + start) + r;
#pragma unroll
                for (int n1 = 0; n1 < 32; ++n1) z[n1] = src[32 * n1];
            } else {                                         // edge frames / odd hop: guarded loads, staged through LDS so that
                wave_lds_fence();                            //  this cold path costs no registers (dynamic index, not unrolled)
#pragma unroll 1
                for (int n1 = 0; n1 < 32; ++n1) scr[n1 * 32 + r] = pcm_at(pcm, start + 64 * n1 + 2 * r, n_samples, pad_mode);
                wave_lds_fence();
#pragma unroll
                for (int n1 = 0; n1 < 32; ++n1) z[n1].x = scr[n1 * 32 + r];
                wave_lds_fence();
#pragma unroll 1
                for (int n1 = 0; n1 < 32; ++n1) scr[n1 * 32 + r] = pcm_at(pcm, start + 64 * n1 + 2 * r + 1, n_samples, pad_mode);
                wave_lds_fence();
#pragma unroll
                for (int n1 = 0; n1 < 32; ++n1) z[n1].y = scr[n1 * 32 + r];
            }
        }
#pragma unroll
        for (int n1 = 0; n1 < 32; ++n1) z[n1] *= s_win[32 * n1 + r];

        // ── pass 1: FFT over n1, twiddle W_1024^{r k1} ──
        fft32(z);
#pragma unroll
        for (int k1 = 1; k1 < 32; ++k1) {
            const int b = brev5(k1);
            z[b] = cmul(z[b], s_tw[k1 * 32 + r]);
        }
        // ── exchange through LDS (real parts, then imaginary parts, the wave's 32 x 65 buffer): lane (half, r) holds
        //    A[r][k1] and stores row k1 lane-linearly; lane (half, c) then needs A[n2][c] = row c, column 32*half + n2 ──
        wave_lds_fence();                                    // the previous frame's mel pass has finished reading the scratch
        exchange_store<0>(xbase, z);
        wave_lds_fence();
        // (single-dword reads straight into the halves of the register pairs: left to hipcc these become ds_read2_b32 into
        // scratch pairs + 64 v_mov to split them — LDS issue slots are cheaper here than vector ones)
        const unsigned xaddr = xbase + (unsigned)(r * LM_ROW_BYTES + 128 * half);          // LDS byte address of xrow
#define LM_XREAD(H)                                                                                              \
    _Pragma("unroll") for (int n2 = 0; n2 < 32; ++n2) {                                                           \
        float t_;                                                                                                \
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(t_) : "v"(xaddr), "i"(n2 * 4));                       \
        z[n2].H = t_;                                                                                            \
    }                                                                                                            \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                           \
    _Pragma("unroll") for (int n2 = 0; n2 < 32; ++n2) asm volatile("" : "+v"(z[n2]));
        LM_XREAD(x)                                          // (the .y halves — the imaginary parts — are still to be stored)
        wave_lds_fence();
        exchange_store<1>(xbase, z);
        wave_lds_fence();
        LM_XREAD(y)
#undef LM_XREAD
        wave_lds_fence();
        // ── pass 2: FFT over n2 -> Z[r + 32 k2] = z[brev5(k2)] ──
        fft32(z);
        // ── pairing: 2 X[k] = (Z[k] + conj Z[N-k]) - i W_2048^k (Z[k] - conj Z[N-k]); lane r owns k = r + 32 k2, k2 < 16, and
        //    its mirror N-k, whose Z lives in lane (32 - r) % 32, register 31 - k2 (lane 0: its own register 32 - k2) ──
        // (the 16 pairing twiddles and the 16 mirrored points are fetched before the first power bin is stored: the stores to
        // `scr` would otherwise fence every later table read behind them, one LDS round trip per k2)
        f2 pwv[16], ppv[16];
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) {
            const int bp = brev5(31 - k2), b0 = brev5((32 - k2) & 31);
            pwv[k2] = s_pw[r + 32 * k2];
            f2 pp = {__shfl(z[bp].x, partner, 64), __shfl(z[bp].y, partner, 64)};
            if (r == 0) pp = z[b0];
            ppv[k2] = pp;
        }
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) {
            const f2 zz = z[brev5(k2)], pp = ppv[k2];
            const int k = r + 32 * k2;
            const f2 e = add_conj(zz, pp);                   // Z[k] + conj Z[N-k]
            const f2 q = rot_sub_conj(zz, pp);               // -i (Z[k] - conj Z[N-k]) = (zi + pi, pr - zr)
            const f2 t = cmul(q, pwv[k2]);
            const f2 a = e + t, bb = e - t;
            const f2 a2 = a * a, b2 = bb * bb;
            scr[k] = a2.x + a2.y;                            // 4 |X[k]|^2   (the 1/4 is folded into the mel weights)
            scr[LM_N - k] = b2.x + b2.y;                     // 4 |X[N-k]|^2
        }
        if (r == 0) { const f2 zz = z[brev5(16)]; scr[512] = 4.f * (zz.x * zz.x + zz.y * zz.y); }
        wave_lds_fence();
        if (!DB) {
            nloaded = pair + stride < n_pairs && fast_ok(pair + stride);
            if (nloaded) {                                   // single set: the FFT registers are dead here; first use is the next window multiply
                const f2* src = reinterpret_cast<const f2*>(pcm + ((pair + stride) * 2 + half) * hop - LM_NFFT / 2) + r;
#pragma unroll
                for (int n1 = 0; n1 < 32; ++n1) z[n1] = src[32 * n1];
            }
        }
        if (two_band) {
            // ── two-band plan: bins 33r .. 33r+32 (bins past 1024 carry zero weights; their slots are cleared so that a
            //    stale non-finite value cannot turn 0*x into NaN) ──
            if (r > 0) scr[LM_N + r] = 0.f;
            if (r == 0) part[LM_PART - 1] = 0.f;             // the always-zero slot the band lists pad with
            wave_lds_fence();
            const f32x4* ent = reinterpret_cast<const f32x4*>(s_ent) + r;
            const float* pb = scr + LM_TRI_BINS * r;
            float lo = 0.f, hi = 0.f;
            // The table entries and power bins of a whole chunk are read BEFORE its first partial sum is stored: hipcc cannot
            // prove that a store to `part` leaves `scr` and the table alone, so with read / use / store per bin (round 2) every
            // bin paid two LDS round trips in sequence — 33 x ~250 cycles, a quarter of the kernel (ablation: 0.11 of 0.40 ms).
            constexpr int CH = DB ? 11 : 3;                   // (12 waves per CU: 168 registers, 64 of them hold the prefetched PCM)
            static_assert(LM_TRI_BINS % CH == 0, "chunks");
#pragma unroll
            for (int c0 = 0; c0 < LM_TRI_BINS; c0 += CH) {
                f32x4 e[CH];
                float pv[CH];
#pragma unroll
                for (int u = 0; u < CH; ++u) { e[u] = ent[(c0 + u) * 32]; pv[u] = pb[c0 + u]; }
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    lo = fmaf(e[u][0], pv[u], lo);
                    hi = fmaf(e[u][1], pv[u], hi);
                    const uint32_t slot = __float_as_uint(e[u][2]);
                    if ((int32_t)slot >= 0) {
                        *reinterpret_cast<float2*>(part + slot) = make_float2(lo, hi);
                        lo = 0.f; hi = 0.f;
                    }
                }
            }
            wave_lds_fence();
            const uint4* blist = reinterpret_cast<const uint4*>(lds + LM_OFF_ENT + LM_TRI_BINS * 32 * 4);
            for (int m = r; m < n_mels; m += 32) {
                const uint4 l = blist[m];
                float v = part[l.x & 0xffffu];                // fixed summation order: deterministic
                v += part[l.x >> 16];
                v += part[l.y & 0xffffu];
                v += part[l.y >> 16];
                v += part[l.z & 0xffffu];
                v += part[l.z >> 16];
                v += part[l.w & 0xffffu];
                v += part[l.w >> 16];
                v = logf(v);
                if (mu) v = (v - s_mu[m]) * s_is[m];
                if (live) out[frame * n_mels_out + m] = v;
            }
        } else {
            // ── list plan: lane r walks entries [r*iters, (r+1)*iters) of the band-major list ──
            float acc = 0.f;
            for (int i0 = 0; i0 < iters; i0 += 8) {          // iters is a multiple of 8; 8 independent LDS reads in flight
                float2 e[8];
                float pv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) e[u] = s_ent[(i0 + u) * 32 + r];
#pragma unroll
                for (int u = 0; u < 8; ++u) pv[u] = scr[__float_as_uint(e[u].y) & 0x7ffu];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const uint32_t meta = __float_as_uint(e[u].y);
                    acc = fmaf(e[u].x, pv[u], acc);
                    if (meta & 0x800u) { part[meta >> 12] = acc; acc = 0.f; }
                }
            }
            wave_lds_fence();
            for (int m = r; m < n_mels; m += 32) {
                const uint32_t bt = s_band[m];
                const int f0 = (int)(bt & 0xffffu), cnt = (int)(bt >> 16);
                float v = 0.f;
                for (int j = 0; j < cnt; ++j) v += part[f0 + j];
                v = logf(v);
                if (mu) v = (v - s_mu[m]) * s_is[m];
                if (live) out[frame * n_mels_out + m] = v;
            }
        }
    }
}

struct LmPlan { int n_mels, iters, n_slots, words, tri; std::vector<uint32_t> ent, band; };

// plan 0: band-major non-zero list of a dense [n_mels][1025] bank, 1/32 per lane; returns 0 when it cannot be planned
int plan_list(const float* fb, int n_mels, LmPlan* p) {
    const int nb = LM_NFFT / 2 + 1;
    std::vector<int> ek, eb;
    std::vector<float> ew;
    for (int m = 0; m < n_mels; ++m) {
        bool any = false;
        for (int k = 0; k < nb; ++k)
            if (fb[(size_t)m * nb + k] != 0.f) { ek.push_back(k); eb.push_back(m); ew.push_back(fb[(size_t)m * nb + k]); any = true; }
        if (!any) { ek.push_back(0); eb.push_back(m); ew.push_back(0.f); }     // an empty band still owns a slot (log 0 = -inf)
    }
    const int nnz = (int)ek.size();
    const int iters = ((nnz + 31) / 32 + 7) & ~7;             // per-lane entries, padded to the kernel's unroll of 8
    if (iters > 256) return 0;
    p->ent.assign((size_t)iters * 64, 0u);
    p->band.assign(n_mels, 0u);
    int slot = 0;
    std::vector<int> first(n_mels, -1), cnt(n_mels, 0);
    for (int lane = 0; lane < 32; ++lane) {
        for (int i = 0; i < iters; ++i) {
            const int e = lane * iters + i;
            if (e >= nnz) break;
            const bool emit = (i == iters - 1) || (e == nnz - 1) || (eb[e + 1] != eb[e]);
            const size_t at = ((size_t)i * 32 + lane) * 2;
            const float w = 0.25f * ew[e];
            memcpy(&p->ent[at], &w, 4);
            p->ent[at + 1] = (uint32_t)ek[e] | (emit ? 0x800u : 0u) | ((uint32_t)slot << 12);
            if (emit) {
                const int m = eb[e];
                if (first[m] < 0) first[m] = slot;
                ++cnt[m];
                ++slot;
            }
        }
    }
    if (slot > LM_PART) return 0;
    for (int m = 0; m < n_mels; ++m) p->band[m] = (uint32_t)first[m] | ((uint32_t)cnt[m] << 16);
    p->n_mels = n_mels; p->iters = iters; p->n_slots = slot; p->tri = 0;
    p->words = (LM_OFF_ENT + iters * 64 + n_mels + 3) & ~3;
    return 1;
}

// plan 1: every bin feeds at most two adjacent bands; returns 0 when the bank is not of that shape
int plan_two_band(const float* fb, int n_mels, LmPlan* p) {
    const int nb = LM_NFFT / 2 + 1;
    std::vector<int> lower(nb, 0);                            // band of the "lower" accumulator at bin k (upper = lower + 1)
    int prev = 0;
    for (int k = 0; k < nb; ++k) {
        int b0 = -1, b1 = -1, n = 0;
        for (int m = 0; m < n_mels; ++m)
            if (fb[(size_t)m * nb + k] != 0.f) { if (n == 0) b0 = m; else b1 = m; ++n; }
        if (n > 2 || (n == 2 && b1 != b0 + 1)) return 0;
        if (n == 2) prev = b0;
        else if (n == 1 && b0 != prev && b0 != prev + 1) prev = b0;
        lower[k] = prev;
    }
    p->ent.assign((size_t)LM_TRI_BINS * 32 * 4, 0u);
    std::vector<std::vector<int>> slots(n_mels);
    int pair = 0;
    for (int lane = 0; lane < 32; ++lane) {
        bool used_lo = false, used_hi = false;
        for (int i = 0; i < LM_TRI_BINS; ++i) {
            const int k = LM_TRI_BINS * lane + i;
            const size_t at = ((size_t)i * 32 + lane) * 4;
            p->ent[at + 2] = 0xffffffffu;
            if (k >= nb) continue;
            const int L = lower[k];
            const float wl = 0.25f * fb[(size_t)L * nb + k];
            const float wh = (L + 1 < n_mels) ? 0.25f * fb[(size_t)(L + 1) * nb + k] : 0.f;
            memcpy(&p->ent[at], &wl, 4);
            memcpy(&p->ent[at + 1], &wh, 4);
            used_lo |= wl != 0.f; used_hi |= wh != 0.f;
            const bool last = (i == LM_TRI_BINS - 1) || (k == nb - 1) || (lower[k + 1] != L);
            if (!last) continue;
            if (used_lo || used_hi) {
                p->ent[at + 2] = (uint32_t)(2 * pair);
                if (used_lo) slots[L].push_back(2 * pair);
                if (used_hi) slots[L + 1].push_back(2 * pair + 1);
                ++pair;
            } else {
                p->ent[at + 2] = 0xfffffffeu;                 // nothing accumulated: no store (accumulators are zero anyway)
            }
            used_lo = used_hi = false;
        }
    }
    if (2 * pair > LM_PART - 1) return 0;
    p->band.assign((size_t)n_mels * 4, 0u);
    const uint32_t zero_slot = LM_PART - 1;
    for (int m = 0; m < n_mels; ++m) {
        if (slots[m].size() > 8) return 0;
        uint32_t l[8];
        for (int j = 0; j < 8; ++j) l[j] = j < (int)slots[m].size() ? (uint32_t)slots[m][j] : zero_slot;
        for (int j = 0; j < 4; ++j) p->band[(size_t)m * 4 + j] = l[2 * j] | (l[2 * j + 1] << 16);
    }
    p->n_mels = n_mels; p->iters = LM_TRI_BINS; p->n_slots = 2 * pair; p->tri = 1;
    p->words = (LM_OFF_ENT + LM_TRI_BINS * 32 * 4 + n_mels * 4 + 3) & ~3;
    return 1;
}

int plan_mel(const float* fb, int n_mels, LmPlan* p) {
    if (plan_two_band(fb, n_mels, p)) return 1;
    return plan_list(fb, n_mels, p);
}

}  // namespace

extern "C" size_t sed_logmel_tables_bytes(const float* melfb_host, int n_fft, int n_mels) {
    if (!melfb_host || n_fft != LM_NFFT || n_mels <= 0 || n_mels > LM_MAX_MELS) return 0;
    LmPlan p;
    if (!plan_mel(melfb_host, n_mels, &p)) return 0;
    return (size_t)p.words * 4;
}

extern "C" int sed_logmel_build_tables(const float* window_host, const float* melfb_host, int n_fft, int n_mels,
                                       void* tables_host, size_t tables_bytes) {
    SED_REQUIRE(window_host && melfb_host && tables_host, "logmel_build_tables: null pointer");
    SED_REQUIRE(n_fft == LM_NFFT, "logmel_build_tables: n_fft must be %d (got %d)", LM_NFFT, n_fft);
    SED_REQUIRE(n_mels > 0 && n_mels <= LM_MAX_MELS, "logmel_build_tables: n_mels must be in [1,%d]", LM_MAX_MELS);
    LmPlan p;
    SED_REQUIRE(plan_mel(melfb_host, n_mels, &p),
                "logmel_build_tables: filterbank has too many non-zeros for the LDS-resident sparse plan");
    SED_REQUIRE(tables_bytes >= (size_t)p.words * 4, "logmel_build_tables: buffer too small (%zu < %zu)", tables_bytes, (size_t)p.words * 4);
    uint32_t* t = (uint32_t*)tables_host;
    memset(t, 0, (size_t)p.words * 4);
    t[0] = LM_MAGIC; t[1] = (uint32_t)n_mels; t[2] = (uint32_t)p.iters; t[3] = (uint32_t)p.n_slots; t[4] = (uint32_t)p.words;
    t[5] = (uint32_t)p.tri;
    float* f = (float*)tables_host;
    memcpy(f + LM_OFF_WIN, window_host, LM_NFFT * sizeof(float));
    const double two_pi = 6.283185307179586476925286766559;
    for (int q = 0; q < 32; ++q)
        for (int r = 0; r < 32; ++r) {
            const double a = two_pi * (double)((r * q) % LM_N) / (double)LM_N;
            f[LM_OFF_TW + (q * 32 + r) * 2] = (float)cos(a);
            f[LM_OFF_TW + (q * 32 + r) * 2 + 1] = (float)(-sin(a));
        }
    for (int k = 0; k <= 512; ++k) {
        const double a = two_pi * (double)k / (double)LM_NFFT;
        f[LM_OFF_PW + 2 * k] = (float)cos(a);
        f[LM_OFF_PW + 2 * k + 1] = (float)(-sin(a));
    }
    memcpy(t + LM_OFF_ENT, p.ent.data(), p.ent.size() * 4);
    memcpy(t + LM_OFF_ENT + p.ent.size(), p.band.data(), p.band.size() * 4);
    return 0;
}

// 12 waves per CU: the most that fit beside the tables (12 x 9.7 KB of exchange / power scratch + 38 KB of tables in 160 KB)
#define LM_WPB 12
#define LM_SCALER_BYTES ((size_t)2 * LM_MAX_MELS * sizeof(float))      // mean and 1/sigma of the fused scaler, behind the wave scratch
template <int WPB>
static int launch_logmel(const float* pcm, long n_samples, const void* tables, int words, const float* mu, const float* inv_sigma,
                         float* out, long frames, int hop, int n_mels, int pad_mode, hipStream_t s) {
    const size_t lds = (size_t)words * 4 + (size_t)WPB * 2 * LM_FRAME_SCR * sizeof(float) + LM_SCALER_BYTES;
    SED_REQUIRE(lds <= 160 * 1024, "logmel: tables + scratch (%zu B) exceed the 160 KiB LDS", lds);
    hipError_t e = hipFuncSetAttribute((const void*)logmel_fft_k<WPB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { sed_set_error("logmel: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    const long pairs = (frames + 1) / 2;
    long blocks = (pairs + WPB - 1) / WPB;
    const long resident = 256;                               // one persistent workgroup per CU: the tables are loaded once each
    if (blocks > resident) blocks = resident;
    SedProfScope prof(SED_K_LOGMEL, s, (double)frames * ((double)hop + n_mels) * 4.0);
    logmel_fft_k<WPB><<<(unsigned)blocks, WPB * 64, lds, s>>>(pcm, n_samples, (const uint32_t*)tables, words, mu, inv_sigma,
                                                              out, frames, hop, pad_mode, n_mels);
    SED_LAUNCH_CHECK("logmel");
    return 0;
}

extern "C" int sed_logmel(const float* pcm, long n_samples, const void* tables, size_t tables_bytes, const float* mu,
                          const float* inv_sigma, float* out, int n_fft, int hop, int n_mels, int pad_mode, void* stream) {
    SED_REQUIRE(pcm && tables && out, "logmel: null pointer");
    SED_REQUIRE(n_fft == LM_NFFT, "logmel: n_fft must be %d (got %d)", LM_NFFT, n_fft);
    SED_REQUIRE(n_samples > 0 && hop > 0 && n_mels > 0 && n_mels <= LM_MAX_MELS, "logmel: bad sizes");
    SED_REQUIRE((mu == nullptr) == (inv_sigma == nullptr), "logmel: mu and inv_sigma go together");
    SED_REQUIRE(pad_mode == 0 || pad_mode == 1, "logmel: pad_mode must be 0 (constant) or 1 (reflect)");
    const int words = (int)(tables_bytes / 4);
    SED_REQUIRE(tables_bytes % 16 == 0 && words >= LM_OFF_ENT + 64 + n_mels, "logmel: table blob of %zu bytes is malformed", tables_bytes);
    SED_REQUIRE((size_t)words * 4 + (size_t)2 * 2 * LM_FRAME_SCR * sizeof(float) + LM_SCALER_BYTES <= (size_t)160 * 1024,
                "logmel: a table blob of %zu bytes leaves no room for the FFT scratch in the 160 KiB LDS", tables_bytes);
    const long frames = 1 + n_samples / hop;
    hipStream_t s = as_stream(stream);
    // 12 waves per workgroup when the tables leave room for their scratch (the two-band plan of a Slaney bank: 38 KB); a large
    // list plan (up to 8 192 non-zeros = 67 KB of entries) runs with fewer waves per CU rather than being refused
    const size_t per_wave = (size_t)2 * LM_FRAME_SCR * sizeof(float), room = (size_t)160 * 1024 - LM_SCALER_BYTES;
    const size_t tb = (size_t)words * 4;
    if (tb + 12 * per_wave <= room) return launch_logmel<12>(pcm, n_samples, tables, words, mu, inv_sigma, out, frames, hop, n_mels, pad_mode, s);
    if (tb + 8 * per_wave <= room) return launch_logmel<8>(pcm, n_samples, tables, words, mu, inv_sigma, out, frames, hop, n_mels, pad_mode, s);
    if (tb + 4 * per_wave <= room) return launch_logmel<4>(pcm, n_samples, tables, words, mu, inv_sigma, out, frames, hop, n_mels, pad_mode, s);
    return launch_logmel<2>(pcm, n_samples, tables, words, mu, inv_sigma, out, frames, hop, n_mels, pad_mode, s);
}
