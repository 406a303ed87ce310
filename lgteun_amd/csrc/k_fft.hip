// Global (FFT amplitude/phase) mixer for gfx950 -- reference models/common/LGT.py:149-180.
//
// One workgroup owns one (sample, channel) plane n x n and keeps it in LDS as a complex n x n image
// (n = 128: 128 KiB of the CU's 160 KiB).  rfft2 is restated as fft_H(rfft_W(x)) and irfft2 as
// irfft_W(ifft_H(X)): forward radix-2 DIF leaves bins in bit-reversed positions, the
// amplitude/phase edit is pointwise so it does not care, and the inverse radix-2 DIT consumes bit-reversed
// input -- no reordering pass.  Only columns kx <= n/2 go through the column transforms (half spectrum);
// rows are Hermitian-extended before the last (row) inverse, with Im of the kx = 0 and kx = n/2 columns
// dropped exactly as a c2r transform does.  The four purely-real bins get +0.0 imaginary parts so that
// angle() takes the same branch as pocketfft's r2c (+pi for negative DC).  fp32 throughout.
#include <math.h>
#include <string.h>

#include "kernels.h"
#include "bwd_kernels.h"

#define FFT_LD(n) ((n) + 1)   // row pitch (complex values) of a plane held in LDS, see fft_pass
// exp(-2 pi i k / 512), k < 256, in constant memory (filled once per device by fft_const_twiddles()).  In every pass of a plane with at least 64
// lines the 64 lanes of a wave hold the SAME butterfly group of 64 different lines, so the twiddle index is wave-uniform: read from here it is an
// s_load (scalar cache, SGPR operand of the packed multiply) instead of one ds_read_b64 per lane and twiddle -- which were a third of the LDS
// instructions of a fused pass (15 twiddle + 16 + 16 data accesses per 16-point item) in kernels whose passes are LDS-issue bound (round 5)
__constant__ float2 c_tw512[256];
static int lg_num_cus() {     // compute units of the current device (256 on an MI355X)
    static std::atomic<int> cus[64];
    int d = 0; (void)hipGetDevice(&d); d &= 63;
    int c = cus[d].load(std::memory_order_relaxed);
    if (!c) {
        if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || c <= 0) c = 256;
        cus[d].store(c, std::memory_order_relaxed);
    }
    return c;
}
static int fft_const_twiddles() {
    static DeviceOnce once;
    if (once.need()) {
        float2 h[256];
        for (int k = 0; k < 256; ++k) {
            const double ang = 2.0 * 3.14159265358979323846 * (double)k / 512.0;
            h[k] = make_float2((float)cos(ang), (float)(0.0 - sin(ang)));
        }
        h[0] = make_float2(1.0f, 0.0f); h[128] = make_float2(0.0f, -1.0f);     // (cos(pi/2) in double is 6e-17, not 0)
        hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(c_tw512), h, sizeof(h));
        if (e != hipSuccess) { lg_set_error("fft: hipMemcpyToSymbol(twiddles): %s", hipGetErrorString(e)); return (int)e; }
        once.done();
    }
    return 0;
}
// the kernel's LDS twiddle table exp(-2 pi i k / n), k < n/2, taken from the constant table (so the uniform and the per-lane path of a pass use the same values)
__device__ __forceinline__ void fft_tw_from_const(float2* tw, int lg) {
    for (int k = threadIdx.x; k < (1 << (lg - 1)); k += blockDim.x) tw[k] = c_tw512[k << (9 - lg)];
}
__device__ __forceinline__ float2 cmul(float2 a, float2 w) { return make_float2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x); }
__device__ __forceinline__ float2 cmulc(float2 a, float2 w) { return make_float2(a.x * w.x + a.y * w.y, a.y * w.x - a.x * w.y); }

// complex arithmetic on packed fp32 (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: the (re, im) pair is the register pair): a radix-2
// butterfly is 4 instructions instead of 8.  a * w = (a.x, a.x) * (w.x, w.y) + (a.y, a.y) * (-w.y, w.x) -- the splat and the swap are
// op_sel / neg modifiers of the packed instruction, not moves.
__device__ __forceinline__ lg_v2f pk_cmul(lg_v2f a, lg_v2f w) {
    const lg_v2f t = (lg_v2f){a.x, a.x} * w;
    return (lg_v2f){a.y, a.y} * (lg_v2f){-w.y, w.x} + t;
}
__device__ __forceinline__ lg_v2f pk_cmulc(lg_v2f a, lg_v2f w) {   // a * conj(w)
    const lg_v2f t = (lg_v2f){a.x, a.x} * (lg_v2f){w.x, -w.y};
    return (lg_v2f){a.y, a.y} * (lg_v2f){w.y, w.x} + t;
}
// the S butterfly levels of one fused pass on the 2^S points a thread holds
template <bool INVERSE, int S, bool UNI = false>
__device__ __forceinline__ void fft_butterflies(float2 (&v)[1 << S], const float2* tw, int lo, int lgmL, int lg) {
    if (UNI) lg = 9;     // twiddles from c_tw512 (lo is wave-uniform: scalar loads)
    constexpr int R = 1 << S;
    lg_v2f u[R];
#pragma unroll
    for (int c = 0; c < R; ++c) u[c] = (lg_v2f){v[c].x, v[c].y};
#pragma unroll
    for (int k = 0; k < S; ++k) {
        // level k pairs (c, c + d); span of this level m = mL * d
        const int dsh = INVERSE ? k : (S - 1 - k);
        const int d = 1 << dsh;
        const int twshift = lg - 1 - (lgmL + dsh);
#pragma unroll
        for (int c = 0; c < R; ++c) {
            if (c & d) continue;
            const int j = lo + ((c & (d - 1)) << lgmL);
            const float2 wf = UNI ? c_tw512[j << twshift] : tw[j << twshift];
            const lg_v2f w = (lg_v2f){wf.x, wf.y};
            const lg_v2f a = u[c], b = u[c + d];
            if (!INVERSE) {
                u[c] = a + b;
                u[c + d] = pk_cmul(a - b, w);
            } else {
                const lg_v2f bw = pk_cmulc(b, w);
                u[c] = a + bw;
                u[c + d] = a - bw;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < R; ++c) v[c] = make_float2(u[c].x, u[c].y);
}

// batched in-place radix-2 butterflies over LDS.  `lines` transforms of length n; element i of line l is at
// buf[l * ls + i * es].  Lines for which skip(l) is true are left untouched.
// S fused radix-2 stages in one LDS pass: a thread loads 2^S points, runs the S butterfly levels in registers and writes
// them back (same data flow as S consecutive radix-2 passes -> same bit-reversed positions), so the plane crosses LDS
// 3 times per 1-D transform (7 = 3 + 3 + 1 stages at n = 128) instead of 7.
// Lines: `nlines` transforms of length n = 2^lg; element i of line l sits at buf[l * ls + i * es].
// SKIP (square in-LDS planes only): lines are columns in bit-reversed kx order; columns with kx > n/2 are not needed.
// position of element p of a line inside a row tile of the split path (SWZ): the element's 32-block keeps its place, the position inside the block
// is rotated by the block's number.  Any fixed permutation of a line leaves the passes conflict-free (all lanes of a wave touch the same element of
// DIFFERENT lines, a conflict-free pitch apart); this one also spreads over all banks what the staging loops touch with consecutive lanes: 32
// consecutive elements, and the bit-reversed images of 32 consecutive bins (8 m + c becomes 8 m + c + (m >> 2): 32 different residues mod 32)
template <bool SWZ>
__device__ __forceinline__ int fft_pos(int p) { return SWZ ? ((p & ~31) | ((p + (p >> 5)) & 31)) : p; }
// SKIP = 2 (compact half-width plane of the real-input kernels): the lines are the n/2 + 1 columns 0 .. n/2 of a plane of n rows.
// twlg: log2 of the length the twiddle table was made for (exp(-2 pi i k / 2^twlg)); a half-length transform on the full table passes lg + 1
template <bool INVERSE, int SKIP, int S, bool LINESFAST = false, bool SWZ = false, bool UNI = false>
__device__ __forceinline__ void fft_fused(float2* buf, const float2* tw, int lg, int lgnl, int ls, int es, int st, int twlg) {
    const int nlines = 1 << lgnl;
    constexpr int R = 1 << S;
    const int n = 1 << lg, half = n >> 1;
    // spans of the fused levels: forward (DIF) largest first, inverse (DIT) smallest first; mL = smallest span
    const int lgmL = INVERSE ? st : (lg - st - S);
    const int mL = 1 << lgmL;
    const int lgpl = lg - S;                     // log2(items per line)
    // SKIP: only the half + 1 needed columns are enumerated (bit-reversed positions: kx < n/2 <-> even p, kx = n/2 <-> p = 1),
    // so every lane of a wave works (skipping by predicate left half of each wave idle)
    const int items = SKIP ? ((half + 1) << lgpl) : (nlines << lgpl);
#pragma unroll 1
    for (int it = threadIdx.x; it < items; it += blockDim.x) {
        int line, t;
        if (SKIP) {
            if (it < (half << lgpl)) { t = it >> (lgnl - 1); line = (SKIP == 2 ? 1 : 2) * (it & (half - 1)); }
            else { t = it - (half << lgpl); line = SKIP == 2 ? half : 1; }
        } else if (es != 1 || LINESFAST) { t = it >> lgnl; line = it & (nlines - 1); }   // consecutive threads -> consecutive lines
        else { line = it >> lgpl; t = it & ((1 << lgpl) - 1); }             // row transforms: consecutive threads -> consecutive items
        float2* base = buf + line * ls;
        float2 v[R];
        const int t0 = UNI ? __builtin_amdgcn_readfirstlane(t) : t;
        if (UNI && __builtin_amdgcn_ballot_w64(t != t0) == 0) {      // the wave's lanes hold the same butterfly group of different lines
            const int lo = t0 & (mL - 1), hi = t0 >> lgmL;
            const int i_base = (hi << (lgmL + S)) + lo;
#pragma unroll
            for (int c = 0; c < R; ++c) v[c] = base[fft_pos<SWZ>(i_base + (c << lgmL)) * es];
            fft_butterflies<INVERSE, S, true>(v, tw, lo, lgmL, twlg);
#pragma unroll
            for (int c = 0; c < R; ++c) base[fft_pos<SWZ>(i_base + (c << lgmL)) * es] = v[c];
        } else {
            const int lo = t & (mL - 1), hi = t >> lgmL;
            const int i_base = (hi << (lgmL + S)) + lo;
#pragma unroll
            for (int c = 0; c < R; ++c) v[c] = base[fft_pos<SWZ>(i_base + (c << lgmL)) * es];
            fft_butterflies<INVERSE, S>(v, tw, lo, lgmL, twlg);
#pragma unroll
            for (int c = 0; c < R; ++c) base[fft_pos<SWZ>(i_base + (c << lgmL)) * es] = v[c];
        }
    }
    __syncthreads();
}

// First pass of the inverse ROW transform of an in-LDS plane (stages 0 .. S-1, span 1: a thread owns 2^S consecutive bit-reversed
// positions of its row) with the Hermitian extension folded into its loads: even positions (kx < n/2) and position 1 (kx = n/2)
// hold the half spectrum the column passes produced; an odd position p > 1 is bin kx = brev(p) > n/2 = conj of bin n - kx, read
// from the even position brev(n - kx) of the same row; Im of the kx = 0 and kx = n/2 bins is dropped (what a c2r transform
// does).  All loads finish before any store (one thread per item, barrier in between), so the separate extension pass and
// its barrier are gone.
template <int S>
__device__ __forceinline__ void fft_rows_inv_first(float2* buf, const float2* tw, int lg, int ld) {
    constexpr int R = 1 << S;
    const int n = 1 << lg;
    const int items = n << (lg - S);           // rows x groups per row; <= blockDim for every plane size launched
    const int it = threadIdx.x;
    const bool act = it < items;
    const int line = it & (n - 1), t = it >> lg;
    float2* base = buf + line * ld;
    float2 v[R];
    if (act) {
#pragma unroll
        for (int c = 0; c < R; ++c) {
            const int pos = (t << S) + c;
            if ((c & 1) == 0) {
                v[c] = base[pos];
                if (pos == 0) v[c].y = 0.0f;
            } else if (pos == 1) {
                v[c] = make_float2(base[1].x, 0.0f);
            } else {
                const int kx = (int)(__brev((unsigned)pos) >> (32 - lg));
                const int pm = (int)(__brev((unsigned)(n - kx)) >> (32 - lg));
                const float2 u = base[pm];
                v[c] = make_float2(u.x, -u.y);
            }
        }
    }
    __syncthreads();
    if (act) {
        fft_butterflies<true, S>(v, tw, 0, 0, lg);
#pragma unroll
        for (int c = 0; c < R; ++c) base[(t << S) + c] = v[c];
    }
    __syncthreads();
}

// full 1-D transform of every line in ceil(lg / 4) LDS passes of up to four fused radix-2 stages (16 points per thread in
// registers): 7 = 4 + 3, 6 = 3 + 3, 8 = 4 + 4, 9 = 3 + 3 + 3.  Every pass costs one read and one write of the plane plus a
// barrier, so at n = 128 a transform is 2 passes instead of the 3 of the (3, 3, 1) grouping.
template <bool INVERSE, int SKIP, bool LINESFAST = false, bool HERMFIRST = false, bool SWZ = false, bool UNI = false>
__device__ __forceinline__ void fft_lines(float2* buf, const float2* tw, int lg, int lgnl, int ls, int es, int twlg = -1) {
    if (twlg < 0) twlg = lg;
    const int npass = (lg + 3) >> 2, base = lg / npass, extra = lg - base * npass;
    int st = 0;
    for (int ps = 0; ps < npass; ++ps) {
        const int S = base + (ps < extra ? 1 : 0);
        if (HERMFIRST && ps == 0) {   // inverse rows of a square in-LDS plane: Hermitian extension folded into the first pass
            if (S == 4) fft_rows_inv_first<4>(buf, tw, lg, ls);
            else if (S == 3) fft_rows_inv_first<3>(buf, tw, lg, ls);
            else if (S == 2) fft_rows_inv_first<2>(buf, tw, lg, ls);
            else fft_rows_inv_first<1>(buf, tw, lg, ls);
        } else if (S == 4) fft_fused<INVERSE, SKIP, 4, LINESFAST, SWZ, UNI>(buf, tw, lg, lgnl, ls, es, st, twlg);
        else if (S == 3) fft_fused<INVERSE, SKIP, 3, LINESFAST, SWZ, UNI>(buf, tw, lg, lgnl, ls, es, st, twlg);
        else if (S == 2) fft_fused<INVERSE, SKIP, 2, LINESFAST, SWZ, UNI>(buf, tw, lg, lgnl, ls, es, st, twlg);
        else fft_fused<INVERSE, SKIP, 1, LINESFAST, SWZ, UNI>(buf, tw, lg, lgnl, ls, es, st, twlg);
        st += S;
    }
}
// square n x n plane resident in LDS, row pitch ld = n + 1 complex values: rows, or the kx <= n/2 columns.  In BOTH directions
// consecutive lanes take consecutive lines: for columns that is adjacent float2s, for rows a stride of n + 1 float2 = 2n + 2
// dwords (= 2 mod 64) -- conflict-free ds_read_b64 / ds_write_b64.  (With pitch n and lanes walking along a row, the
// radix-2 spans put 4..16 lanes on the same banks in every row pass.)
template <bool INVERSE, bool COLS>
__device__ __forceinline__ void fft_pass(float2* buf, const float2* tw, int n, int lg) {
    if (COLS) fft_lines<INVERSE, 1>(buf, tw, lg, lg, 1, FFT_LD(n));
    else fft_lines<INVERSE, 0, true, INVERSE>(buf, tw, lg, lg, FFT_LD(n), 1);   // inverse rows: Hermitian extension fused in
}

// ---- the edit's elementary functions.  The bin edit is pure vector arithmetic (no LDS, no matrix pipe): at 128 x 128 it was 26 % of k_fftmix_r and
// 150 instructions per bin with the library's hypotf / atan2f / sincosf (argument paths for every float, IEEE square root and division).  The bins of
// an image spectrum are ordinary numbers and the edited phase is a few radians, so each function is a short straight-line form here, with the
// library function kept behind a wave-uniform branch for the arguments the short form does not cover (measured in tools/micro/fft_check.hip
// against double precision: errors in the last place or two, as the library's).
// sin and cos of x, |x| <= 100: Cody-Waite reduction by pi/2 in three pieces (k * piece exact for |k| < 2^12), Cephes' minimax pair on [-pi/4, pi/4]
__device__ __forceinline__ void edit_sincos(float x, float& sn, float& cs) {
    if (__builtin_amdgcn_ballot_w64(!(fabsf(x) <= 100.0f)) != 0) { sincosf(x, &sn, &cs); return; }
    const float k = rintf(x * 0.63661977236758134f);
    float r = fmaf(k, -1.57073974609375f, x);
    r = fmaf(k, -5.657970905303955078e-05f, r);
    r = fmaf(k, -9.920936294705029468e-10f, r);
    const float s = r * r;
    const float ps = fmaf(fmaf(-1.9515295891e-4f, s, 8.3321608736e-3f), s, -1.6666654611e-1f);
    const float si = fmaf(ps * s, r, r);
    const float pc = fmaf(fmaf(2.443315711809948e-5f, s, -1.388731625493765e-3f), s, 4.166664568298827e-2f);
    const float co = fmaf(pc * s, s, fmaf(-0.5f, s, 1.0f));
    const unsigned kq = (unsigned)(int)k;
    const float a = (kq & 1u) ? co : si, b = (kq & 1u) ? si : co;
    sn = __uint_as_float(__float_as_uint(a) ^ ((kq & 2u) << 30));
    cs = __uint_as_float(__float_as_uint(b) ^ (((kq + 1u) & 2u) << 30));
}
// |f| and angle(f) (torch.abs / torch.angle of a complex value: hypot and atan2 with IEEE signed-zero conventions -- angle(-a + 0 i) = +pi,
// angle(0) = 0) for 1e-18 <= max(|re|, |im|) <= 1e18 or f = 0; anything else in the wave sends the wave to the library functions
__device__ __forceinline__ void edit_abs_angle(float2 f, float& amp, float& pha) {
    const float ax = fabsf(f.x), ay = fabsf(f.y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    if (__builtin_amdgcn_ballot_w64(!(mx <= 1e18f) || (mx < 1e-18f && mx > 0.0f)) != 0) { amp = hypotf(f.x, f.y); pha = atan2f(f.y, f.x); return; }
    amp = __builtin_amdgcn_sqrtf(fmaf(f.x, f.x, f.y * f.y));
    const float r = __builtin_amdgcn_rcpf(mx);
    float q = mn * r;
    q = fmaf(fmaf(-mx, q, mn), r, q);            // mn / mx to the last place
    q = mx == 0.0f ? 0.0f : q;
    const float t = q * q;
    float p = -0.0017540286062285304f;           // atan(q) = q + q t P(t) on [0, 1]: degree-9 minimax in t (relative error 2.4e-9)
    p = fmaf(p, t, 0.010727421380579472f); p = fmaf(p, t, -0.030805569142103195f); p = fmaf(p, t, 0.05755317956209183f);
    p = fmaf(p, t, -0.08377385884523392f); p = fmaf(p, t, 0.10942058265209198f); p = fmaf(p, t, -0.14261938631534576f);
    p = fmaf(p, t, 0.1999826729297638f); p = fmaf(p, t, -0.3333328366279602f);
    p = fmaf(p * t, q, q);
    p = ay > ax ? (1.57079637050628662f - p) + -4.37113883e-8f : p;
    p = (__float_as_uint(f.x) >> 31) ? (3.14159274101257324f - p) + -8.74227766e-8f : p;
    pha = copysignf(p, f.y);
}

// amplitude / phase edit of one bin (LGT.py:168-177) and its backward, shared by the in-LDS and the split (256^2) paths
__device__ __forceinline__ float2 bin_edit_fwd(float2 f, float aw, float ab, float pw, float pb, float& amp, float& pha) {
    edit_abs_angle(f, amp, pha);
    const float am = aw * amp + ab, ph = pw * pha + pb;
    float sn, cs;
    edit_sincos(ph, sn, cs);
    return make_float2((am * cs + 1e-8f) + 1e-8f, am * sn + 1e-8f);
}
// f = c/n^2 * rfft2(dt) bin; returns dF * n^2 / c and accumulates the four parameter-gradient partials
__device__ __forceinline__ float2 bin_edit_bwd(float2 f, float c, float nn, float A, float PH, float aw, float ab, float pw, float pb,
                                               float& s_aw, float& s_ab, float& s_pw, float& s_pb) {
    const float dR = f.x * (c / nn), dI = f.y * (c / nn);
    const float Am = aw * A + ab, Ph = pw * PH + pb;
    float sn, cs;
    edit_sincos(Ph, sn, cs);
    const float dAm = dR * cs + dI * sn;
    const float dPh = Am * (dI * cs - dR * sn);
    s_aw += dAm * A; s_ab += dAm; s_pw += dPh * PH; s_pb += dPh;
    const float dA = aw * dAm, dPH = pw * dPh;
    float sn0, cs0;
    edit_sincos(PH, sn0, cs0);
    float dFr = 0.f, dFi = 0.f;
    if (A > 0.f) {
        const float ia = 1.0f / A;
        dFr = dA * cs0 - dPH * sn0 * ia;
        dFi = dA * sn0 + dPH * cs0 * ia;
    }
    const float k2 = nn / c;
    return make_float2(dFr * k2, dFi * k2);
}

// enumeration of the half-spectrum bins of an in-LDS plane (bit-reversed layout): item it in [0, n (n/2 + 1)) -> row q, LDS
// column p, compact column c (c < n/2: p = 2c, kx = brev(c); c = n/2: p = 1, kx = n/2).  amp / pha are stored at [q][c]
// (coalesced; the backward of the same plane size reads them back with the same map)
__device__ __forceinline__ void half_bin(int it, int n, int lg, int& q, int& p, int& c) {
    const int half = n >> 1;
    if (it < n * half) { q = it >> (lg - 1); c = it & (half - 1); p = 2 * c; }
    else { q = it - n * half; c = half; p = 1; }
}

// LG = log2(n) is a template parameter: every shift / mask / stride of the (force-inlined) passes becomes an immediate and the
// stage loops unroll -- about a third of the butterfly loops' instructions were runtime index arithmetic
// threads of the in-LDS mixer kernels for a 2^LG-point side (launch_fftmix / launch_fftmix_bwd use the same function)
__host__ __device__ constexpr int fft_threads(int lg) { return lg >= 7 ? 1024 : (lg >= 6 ? 512 : 256); }

template <int LG>
__global__ void k_fftmix(FftArgs a) {
    extern __shared__ float2 smem2[];
    constexpr int lg = LG, n = 1 << LG, half = n >> 1;
    constexpr int LD = FFT_LD(n);
    float2* buf = smem2;          // [n][LD]
    float2* tw = smem2 + n * LD;   // [n/2]  exp(-2 pi i k / n)
    const int plane = blockIdx.x;
    const int ch = plane % a.ch;
    const float* g = a.g + (size_t)plane * n * n;
    for (int k = threadIdx.x; k < half; k += blockDim.x) {
        float ang = 2.0f * (float)k / (float)n;
        tw[k] = make_float2(cospif(ang), 0.0f - sinpif(ang));
    }
    {   // 16-byte global loads (n >= 8: a quad never straddles rows), ALL of a thread's requested before the first is stored: as a loop over
        // blockDim the plane came in as four dependent HBM round trips (load, s_waitcnt vmcnt(0), store per trip) with every workgroup
        // of the launch loading at the same time (round 4)
        constexpr int NTH = fft_threads(LG), NLD = (n * n / 4 + NTH - 1) / NTH;
        float4 v[NLD];
#pragma unroll
        for (int k = 0; k < NLD; ++k) { const int i = (k * NTH + (int)threadIdx.x) * 4; v[k] = *reinterpret_cast<const float4*>(g + (i < n * n ? i : 0)); }
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int i = (k * NTH + (int)threadIdx.x) * 4;
            if (i < n * n) {
                float2* d = buf + (i >> lg) * LD + (i & (n - 1));
                d[0] = make_float2(v[k].x, 0.0f); d[1] = make_float2(v[k].y, 0.0f); d[2] = make_float2(v[k].z, 0.0f); d[3] = make_float2(v[k].w, 0.0f);
            }
        }
    }
    __syncthreads();
    // ---- rfft2: rows then columns
    fft_pass<false, false>(buf, tw, n, lg);
    for (int y = threadIdx.x; y < n; y += blockDim.x) { buf[y * LD + 0].y = 0.0f; buf[y * LD + 1].y = 0.0f; }  // kx = 0, n/2 are real
    __syncthreads();
    fft_pass<false, true>(buf, tw, n, lg);
    if (threadIdx.x < 4) buf[(threadIdx.x >> 1) * LD + (threadIdx.x & 1)].y = 0.0f;  // the four purely-real bins
    __syncthreads();
    // ---- amplitude / phase edit (LGT.py:168-177)
    const float aw = a.ampw[ch], ab = a.ampb[ch], pw = a.phaw[ch], pb = a.phab[ch];
    for (int it = threadIdx.x; it < n * (half + 1); it += blockDim.x) {
        int q, p, c;
        half_bin(it, n, lg, q, p, c);
        float amp, pha;
        const float2 ed = bin_edit_fwd(buf[q * LD + p], aw, ab, pw, pb, amp, pha);
        if (a.amp) {
            size_t o = ((size_t)plane * n + q) * (half + 1) + c;
            __builtin_nontemporal_store(amp, a.amp + o);      // saved for the backward only: streaming stores
            __builtin_nontemporal_store(pha, a.pha + o);
        }
        buf[q * LD + p] = ed;
    }
    __syncthreads();
    // ---- irfft2: columns (complex), Hermitian extension, rows
    fft_pass<true, true>(buf, tw, n, lg);
    fft_pass<true, false>(buf, tw, n, lg);
    const float sc = 1.0f / ((float)n * (float)n);
    float* o = a.o + (size_t)plane * n * n;
    for (int i = threadIdx.x * 4; i < n * n; i += blockDim.x * 4) {
        const float2* r = buf + (i >> lg) * LD + (i & (n - 1));
        const float v[4] = {r[0].x * sc, r[1].x * sc, r[2].x * sc, r[3].x * sc};
        *reinterpret_cast<float4*>(o + i) = make_float4(fabsf(v[0]), fabsf(v[1]), fabsf(v[2]), fabsf(v[3]));
        if (a.sgn) {
            float sg[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) sg[k] = (v[k] > 0.f) ? 1.0f : ((v[k] < 0.f) ? -1.0f : 0.0f);
            { typedef float f4 __attribute__((ext_vector_type(4))); __builtin_nontemporal_store((f4){sg[0], sg[1], sg[2], sg[3]}, reinterpret_cast<f4*>(a.sgn + (size_t)plane * n * n + i)); }
        }
    }
}


// ================================================================================================
// Real-input form of the in-LDS mixer (round 5; the default): the rows are REAL, so a row of n reals is transformed as n/2 complex
// values z[j] = x[2j] + i x[2j+1] -- an n/2-point transform and one unpack step (X[k] = E[k] + w^k O[k], E / O from Z[k], conj Z[n/2-k]) give
// the half spectrum X[0 .. n/2].  The plane is kept COMPACT, n rows x (n/2 + 1) complex: 65 KiB at n = 128 instead of 129 KiB, so TWO
// workgroups (512 threads each) share a CU and one's global loads / stores / barriers run under the other's butterflies; the row passes
// move half the LDS bytes and do half the butterflies of the complex-row form (k_fftmix above: the A/B variant LG_VAR_FFT_FULL).
// Slots: Z[k] sits at slot brev_{lg-1}(k) of its row after the DIF row transform; X[k] (k < n/2) goes to the SAME slot -- which is the
// compact column c = brev_{lg-1}(kx) the half_bin map of the complex-row kernels uses (saved amplitude / phase files are identical) --
// and X[n/2] to slot n/2.  The pair (k, n/2 - k) is read and written by one thread: in place, no barrier inside the step.
// ================================================================================================
#define FFT_HP(n) ((n) / 2 + 1)
__host__ __device__ constexpr int fftr_threads(int lg) { return lg >= 7 ? 1024 : (lg >= 6 ? 512 : 256); }   // as the complex-row kernels: the launches are one or two planes per CU, a plane's latency is what is timed

template <int LG>
__device__ __forceinline__ void fftr_unpack(float2* buf, const float2* tw) {
    constexpr int n = 1 << LG, half = n >> 1, quarter = n >> 2, HP = FFT_HP(n);
    for (int it = threadIdx.x; it < n * (quarter + 1); it += blockDim.x) {
        const int y = it & (n - 1), k = it >> LG;
        float2* row = buf + y * HP;
        if (k == 0) {
            const float2 A = row[0];
            row[0] = make_float2(A.x + A.y, 0.0f);        // kx = 0 and kx = n/2 are real: Im = +0 (the branch-cut convention of the edit)
            row[half] = make_float2(A.x - A.y, 0.0f);
        } else {
            const int qa = (int)(__brev((unsigned)k) >> (33 - LG)), qb = (int)(__brev((unsigned)(half - k)) >> (33 - LG));
            const float2 A = row[qa], B = row[qb], w = tw[k];
            const float ex = 0.5f * (A.x + B.x), ey = 0.5f * (A.y - B.y);      // E = (A + conj B) / 2
            const float ox = 0.5f * (A.y + B.y), oy = 0.5f * (B.x - A.x);      // O = -i (A - conj B) / 2
            const float px = w.x * ox - w.y * oy, py = w.x * oy + w.y * ox;    // w^k O
            row[qa] = make_float2(ex + px, ey + py);                           // X[k] = E + w^k O
            row[qb] = make_float2(ex - px, py - ey);                           // X[n/2 - k] = conj(E - w^k O)
        }
    }
    __syncthreads();
}
// inverse of the step above for a c2r row transform: half spectrum Y[0 .. n/2] -> Z'[k] = E' + i O' (unnormalised: the n/2-point inverse
// then yields n y[2j] + i n y[2j+1], the same scale as the n-point inverse of the complex-row form).  Im Y[0], Im Y[n/2] are dropped (c2r).
template <int LG>
__device__ __forceinline__ void fftr_pack(float2* buf, const float2* tw) {
    constexpr int n = 1 << LG, half = n >> 1, quarter = n >> 2, HP = FFT_HP(n);
    for (int it = threadIdx.x; it < n * (quarter + 1); it += blockDim.x) {
        const int y = it & (n - 1), k = it >> LG;
        float2* row = buf + y * HP;
        if (k == 0) {
            const float a = row[0].x, b = row[half].x;
            row[0] = make_float2(a + b, a - b);
        } else {
            const int qa = (int)(__brev((unsigned)k) >> (33 - LG)), qb = (int)(__brev((unsigned)(half - k)) >> (33 - LG));
            const float2 A = row[qa], B = row[qb], w = tw[k];
            const float ex = A.x + B.x, ey = A.y - B.y;                        // E' = Y[k] + conj Y[n/2 - k]
            const float dx = A.x - B.x, dy = A.y + B.y;                        // D  = Y[k] - conj Y[n/2 - k]
            const float ox = w.x * dx + w.y * dy, oy = w.x * dy - w.y * dx;    // O' = conj(w^k) D
            row[qa] = make_float2(ex - oy, ey + ox);                           // Z'[k] = E' + i O'
            row[qb] = make_float2(ex + oy, ox - ey);                           // Z'[n/2 - k] = conj E' + i conj O'
        }
    }
    __syncthreads();
}
template <int LG> __device__ __forceinline__ void fftr_rows_fwd(float2* buf, const float2* tw) {
    fft_lines<false, 0, true, false, false, true>(buf, tw, LG - 1, LG, FFT_HP(1 << LG), 1, LG);
    fftr_unpack<LG>(buf, tw);
}
template <int LG> __device__ __forceinline__ void fftr_rows_inv(float2* buf, const float2* tw) {
    fftr_pack<LG>(buf, tw);
    fft_lines<true, 0, true, false, false, true>(buf, tw, LG - 1, LG, FFT_HP(1 << LG), 1, LG);
}
template <bool INVERSE, int LG> __device__ __forceinline__ void fftr_cols(float2* buf, const float2* tw) {
    fft_lines<INVERSE, 2, false, false, false, true>(buf, tw, LG, LG, 1, FFT_HP(1 << LG));
}
// half-spectrum bins of the compact plane: item it in [0, n (n/2 + 1)) -> row q (bit-reversed ky), column c (c < n/2: kx = brev(c); c = n/2: kx = n/2)
__device__ __forceinline__ void fftr_bin(int it, int n, int lg, int& q, int& c) {
    const int half = n >> 1;
    if (it < n * half) { q = it >> (lg - 1); c = it & (half - 1); }
    else { q = it - n * half; c = half; }
}

#ifdef LG_FFT_STAMPS     // tools/micro/fft_check.hip: s_memtime at the phase boundaries of workgroup 0 .. 255, wave 0
__device__ unsigned long long fft_stamps[256][12];
#define FFT_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 256) fft_stamps[blockIdx.x][i] = __builtin_readcyclecounter(); } while (0)
#else
#define FFT_STAMP(i) do { } while (0)
#endif
// NTH = 1024 at n = 128 when the launch has at most one plane per CU (a plane's latency is the launch's duration); NTH = 512 otherwise: two workgroups
// per CU (2 x 65.5 KiB of LDS, 128 registers per lane), one's loads / stores / barriers under the other's butterflies
template <int LG, int NTH>
__global__ void __launch_bounds__(NTH, (LG >= 7 && NTH == 512) ? 4 : 1) k_fftmix_r(FftArgs a) {
    extern __shared__ float2 smem2[];
    constexpr int lg = LG, n = 1 << LG, half = n >> 1, HP = FFT_HP(n);
    float2* buf = smem2;            // [n][HP]
    float2* tw = smem2 + n * HP;    // [n/2]  exp(-2 pi i k / n)
    FFT_STAMP(0);
    const int plane = blockIdx.x;
    const int ch = plane % a.ch;
    const float* g = a.g + (size_t)plane * n * n;
    constexpr int NLD = (n * n / 4 + NTH - 1) / NTH;     // (n = 8: 16 of the 64 threads hold a load)
    {   // every 16-byte load of the thread requested before the first store; four reals = two packed complex values of a row
        float4 v[NLD];
#pragma unroll
        for (int k = 0; k < NLD; ++k) { const int i = (k * NTH + (int)threadIdx.x) * 4; v[k] = *reinterpret_cast<const float4*>(g + (i < n * n ? i : 0)); }
        fft_tw_from_const(tw, lg);
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int i = (k * NTH + (int)threadIdx.x) * 4;
            if (i < n * n) {
                float2* d = buf + (i >> lg) * HP + ((i & (n - 1)) >> 1);
                d[0] = make_float2(v[k].x, v[k].y); d[1] = make_float2(v[k].z, v[k].w);
            }
        }
    }
    __syncthreads();
    FFT_STAMP(1);
    // ---- rfft2: rows (real-input form), then the n/2 + 1 columns
    fftr_rows_fwd<LG>(buf, tw);
    FFT_STAMP(2);
    fftr_cols<false, LG>(buf, tw);
    if (threadIdx.x < 4) buf[(threadIdx.x >> 1) * HP + (threadIdx.x & 1) * half].y = 0.0f;  // the four purely-real bins (ky in {0, n/2}: rows 0, 1)
    __syncthreads();
    FFT_STAMP(3);
    // ---- amplitude / phase edit (LGT.py:168-177)
    const float aw = a.ampw[ch], ab = a.ampb[ch], pw = a.phaw[ch], pb = a.phab[ch];
    for (int it = threadIdx.x; it < n * (half + 1); it += NTH) {
        int q, c;
        fftr_bin(it, n, lg, q, c);
        float amp, pha;
        const float2 ed = bin_edit_fwd(buf[q * HP + c], aw, ab, pw, pb, amp, pha);
        if (a.amp) {
            size_t o = ((size_t)plane * n + q) * (half + 1) + c;
            __builtin_nontemporal_store(amp, a.amp + o);      // saved for the backward only: streaming stores
            __builtin_nontemporal_store(pha, a.pha + o);
        }
        buf[q * HP + c] = ed;
    }
    __syncthreads();
    FFT_STAMP(4);
    // ---- irfft2: columns (complex), rows (c2r: pack + n/2-point inverse)
    fftr_cols<true, LG>(buf, tw);
    FFT_STAMP(5);
    fftr_rows_inv<LG>(buf, tw);
    FFT_STAMP(6);
    const float sc = 1.0f / ((float)n * (float)n);
    float* o = a.o + (size_t)plane * n * n;
    int tid_s = threadIdx.x;
    asm volatile("" : "+v"(tid_s));      // the store offsets are recomputed here, not kept in registers (or scratch) since the loads
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int i = (k * NTH + tid_s) * 4;
        if (i >= n * n) break;
        const float2* r = buf + (i >> lg) * HP + ((i & (n - 1)) >> 1);
        const float2 r0 = r[0], r1 = r[1];
        const float v[4] = {r0.x * sc, r0.y * sc, r1.x * sc, r1.y * sc};
        *reinterpret_cast<float4*>(o + i) = make_float4(fabsf(v[0]), fabsf(v[1]), fabsf(v[2]), fabsf(v[3]));
        if (a.sgn) {
            float sg[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) sg[u] = (v[u] > 0.f) ? 1.0f : ((v[u] < 0.f) ? -1.0f : 0.0f);
            { typedef float f4 __attribute__((ext_vector_type(4))); __builtin_nontemporal_store((f4){sg[0], sg[1], sg[2], sg[3]}, reinterpret_cast<f4*>(a.sgn + (size_t)plane * n * n + i)); }
        }
    }
    FFT_STAMP(7);
}


// ================================================================================================
// Split path for planes that do not fit LDS (n = 256, 512): rows -> global half spectrum S[plane][n][n/2+1] -> columns
// (+ amplitude/phase edit, in place) -> rows.  Same arithmetic as the in-LDS kernels; three launches per direction.
// ================================================================================================
#define FFT_ROWS_PER_WG(n) (8192 / (n))    // 64 KiB complex tile, 512 threads: TWO workgroups per CU, so one's loads / stores run under the other's
                                           // transform (as one 128 KiB tile of 1 024 threads per CU the three phases of a tile were serial)
#define FFT_ROWS_NT 512
#define FFT_COLS_PER_WG(n) (8192 / (n))    // 64 KiB complex tile

__device__ __forceinline__ void make_twiddles(float2* tw, int n) {
    for (int k = threadIdx.x; k < (n >> 1); k += blockDim.x) {
        float ang = 2.0f * (float)k / (float)n;
        tw[k] = make_float2(cospif(ang), 0.0f - sinpif(ang));
    }
}

// real rows -> half spectrum (natural kx order).  mul: optional elementwise factor (backward: sign of the forward output)
// (LG = log2 n is a template parameter here too: the passes' shifts and the divisions by n/2 + 1 of the staging loops become immediates)
template <int LG>
__global__ void k_fft_rows_fwd(const float* __restrict__ in, const float* __restrict__ mul, float2* __restrict__ S, int force_real) {
    extern __shared__ float2 smem2[];
    constexpr int n = 1 << LG, lg = LG, R = FFT_ROWS_PER_WG(n), half = n >> 1, lgR = 13 - LG;
    static_assert((1 << lgR) == R, "rows per tile");
    // row pitch n + 1 complex and consecutive lanes on consecutive ROWS: a lane stride of 2n + 2 dwords (= 2 mod 64) is conflict-free in every pass;
    // with the rows back to back and lanes walking along a row the second pass (span-1 groups of 16 points) was a 16-way bank conflict
    constexpr int LD = n + 1;
    float2* buf = smem2;
    float2* tw = smem2 + R * LD;
    const size_t plane = blockIdx.x;
    const int row0 = blockIdx.y * R;
    const size_t base = (plane * n + row0) * n;
    {   // the tile's 8 192 reals = four 16-byte loads per thread (512 threads), ALL requested before the twiddles are made and the first value
        // is stored: as sixteen load -> store trips of one float the kernel was sixteen dependent round trips long (one workgroup per CU)
        const float4* __restrict__ in4 = reinterpret_cast<const float4*>(in + base);
        float4 v[4], m[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = in4[threadIdx.x + FFT_ROWS_NT * k];
        if (mul) {
            const float4* __restrict__ mul4 = reinterpret_cast<const float4*>(mul + base);
#pragma unroll
            for (int k = 0; k < 4; ++k) m[k] = mul4[threadIdx.x + FFT_ROWS_NT * k];
        }
        make_twiddles(tw, n);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float4 t = v[k];
            if (mul) { t.x *= m[k].x; t.y *= m[k].y; t.z *= m[k].z; t.w *= m[k].w; }
            const int i4 = threadIdx.x + FFT_ROWS_NT * k;              // float4 index of the tile: row i4 / (n / 4), columns 4 (i4 % (n / 4)) ...
            const int c0 = 4 * (i4 % (n / 4));
            float2* row = buf + (i4 / (n / 4)) * LD;
            row[fft_pos<true>(c0)] = make_float2(t.x, 0.0f); row[fft_pos<true>(c0 + 1)] = make_float2(t.y, 0.0f);
            row[fft_pos<true>(c0 + 2)] = make_float2(t.z, 0.0f); row[fft_pos<true>(c0 + 3)] = make_float2(t.w, 0.0f);
        }
    }
    __syncthreads();
    fft_lines<false, false, true, false, true>(buf, tw, lg, lgR, LD, 1);
    for (int i = threadIdx.x; i < R * (half + 1); i += blockDim.x) {
        const int r = i / (half + 1), kx = i - r * (half + 1);
        float2 v = buf[r * LD + fft_pos<true>((int)(__brev((unsigned)kx) >> (32 - lg)))];
        if (force_real && (kx == 0 || kx == half)) v.y = 0.0f;
        S[(plane * n + row0 + r) * (half + 1) + kx] = v;
    }
}

// columns of the half spectrum, in place: forward column FFT, bin edit (forward or backward), inverse column FFT
template <bool BWD, int LG>
__global__ void k_fft_cols(FftArgs fa, FftBwdArgs ba, float2* __restrict__ S) {
    extern __shared__ float2 smem2[];
    constexpr int n = 1 << LG, lg = LG, CB = FFT_COLS_PER_WG(n), half = n >> 1, lgCB = 13 - LG;
    static_assert((1 << lgCB) == CB, "columns per tile");
    float2* buf = smem2;            // [n rows][CB cols]
    float2* tw = smem2 + n * CB;
    float* red = reinterpret_cast<float*>(tw + half);
    const size_t plane = blockIdx.x;
    const int chn = BWD ? ba.ch : fa.ch;
    const int ch = (int)(plane % chn);
    const int kx0 = blockIdx.y * CB;
    const int ncols = min(CB, half + 1 - kx0);
    {   // the tile's 8 192 bins = sixteen 8-byte loads per thread (512 threads), all requested first (clamped columns, zeroed afterwards)
        float2 v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int i = threadIdx.x + 512 * k, y = i >> lgCB, c = i & (CB - 1);
            v[k] = S[(plane * n + y) * (half + 1) + kx0 + min(c, ncols - 1)];
        }
        make_twiddles(tw, n);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int i = threadIdx.x + 512 * k, c = i & (CB - 1);
            buf[i] = (c < ncols) ? v[k] : make_float2(0.f, 0.f);
        }
    }
    __syncthreads();
    fft_lines<false, false>(buf, tw, lg, lgCB, 1, CB);
    if (!BWD) {   // the four purely-real bins: ky in {0, n/2} (positions 0, 1) x kx in {0, n/2}
        if (threadIdx.x < 2 * CB) {
            const int q = threadIdx.x / CB, c = threadIdx.x - q * CB;
            if (c < ncols && (kx0 + c == 0 || kx0 + c == half)) buf[q * CB + c].y = 0.0f;
        }
        __syncthreads();
    }
    const float aw = (BWD ? ba.ampw : fa.ampw)[ch], ab = (BWD ? ba.ampb : fa.ampb)[ch];
    const float pw = (BWD ? ba.phaw : fa.phaw)[ch], pb = (BWD ? ba.phab : fa.phab)[ch];
    const float nn = (float)n * (float)n;
    float s_aw = 0.f, s_ab = 0.f, s_pw = 0.f, s_pb = 0.f;
#pragma unroll 1
    for (int g = 0; g < 4; ++g) {   // four groups of four bins per thread; the backward's saved amplitude / phase of a group in one batch
        float sa[4], sp[4];
        if (BWD) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = threadIdx.x + 512 * (4 * g + u), q = i >> lgCB, c = min(i & (CB - 1), ncols - 1);
                const size_t o = (plane * n + (int)(__brev((unsigned)q) >> (32 - lg))) * (half + 1) + kx0 + c;
                sa[u] = ba.amp[o]; sp[u] = ba.pha[o];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = threadIdx.x + 512 * (4 * g + u);
            const int q = i >> lgCB, c = i & (CB - 1);
            if (c >= ncols) continue;
            const int ky = (int)(__brev((unsigned)q) >> (32 - lg)), kx = kx0 + c;
            const size_t o = (plane * n + ky) * (half + 1) + kx;
            if (!BWD) {
                float amp, pha;
                buf[i] = bin_edit_fwd(buf[i], aw, ab, pw, pb, amp, pha);
                if (fa.amp) { __builtin_nontemporal_store(amp, fa.amp + o); __builtin_nontemporal_store(pha, fa.pha + o); }
            } else {
                const float cw = (kx == 0 || kx == half) ? 1.0f : 2.0f;
                buf[i] = bin_edit_bwd(buf[i], cw, nn, sa[u], sp[u], aw, ab, pw, pb, s_aw, s_ab, s_pw, s_pb);
            }
        }
    }
    __syncthreads();
    fft_lines<true, false>(buf, tw, lg, lgCB, 1, CB);
    for (int i = threadIdx.x; i < n * CB; i += blockDim.x) {
        const int y = i >> lgCB, c = i & (CB - 1);
        if (c < ncols) S[(plane * n + y) * (half + 1) + kx0 + c] = buf[i];
    }
    if (BWD) {
        float v[4] = {s_aw, s_ab, s_pw, s_pb};
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v[i] += __shfl_xor(v[i], off);
            if (lane == 0) red[wave * 4 + i] = v[i];
        }
        __syncthreads();
        if (threadIdx.x < 4) {
            float sum = 0.f;
            for (int w = 0; w < nw; ++w) sum += red[w * 4 + threadIdx.x];
            // partial row [sample][column group][channel][4]: summed in a fixed order by launch_reduce_job (no float atomics)
            ba.part[(((plane / chn) * gridDim.y + blockIdx.y) * chn + ch) * 4 + threadIdx.x] = sum;
        }
    }
}

// half spectrum rows -> real rows (c2r): out = |x| (+ sign save) when absout, else the signed value
template <int LG>
__global__ void k_fft_rows_inv(const float2* __restrict__ S, float* __restrict__ out, float* __restrict__ sgn, int absout) {
    extern __shared__ float2 smem2[];
    constexpr int n = 1 << LG, lg = LG, R = FFT_ROWS_PER_WG(n), half = n >> 1, lgR = 13 - LG;
    static_assert((1 << lgR) == R, "rows per tile");
    constexpr int LD = n + 1;   // as in k_fft_rows_fwd
    float2* buf = smem2;
    float2* tw = smem2 + R * LD;
    const size_t plane = blockIdx.x;
    const int row0 = blockIdx.y * R;
    {   // the tile's R (n/2 + 1) bins (contiguous in S): nine 8-byte loads per thread, all requested first
        const float2* __restrict__ Sb = S + (plane * n + row0) * (size_t)(half + 1);
        const int total = R * (half + 1);
        float2 v[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) v[k] = Sb[min((int)threadIdx.x + FFT_ROWS_NT * k, total - 1)];
        make_twiddles(tw, n);
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int i = threadIdx.x + FFT_ROWS_NT * k;
            if (i < total) {
                const int r = i / (half + 1), kx = i - r * (half + 1);
                const int p = fft_pos<true>((int)(__brev((unsigned)kx) >> (32 - lg)));
                if (kx == 0 || kx == half) {
                    buf[r * LD + p] = make_float2(v[k].x, 0.0f);           // c2r drops these imaginary parts
                } else {
                    buf[r * LD + p] = v[k];
                    buf[r * LD + fft_pos<true>((int)(__brev((unsigned)(n - kx)) >> (32 - lg)))] = make_float2(v[k].x, -v[k].y);
                }
            }
        }
    }
    __syncthreads();
    fft_lines<true, false, true, false, true>(buf, tw, lg, lgR, LD, 1);
    const float sc = 1.0f / ((float)n * (float)n);
    const size_t base = (plane * n + row0) * n;
#pragma unroll
    for (int k = 0; k < 4; ++k) {   // 8 192 reals = four 16-byte stores per thread
        const int i4 = threadIdx.x + FFT_ROWS_NT * k;
        const float2* row = buf + (i4 / (n / 4)) * LD;
        const int c0 = 4 * (i4 % (n / 4));
        const float v[4] = {row[fft_pos<true>(c0)].x * sc, row[fft_pos<true>(c0 + 1)].x * sc, row[fft_pos<true>(c0 + 2)].x * sc, row[fft_pos<true>(c0 + 3)].x * sc};
        if (absout) {
            reinterpret_cast<float4*>(out + base)[i4] = make_float4(fabsf(v[0]), fabsf(v[1]), fabsf(v[2]), fabsf(v[3]));
            if (sgn) {
                float sg[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) sg[u] = (v[u] > 0.f) ? 1.0f : ((v[u] < 0.f) ? -1.0f : 0.0f);
                { typedef float f4 __attribute__((ext_vector_type(4))); __builtin_nontemporal_store((f4){sg[0], sg[1], sg[2], sg[3]}, reinterpret_cast<f4*>(sgn + base) + i4); }
            }
        } else {
            reinterpret_cast<float4*>(out + base)[i4] = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

size_t fft_scratch_floats(int planes, int n) { return n > 128 ? (size_t)planes * n * (n / 2 + 1) * 2 : 0; }
bool fft_is_generic(int h, int w);
// half-spectrum scratch of the paths that go through HBM (split: square powers of two above 128; generic: everything else)
size_t fft_scratch_floats_hw(int planes, int h, int w) {
    return (fft_is_generic(h, w) || h > 128) ? (size_t)planes * h * (w / 2 + 1) * 2 : 0;
}

template <int LG>
static int launch_fft_split_t(const FftArgs* fa, const FftBwdArgs* ba, float2* S, int planes, hipStream_t s) {
    constexpr int n = 1 << LG;
    static DeviceOnce attr_once;
    if (attr_once.need()) {
        const void* fns[4] = {(const void*)k_fft_rows_fwd<LG>, (const void*)k_fft_cols<false, LG>, (const void*)k_fft_cols<true, LG>, (const void*)k_fft_rows_inv<LG>};
        for (int i = 0; i < 4; ++i) {
            hipError_t e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
            if (e != hipSuccess) { lg_set_error("fft split: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
        }
        attr_once.done();
    }
    const int R = FFT_ROWS_PER_WG(n), CB = FFT_COLS_PER_WG(n), half = n / 2;
    const size_t lds_rows = ((size_t)R * (n + 1) + half) * sizeof(float2);
    const size_t lds_cols = ((size_t)n * CB + half) * sizeof(float2) + 64 * sizeof(float);
    dim3 grows(planes, n / R), gcols(planes, (half + 1 + CB - 1) / CB);
    if (fa) {
        k_fft_rows_fwd<LG><<<grows, FFT_ROWS_NT, lds_rows, s>>>(fa->g, nullptr, S, 1);
        LG_CHECK_LAUNCH();
        FftBwdArgs dummy;
        memset(&dummy, 0, sizeof(dummy));
        k_fft_cols<false, LG><<<gcols, 512, lds_cols, s>>>(*fa, dummy, S);
        LG_CHECK_LAUNCH();
        k_fft_rows_inv<LG><<<grows, FFT_ROWS_NT, lds_rows, s>>>(S, fa->o, fa->sgn, 1);
        LG_CHECK_LAUNCH();
    } else {
        k_fft_rows_fwd<LG><<<grows, FFT_ROWS_NT, lds_rows, s>>>(ba->do2, ba->sgn, S, 0);
        LG_CHECK_LAUNCH();
        FftArgs dummy;
        memset(&dummy, 0, sizeof(dummy));
        k_fft_cols<true, LG><<<gcols, 512, lds_cols, s>>>(dummy, *ba, S);
        LG_CHECK_LAUNCH();
        k_fft_rows_inv<LG><<<grows, FFT_ROWS_NT, lds_rows, s>>>(S, ba->dg, nullptr, 0);
        LG_CHECK_LAUNCH();
    }
    return 0;
}
static int launch_fft_split(const FftArgs* fa, const FftBwdArgs* ba, hipStream_t s) {
    const int n = fa ? fa->n : ba->n, planes = fa ? fa->planes : ba->planes;
    float2* S = reinterpret_cast<float2*>(fa ? fa->scratch : ba->scratch);
    if (!S) { lg_set_error("fftmix: plane size %d needs the split path but no scratch buffer was given", n); return -2; }
    if (n == 256) return launch_fft_split_t<8>(fa, ba, S, planes, s);
    if (n == 512) return launch_fft_split_t<9>(fa, ba, S, planes, s);
    lg_set_error("fftmix: the split path exists for planes of 256 and 512 (got %d)", n);
    return -2;
}


// ================================================================================================
// Generic path: any plane h x w (sides multiples of 8 up to 1024: 400 x 400 full-resolution scenes, rectangular
// crops, ...).  Same three-kernel structure as the split path (rows -> global half spectrum -> columns + edit -> rows), but every
// line transform is a Bluestein chirp-z DFT evaluated with the power-of-two passes above:
//     X_k = w_k * sum_j (x_j w_j) conj(w)_{k-j},   w_k = exp(-i pi k^2 / n)
// i.e. one forward and one inverse FFT of length M >= 2n - 1 per line; the chirp filter's spectrum is computed once per
// workgroup.  The DIF forward leaves bit-reversed order and the DIT inverse consumes it, so the pointwise product needs no
// reordering.  Inverse transforms use IDFT(x) = conj(DFT(conj(x))).  Built for coverage, not speed: the square power-of-two
// sizes (all BASELINE configs) never come here.
// ================================================================================================
struct GDft { int n, M, lgM, L, lgL; };   // line length, FFT length, lines per workgroup
static GDft gdft_plan(int n) {
    GDft d;
    d.n = n; d.lgM = 0;
    while ((1 << d.lgM) < 2 * n - 1) ++d.lgM;
    d.M = 1 << d.lgM;
    d.L = 8192 / d.M;          // 64 KiB of lines per workgroup
    if (d.L > 16) d.L = 16;
    d.lgL = 0;
    while ((1 << d.lgL) < d.L) ++d.lgL;
    return d;
}
static size_t gdft_lds_bytes(const GDft& d) { return ((size_t)d.L * d.M + d.M + d.n + d.M / 2) * sizeof(float2) + 64 * sizeof(float); }
bool fft_is_generic(int h, int w) { return !(h == w && (h & (h - 1)) == 0 && h >= 8 && h <= 512); }

struct GLds { float2 *lines, *bf, *wch, *tw; float* red; };
__device__ __forceinline__ GLds gdft_carve(float2* smem, const GDft& d) {
    GLds g;
    g.lines = smem; g.bf = smem + ((size_t)d.L << d.lgM); g.wch = g.bf + d.M; g.tw = g.wch + d.n;
    g.red = reinterpret_cast<float*>(g.tw + d.M / 2);
    return g;
}
// twiddles of length M, chirp w_k (k < n, k^2 reduced mod 2n in integers so the angle keeps full precision) and the spectrum of
// the chirp filter b_j = conj(w_j), |j| < n, wrapped to length M -- in the bit-reversed order the lines' spectra come out in
__device__ __forceinline__ void gdft_setup(const GLds& g, const GDft& d) {
    make_twiddles(g.tw, d.M);
    for (int k = threadIdx.x; k < d.n; k += blockDim.x) {
        const int k2 = (k * k) % (2 * d.n);
        const float ang = (float)k2 / (float)d.n;
        g.wch[k] = make_float2(cospif(ang), 0.0f - sinpif(ang));
    }
    for (int k = threadIdx.x; k < d.M; k += blockDim.x) g.bf[k] = make_float2(0.f, 0.f);
    __syncthreads();
    for (int k = threadIdx.x; k < d.n; k += blockDim.x) {
        const float2 c = make_float2(g.wch[k].x, -g.wch[k].y);
        g.bf[k] = c;
        if (k) g.bf[d.M - k] = c;
    }
    __syncthreads();
    fft_lines<false, false>(g.bf, g.tw, d.lgM, 0, d.M, 1);
}
// unnormalised forward DFT (length n) of the L lines [L][M] (entries k < n are the input; the rest is scratch), in place,
// natural order on both sides.  Caller has synchronised after writing the input.
__device__ __forceinline__ void gdft_lines(const GLds& g, const GDft& d) {
    const int tot = d.L << d.lgM, msk = d.M - 1;
    const float inv = 1.0f / (float)d.M;
    for (int i = threadIdx.x; i < tot; i += blockDim.x) {
        const int k = i & msk;
        g.lines[i] = (k < d.n) ? cmul(g.lines[i], g.wch[k]) : make_float2(0.f, 0.f);
    }
    __syncthreads();
    fft_lines<false, false>(g.lines, g.tw, d.lgM, d.lgL, d.M, 1);
    for (int i = threadIdx.x; i < tot; i += blockDim.x) g.lines[i] = cmul(g.lines[i], g.bf[i & msk]);
    __syncthreads();
    fft_lines<true, false>(g.lines, g.tw, d.lgM, d.lgL, d.M, 1);
    for (int i = threadIdx.x; i < tot; i += blockDim.x) {
        const int k = i & msk;
        if (k < d.n) {
            const float2 v = cmul(g.lines[i], g.wch[k]);
            g.lines[i] = make_float2(v.x * inv, v.y * inv);
        }
    }
    __syncthreads();
}

// real rows (length w) -> half spectrum S[plane][h][w/2+1]
__global__ void k_gfft_rows_fwd(const float* __restrict__ in, const float* __restrict__ mul, float2* __restrict__ S, int h, int w, GDft d,
                                int force_real) {
    extern __shared__ float2 smem2[];
    const GLds g = gdft_carve(smem2, d);
    const size_t plane = blockIdx.x;
    const int row0 = blockIdx.y * d.L, halfw = w >> 1;
    gdft_setup(g, d);
    for (int i = threadIdx.x; i < d.L * w; i += blockDim.x) {
        const int l = i / w, x = i - l * w, row = row0 + l;
        float v = 0.f;
        if (row < h) {
            const size_t o = (plane * h + row) * w + x;
            v = in[o];
            if (mul) v *= mul[o];
        }
        g.lines[(l << d.lgM) + x] = make_float2(v, 0.0f);
    }
    __syncthreads();
    gdft_lines(g, d);
    for (int i = threadIdx.x; i < d.L * (halfw + 1); i += blockDim.x) {
        const int l = i / (halfw + 1), kx = i - l * (halfw + 1), row = row0 + l;
        if (row >= h) continue;
        float2 v = g.lines[(l << d.lgM) + kx];
        if (force_real && (kx == 0 || kx == halfw)) v.y = 0.0f;
        S[(plane * h + row) * (halfw + 1) + kx] = v;
    }
}

// columns (length h) of the half spectrum, in place: forward DFT, bin edit (forward or backward), inverse DFT
template <bool BWD>
__global__ void k_gfft_cols(FftArgs fa, FftBwdArgs ba, float2* __restrict__ S, int h, int w, GDft d) {
    extern __shared__ float2 smem2[];
    const GLds g = gdft_carve(smem2, d);          // line c = column kx0 + c, element = row
    const size_t plane = blockIdx.x;
    const int chn = BWD ? ba.ch : fa.ch;
    const int ch = (int)(plane % chn);
    const int halfw = w >> 1, kx0 = blockIdx.y * d.L;
    const int ncols = min(d.L, halfw + 1 - kx0);
    gdft_setup(g, d);
    for (int i = threadIdx.x; i < h * d.L; i += blockDim.x) {
        const int y = i >> d.lgL, c = i & (d.L - 1);
        g.lines[(c << d.lgM) + y] = (c < ncols) ? S[(plane * h + y) * (halfw + 1) + kx0 + c] : make_float2(0.f, 0.f);
    }
    __syncthreads();
    gdft_lines(g, d);
    if (!BWD) {   // the four purely-real bins: ky in {0, h/2} x kx in {0, w/2}
        if (threadIdx.x < 2 * d.L) {
            const int q = threadIdx.x >> d.lgL, c = threadIdx.x & (d.L - 1);
            if (c < ncols && (kx0 + c == 0 || kx0 + c == halfw)) g.lines[(c << d.lgM) + (q ? (h >> 1) : 0)].y = 0.0f;
        }
        __syncthreads();
    }
    const float aw = (BWD ? ba.ampw : fa.ampw)[ch], ab = (BWD ? ba.ampb : fa.ampb)[ch];
    const float pw = (BWD ? ba.phaw : fa.phaw)[ch], pb = (BWD ? ba.phab : fa.phab)[ch];
    const float nn = (float)h * (float)w;
    float s_aw = 0.f, s_ab = 0.f, s_pw = 0.f, s_pb = 0.f;
    for (int i = threadIdx.x; i < d.L * h; i += blockDim.x) {
        const int c = i / h, ky = i - c * h;
        if (c >= ncols) continue;
        const int kx = kx0 + c, idx = (c << d.lgM) + ky;
        const size_t o = (plane * h + ky) * (halfw + 1) + kx;
        float2 v;
        if (!BWD) {
            float amp, pha;
            v = bin_edit_fwd(g.lines[idx], aw, ab, pw, pb, amp, pha);
            if (fa.amp) { __builtin_nontemporal_store(amp, fa.amp + o); __builtin_nontemporal_store(pha, fa.pha + o); }
        } else {
            const float cw = (kx == 0 || kx == halfw) ? 1.0f : 2.0f;
            v = bin_edit_bwd(g.lines[idx], cw, nn, ba.amp[o], ba.pha[o], aw, ab, pw, pb, s_aw, s_ab, s_pw, s_pb);
        }
        g.lines[idx] = make_float2(v.x, -v.y);    // conj: the inverse transform is conj(DFT(conj(.)))
    }
    __syncthreads();
    gdft_lines(g, d);
    for (int i = threadIdx.x; i < h * d.L; i += blockDim.x) {
        const int y = i >> d.lgL, c = i & (d.L - 1);
        if (c < ncols) {
            const float2 v = g.lines[(c << d.lgM) + y];
            S[(plane * h + y) * (halfw + 1) + kx0 + c] = make_float2(v.x, -v.y);
        }
    }
    if (BWD) {
        float v[4] = {s_aw, s_ab, s_pw, s_pb};
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v[i] += __shfl_xor(v[i], off);
            if (lane == 0) g.red[wave * 4 + i] = v[i];
        }
        __syncthreads();
        if (threadIdx.x < 4) {
            float sum = 0.f;
            for (int k = 0; k < nw; ++k) sum += g.red[k * 4 + threadIdx.x];
            ba.part[(((plane / chn) * gridDim.y + blockIdx.y) * chn + ch) * 4 + threadIdx.x] = sum;   // [sample][column group][channel][4]
        }
    }
}

// half spectrum rows -> real rows (c2r)
__global__ void k_gfft_rows_inv(const float2* __restrict__ S, float* __restrict__ out, float* __restrict__ sgn, int h, int w, GDft d, int absout) {
    extern __shared__ float2 smem2[];
    const GLds g = gdft_carve(smem2, d);
    const size_t plane = blockIdx.x;
    const int row0 = blockIdx.y * d.L, halfw = w >> 1;
    gdft_setup(g, d);
    for (int i = threadIdx.x; i < d.L * (halfw + 1); i += blockDim.x) {
        const int l = i / (halfw + 1), kx = i - l * (halfw + 1), row = row0 + l;
        const float2 v = (row < h) ? S[(plane * h + row) * (halfw + 1) + kx] : make_float2(0.f, 0.f);
        float2* ln = g.lines + (l << d.lgM);
        if (kx == 0 || kx == halfw) {
            ln[kx] = make_float2(v.x, 0.0f);                 // c2r drops these imaginary parts
        } else {
            ln[kx] = make_float2(v.x, -v.y);                 // conj(X_kx)
            ln[w - kx] = v;                                  // conj(X_{w-kx}) = conj(conj(X_kx))
        }
    }
    __syncthreads();
    gdft_lines(g, d);
    const float sc = 1.0f / ((float)h * (float)w);
    for (int i = threadIdx.x; i < d.L * w; i += blockDim.x) {
        const int l = i / w, x = i - l * w, row = row0 + l;
        if (row >= h) continue;
        const float v = g.lines[(l << d.lgM) + x].x * sc;
        const size_t o = (plane * h + row) * w + x;
        if (absout) {
            out[o] = fabsf(v);
            if (sgn) sgn[o] = (v > 0.f) ? 1.0f : ((v < 0.f) ? -1.0f : 0.0f);
        } else {
            out[o] = v;
        }
    }
}

static int fft_generic_col_groups(int h, int w) { const GDft dh = gdft_plan(h); return (w / 2 + 1 + dh.L - 1) / dh.L; }

static int launch_fft_generic(const FftArgs* fa, const FftBwdArgs* ba, int h, int w, hipStream_t s) {
    const int planes = fa ? fa->planes : ba->planes;
    float2* S = reinterpret_cast<float2*>(fa ? fa->scratch : ba->scratch);
    if (!S) { lg_set_error("fftmix: plane %dx%d needs the generic path but no scratch buffer was given", h, w); return -2; }
    if (h < 8 || w < 8 || h > 1024 || w > 1024 || (h & 7) || (w & 7)) {
        lg_set_error("fftmix: plane %dx%d unsupported (sides: multiples of 8, 8..1024)", h, w);
        return -2;
    }
    static DeviceOnce attr_once;
    if (attr_once.need()) {
        const void* fns[4] = {(const void*)k_gfft_rows_fwd, (const void*)k_gfft_cols<false>, (const void*)k_gfft_cols<true>, (const void*)k_gfft_rows_inv};
        for (int i = 0; i < 4; ++i) {
            hipError_t e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
            if (e != hipSuccess) { lg_set_error("fft generic: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
        }
        attr_once.done();
    }
    const GDft dw = gdft_plan(w), dh = gdft_plan(h);
    const size_t lds_r = gdft_lds_bytes(dw), lds_c = gdft_lds_bytes(dh);
    dim3 grows(planes, (h + dw.L - 1) / dw.L), gcols(planes, fft_generic_col_groups(h, w));
    if (fa) {
        k_gfft_rows_fwd<<<grows, 1024, lds_r, s>>>(fa->g, nullptr, S, h, w, dw, 1);
        LG_CHECK_LAUNCH();
        FftBwdArgs dummy;
        memset(&dummy, 0, sizeof(dummy));
        k_gfft_cols<false><<<gcols, 1024, lds_c, s>>>(*fa, dummy, S, h, w, dh);
        LG_CHECK_LAUNCH();
        k_gfft_rows_inv<<<grows, 1024, lds_r, s>>>(S, fa->o, fa->sgn, h, w, dw, 1);
        LG_CHECK_LAUNCH();
    } else {
        k_gfft_rows_fwd<<<grows, 1024, lds_r, s>>>(ba->do2, ba->sgn, S, h, w, dw, 0);
        LG_CHECK_LAUNCH();
        FftArgs dummy;
        memset(&dummy, 0, sizeof(dummy));
        k_gfft_cols<true><<<gcols, 1024, lds_c, s>>>(dummy, *ba, S, h, w, dh);
        LG_CHECK_LAUNCH();
        k_gfft_rows_inv<<<grows, 1024, lds_r, s>>>(S, ba->dg, nullptr, h, w, dw, 0);
        LG_CHECK_LAUNCH();
    }
    return 0;
}

int launch_fftmix(const FftArgs& a, hipStream_t s) {
    ProfScope prof__(LG_K_FFT, s);
    const int ph = a.h ? a.h : a.n, pw = a.w ? a.w : a.n;
    if (fft_is_generic(ph, pw)) return launch_fft_generic(&a, nullptr, ph, pw, s);
    int n = ph, lg = 0;
    while ((1 << lg) < n) ++lg;
    if (n > 128) return launch_fft_split(&a, nullptr, s);
    if (!a.full) {
        if (int rc = fft_const_twiddles()) return rc;
        static DeviceOnce attr_r;
        if (attr_r.need()) {
            hipError_t e = hipFuncSetAttribute((const void*)k_fftmix_r<7, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024 - 512);
            if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_fftmix_r<7, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024 - 512);
            if (e != hipSuccess) { lg_set_error("fftmix: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
            attr_r.done();
        }
        const size_t ldsr = ((size_t)n * FFT_HP(n) + n / 2) * sizeof(float2);
        switch (lg) {
            case 3: k_fftmix_r<3, fftr_threads(3)><<<a.planes, fftr_threads(3), ldsr, s>>>(a); break;
            case 4: k_fftmix_r<4, fftr_threads(4)><<<a.planes, fftr_threads(4), ldsr, s>>>(a); break;
            case 5: k_fftmix_r<5, fftr_threads(5)><<<a.planes, fftr_threads(5), ldsr, s>>>(a); break;
            case 6: k_fftmix_r<6, fftr_threads(6)><<<a.planes, fftr_threads(6), ldsr, s>>>(a); break;
            default:
                if (a.planes <= lg_num_cus()) k_fftmix_r<7, 1024><<<a.planes, 1024, ldsr, s>>>(a);
                else k_fftmix_r<7, 512><<<a.planes, 512, ldsr, s>>>(a);
                break;
        }
        LG_CHECK_LAUNCH();
        return 0;
    }
    size_t lds = ((size_t)n * FFT_LD(n) + n / 2) * sizeof(float2);
    static DeviceOnce attr_once;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)k_fftmix<7>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
        if (e != hipSuccess) { lg_set_error("fftmix: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
        attr_once.done();
    }
    int threads = fft_threads(lg);
    switch (lg) {
        case 3: k_fftmix<3><<<a.planes, threads, lds, s>>>(a); break;
        case 4: k_fftmix<4><<<a.planes, threads, lds, s>>>(a); break;
        case 5: k_fftmix<5><<<a.planes, threads, lds, s>>>(a); break;
        case 6: k_fftmix<6><<<a.planes, threads, lds, s>>>(a); break;
        default: k_fftmix<7><<<a.planes, threads, lds, s>>>(a); break;
    }
    LG_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// backward (autograd of LGT.py:149-180):  dt = do * sign(t);  dY = (c_kx / n^2) rfft2(dt)  [adjoint of the c2r irfft2];
// through re/im = amp' (cos,sin)(pha') and amp' = aw*amp + ab, pha' = pw*pha + pb (parameter gradients);
// through amp = |F|, pha = angle(F);  dg = Re sum_{kx<=n/2} dF e^{+i theta}  [adjoint of the r2c rfft2], which is the
// same c2r machinery applied to dF * n^2 / c_kx.
// ------------------------------------------------------------------------------------------------
template <int LG>
__global__ void k_fftmix_bwd(FftBwdArgs a) {
    extern __shared__ float2 smem2[];
    constexpr int lg = LG, n = 1 << LG, half = n >> 1;
    constexpr int LD = FFT_LD(n);
    float2* buf = smem2;           // [n][LD]
    float2* tw = smem2 + n * LD;
    float* red = reinterpret_cast<float*>(tw + half);  // [16][4]
    const int plane = blockIdx.x;
    const int ch = plane % a.ch;
    const size_t base = (size_t)plane * n * n;
    for (int k = threadIdx.x; k < half; k += blockDim.x) {
        float ang = 2.0f * (float)k / (float)n;
        tw[k] = make_float2(cospif(ang), 0.0f - sinpif(ang));
    }
    // Every global operand of a thread is requested up front (round 4): the plane (do2, sgn) before the first store into LDS, and the saved
    // spectrum (amp, pha) of the thread's bins in one batch in front of the edit.  As loops over blockDim the plane was 4 and the spectrum 9
    // dependent HBM round trips (load, wait, use per trip) in a 35 us kernel.
    constexpr int NTH = fft_threads(LG), NLD = (n * n / 4 + NTH - 1) / NTH, NBIN = n * (half + 1), NIT = (NBIN + NTH - 1) / NTH;
    {
        float4 u[NLD], sg[NLD];
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int i = (k * NTH + (int)threadIdx.x) * 4, ic = i < n * n ? i : 0;
            u[k] = *reinterpret_cast<const float4*>(a.do2 + base + ic);
            sg[k] = *reinterpret_cast<const float4*>(a.sgn + base + ic);
        }
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int i = (k * NTH + (int)threadIdx.x) * 4;
            if (i < n * n) {
                float2* d = buf + (i >> lg) * LD + (i & (n - 1));
                d[0] = make_float2(u[k].x * sg[k].x, 0.0f); d[1] = make_float2(u[k].y * sg[k].y, 0.0f);
                d[2] = make_float2(u[k].z * sg[k].z, 0.0f); d[3] = make_float2(u[k].w * sg[k].w, 0.0f);
            }
        }
    }
    __syncthreads();
    fft_pass<false, false>(buf, tw, n, lg);
    fft_pass<false, true>(buf, tw, n, lg);
    const float aw = a.ampw[ch], ab = a.ampb[ch], pw = a.phaw[ch], pb = a.phab[ch];
    const float nn = (float)n * (float)n;
    float s_aw = 0.f, s_ab = 0.f, s_pw = 0.f, s_pb = 0.f;
    // (requested here, in ONE batch, not in front of the transform: 16 waves per CU leave 128 registers per lane and the butterflies use them)
    float ampv[NIT], phav[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int it = k * NTH + (int)threadIdx.x, itc = it < NBIN ? it : 0;
        int q, p, c;
        half_bin(itc, n, lg, q, p, c);
        const size_t o = ((size_t)plane * n + q) * (half + 1) + c;
        ampv[k] = a.amp[o];
        phav[k] = a.pha[o];
    }
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int it = k * NTH + (int)threadIdx.x;
        if (it < NBIN) {
            int q, p, c;
            half_bin(it, n, lg, q, p, c);
            const float cf = (c == 0 || c == half) ? 1.0f : 2.0f;
            buf[q * LD + p] = bin_edit_bwd(buf[q * LD + p], cf, nn, ampv[k], phav[k], aw, ab, pw, pb, s_aw, s_ab, s_pw, s_pb);
        }
    }
    __syncthreads();
    fft_pass<true, true>(buf, tw, n, lg);
    fft_pass<true, false>(buf, tw, n, lg);
    const float sc = 1.0f / nn;
    for (int i = threadIdx.x * 4; i < n * n; i += blockDim.x * 4) {
        const float2* r = buf + (i >> lg) * LD + (i & (n - 1));
        *reinterpret_cast<float4*>(a.dg + base + i) = make_float4(r[0].x * sc, r[1].x * sc, r[2].x * sc, r[3].x * sc);
    }
    // parameter gradient partials
    float v[4] = {s_aw, s_ab, s_pw, s_pb};
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[i] += __shfl_xor(v[i], off);
        if (lane == 0) red[wave * 4 + i] = v[i];
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        float s = 0.f;
        for (int w = 0; w < nw; ++w) s += red[w * 4 + threadIdx.x];
        a.part[(size_t)plane * 4 + threadIdx.x] = s;   // [sample][channel][4] partial row (plane = sample * ch + channel)
    }
}

// real-input form of the backward (see k_fftmix_r): the same two real <-> half-spectrum row steps around the same bin edit
template <int LG, int NTH>
__global__ void __launch_bounds__(NTH, (LG >= 7 && NTH == 512) ? 4 : 1) k_fftmix_bwd_r(FftBwdArgs a) {
    extern __shared__ float2 smem2[];
    constexpr int lg = LG, n = 1 << LG, half = n >> 1, HP = FFT_HP(n);
    float2* buf = smem2;           // [n][HP]
    float2* tw = smem2 + n * HP;
    float* red = reinterpret_cast<float*>(tw + half);  // [waves][4]
    const int plane = blockIdx.x;
    const int ch = plane % a.ch;
    const size_t base = (size_t)plane * n * n;
    constexpr int NLD = (n * n / 4 + NTH - 1) / NTH, NBIN = n * (half + 1), NIT = (NBIN + NTH - 1) / NTH;
    constexpr int LB = NLD > 4 ? 4 : NLD;     // loads in batches of four (do2 and sgn: 32 registers in flight)
#pragma unroll
    for (int b = 0; b < NLD; b += LB) {
        float4 u[LB], sg[LB];
#pragma unroll
        for (int k = 0; k < LB; ++k) {
            const int i = ((b + k) * NTH + (int)threadIdx.x) * 4, ic = i < n * n ? i : 0;
            u[k] = *reinterpret_cast<const float4*>(a.do2 + base + ic);
            sg[k] = *reinterpret_cast<const float4*>(a.sgn + base + ic);
        }
        if (b == 0) fft_tw_from_const(tw, lg);
#pragma unroll
        for (int k = 0; k < LB; ++k) {
            const int i = ((b + k) * NTH + (int)threadIdx.x) * 4;
            if (i >= n * n) continue;
            float2* d = buf + (i >> lg) * HP + ((i & (n - 1)) >> 1);
            d[0] = make_float2(u[k].x * sg[k].x, u[k].y * sg[k].y); d[1] = make_float2(u[k].z * sg[k].z, u[k].w * sg[k].w);
        }
    }
    __syncthreads();
    fftr_rows_fwd<LG>(buf, tw);
    fftr_cols<false, LG>(buf, tw);
    const float aw = a.ampw[ch], ab = a.ampb[ch], pw = a.phaw[ch], pb = a.phab[ch];
    const float nn = (float)n * (float)n;
    float s_aw = 0.f, s_ab = 0.f, s_pw = 0.f, s_pb = 0.f;
    constexpr int EB = NIT > 9 ? 9 : NIT;     // saved amplitude / phase of the thread's bins in batches of nine
#pragma unroll 1
    for (int b = 0; b < NIT; b += EB) {
        float ampv[EB], phav[EB];
#pragma unroll
        for (int k = 0; k < EB; ++k) {
            const int it = (b + k) * NTH + (int)threadIdx.x, itc = it < NBIN ? it : 0;
            int q, c;
            fftr_bin(itc, n, lg, q, c);
            const size_t o = ((size_t)plane * n + q) * (half + 1) + c;
            ampv[k] = a.amp[o];
            phav[k] = a.pha[o];
        }
#pragma unroll
        for (int k = 0; k < EB; ++k) {
            const int it = (b + k) * NTH + (int)threadIdx.x;
            if (it < NBIN) {
                int q, c;
                fftr_bin(it, n, lg, q, c);
                const float cf = (c == 0 || c == half) ? 1.0f : 2.0f;
                buf[q * HP + c] = bin_edit_bwd(buf[q * HP + c], cf, nn, ampv[k], phav[k], aw, ab, pw, pb, s_aw, s_ab, s_pw, s_pb);
            }
        }
    }
    __syncthreads();
    fftr_cols<true, LG>(buf, tw);
    fftr_rows_inv<LG>(buf, tw);
    const float sc = 1.0f / nn;
    int tid_s = threadIdx.x;
    asm volatile("" : "+v"(tid_s));      // (see k_fftmix_r)
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int i = (k * NTH + tid_s) * 4;
        if (i >= n * n) break;
        const float2* r = buf + (i >> lg) * HP + ((i & (n - 1)) >> 1);
        const float2 r0 = r[0], r1 = r[1];
        *reinterpret_cast<float4*>(a.dg + base + i) = make_float4(r0.x * sc, r0.y * sc, r1.x * sc, r1.y * sc);
    }
    // parameter gradient partials
    float v[4] = {s_aw, s_ab, s_pw, s_pb};
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = NTH >> 6;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[i] += __shfl_xor(v[i], off);
        if (lane == 0) red[wave * 4 + i] = v[i];
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        float s = 0.f;
        for (int w = 0; w < nw; ++w) s += red[w * 4 + threadIdx.x];
        a.part[(size_t)plane * 4 + threadIdx.x] = s;
    }
}

static int fft_bwd_col_groups(int h, int w) {
    if (fft_is_generic(h, w)) return fft_generic_col_groups(h, w);
    return h > 128 ? (h / 2 + 1 + FFT_COLS_PER_WG(h) - 1) / FFT_COLS_PER_WG(h) : 1;
}
size_t fft_bwd_part_floats(int planes, int h, int w) { return (size_t)planes * fft_bwd_col_groups(h, w) * 4; }
// the four per-channel parameter gradients from the partial rows [sample][column group][channel][4]
static int fft_bwd_reduce(const FftBwdArgs& a, hipStream_t s) {
    float* dst[4] = {a.d_ampw, a.d_ampb, a.d_phaw, a.d_phab};
    for (int k = 0; k < 4; ++k) {
        ReduceJob j;
        j.slab = a.part + k; j.dst = dst[k]; j.dst2 = nullptr;
        j.nslices = (long)(a.planes / a.ch) * fft_bwd_col_groups(a.h ? a.h : a.n, a.w ? a.w : a.n); j.slice_stride = (long)a.ch * 4;
        j.rows = a.ch; j.cols = 1; j.row_stride = 4; j.ld = 1; j.rows_valid = a.ch; j.cols_valid = 1;
        int rc = launch_reduce_job(j, s);
        if (rc) return rc;
    }
    return 0;
}
static int launch_fftmix_bwd_kernels(const FftBwdArgs& a, hipStream_t s);
int launch_fftmix_bwd(const FftBwdArgs& a, hipStream_t s) {
    if (!a.part) { lg_set_error("fftmix_bwd: partial-sum scratch missing"); return -2; }
    int rc = launch_fftmix_bwd_kernels(a, s);
    return rc ? rc : fft_bwd_reduce(a, s);
}
static int launch_fftmix_bwd_kernels(const FftBwdArgs& a, hipStream_t s) {
    ProfScope prof__(LG_K_FFT_BWD, s);
    const int ph = a.h ? a.h : a.n, pw = a.w ? a.w : a.n;
    if (fft_is_generic(ph, pw)) return launch_fft_generic(nullptr, &a, ph, pw, s);
    int n = ph, lg = 0;
    while ((1 << lg) < n) ++lg;
    if (n > 128) return launch_fft_split(nullptr, &a, s);
    if (!a.full) {
        if (int rc = fft_const_twiddles()) return rc;
        static DeviceOnce attr_r;
        if (attr_r.need()) {
            hipError_t e = hipFuncSetAttribute((const void*)k_fftmix_bwd_r<7, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024 - 512);
            if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_fftmix_bwd_r<7, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024 - 512);
            if (e != hipSuccess) { lg_set_error("fftmix_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
            attr_r.done();
        }
        const size_t ldsr = ((size_t)n * FFT_HP(n) + n / 2) * sizeof(float2) + 64 * sizeof(float);
        switch (lg) {
            case 3: k_fftmix_bwd_r<3, fftr_threads(3)><<<a.planes, fftr_threads(3), ldsr, s>>>(a); break;
            case 4: k_fftmix_bwd_r<4, fftr_threads(4)><<<a.planes, fftr_threads(4), ldsr, s>>>(a); break;
            case 5: k_fftmix_bwd_r<5, fftr_threads(5)><<<a.planes, fftr_threads(5), ldsr, s>>>(a); break;
            case 6: k_fftmix_bwd_r<6, fftr_threads(6)><<<a.planes, fftr_threads(6), ldsr, s>>>(a); break;
            default:
                if (a.planes <= lg_num_cus()) k_fftmix_bwd_r<7, 1024><<<a.planes, 1024, ldsr, s>>>(a);
                else k_fftmix_bwd_r<7, 512><<<a.planes, 512, ldsr, s>>>(a);
                break;
        }
        LG_CHECK_LAUNCH();
        return 0;
    }
    size_t lds = ((size_t)n * FFT_LD(n) + n / 2) * sizeof(float2) + 64 * sizeof(float);
    static DeviceOnce attr_once;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)k_fftmix_bwd<7>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
        if (e != hipSuccess) { lg_set_error("fftmix_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
        attr_once.done();
    }
    int threads = fft_threads(lg);
    switch (lg) {
        case 3: k_fftmix_bwd<3><<<a.planes, threads, lds, s>>>(a); break;
        case 4: k_fftmix_bwd<4><<<a.planes, threads, lds, s>>>(a); break;
        case 5: k_fftmix_bwd<5><<<a.planes, threads, lds, s>>>(a); break;
        case 6: k_fftmix_bwd<6><<<a.planes, threads, lds, s>>>(a); break;
        default: k_fftmix_bwd<7><<<a.planes, threads, lds, s>>>(a); break;
    }
    LG_CHECK_LAUNCH();
    return 0;
}
