// Standalone check + timing of the in-LDS FFT mixer kernels: the real-input form (k_fftmix_r) against the complex-row form (k_fftmix) on random planes,
// and (-DLG_FFT_STAMPS) the cycle counts between the phase boundaries of k_fftmix_r.
//   hipcc -O3 -std=c++17 -fno-slp-vectorize --offload-arch=gfx950 [-DLG_FFT_STAMPS] -Ilgteun_amd/csrc tools/micro/fft_check.hip -o /tmp/fft_check
//   /tmp/fft_check [n=128] [planes=256] [reps=50]
#include "../../lgteun_amd/csrc/k_fft.hip"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <cstdarg>
void lg_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
void lg_prof_begin(int, hipStream_t) {}
void lg_prof_end(int, hipStream_t) {}
int launch_reduce_job(const ReduceJob&, hipStream_t) { return 0; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
static uint64_t rs = 0x1234567ull;
static double urand() { rs = rs * 6364136223846793005ull + 1442695040888963407ull; return (double)(rs >> 11) / 9007199254740992.0; }
static double nrand() { double u = urand() + 1e-12, v = urand(); return sqrt(-2 * log(u)) * cos(6.283185307179586 * v); }
template <class T> static T* dev(const std::vector<T>& h) { T* d; CK(hipMalloc(&d, h.size() * sizeof(T))); CK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); return d; }

// accuracy of the edit's short-form functions and of the library's against double precision, in units of the last place of the exact value
__global__ void k_edit_acc(const float2* in, int n, unsigned* worst) {   // worst[0..5]: 1000 x max ulp error of {amp, pha, sin, cos} ours, then library {pha, sin}
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float2 f = in[i];
    float amp, pha, sn, cs;
    edit_abs_angle(f, amp, pha);
    const float x = f.x * 3.0f;      // sincos argument: |x| up to ~ 4 pi and beyond
    edit_sincos(x, sn, cs);
    const double da = hypot((double)f.x, (double)f.y), dp = atan2((double)f.y, (double)f.x), ds = sin((double)x), dc = cos((double)x);
    auto ulps = [](float got, double want) { const float w = (float)want; const double u = fmax((double)fabsf(w) * 1.1920929e-7 * 0.5, 1e-45); return (unsigned)fmin(4e9, fabs((double)got - want) / u * 1000.0); };
    // (half an epsilon of |want| underestimates the ulp by up to 2x just above a power of two: these are upper bounds on the error in ulps)
    atomicMax(worst + 0, ulps(amp, da)); atomicMax(worst + 1, ulps(pha, dp));
    // sin / cos: absolute error against the ulp of 1 (near a zero of sin the relative error of ANY reduced-argument form is unbounded)
    atomicMax(worst + 2, (unsigned)(fabs((double)sn - ds) / 5.96e-8 * 1000.0)); atomicMax(worst + 3, (unsigned)(fabs((double)cs - dc) / 5.96e-8 * 1000.0));
    float ls, lc; sincosf(x, &ls, &lc);
    atomicMax(worst + 4, ulps(atan2f(f.y, f.x), dp)); atomicMax(worst + 5, (unsigned)(fabs((double)ls - ds) / 5.96e-8 * 1000.0));
    atomicMax(worst + 6, ulps(hypotf(f.x, f.y), da));
}
static void edit_accuracy() {
    const int n = 1 << 22;
    std::vector<float2> h(n);
    for (int i = 0; i < n; ++i) {
        const double mag = pow(10.0, 8.0 * urand() - 5.0);
        double x = nrand() * mag, y = nrand() * mag;
        if (i % 17 == 0) y = (i & 1) ? 0.0 : -0.0;
        if (i % 19 == 0) x = (i & 1) ? 0.0 : -0.0;
        if (i % 23 == 0) y = x;
        if (i % 29 == 0) y = x * 1e-6;
        h[i] = make_float2((float)x, (float)y);
    }
    float2* d = dev(h);
    unsigned* w; CK(hipMalloc(&w, 8 * 4)); CK(hipMemset(w, 0, 8 * 4));
    k_edit_acc<<<n / 256, 256>>>(d, n, w);
    unsigned hw[8]; CK(hipMemcpy(hw, w, 32, hipMemcpyDeviceToHost));
    printf("edit functions vs double (max error, units of the last place): |f| ours %.2f library %.2f; angle ours %.2f library %.2f; sin ours %.2f library %.2f; cos ours %.2f\n",
           hw[0] / 1000.0, hw[6] / 1000.0, hw[1] / 1000.0, hw[4] / 1000.0, hw[2] / 1000.0, hw[5] / 1000.0, hw[3] / 1000.0);
}

int main(int argc, char** argv) {
    if (argc > 1 && atoi(argv[1]) == 0) { edit_accuracy(); return 0; }
    const int n = argc > 1 ? atoi(argv[1]) : 128, planes = argc > 2 ? atoi(argv[2]) : 256, reps = argc > 3 ? atoi(argv[3]) : 50;
    const int ch = 8;
    const size_t N = (size_t)planes * n * n, NB = (size_t)planes * n * (n / 2 + 1);
    std::vector<float> g(N), aw(ch), ab(ch), pw(ch), pb(ch);
    for (auto& v : g) v = (float)nrand();
    for (int c = 0; c < ch; ++c) { aw[c] = (float)(1 + 0.1 * nrand()); ab[c] = (float)(0.05 * nrand()); pw[c] = (float)(1 + 0.1 * nrand()); pb[c] = (float)(0.05 * nrand()); }
    float *dg = dev(g), *daw = dev(aw), *dab = dev(ab), *dpw = dev(pw), *dpb = dev(pb);
    float *o[2], *amp[2], *pha[2], *sgn[2];
    for (int k = 0; k < 2; ++k) { CK(hipMalloc(&o[k], N * 4)); CK(hipMalloc(&sgn[k], N * 4)); CK(hipMalloc(&amp[k], NB * 4)); CK(hipMalloc(&pha[k], NB * 4)); }
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int k = 0; k < 2; ++k) {
        FftArgs a; a.g = dg; a.o = o[k]; a.amp = amp[k]; a.pha = pha[k]; a.sgn = sgn[k]; a.scratch = nullptr; a.ampw = daw; a.ampb = dab; a.phaw = dpw; a.phab = dpb;
        a.planes = planes; a.ch = ch; a.n = n; a.h = n; a.w = n; a.full = k == 0;
        for (int r = 0; r < 3; ++r) if (launch_fftmix(a, s)) return 1;
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < reps; ++r) launch_fftmix(a, s);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s n=%d planes=%d: %.2f us per launch\n", k == 0 ? "k_fftmix  (complex rows)" : "k_fftmix_r (real input) ", n, planes, ms * 1e3 / reps);
    }
    std::vector<float> h0(N), h1(N), a0(NB), a1(NB);
    CK(hipMemcpy(h0.data(), o[0], N * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), o[1], N * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(a0.data(), amp[0], NB * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(a1.data(), amp[1], NB * 4, hipMemcpyDeviceToHost));
    double num = 0, den = 0, mx = 0, anum = 0, aden = 0;
    for (size_t i = 0; i < N; ++i) { const double d = (double)h0[i] - h1[i]; num += d * d; den += (double)h0[i] * h0[i]; mx = fmax(mx, fabs(d)); }
    for (size_t i = 0; i < NB; ++i) { const double d = (double)a0[i] - a1[i]; anum += d * d; aden += (double)a0[i] * a0[i]; }
    printf("output: rel l2 of the difference %.3e, max abs %.3e;  saved amplitude: rel l2 %.3e\n", sqrt(num / den), mx, sqrt(anum / aden));
#ifdef LG_FFT_STAMPS
    static unsigned long long st[256][12];
    CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(fft_stamps), sizeof(st)));
    const char* nm[7] = {"load + store to LDS", "rows forward (n/2-point + unpack)", "columns forward", "bin edit", "columns inverse", "rows inverse (pack + n/2-point)", "store"};
    for (int i = 0; i < 7; ++i) {
        double sum = 0; int cnt = 0;
        for (int b = 0; b < 256 && b < planes; ++b) { sum += (double)(st[b][i + 1] - st[b][i]); ++cnt; }
        printf("  %-36s %8.0f cycles (s_memtime, 100 MHz ticks x ?)\n", nm[i], sum / cnt);
    }
    printf("  total %8.0f\n", (double)(st[0][7] - st[0][0]));
#endif
    return 0;
}
