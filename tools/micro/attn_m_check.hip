// Standalone check + timing of the matrix-pipe window-attention kernel (k_attn_m.hip) against a plain double-precision CPU
// restatement written here (LGT.py:112-146,183-219: LayerNorm -> local window MSA | FFT half as given -> proj -> + x) and against
// round 2's vector-pipe kernel (k_attn.hip).
//   hipcc -O3 -std=c++17 -fno-slp-vectorize --offload-arch=gfx950 -Ilgteun_amd/csrc tools/micro/attn_m_check.hip -o /tmp/attn_m_check
//   /tmp/attn_m_check [HC=8] [B=2] [H=32] [W=32] [bf16=0] [reps=0]
#include "../../lgteun_amd/csrc/k_attn.hip"
#include "../../lgteun_amd/csrc/k_attn_m.hip"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <cstdarg>
#include <cstring>
void lg_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
void lg_prof_begin(int, hipStream_t) {}
void lg_prof_end(int, hipStream_t) {}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

static uint64_t rs = 0x1234567ull;
static double urand() { rs = rs * 6364136223846793005ull + 1442695040888963407ull; return (double)(rs >> 11) / 9007199254740992.0; }
static double nrand() { double u = urand() + 1e-12, v = urand(); return sqrt(-2 * log(u)) * cos(6.283185307179586 * v); }

template <class T> static T* dev(const std::vector<T>& h) { T* d; CK(hipMalloc(&d, h.size() * sizeof(T))); CK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); return d; }

int main(int argc, char** argv) {
    const int HC = argc > 1 ? atoi(argv[1]) : 8, B = argc > 2 ? atoi(argv[2]) : 2, H = argc > 3 ? atoi(argv[3]) : 32, W = argc > 4 ? atoi(argv[4]) : 32;
    const int bf = argc > 5 ? atoi(argv[5]) : 0, reps = argc > 6 ? atoi(argv[6]) : 0;
    const int E = 2 * HC, D = HC / 2;
    const long P = (long)B * H * W;
    std::vector<float> x(P * E), o2(P * HC), pos(2 * 64 * 64), posT(2 * 64 * 64), g(E), bt(E), wq(3 * HC * HC), bq(3 * HC), wp(E * E), bp(E);
    for (auto& v : x) v = (float)(nrand() * 1.5 + 0.3);
    for (auto& v : o2) v = (float)fabs(nrand());
    for (auto& v : pos) v = (float)nrand();
    for (int h = 0; h < 2; ++h) for (int i = 0; i < 64; ++i) for (int j = 0; j < 64; ++j) posT[(h * 64 + j) * 64 + i] = pos[(h * 64 + i) * 64 + j];
    for (auto& v : g) v = (float)(1.0 + 0.2 * nrand());
    for (auto& v : bt) v = (float)(0.1 * nrand());
    for (auto& v : wq) v = (float)(nrand() * 1.2 / sqrt((double)HC));
    for (auto& v : bq) v = (float)(0.3 * nrand());
    for (auto& v : wp) v = (float)(nrand() / sqrt((double)E));
    for (auto& v : bp) v = (float)(0.1 * nrand());
    // ---- CPU restatement in double
    std::vector<double> ref(P * E);
    {
        std::vector<double> q(64 * HC), k(64 * HC), v(64 * HC), o1(64 * HC);
        for (int b = 0; b < B; ++b) for (int wy = 0; wy < H / 8; ++wy) for (int wx = 0; wx < W / 8; ++wx) {
            for (int t = 0; t < 64; ++t) {
                const long p = ((long)b * H + wy * 8 + t / 8) * W + wx * 8 + t % 8;
                double mu = 0, var = 0;
                for (int c = 0; c < E; ++c) mu += x[p * E + c];
                mu /= E;
                for (int c = 0; c < E; ++c) var += (x[p * E + c] - mu) * (x[p * E + c] - mu);
                const double rstd = 1.0 / sqrt(var / E + 1e-5);
                double y[64];
                for (int c = 0; c < HC; ++c) y[c] = (x[p * E + c] - mu) * rstd * g[c] + bt[c];
                for (int c = 0; c < HC; ++c) {
                    double aq = bq[c], ak = bq[HC + c], av = bq[2 * HC + c];
                    for (int kk = 0; kk < HC; ++kk) { aq += (double)wq[c * HC + kk] * y[kk]; ak += (double)wq[(HC + c) * HC + kk] * y[kk]; av += (double)wq[(2 * HC + c) * HC + kk] * y[kk]; }
                    q[t * HC + c] = aq / sqrt((double)D); k[t * HC + c] = ak; v[t * HC + c] = av;
                }
            }
            for (int h = 0; h < 2; ++h) for (int i = 0; i < 64; ++i) {
                double s[64], mx = -1e300, l = 0;
                for (int j = 0; j < 64; ++j) {
                    double a = pos[(h * 64 + i) * 64 + j];
                    for (int d = 0; d < D; ++d) a += q[i * HC + h * D + d] * k[j * HC + h * D + d];
                    s[j] = a; mx = fmax(mx, a);
                }
                for (int j = 0; j < 64; ++j) { s[j] = exp(s[j] - mx); l += s[j]; }
                for (int d = 0; d < D; ++d) {
                    double a = 0;
                    for (int j = 0; j < 64; ++j) a += s[j] * v[j * HC + h * D + d];
                    o1[i * HC + h * D + d] = a / l;
                }
            }
            for (int t = 0; t < 64; ++t) {
                const int yy = wy * 8 + t / 8, xx = wx * 8 + t % 8;
                const long p = ((long)b * H + yy) * W + xx;
                for (int n = 0; n < E; ++n) {
                    double a = bp[n];
                    for (int c = 0; c < HC; ++c) a += (double)wp[n * E + c] * o1[t * HC + c];
                    for (int c = 0; c < HC; ++c) a += (double)wp[n * E + HC + c] * o2[((long)b * HC + c) * H * W + yy * W + xx];
                    ref[p * E + n] = x[p * E + n] + a;
                }
            }
        }
    }
    AttnArgs a;
    float *dy, *dy0;
    CK(hipMalloc(&dy, P * E * sizeof(float))); CK(hipMalloc(&dy0, P * E * sizeof(float)));
    a.x = dev(x); a.o2 = dev(o2); a.y = dy; a.posT = dev(posT); a.pos = dev(pos);
    a.ln1g = dev(g); a.ln1b = dev(bt); a.qkvw = dev(wq); a.qkvb = dev(bq); a.projw = dev(wp); a.projb = dev(bp);
    a.B = B; a.h = H; a.w = W; a.dropout = 0; a.seed = 0x1234; a.bf16 = bf;
    CK(hipMemset(dy, 0xff, P * E * sizeof(float)));
    if (launch_attn_m(E, a, 0)) return 2;
    CK(hipDeviceSynchronize());
    std::vector<float> y(P * E), y0(P * E);
    CK(hipMemcpy(y.data(), dy, P * E * sizeof(float), hipMemcpyDeviceToHost));
    a.y = dy0;
    if (launch_attn(E, a, 0)) return 2;
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(y0.data(), dy0, P * E * sizeof(float), hipMemcpyDeviceToHost));
    double e1 = 0, e0 = 0, nr = 0, m1 = 0, m0 = 0;
    long bad = -1;
    for (long i = 0; i < P * E; ++i) {
        const double r = ref[i] - x[i];   // compare the mixer's contribution (the residual x would hide its error)
        const double d1 = (double)y[i] - ref[i], d0 = (double)y0[i] - ref[i];
        if (!(fabs(d1) < 1e30)) { if (bad < 0) bad = i; continue; }
        e1 += d1 * d1; e0 += d0 * d0; nr += r * r; m1 = fmax(m1, fabs(d1)); m0 = fmax(m0, fabs(d0));
    }
    printf("HC=%d B=%d %dx%d bf16=%d: k_attn_m relL2 %.3e max %.3e | k_attn (vector pipe) relL2 %.3e max %.3e | first non-finite %ld\n", HC, B, H, W, bf,
           sqrt(e1 / nr), m1, sqrt(e0 / nr), m0, bad);
    if (bad >= 0 || sqrt(e1 / nr) > (bf ? 3e-2 : 2e-6)) {
        int shown = 0;
        for (long i = 0; i < P * E && shown < 24; ++i) if (!(fabs((double)y[i] - ref[i]) < 1e-4 * (bf ? 1e3 : 1))) { printf("  [pix %ld ch %ld] got %g want %g (old %g)\n", i / E, i % E, y[i], ref[i], y0[i]); ++shown; }
    }
    // dropout: the two kernels must agree on the mask
    a.dropout = 1; a.y = dy;
    if (launch_attn_m(E, a, 0)) return 2;
    a.y = dy0;
    if (launch_attn(E, a, 0)) return 2;
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(y.data(), dy, P * E * sizeof(float), hipMemcpyDeviceToHost));
    CK(hipMemcpy(y0.data(), dy0, P * E * sizeof(float), hipMemcpyDeviceToHost));
    long dropped = 0, mism = 0; double md = 0;
    for (long i = 0; i < P * E; ++i) {
        const bool z1 = y[i] == x[i], z0 = y0[i] == x[i];
        dropped += z1; mism += (z1 != z0);
        md = fmax(md, fabs((double)y[i] - y0[i]));
    }
    printf("dropout: %.4f dropped, %ld mask mismatches, max |new - old| %.3e\n", (double)dropped / (P * E), mism, md);
    if (reps > 0) {
        hipEvent_t e0_, e1_; CK(hipEventCreate(&e0_)); CK(hipEventCreate(&e1_));
        for (int which = 0; which < 2; ++which) {
            a.y = which ? dy : dy0; a.dropout = 1;
            for (int i = 0; i < 3; ++i) which ? launch_attn_m(E, a, 0) : launch_attn(E, a, 0);
            CK(hipEventRecord(e0_, 0));
            for (int i = 0; i < reps; ++i) which ? launch_attn_m(E, a, 0) : launch_attn(E, a, 0);
            CK(hipEventRecord(e1_, 0)); CK(hipEventSynchronize(e1_));
            float ms; CK(hipEventElapsedTime(&ms, e0_, e1_));
            printf("%s: %.2f us per launch\n", which ? "k_attn_m" : "k_attn  ", ms * 1e3 / reps);
        }
#ifdef LG_ATTN_STAMPS
        {   // ticks per phase, averaged over the waves of ONE launch
            static unsigned long long all[1024][4][8];
            memset(all, 0, sizeof all);
            CK(hipMemcpyToSymbol(HIP_SYMBOL(am_stamps), all, sizeof all));
            a.y = dy; launch_attn_m(E, a, 0);
            CK(hipDeviceSynchronize());
            CK(hipMemcpyFromSymbol(all, HIP_SYMBOL(am_stamps), sizeof all));
            unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int b_ = 0; b_ < 1024; ++b_) for (int w_ = 0; w_ < 4; ++w_) for (int i = 0; i < 8; ++i) st[i] += all[b_][w_][i];
            st[7] /= 4;
            const double nw = (double)st[7] * 4;
            const char* nm[6] = {"staging + barrier", "window setup, x requested", "x wait, LayerNorm, to_qkv", "o2 requests, V^T fragments", "scores, softmax, P V", "proj, dropout, residual"};
            double tot = 0;
            for (int i = 0; i < 6; ++i) tot += st[i] / nw;
            for (int i = 0; i < 6; ++i) printf("  stamps: %-28s %9.0f ticks per wave (%4.1f %%)\n", nm[i], st[i] / nw, 100.0 * st[i] / nw / tot);
            printf("  stamps: %.0f workgroups, %.0f ticks per wave in all (s_memtime ticks = shader cycles)\n", (double)st[7], tot);
        }
#endif
    }
    return 0;
}
