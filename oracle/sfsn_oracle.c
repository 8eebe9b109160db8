/*
 * sfsn_oracle.c -- CPU restatement of the Spiking-FullSubNet inference hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the
 * checker / the timed CPU baseline.  The product path (spiking_fullsubnet_amd/) never imports it and
 * has no CPU fallback.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function here against golden
 * vectors produced by importing the reference itself (tests/golden/make_golden.py, run in the build
 * container where /root/reference is mounted).  The reference holds no golden vectors of its own
 * (its tests/ covers nothing on this path).
 *
 * Each function cites the reference lines it restates (paths relative to the reference root):
 *   NEURON = audiozen/models/spiking_fullsubnet/efficient_spiking_neuron.py
 *   MODEL  = audiozen/models/spiking_fullsubnet/modeling_spiking_fullsubnet.py
 *   FROZEN = recipes/intel_ndns/spiking_fullsubnet_freeze_phase/model_low_freq.py
 *
 * The file is compiled twice (see Makefile): REAL=float gives the sfsn_oracle_f32_* symbols (the
 * oracle proper: every elementwise operation is a separately rounded fp32 operation in the order the
 * reference's ATen ops apply them; dot products are accumulated in double and rounded once, i.e. the
 * centre of the rounding-noise cloud of any fp32 GEMM), REAL=double gives sfsn_oracle_f64_* (used
 * only to measure the fp32-vs-fp64 noise floor of the discontinuous spike map).
 *
 * Layouts are the reference's own: time-major [T][R][feat] for everything inside a sequence model,
 * [B][F][T] for spectra, interleaved (re,im) for complex.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef REAL
#define REAL float
#endif
#ifndef SUFFIX
#define SUFFIX f32
#endif
#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define ORA(name) CAT(CAT(CAT(sfsn_oracle_, SUFFIX), _), name)

typedef REAL real;

static inline real r_exp(real x) { return sizeof(real) == 4 ? (real)expf((float)x) : (real)exp((double)x); }
static inline real r_sqrt(real x) { return sizeof(real) == 4 ? (real)sqrtf((float)x) : (real)sqrt((double)x); }
static inline real r_fma(real a, real b, real c) {
    return sizeof(real) == 4 ? (real)fmaf((float)a, (float)b, (float)c) : (real)fma((double)a, (double)b, (double)c);
}
static inline real r_hypot(real a, real b) {
    return sizeof(real) == 4 ? (real)hypotf((float)a, (float)b) : (real)hypot((double)a, (double)b);
}
static inline real r_pow(real a, real b) { return sizeof(real) == 4 ? (real)powf((float)a, (float)b) : (real)pow((double)a, (double)b); }

/* Dot products (torch.mm NEURON:141,143; nn.Linear MODEL:50) are accumulated in double and rounded once:
 * see matvec_acc below. */
int ORA(version)(void) { return 1; }

/* acc[n] += sum_k x[k] * wT[k][n]; double accumulation, vectorisable over n without reassociation.
 * Zero inputs are skipped (exact: adding 0*w changes nothing) -- spikes are 25-58 % active.
 * (The inner loop is compiled for AVX-512 / AVX2 / baseline x86-64 and picked at run time: vector width changes how many of the
 *  independent accumulators acc[n] advance per instruction, not the order of the additions into any one of them; with
 *  -ffp-contract=off the multiply and the add stay separately rounded in every clone -- bit-identical results, checked against the
 *  fixtures.  It is what makes the oracle a credible CPU baseline for bench.py: 2.4 x one core's speed of the SSE2 build.) */
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
__attribute__((target_clones("avx512f", "avx2", "default")))
#endif
static void matvec_acc(double* acc, const real* x, int K, const double* wT, int N) {
    for (int k = 0; k < K; ++k) {
        const double xk = (double)x[k];
        if (xk == 0.0) continue;
        const double* w = wT + (size_t)k * N;
        for (int n = 0; n < N; ++n) acc[n] += xk * w[n];
    }
}

/* acc[n] += xk * w[n] for one k (the inner loop of matvec_acc, for callers that walk k outside: gsn_layer's row blocks) */
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
__attribute__((target_clones("avx512f", "avx2", "default")))
#endif
static void axpy_acc(double* acc, double xk, const double* w, int N) {
    if (xk == 0.0) return; /* exact: adding 0*w changes nothing */
    for (int n = 0; n < N; ++n) acc[n] += xk * w[n];
}

static double* transpose_to_double(const real* w, int N, int K) { /* w [N][K] -> wT [K][N] */
    double* wT = (double*)malloc(sizeof(double) * (size_t)N * K);
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) wT[(size_t)k * N + n] = (double)w[(size_t)n * K + k];
    return wT;
}

/* ------------------------------------------------------------------------------------------------
 * GSN layer scan.  NEURON:75-81 (GSULayer.forward: python loop over T), NEURON:132-153
 * (GSUCell.forward), NEURON:84-92 (Triangle.forward: spike = (membrane >= 0)).
 *
 *   gates = (x_t . W_ih^T + bias_ih) + h . W_hh^T          NEURON:140-145 (note the association)
 *   f = sigmoid(gates[:, :H]);  g = gates[:, H:]            NEURON:146-147
 *   c' = f*c + (1-f)*g                                      NEURON:148   (four separately rounded ops)
 *   c'' = BatchNorm1d_eval(c')                              NEURON:149-150
 *   h' = (c'' >= 0) ? 1 : 0 ; carry (h', c'')               NEURON:151-153
 *
 * shared != 0: W_ih [H][I], W_hh [H][H] are used for both gates (NEURON:134-136, `.repeat((2,1))`),
 * bias_ih is always [2H].  shared == 0: W_ih [2H][I], W_hh [2H][H] (NEURON:137-139).
 *
 * BatchNorm in eval mode is restated the way this container's ATen CPU kernel evaluates it (probed
 * bit-exact, torch 2.10.0 CPU/AVX512): invstd = 1/sqrt(var+eps); alpha = invstd*gamma;
 * beta = fma(-mean, alpha, bias); y = fma(x, alpha, beta).
 *
 * x [T][R][I];  h, c [R][H] are read as the initial state and overwritten with the final state
 * (NEURON:50-62 StackedGSU passes states in and out);  spikes [T][R][H];  membrane (nullable)
 * receives the post-BN membrane c'' per step so tests can mask near-threshold cases.
 * ---------------------------------------------------------------------------------------------- */
void ORA(gsn_layer)(const real* x, int T, int R, int I, int H, int shared, const real* w_ih, const real* w_hh,
                    const real* bias, int use_bn, const real* bn_w, const real* bn_b, const real* bn_rm,
                    const real* bn_rv, double eps, real* h, real* c, real* spikes, real* membrane) {
    const int G = shared ? 1 : 2; /* weight rows = G*H; gate g uses rows g*H.. when not shared */
    const int GH = G * H;
    real* alpha = (real*)malloc(sizeof(real) * (size_t)H);
    real* beta = (real*)malloc(sizeof(real) * (size_t)H);
    for (int j = 0; j < H; ++j) {
        if (use_bn) {
            real invstd = (real)1 / r_sqrt(bn_rv[j] + (real)eps);
            alpha[j] = invstd * bn_w[j];
            beta[j] = r_fma(-bn_rm[j], alpha[j], bn_b[j]);
        } else {
            alpha[j] = 1;
            beta[j] = 0;
        }
    }
    double* wihT = transpose_to_double(w_ih, GH, I);
    double* whhT = transpose_to_double(w_hh, GH, H);
    /* Rows are independent; a thread takes a BLOCK of up to GSN_RB rows through all T steps and walks k in the outer loop of the
     * two products, so that a row of W^T fetched once serves every row of the block (per row the additions into acc[n] still
     * happen in the order k = 0, 1, ...: bit-identical to a row-at-a-time evaluation; checked against the fixtures).  Without the
     * blocking every row streams both matrices (0.5-1.6 MB as double) per step and 256 threads are bound by the shared caches. */
#define GSN_RB 8
    const int nblk = (R + GSN_RB - 1) / GSN_RB;
#pragma omp parallel for schedule(dynamic, 1)
    for (int blk = 0; blk < nblk; ++blk) {
        const int r0 = blk * GSN_RB, nr = (R - r0 < GSN_RB) ? R - r0 : GSN_RB;
        double* zi = (double*)malloc(sizeof(double) * (size_t)GH * GSN_RB);
        double* zr = (double*)malloc(sizeof(double) * (size_t)GH * GSN_RB);
        for (int t = 0; t < T; ++t) {
            memset(zi, 0, sizeof(double) * (size_t)GH * nr);
            memset(zr, 0, sizeof(double) * (size_t)GH * nr);
            for (int k = 0; k < I; ++k) { /* torch.mm(input, weight_ih.t())  NEURON:141 */
                const double* w = wihT + (size_t)k * GH;
                for (int q = 0; q < nr; ++q) axpy_acc(zi + (size_t)q * GH, (double)x[((size_t)t * R + r0 + q) * I + k], w, GH);
            }
            for (int k = 0; k < H; ++k) { /* torch.mm(hx, weight_hh.t())     NEURON:143 */
                const double* w = whhT + (size_t)k * GH;
                for (int q = 0; q < nr; ++q) axpy_acc(zr + (size_t)q * GH, (double)h[(size_t)(r0 + q) * H + k], w, GH);
            }
            for (int q = 0; q < nr; ++q) {
                const int r = r0 + q;
                real* hr = h + (size_t)r * H;
                real* cr = c + (size_t)r * H;
                const double* zir = zi + (size_t)q * GH;
                const double* zrr = zr + (size_t)q * GH;
                for (int j = 0; j < H; ++j) {
                    const int jg = shared ? j : H + j;
                    real pf = ((real)zir[j] + bias[j]) + (real)zrr[j];
                    real pg = ((real)zir[jg] + bias[H + j]) + (real)zrr[jg];
                    real f = (real)1 / ((real)1 + r_exp(-pf));
                    real a = f * cr[j];
                    real b = (real)1 - f;
                    real d = b * pg;
                    real cy = a + d;
                    if (use_bn) cy = r_fma(cy, alpha[j], beta[j]);
                    cr[j] = cy;
                    size_t o = ((size_t)t * R + r) * H + j;
                    spikes[o] = (cy >= (real)0) ? (real)1 : (real)0;
                    if (membrane) membrane[o] = cy;
                }
                memcpy(hr, spikes + ((size_t)t * R + r) * H, sizeof(real) * (size_t)H);
            }
        }
        free(zi);
        free(zr);
    }
    free(wihT);
    free(whhT);
    free(alpha);
    free(beta);
}

/* y[M][N] = x[M][K] . w[N][K]^T + b   (nn.Linear: MODEL:49-52,118; FROZEN:71-76,125) ; b nullable */
void ORA(linear)(const real* x, int M, int K, int N, const real* w, const real* b, real* y) {
    double* wT = transpose_to_double(w, N, K);
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m) {
        double* acc = (double*)calloc((size_t)N, sizeof(double));
        matvec_acc(acc, x + (size_t)m * K, K, wT, N);
        for (int n = 0; n < N; ++n) y[(size_t)m * N + n] = b ? (real)acc[n] + b[n] : (real)acc[n];
        free(acc);
    }
    free(wT);
}

/* nn.LayerNorm(I) over the last dim, eps inside sqrt, affine (MODEL:27-28,111-112). In place. */
void ORA(layer_norm)(real* x, int M, int I, const real* g, const real* b, double eps) {
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m) {
        real* v = x + (size_t)m * I;
        double s = 0, ss = 0;
        for (int i = 0; i < I; ++i) s += v[i];
        double mean = s / I;
        for (int i = 0; i < I; ++i) ss += ((double)v[i] - mean) * ((double)v[i] - mean);
        real rstd = (real)(1.0 / sqrt(ss / I + eps));
        real mu = (real)mean;
        for (int i = 0; i < I; ++i) v[i] = ((v[i] - mu) * rstd) * g[i] + b[i];
    }
}

/* |X|^fdrc on bins 0..F-2 (the Nyquist bin is dropped): MODEL:429-436 / FROZEN:569-576.
 * torch.abs(complex) is hypot; pow(x, 0.5) is evaluated as sqrt by ATen's pow kernel.
 * stft_ri [B][F][T][2] -> mag [B][F-1][T]. */
void ORA(front_mag)(const real* stft_ri, int B, int F, int T, double fdrc, real* mag) {
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b)
        for (int f = 0; f < F - 1; ++f)
            for (int t = 0; t < T; ++t) {
                const real* p = stft_ri + (((size_t)b * F + f) * T + t) * 2;
                real m = r_hypot(p[0], p[1]);
                mag[((size_t)b * (F - 1) + f) * T + t] = fdrc == 0.5 ? r_sqrt(m) : r_pow(m, (real)fdrc);
            }
}

static inline int reflect_bin(int f, int nf) { return f < 0 ? -f : (f > nf - 1 ? 2 * (nf - 1) - f : f); }

/* Sub-band feature gather for one group: MODEL:239-258 (two _freq_unfold calls + cat), _freq_unfold
 * MODEL:265-312 (slice, F.pad(mode="reflect"), F.unfold(kernel=(c+2n,T), stride=(c,T)), rearrange);
 * frozen twin FROZEN:350-431,451-474.
 *   unit k in [0,(hi-lo)/ctr), feature j in [0,ctr+2*nbr): bin f = lo + k*ctr - nbr + j, reflected
 *   (without edge repeat) at 0 and nf-1; then the full-band features with their own (ctr_fb, nbr_fb)
 *   appended (MODEL:258 cat dim=-2).  The full-band input is the full-band model's projection
 *   fb_tbf [T][B][FB], tiled along frequency: bin f reads fb_tbf[t][b][f % FB] (MODEL:442-443 repeat).
 * mag [B][nf][T] ; out x [T][B*N][I], I = (ctr+2nbr) + (ctr_fb+2nbr_fb), row r = b*N + k (MODEL:155).
 * Returns 0, or -1 if (hi-lo) % ctr != 0 (the reference raises ValueError, MODEL:283-287). */
int ORA(gather_group)(const real* mag, const real* fb_tbf, int B, int nf, int T, int FB, int lo, int hi, int ctr,
                      int nbr, int ctr_fb, int nbr_fb, real* x) {
    if ((hi - lo) % ctr != 0 || (hi - lo) % ctr_fb != 0) return -1;
    const int N = (hi - lo) / ctr;
    if ((hi - lo) / ctr_fb != N) return -1;
    const int I1 = ctr + 2 * nbr, I2 = ctr_fb + 2 * nbr_fb, I = I1 + I2;
#pragma omp parallel for schedule(static)
    for (int t = 0; t < T; ++t)
        for (int b = 0; b < B; ++b)
            for (int k = 0; k < N; ++k) {
                real* row = x + ((size_t)t * B * N + (size_t)b * N + k) * I;
                for (int j = 0; j < I1; ++j) {
                    int f = reflect_bin(lo + k * ctr - nbr + j, nf);
                    row[j] = mag[((size_t)b * nf + f) * T + t];
                }
                for (int j = 0; j < I2; ++j) {
                    int f = reflect_bin(lo + k * ctr_fb - nbr_fb + j, nf);
                    row[I1 + j] = fb_tbf[((size_t)t * B + b) * FB + (f % FB)];
                }
            }
    return 0;
}

/* Full-band input: bins 0..FB-1 of mag, time-major: MODEL:438-440,108 ("b f t -> t b f").
 * mag [B][nf][T] -> x [T][B][FB]. */
void ORA(gather_fullband)(const real* mag, int B, int nf, int T, int FB, real* x) {
#pragma omp parallel for schedule(static)
    for (int t = 0; t < T; ++t)
        for (int b = 0; b < B; ++b)
            for (int f = 0; f < FB; ++f) x[((size_t)t * B + b) * FB + f] = mag[((size_t)b * nf + f) * T + t];
}

/* offline_laplace_norm (FROZEN:147-169): x / (mean over all non-batch dims + EPSILON), one scalar per
 * batch item; EPSILON = np.finfo(float).eps (audiozen/constant.py:11).  Applied to the time-major
 * tensor x [T][B*N][I] whose batch item b owns rows b*N..b*N+N-1 at every t.  In place; mu_out [B]
 * (nullable) receives the means. */
void ORA(laplace_norm)(real* x, int T, int B, int N, int I, real* mu_out) {
    for (int b = 0; b < B; ++b) {
        double s = 0;
        for (int t = 0; t < T; ++t) {
            const real* p = x + ((size_t)t * B * N + (size_t)b * N) * I;
            for (int i = 0; i < N * I; ++i) s += p[i];
        }
        real mu = (real)(s / ((double)T * N * I));
        real den = mu + (real)2.220446049250313e-16;
        if (mu_out) mu_out[b] = mu;
        for (int t = 0; t < T; ++t) {
            real* p = x + ((size_t)t * B * N + (size_t)b * N) * I;
            for (int i = 0; i < N * I; ++i) p[i] = p[i] / den;
        }
    }
}

/* offline_gaussian_norm (FROZEN:205-218): (x - mean) / (std + EPSILON) with torch.mean / torch.std (the UNBIASED estimate) over all
 * non-batch dims, one pair per batch item; two passes in double.  Layout and in-place convention as laplace_norm; mu_out / sd_out
 * [B] nullable. */
void ORA(gaussian_norm)(real* x, int T, int B, int N, int I, real* mu_out, real* sd_out) {
    for (int b = 0; b < B; ++b) {
        double s = 0;
        const double n = (double)T * N * I;
        for (int t = 0; t < T; ++t) {
            const real* p = x + ((size_t)t * B * N + (size_t)b * N) * I;
            for (int i = 0; i < N * I; ++i) s += p[i];
        }
        const double m = s / n;
        double q = 0;
        for (int t = 0; t < T; ++t) {
            const real* p = x + ((size_t)t * B * N + (size_t)b * N) * I;
            for (int i = 0; i < N * I; ++i) q += ((double)p[i] - m) * ((double)p[i] - m);
        }
        const real mu = (real)m, sd = (real)sqrt(q / (n - 1.0));
        const real den = sd + (real)2.220446049250313e-16;
        if (mu_out) mu_out[b] = mu;
        if (sd_out) sd_out[b] = sd;
        for (int t = 0; t < T; ++t) {
            real* p = x + ((size_t)t * B * N + (size_t)b * N) * I;
            for (int i = 0; i < N * I; ++i) p[i] = (p[i] - mu) / den;
        }
    }
}

/* cumulative_laplace_norm (FROZEN:172-202; the form that accepts the 5-D sub-band tensor:
 * recipes/intel_ndns/spiking_fullsubnet_freeze_phase/model_low_freq_count_time.py:182-204 -- FROZEN's own
 * version unpacks four dimensions and raises on the sub-band input): every row (clip x unit) is divided, frame by
 * frame, by the mean of everything that row has seen so far:
 *   step_sum[r][t] = sum_i x[t][r][i];  cum[r][t] = cumsum_t step_sum;  mean = cum / (I * (t + 1));  x /= mean + EPSILON.
 * Sums in the working precision, sequentially over t (torch.cumsum on the CPU).  x [T][R][I], in place. */
void ORA(cum_laplace_norm)(real* x, int T, int R, int I) {
    for (int r = 0; r < R; ++r) {
        real cum = 0;
        for (int t = 0; t < T; ++t) {
            real* p = x + ((size_t)t * R + r) * I;
            real s = 0;
            for (int i = 0; i < I; ++i) s += p[i];
            cum += s;
            const real mean = cum / (real)((double)I * (t + 1));
            const real den = mean + (real)2.220446049250313e-16;
            for (int i = 0; i < I; ++i) p[i] = p[i] / den;
        }
    }
}

/* Deep filtering of one group and write-back into the enhanced spectrum.
 * Output re-index MODEL:160-167 "(b n) (c fc df s) t -> b df s (n fc) t c" (frozen FROZEN:259-265 has
 * no s): projection channel p = ((ci*fc + fci)*df + di)*S + si, ci=0 real / 1 imag.
 * deepfiltering MODEL:315-346 (frozen FROZEN:15-39): Y[f,t] = sum_{d<df} X[f, t-(df-1)+d] * C[d,f,t],
 * zero left padding (F.pad (order-1, 0)), complex product via einsum.
 * stft_ri [B][F][T][2] noisy spectrum; proj [T][B*N][P], P = 2*fc*df*S; group covers bins
 * lo .. lo+N*fc-1; enh_ri [B][S][F][T][2] is written for those bins only (MODEL:450-471). */
void ORA(deepfilter_group)(const real* stft_ri, const real* proj, int B, int F, int T, int lo, int N, int fc, int df,
                           int S, real* enh_ri) {
    const int P = 2 * fc * df * S;
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b)
        for (int k = 0; k < N; ++k)
            for (int fci = 0; fci < fc; ++fci) {
                const int f = lo + k * fc + fci;
                for (int s = 0; s < S; ++s)
                    for (int t = 0; t < T; ++t) {
                        const real* pr = proj + ((size_t)t * B * N + (size_t)b * N + k) * P;
                        real yr = 0, yi = 0;
                        for (int d = 0; d < df; ++d) {
                            int tt = t - (df - 1) + d;
                            real xr = 0, xi = 0;
                            if (tt >= 0) {
                                const real* px = stft_ri + (((size_t)b * F + f) * T + tt) * 2;
                                xr = px[0];
                                xi = px[1];
                            }
                            real cr = pr[((0 * fc + fci) * df + d) * S + s];
                            real ci = pr[((1 * fc + fci) * df + d) * S + s];
                            yr += xr * cr - xi * ci;
                            yi += xr * ci + xi * cr;
                        }
                        real* po = enh_ri + ((((size_t)b * S + s) * F + f) * T + t) * 2;
                        po[0] = yr;
                        po[1] = yi;
                    }
            }
}

/* Bins the groups do not cover (at least the Nyquist bin F-1) pass through from the noisy spectrum:
 * MODEL:461-470 (clone, then assign [..., :-1, :]); |.| for enh_mag MODEL:472.
 * enh_ri [B][S][F][T][2] bins [f0,F) are copied from stft_ri; mag_out [B][S][F][T] = hypot of all. */
void ORA(finish_spectrum)(const real* stft_ri, int B, int S, int F, int T, int f0, real* enh_ri, real* mag_out) {
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b)
        for (int s = 0; s < S; ++s)
            for (int f = 0; f < F; ++f)
                for (int t = 0; t < T; ++t) {
                    size_t o = (((size_t)b * S + s) * F + f) * T + t;
                    if (f >= f0) {
                        const real* px = stft_ri + (((size_t)b * F + f) * T + t) * 2;
                        enh_ri[2 * o] = px[0];
                        enh_ri[2 * o + 1] = px[1];
                    }
                    if (mag_out) mag_out[o] = r_hypot(enh_ri[2 * o], enh_ri[2 * o + 1]);
                }
}
