/* ORACLE (test infrastructure, not product code): plain-C sequential S6 selective scan,
 * forward and analytic backward, fp32 arithmetic (fp64 only for the sums over batch*length).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * Restates the algorithm whose text is embedded in the reference at
 * code/networks/mamba_sys.py:62-83 (bias, softplus, zero state), :92-105 (discretisation,
 * grouped B/C), :115-131 (recurrence, y = <h, C>) and :139-144 (D skip, silu(z) gate).
 * The backward is the hand-derived adjoint of exactly that forward (the reference obtains it
 * from autograd / the absent mamba_ssm CUDA kernel); tests/test_oracle.py checks it against
 * torch.autograd of oracle/selective_scan_ref.py.
 *
 * Layouts (all contiguous, row-major):
 *   u, delta, z, out, dout, du, ddelta, dz : (batch, dim, L)
 *   A, dA : (dim, N)      B, C, dB, dC : (batch, G, N, L), dim % G == 0, group(d) = d / (dim/G)
 *   D, delta_bias, dD, ddelta_bias : (dim)     last_state : (batch, dim, N)
 * Parallelism: OpenMP over (batch, group); `nthreads` <= 0 keeps the OpenMP default.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline float softplus_f(float x) { return x > 20.0f ? x : log1pf(expf(x)); } /* torch default threshold */
static inline float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

int scan_ref_fwd_f32(const float *u, const float *delta, const float *A, const float *B, const float *C,
                     const float *D, const float *z, const float *delta_bias, float *out, float *last_state,
                     int batch, int dim, int L, int N, int G, int delta_softplus, int nthreads)
{
    if (dim % G) return -1;
    const int dpg = dim / G;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
    {
        float *h = (float *)malloc(sizeof(float) * (size_t)N);
#pragma omp for collapse(2) schedule(dynamic, 1)
        for (int b = 0; b < batch; ++b)
            for (int d = 0; d < dim; ++d) {
                const int g = d / dpg;
                const float *ur = u + ((size_t)b * dim + d) * L, *dr = delta + ((size_t)b * dim + d) * L;
                const float *Bg = B + ((size_t)b * G + g) * N * L, *Cg = C + ((size_t)b * G + g) * N * L;
                const float *zr = z ? z + ((size_t)b * dim + d) * L : NULL;
                float *orow = out + ((size_t)b * dim + d) * L;
                const float bias = delta_bias ? delta_bias[d] : 0.0f, Dd = D ? D[d] : 0.0f;
                for (int n = 0; n < N; ++n) h[n] = 0.0f;
                for (int l = 0; l < L; ++l) {
                    float dt = dr[l] + bias;
                    if (delta_softplus) dt = softplus_f(dt);
                    const float du_ = dt * ur[l];
                    float y = 0.0f;
                    for (int n = 0; n < N; ++n) {
                        h[n] = expf(dt * A[(size_t)d * N + n]) * h[n] + du_ * Bg[(size_t)n * L + l];
                        y += h[n] * Cg[(size_t)n * L + l];
                    }
                    y += Dd * ur[l];
                    if (zr) y *= zr[l] * sigmoid_f(zr[l]);
                    orow[l] = y;
                }
                if (last_state)
                    for (int n = 0; n < N; ++n) last_state[((size_t)b * dim + d) * N + n] = h[n];
            }
        free(h);
    }
    return 0;
}

/* dA, dD, ddelta_bias, dB, dC are OVERWRITTEN (not accumulated into). */
int scan_ref_bwd_f32(const float *u, const float *delta, const float *A, const float *B, const float *C,
                     const float *D, const float *z, const float *delta_bias, const float *dout,
                     float *du, float *ddelta, float *dA, float *dB, float *dC, float *dD, float *dz,
                     float *ddelta_bias, int batch, int dim, int L, int N, int G, int delta_softplus, int nthreads)
{
    if (dim % G) return -1;
    const int dpg = dim / G;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    double *dA_acc = (double *)calloc((size_t)dim * N, sizeof(double));
    double *dD_acc = (double *)calloc((size_t)dim, sizeof(double));
    double *db_acc = (double *)calloc((size_t)dim, sizeof(double));
    memset(dB, 0, sizeof(float) * (size_t)batch * G * N * L);
    memset(dC, 0, sizeof(float) * (size_t)batch * G * N * L);
#pragma omp parallel
    {
        float *hs = (float *)malloc(sizeof(float) * (size_t)(L + 1) * N); /* hs[(l+1)*N+n] = h_l, hs[0..N) = 0 */
        float *as = (float *)malloc(sizeof(float) * (size_t)L * N);
        float *dts = (float *)malloc(sizeof(float) * (size_t)L);
        float *dh = (float *)malloc(sizeof(float) * (size_t)N);
        double *dA_loc = (double *)malloc(sizeof(double) * (size_t)N);
#pragma omp for collapse(2) schedule(dynamic, 1)
        for (int b = 0; b < batch; ++b)
            for (int g = 0; g < G; ++g) {
                const float *Bg = B + ((size_t)b * G + g) * N * L, *Cg = C + ((size_t)b * G + g) * N * L;
                float *dBg = dB + ((size_t)b * G + g) * N * L, *dCg = dC + ((size_t)b * G + g) * N * L;
                for (int d = g * dpg; d < (g + 1) * dpg; ++d) {
                    const size_t row = ((size_t)b * dim + d) * L;
                    const float *ur = u + row, *dr = delta + row, *gor = dout + row;
                    const float *zr = z ? z + row : NULL;
                    const float bias = delta_bias ? delta_bias[d] : 0.0f, Dd = D ? D[d] : 0.0f;
                    /* forward sweep, keeping every state */
                    for (int n = 0; n < N; ++n) hs[n] = 0.0f;
                    for (int l = 0; l < L; ++l) {
                        float dt = dr[l] + bias;
                        if (delta_softplus) dt = softplus_f(dt);
                        dts[l] = dt;
                        const float du_ = dt * ur[l];
                        for (int n = 0; n < N; ++n) {
                            const float a = expf(dt * A[(size_t)d * N + n]);
                            as[(size_t)l * N + n] = a;
                            hs[(size_t)(l + 1) * N + n] = a * hs[(size_t)l * N + n] + du_ * Bg[(size_t)n * L + l];
                        }
                    }
                    /* reverse sweep */
                    for (int n = 0; n < N; ++n) { dh[n] = 0.0f; dA_loc[n] = 0.0; }
                    double dD_loc = 0.0, db_loc = 0.0;
                    for (int l = L - 1; l >= 0; --l) {
                        float dy = gor[l];
                        if (zr) {
                            float y = 0.0f;
                            for (int n = 0; n < N; ++n) y += hs[(size_t)(l + 1) * N + n] * Cg[(size_t)n * L + l];
                            y += Dd * ur[l];
                            const float s = sigmoid_f(zr[l]);
                            dz[row + l] = dy * y * (s * (1.0f + zr[l] * (1.0f - s)));
                            dy *= zr[l] * s;
                        }
                        const float dt = dts[l];
                        float ddt = 0.0f, ddu = 0.0f; /* grads wrt dt and wrt (dt*u) */
                        for (int n = 0; n < N; ++n) {
                            const float a = as[(size_t)l * N + n];
                            const float hcur = hs[(size_t)(l + 1) * N + n], hprev = hs[(size_t)l * N + n];
                            const float Cn = Cg[(size_t)n * L + l], Bn = Bg[(size_t)n * L + l];
                            dCg[(size_t)n * L + l] += dy * hcur;
                            const float dhn = dy * Cn + dh[n];          /* dL/dh_l */
                            const float w = dhn * hprev * a;            /* dL/da * a */
                            ddt += w * A[(size_t)d * N + n];
                            dA_loc[n] += (double)(w * dt);
                            dBg[(size_t)n * L + l] += dhn * dt * ur[l];
                            ddu += dhn * Bn;
                            dh[n] = a * dhn;                            /* carried to l-1 */
                        }
                        ddt += ddu * ur[l];
                        du[row + l] = ddu * dt + dy * Dd;
                        dD_loc += (double)(dy * ur[l]);
                        if (delta_softplus) {
                            const float x = dr[l] + bias;
                            if (x <= 20.0f) ddt *= sigmoid_f(x);
                        }
                        ddelta[row + l] = ddt;
                        db_loc += (double)ddt;
                    }
#pragma omp critical
                    {
                        for (int n = 0; n < N; ++n) dA_acc[(size_t)d * N + n] += dA_loc[n];
                        dD_acc[d] += dD_loc;
                        db_acc[d] += db_loc;
                    }
                }
            }
        free(hs); free(as); free(dts); free(dh); free(dA_loc);
    }
    for (size_t i = 0; i < (size_t)dim * N; ++i) dA[i] = (float)dA_acc[i];
    for (int d = 0; d < dim; ++d) {
        if (dD) dD[d] = (float)dD_acc[d];
        if (ddelta_bias) ddelta_bias[d] = (float)db_acc[d];
    }
    free(dA_acc); free(dD_acc); free(db_acc);
    return 0;
}
