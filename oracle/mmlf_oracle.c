/*
 * mmlf_oracle.c -- TEST INFRASTRUCTURE ONLY (the parity checker; never the product path).
 *
 * Plain-C CPU restatement of the convolution / batch-norm arithmetic that the
 * reference's hot path reaches through stock torch.nn modules:
 *   nn.Conv2d(k=2, stride 1, zero pad p in {0,1}, bias)   reference mmlf/model/feed_forward.py:123,125
 *   nn.ReLU                                                 reference feed_forward.py:124,135
 *   nn.BatchNorm2d(C, eps=1e-5, momentum) train and eval   reference feed_forward.py:134
 * and of their backward passes (autograd, reference mmlf/train/cli.py:257).
 *
 * The arithmetic itself lives in a third-party dependency of the reference (torch,
 * requirements.txt:18, unpinned); this file restates the published definitions
 * (cross-correlation; biased batch variance for normalisation, unbiased for the running
 * estimate).  It is pinned by the golden vectors in tests/golden/, which were produced by
 * importing the reference itself (tests/golden/make_golden.py).
 *
 * Layout: NCHW float32, like the reference.  Accumulation type is double by default (the
 * checker should be more accurate than either side it referees); -DORC_ACC_FLOAT builds the
 * float-accumulating variant that bench.py times as the "port" CPU baseline.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef ORC_ACC_FLOAT
typedef float acc_t;
#else
typedef double acc_t;
#endif

#define IDX4(b, c, y, x, C, H, W) ((((size_t)(b) * (C) + (c)) * (H) + (y)) * (W) + (x))

int orc_acc_bytes(void) { return (int)sizeof(acc_t); }

/* out[b,co,y,x] = bias[co] + sum_{ci,dy,dx} w[co,ci,dy,dx] * in[b,ci,y+dy-p,x+dx-p]
 * (zero outside), Ho = H + 2p - 1, Wo = W + 2p - 1.  reference feed_forward.py:123,125 */
void orc_conv2x2_fwd(const float *in, const float *w, const float *bias, float *out,
                     int B, int Cin, int H, int W, int Cout, int pad, int relu)
{
    const int Ho = H + 2 * pad - 1, Wo = W + 2 * pad - 1;
#pragma omp parallel
    {
        acc_t *plane = (acc_t *)malloc(sizeof(acc_t) * (size_t)Ho * Wo);
#pragma omp for collapse(2) schedule(static)
        for (int b = 0; b < B; ++b)
            for (int co = 0; co < Cout; ++co) {
                const acc_t bv = bias ? (acc_t)bias[co] : (acc_t)0;
                for (int i = 0; i < Ho * Wo; ++i) plane[i] = bv;
                for (int ci = 0; ci < Cin; ++ci) {
                    const float *ip = in + IDX4(b, ci, 0, 0, Cin, H, W);
                    const float *wp = w + (((size_t)co * Cin + ci) * 4);
                    for (int dy = 0; dy < 2; ++dy)
                        for (int dx = 0; dx < 2; ++dx) {
                            const acc_t wv = (acc_t)wp[dy * 2 + dx];
                            /* valid output range so that input coords stay inside */
                            int y0 = pad - dy; if (y0 < 0) y0 = 0;
                            int y1 = H - 1 + pad - dy; if (y1 > Ho - 1) y1 = Ho - 1;
                            int x0 = pad - dx; if (x0 < 0) x0 = 0;
                            int x1 = W - 1 + pad - dx; if (x1 > Wo - 1) x1 = Wo - 1;
                            for (int y = y0; y <= y1; ++y) {
                                const float *irow = ip + (size_t)(y + dy - pad) * W + (dx - pad);
                                acc_t *prow = plane + (size_t)y * Wo;
                                for (int x = x0; x <= x1; ++x)
                                    prow[x] += wv * (acc_t)irow[x];
                            }
                        }
                }
                float *op = out + IDX4(b, co, 0, 0, Cout, Ho, Wo);
                for (int i = 0; i < Ho * Wo; ++i) {
                    float v = (float)plane[i];
                    op[i] = (relu && v < 0.f) ? 0.f : v;
                }
            }
        free(plane);
    }
}

/* Backward of the convolution above.  gout: (B,Cout,Ho,Wo).
 * gin (nullable): (B,Cin,H,W); gw: (Cout,Cin,2,2); gb: (Cout). */
void orc_conv2x2_bwd(const float *in, const float *w, const float *gout,
                     float *gin, float *gw, float *gb,
                     int B, int Cin, int H, int W, int Cout, int pad)
{
    const int Ho = H + 2 * pad - 1, Wo = W + 2 * pad - 1;
    if (gb) {
#pragma omp parallel for schedule(static)
        for (int co = 0; co < Cout; ++co) {
            double s = 0;
            for (int b = 0; b < B; ++b) {
                const float *gp = gout + IDX4(b, co, 0, 0, Cout, Ho, Wo);
                acc_t sb = 0;
                for (int i = 0; i < Ho * Wo; ++i) sb += (acc_t)gp[i];
                s += (double)sb;
            }
            gb[co] = (float)s;
        }
    }
    if (gw) {
#pragma omp parallel for collapse(2) schedule(static)
        for (int co = 0; co < Cout; ++co)
            for (int ci = 0; ci < Cin; ++ci) {
                double s[4] = {0, 0, 0, 0};
                for (int b = 0; b < B; ++b) {
                    const float *gp = gout + IDX4(b, co, 0, 0, Cout, Ho, Wo);
                    const float *ip = in + IDX4(b, ci, 0, 0, Cin, H, W);
                    for (int dy = 0; dy < 2; ++dy)
                        for (int dx = 0; dx < 2; ++dx) {
                            int y0 = pad - dy; if (y0 < 0) y0 = 0;
                            int y1 = H - 1 + pad - dy; if (y1 > Ho - 1) y1 = Ho - 1;
                            int x0 = pad - dx; if (x0 < 0) x0 = 0;
                            int x1 = W - 1 + pad - dx; if (x1 > Wo - 1) x1 = Wo - 1;
                            acc_t t = 0;
                            for (int y = y0; y <= y1; ++y) {
                                const float *irow = ip + (size_t)(y + dy - pad) * W + (dx - pad);
                                const float *grow = gp + (size_t)y * Wo;
                                for (int x = x0; x <= x1; ++x)
                                    t += (acc_t)grow[x] * (acc_t)irow[x];
                            }
                            s[dy * 2 + dx] += (double)t;
                        }
                }
                float *o = gw + (((size_t)co * Cin + ci) * 4);
                for (int k = 0; k < 4; ++k) o[k] = (float)s[k];
            }
    }
    if (gin) {
#pragma omp parallel
        {
            acc_t *plane = (acc_t *)malloc(sizeof(acc_t) * (size_t)H * W);
#pragma omp for collapse(2) schedule(static)
            for (int b = 0; b < B; ++b)
                for (int ci = 0; ci < Cin; ++ci) {
                    for (int i = 0; i < H * W; ++i) plane[i] = 0;
                    for (int co = 0; co < Cout; ++co) {
                        const float *gp = gout + IDX4(b, co, 0, 0, Cout, Ho, Wo);
                        const float *wp = w + (((size_t)co * Cin + ci) * 4);
                        for (int dy = 0; dy < 2; ++dy)
                            for (int dx = 0; dx < 2; ++dx) {
                                const acc_t wv = (acc_t)wp[dy * 2 + dx];
                                int y0 = pad - dy; if (y0 < 0) y0 = 0;
                                int y1 = H - 1 + pad - dy; if (y1 > Ho - 1) y1 = Ho - 1;
                                int x0 = pad - dx; if (x0 < 0) x0 = 0;
                                int x1 = W - 1 + pad - dx; if (x1 > Wo - 1) x1 = Wo - 1;
                                for (int y = y0; y <= y1; ++y) {
                                    acc_t *prow = plane + (size_t)(y + dy - pad) * W + (dx - pad);
                                    const float *grow = gp + (size_t)y * Wo;
                                    for (int x = x0; x <= x1; ++x)
                                        prow[x] += wv * (acc_t)grow[x];
                                }
                            }
                    }
                    float *o = gin + IDX4(b, ci, 0, 0, Cin, H, W);
                    for (int i = 0; i < H * W; ++i) o[i] = (float)plane[i];
                }
            free(plane);
        }
    }
}

/* g *= (y > 0): backward of y = relu(.) given the forward OUTPUT y. */
void orc_relu_bwd(const float *y, float *g, size_t n)
{
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i)
        if (!(y[i] > 0.f)) g[i] = 0.f;
}

/* BatchNorm2d, training mode (reference feed_forward.py:134; torch semantics):
 * per channel over (B,H,W): mean, biased var; y = (x-mean)*invstd*gamma + beta (+ReLU);
 * running <- (1-m)*running + m*stat with the UNBIASED variance; saves mean/invstd. */
void orc_bn_train_fwd(const float *x, const float *gamma, const float *beta, float *y,
                      float *save_mean, float *save_invstd,
                      float *running_mean, float *running_var,
                      double momentum, double eps, int B, int C, int HW, int relu)
{
    const double n = (double)B * HW;
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c) {
        double s = 0;
        for (int b = 0; b < B; ++b) {
            const float *p = x + ((size_t)b * C + c) * HW;
            for (int i = 0; i < HW; ++i) s += p[i];
        }
        const double mean = s / n;
        double v = 0;
        for (int b = 0; b < B; ++b) {
            const float *p = x + ((size_t)b * C + c) * HW;
            for (int i = 0; i < HW; ++i) { double d = p[i] - mean; v += d * d; }
        }
        const double var = v / n;
        const double invstd = 1.0 / sqrt(var + eps);
        const float meanf = (float)mean, invf = (float)invstd;
        save_mean[c] = meanf;
        save_invstd[c] = invf;
        if (running_mean) {
            running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
            running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * (n > 1 ? v / (n - 1) : var));
        }
        const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
        for (int b = 0; b < B; ++b) {
            const float *p = x + ((size_t)b * C + c) * HW;
            float *q = y + ((size_t)b * C + c) * HW;
            for (int i = 0; i < HW; ++i) {
                float o = (p[i] - meanf) * invf * g + bt;
                q[i] = (relu && o < 0.f) ? 0.f : o;
            }
        }
    }
}

/* BatchNorm2d, eval mode: y = (x - running_mean) / sqrt(running_var + eps) * gamma + beta. */
void orc_bn_eval_fwd(const float *x, const float *gamma, const float *beta, float *y,
                     const float *running_mean, const float *running_var,
                     double eps, int B, int C, int HW, int relu)
{
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c) {
        const float invf = (float)(1.0 / sqrt((double)running_var[c] + eps));
        const float meanf = running_mean[c];
        const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
        for (int b = 0; b < B; ++b) {
            const float *p = x + ((size_t)b * C + c) * HW;
            float *q = y + ((size_t)b * C + c) * HW;
            for (int i = 0; i < HW; ++i) {
                float o = (p[i] - meanf) * invf * g + bt;
                q[i] = (relu && o < 0.f) ? 0.f : o;
            }
        }
    }
}

/* Backward of training-mode BatchNorm2d.  gy is the gradient w.r.t. the BN output
 * (already multiplied by the ReLU mask if a ReLU followed).
 * gx = gamma*invstd/n * (n*gy - sum(gy) - xhat*sum(gy*xhat)); ggamma = sum(gy*xhat); gbeta = sum(gy). */
void orc_bn_train_bwd(const float *x, const float *gy, const float *gamma,
                      const float *save_mean, const float *save_invstd,
                      float *gx, float *ggamma, float *gbeta, int B, int C, int HW)
{
    const double n = (double)B * HW;
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c) {
        const double mean = save_mean[c], inv = save_invstd[c];
        double sg = 0, sgx = 0;
        for (int b = 0; b < B; ++b) {
            const float *p = x + ((size_t)b * C + c) * HW;
            const float *g = gy + ((size_t)b * C + c) * HW;
            for (int i = 0; i < HW; ++i) { sg += g[i]; sgx += g[i] * ((p[i] - mean) * inv); }
        }
        if (ggamma) ggamma[c] = (float)sgx;
        if (gbeta) gbeta[c] = (float)sg;
        const double k = (gamma ? gamma[c] : 1.0) * inv / n;
        for (int b = 0; b < B; ++b) {
            const float *p = x + ((size_t)b * C + c) * HW;
            const float *g = gy + ((size_t)b * C + c) * HW;
            float *o = gx + ((size_t)b * C + c) * HW;
            for (int i = 0; i < HW; ++i) {
                double xh = (p[i] - mean) * inv;
                o[i] = (float)(k * (n * g[i] - sg - xh * sgx));
            }
        }
    }
}
