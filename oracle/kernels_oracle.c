/*
 * TEST INFRASTRUCTURE ONLY -- plain-C, double-precision restatement of the arithmetic of the attack-side kernels.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the product package never does.
 * Parity pinning: this file is checked (tests/test_oracle_pinning.py) against golden vectors produced by running the
 * unmodified reference in the build container (oracle/make_golden.py -> tests/golden/kernels_*.npz).
 *
 * Each function cites the reference lines it follows (paths relative to the reference checkout,
 * breaching/attacks/auxiliaries/...).  Inputs are fp32 arrays (the reference computes in fp32), all accumulation is
 * fp64, outputs are fp64 so a test can choose its own tolerance.
 *
 * Build: gcc -O2 -shared -fPIC -o oracle/_build/libkernels_oracle.so oracle/kernels_oracle.c -lm   (oracle/build.py)
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

enum { K_COSINE = 0, K_COSINE_MASKED = 1, K_COSINE_FAST = 2, K_ANGULAR = 3, K_L2 = 4, K_L1 = 5, K_TAG = 6 };

static double sgn(double e) { return (e > 0) - (e < 0); }

/*
 * Gradient-matching objective over a list of T tensors given as one concatenated array with offsets off[0..T].
 * objectives.py:89-95 (L2), :133-141 (TAG, weights[t]), :158-166 (L1), :183-196 (cosine), :210-214 (angular),
 * :233-244 (masked cosine, |d| > 1e-6), :259-273 (fast cosine); `* scale` at :86,:126,:155,:178.
 * Writes value and, if grad != NULL, d value / d rec (same layout as rec).
 */
double oracle_gm(int kind, int T, const int64_t* off, const float* rec, const float* data, const double* weights,
                 double scale, double tag_scale, double fudge, double* grad) {
  double dot = 0, rr = 0, dd = 0, ss = 0, acc = 0;
  for (int t = 0; t < T; ++t) {
    double ts = 0, ta = 0;
    for (int64_t i = off[t]; i < off[t + 1]; ++i) {
      double r = rec[i], d = data[i];
      if (kind == K_COSINE_MASKED && !(fabs(d) > 1e-6)) { r = 0; d = 0; }
      dot += r * d; rr += r * r; dd += d * d;
      double e = (double)rec[i] - (double)data[i];
      ts += e * e; ta += fabs(e);
    }
    ss += ts;
    if (kind == K_TAG) acc += ts + tag_scale * weights[t] * ta;
    if (kind == K_L1) acc += ta;
  }
  double value = 0, c1 = 0, c2 = 0; /* cosine family: g = c1*d + c2*r ; L2 family: g = c1*e + c2*w*sign(e) */
  const double pi = 3.14159265358979323846;
  if (kind <= K_ANGULAR) {
    double rn = sqrt(rr), dn = sqrt(dd), cosv = dot / (rn * dn);
    double dcos_d = 1.0 / (rn * dn), dcos_r = -dot / (rr * rn * dn);
    if (kind == K_ANGULAR) {
      double lo = -1 + fudge, hi = 1 - fudge;
      int clamped = !(cosv > lo && cosv < hi);
      double cc = cosv < lo ? lo : (cosv > hi ? hi : cosv);
      value = scale * acos(cc) / pi;
      double dl = clamped ? 0.0 : -scale / (pi * sqrt(1 - cc * cc));
      c1 = dl * dcos_d; c2 = dl * dcos_r;
    } else {
      value = scale * (1 - cosv);
      c1 = -scale * dcos_d;
      c2 = kind == K_COSINE_FAST ? 0.0 : -scale * dcos_r;
    }
  } else if (kind == K_L2) { value = scale * 0.5 * ss; c1 = scale; }
  else if (kind == K_L1) { value = scale * 0.5 * acc; c2 = 0.5 * scale; }
  else { value = scale * 0.5 * acc; c1 = scale; c2 = 0.5 * scale * tag_scale; }
  if (grad) {
    for (int t = 0; t < T; ++t) {
      double w = (kind == K_TAG) ? weights[t] : 1.0;
      for (int64_t i = off[t]; i < off[t + 1]; ++i) {
        double r = rec[i], d = data[i];
        if (kind <= K_ANGULAR) {
          grad[i] = (kind == K_COSINE_MASKED && !(fabs(d) > 1e-6)) ? 0.0 : c1 * d + c2 * r;
        } else {
          double e = r - d;
          grad[i] = c1 * e + c2 * w * sgn(e);
        }
      }
    }
  }
  return value;
}

/*
 * TotalVariation.forward (regularizers.py:130-147) + NormRegularization.forward (:196-197) on x[B,3,H,W].
 * Forward differences with zero extension (the 3x3 conv with padding=1 of :142-144), eps inside the abs term (:145),
 * planes R-G, R-B, G-B appended for double_opponents (:132-141), mean over all planes and pixels (:147).
 * out[0] = tv value, out[1] = norm value; grad (nullable) receives d(tv + norm)/dx.
 */
static double plane_at(const float* x, int64_t base0, int64_t base1, int sub, int H, int W, int i, int j) {
  if (i < 0 || j < 0 || i >= H || j >= W) return 0.0;
  double v = x[base0 + (int64_t)i * W + j];
  if (sub) v -= x[base1 + (int64_t)i * W + j];
  return v;
}

void oracle_tv_norm(const float* x, int B, int H, int W, double tv_scale, double p, double q, double eps, int opponents,
                    double norm_scale, double norm_p, double* out, double* grad) {
  const int64_t plane = (int64_t)H * W;
  const int groups = opponents ? 6 : 3;
  const int pa[6] = {0, 1, 2, 0, 0, 1}, pb[6] = {0, 0, 0, 1, 2, 2};
  const double M = (double)B * groups * plane;
  double tv = 0, nrm = 0;
  if (grad) for (int64_t i = 0; i < (int64_t)B * 3 * plane; ++i) grad[i] = 0;
  for (int b = 0; b < B; ++b) {
    for (int g = 0; g < groups; ++g) {
      int sub = g >= 3;
      int64_t base0 = ((int64_t)b * 3 + pa[g]) * plane, base1 = ((int64_t)b * 3 + pb[g]) * plane;
      for (int i = 0; i < H; ++i) for (int j = 0; j < W; ++j) {
        double c = plane_at(x, base0, base1, sub, H, W, i, j);
        double dv = plane_at(x, base0, base1, sub, H, W, i + 1, j) - c;
        double dh = plane_at(x, base0, base1, sub, H, W, i, j + 1) - c;
        double a = fabs(dv) + eps, bb = fabs(dh) + eps;
        double S = pow(a, p) + pow(bb, p);
        tv += pow(S, q);
        if (grad) {
          double common = q * pow(S, q - 1) * p;
          double fv = common * pow(a, p - 1) * sgn(dv), fh = common * pow(bb, p - 1) * sgn(dh);
          /* dv = u(i+1,j) - u(i,j), dh = u(i,j+1) - u(i,j); scatter onto the plane(s) */
          double coef = tv_scale / M;
          int64_t o = (int64_t)i * W + j;
          double gs[3][2] = {{-(fv + fh), 0}, {fv, 0}, {fh, 0}};
          int64_t os[3] = {o, o + W, o + 1};
          int valid[3] = {1, i + 1 < H, j + 1 < W};
          for (int k = 0; k < 3; ++k) if (valid[k]) {
            grad[base0 + os[k]] += coef * gs[k][0];
            if (sub) grad[base1 + os[k]] -= coef * gs[k][0];
          }
        }
      }
    }
  }
  tv = tv_scale * tv / M;
  if (norm_scale != 0) {
    const double M3 = (double)B * 3 * plane;
    for (int64_t i = 0; i < (int64_t)B * 3 * plane; ++i) {
      double v = x[i];
      nrm += pow(v, norm_p);
      if (grad) grad[i] += norm_scale / M3 * pow(v, norm_p - 1);
    }
    nrm = norm_scale / norm_p * nrm / M3;
  }
  out[0] = tv; out[1] = nrm;
}

/*
 * DeepInversion feature statistic of one BN input x[B,C,HW] (math of deepinversion.py:93-101, restated, not copied):
 * mean_c, biased var_c over (b,hw); r = ||rv - var||_2 + ||rm - mean||_2.  grad (nullable) = dr/dx.
 */
double oracle_bnstat(const float* x, int B, int C, int64_t HW, const float* rm, const float* rv, double* grad,
                     double* mean_out, double* var_out) {
  const double n = (double)B * HW;
  double nv = 0, nm = 0;
  for (int c = 0; c < C; ++c) {
    double s = 0;
    for (int b = 0; b < B; ++b) for (int64_t k = 0; k < HW; ++k) s += x[((int64_t)b * C + c) * HW + k];
    double mean = s / n, v = 0;
    for (int b = 0; b < B; ++b) for (int64_t k = 0; k < HW; ++k) {
      double e = x[((int64_t)b * C + c) * HW + k] - mean; v += e * e;
    }
    v /= n;
    mean_out[c] = mean; var_out[c] = v;
    nv += (rv[c] - v) * (rv[c] - v); nm += (rm[c] - mean) * (rm[c] - mean);
  }
  nv = sqrt(nv); nm = sqrt(nm);
  if (grad) {
    for (int c = 0; c < C; ++c) {
      double pv = nv > 0 ? -(rv[c] - var_out[c]) / nv : 0, pm = nm > 0 ? -(rm[c] - mean_out[c]) / nm : 0;
      for (int b = 0; b < B; ++b) for (int64_t k = 0; k < HW; ++k) {
        int64_t idx = ((int64_t)b * C + c) * HW + k;
        grad[idx] = pm / n + pv * 2 * (x[idx] - mean_out[c]) / n;
      }
    }
  }
  return nv + nm;
}

/*
 * One candidate step: gradient post-processing (optimization_based_attack.py:167-184), torch.optim Adam / AdamW
 * single-tensor update (selected at auxiliaries/common.py:5-12) and box projection (:117-118).
 * sign_mode 0 none / 1 hard / 2 soft; step is 1-based; noise may be NULL; clip < 0 disables clipping (0 is a legal threshold).
 * All state updated in place (fp64 copies of fp32 state supplied by the test).
 */
void oracle_candidate_step(int64_t n, int64_t plane, int channels, double* x, const double* g_in, const double* noise,
                           double* m, double* v, double lr, double beta1, double beta2, double eps, double weight_decay,
                           int decoupled, int step, int sign_mode, int iteration, int max_iterations, double langevin,
                           double clip, int boxed, const double* lo, const double* hi) {
  double nrm = 0;
  for (int64_t i = 0; i < n; ++i) {
    double g = g_in[i] + (noise ? langevin * lr * noise[i] : 0.0);
    nrm += g * g;
  }
  nrm = sqrt(nrm);
  double clip_mul = (clip >= 0 && nrm > clip) ? clip / (nrm + 1e-6) : 1.0;
  double soft = 1.0 - (double)iteration / (double)max_iterations;
  double bc1 = 1 - pow(beta1, step), bc2 = 1 - pow(beta2, step);
  for (int64_t i = 0; i < n; ++i) {
    double g = (g_in[i] + (noise ? langevin * lr * noise[i] : 0.0)) * clip_mul;
    if (sign_mode == 1) g = sgn(g);
    else if (sign_mode == 2) g = tanh(g * soft) / soft;
    if (decoupled && weight_decay != 0) x[i] *= 1 - lr * weight_decay;
    m[i] = m[i] + (1 - beta1) * (g - m[i]);
    v[i] = v[i] * beta2 + (1 - beta2) * g * g;
    double denom = sqrt(v[i]) / sqrt(bc2) + eps;
    x[i] = x[i] - (lr / bc1) * m[i] / denom;
    if (boxed) {
      int c = (int)((i / plane) % channels);
      double t = x[i] < hi[c] ? x[i] : hi[c];
      x[i] = t > lo[c] ? t : lo[c];
    }
  }
}

/*
 * OrthogonalityRegularization on x[B, D] (regularizers.py:170-178): full_products[i][j] = mean_k (x_ik * x_jk)^2, diagonal
 * zeroed, summed over all ordered pairs.  grad (nullable) = d value / d x.
 */
double oracle_orthogonality(const float* x, int B, int64_t D, double* grad) {
  double value = 0;
  for (int i = 0; i < B; ++i)
    for (int j = 0; j < B; ++j) {
      if (i == j) continue;
      double acc = 0;
      for (int64_t k = 0; k < D; ++k) {
        double p = (double)x[(int64_t)i * D + k] * (double)x[(int64_t)j * D + k];
        acc += p * p;
      }
      value += acc / (double)D;
    }
  if (grad) {
    for (int i = 0; i < B; ++i)
      for (int64_t k = 0; k < D; ++k) {
        double others = 0;
        for (int j = 0; j < B; ++j)
          if (j != i) others += (double)x[(int64_t)j * D + k] * (double)x[(int64_t)j * D + k];
        /* the pair (i, j) and the pair (j, i) both contain x_ik: 2 * 2 x_ik x_jk^2 / D */
        grad[(int64_t)i * D + k] = 4.0 * (double)x[(int64_t)i * D + k] * others / (double)D;
      }
  }
  return value;
}

/*
 * PSNR per example (analysis/metrics.py:117-130, batched=False) of de-normalised, optionally clamped images
 * (analysis.py:228-229): img = x * std[c] + mean[c].  out[0] = mean, out[1] = max, out[2 + b] = example b.
 */
void oracle_psnr(const float* rec, const float* ref, int B, int64_t per_example, int64_t plane, int channels,
                 const double* mean, const double* std, double factor, int clip, double* out) {
  double sum = 0, best = -INFINITY;
  for (int b = 0; b < B; ++b) {
    double acc = 0;
    for (int64_t i = 0; i < per_example; ++i) {
      int c = channels > 1 ? (int)((i / plane) % channels) : 0;
      double u = (double)rec[(int64_t)b * per_example + i] * std[c] + mean[c];
      double v = (double)ref[(int64_t)b * per_example + i] * std[c] + mean[c];
      if (clip) {
        u = u < 0 ? 0 : (u > 1 ? 1 : u);
        v = v < 0 ? 0 : (v > 1 ? 1 : v);
      }
      acc += (u - v) * (u - v);
    }
    double p = 10.0 * log10(factor * factor / (acc / (double)per_example));
    out[2 + b] = p;
    sum += p;
    if (p > best) best = p;
  }
  out[0] = sum / B;
  out[1] = best;
}
