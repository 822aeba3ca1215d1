/* C half of the CPU oracle (TEST INFRASTRUCTURE ONLY; see snn_oracle.py).
 *
 * oracle_fseq_matmul: the 'fseq' contraction contract -- for every output a
 *   float32 fmaf chain over k ascending, starting from +0.  It restates
 *   lax.dot_general at flax_qdense.py:87 / lax.conv_general_dilated at
 *   flax_qconv.py:158 with one fixed summation order (XLA leaves it open).
 *
 * oracle_check_div: exhaustive proof obligation for the two-instruction
 *   division used by the HIP epilogues, t = a*r_lo; q = fma(a, r_hi, t) with
 *   r_hi = fl(1/L), r_lo = fl(1/L - r_hi) (a split reciprocal: the fma rounds
 *   a/L * (1 + ~2^-48) once, and a quotient of integers a < 2^24 by L never
 *   lies that close to a rounding boundary): counts integers |a| <= amax for
 *   which q differs from the IEEE quotient fl(a / L) (DuQ dequantisation
 *   x / (n_lvl - 1), quant.py:443).  Must return 0.
 *
 * oracle_fma_rows: acc[i][j] = fmaf(g[i], x[i][j], acc[i][j]) -- one step of the 'gint' chain
 *   (snn_oracle.gated_conv: a sigmoid gate per input channel times the exact integer sum of that
 *   channel's codes over the taps; examples/tcja/models.py:95-97 feeding models.py:149-187).
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC -fopenmp oracle_c.c -lm
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

void oracle_fseq_matmul(const float *x, const float *w, float *y, int64_t m,
                        int64_t k, int64_t n) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < m; ++i) {
    float *acc = y + i * n;
    for (int64_t j = 0; j < n; ++j) acc[j] = 0.0f;
    for (int64_t kk = 0; kk < k; ++kk) {
      const float xv = x[i * k + kk];
      const float *wr = w + kk * n;
      for (int64_t j = 0; j < n; ++j) acc[j] = fmaf(xv, wr[j], acc[j]);
    }
  }
}

int64_t oracle_check_div(int32_t L, int32_t amax) {
  const float Lf = (float)L;
  const float r_hi = 1.0f / Lf;
  const float r_lo = (float)(1.0 / (double)Lf - (double)r_hi);
  int64_t bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(static)
  for (int32_t a = -amax; a <= amax; ++a) {
    const float af = (float)a;
    const float t = af * r_lo;
    const float q = fmaf(af, r_hi, t);
    const float ref = af / Lf;
    if (memcmp(&q, &ref, 4) != 0 && !(q == 0.0f && ref == 0.0f)) ++bad;
  }
  return bad;
}

void oracle_fma_rows(float *acc, const float *g, const float *x, int64_t m, int64_t n) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < m; ++i) {
    const float gi = g[i];
    for (int64_t j = 0; j < n; ++j) acc[i * n + j] = fmaf(gi, x[i * n + j], acc[i * n + j]);
  }
}
