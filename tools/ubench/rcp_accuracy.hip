// Accuracy of v_rcp_f64 / v_rsq_f64 seeds and of 1, 2, 3 Newton steps on gfx950 (development record).
// build: hipcc --offload-arch=gfx950 -O3 -o rcp_accuracy rcp_accuracy.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k(const double* x, double* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = x[i];
  double r = __builtin_amdgcn_rcp(v);
  out[i] = r;
  for (int s = 1; s <= 3; ++s) {
    r = __builtin_fma(__builtin_fma(-v, r, 1.0), r, r);
    out[s * n + i] = r;
  }
  // cubic step from the seed, then one Newton step
  double r0 = __builtin_amdgcn_rcp(v);
  double e = __builtin_fma(-v, r0, 1.0);
  double t = __builtin_fma(e, e, e);
  double rc = __builtin_fma(r0, t, r0);
  out[4 * n + i] = rc;
  out[5 * n + i] = __builtin_fma(__builtin_fma(-v, rc, 1.0), rc, rc);
  double q = __builtin_amdgcn_rsq(v > 0 ? v : -v);
  out[6 * n + i] = q;
  const double ax = v > 0 ? v : -v;
  for (int s = 1; s <= 3; ++s) {
    q = __builtin_fma(__builtin_fma(-0.5 * ax * q, q, 0.5), q, q);
    out[(6 + s) * n + i] = q;
  }
}
int main() {
  const int n = 1 << 20;
  std::vector<double> x(n), o(10 * n);
  srand(1);
  for (int i = 0; i < n; ++i) {
    double m = 1.0 + (double)rand() / RAND_MAX + (double)rand() / RAND_MAX / RAND_MAX;
    int e = rand() % 80 - 40;
    x[i] = ldexp(m, e) * ((rand() & 1) ? 1 : -1);
  }
  double *dx, *dout;
  hipMalloc(&dx, n * 8); hipMalloc(&dout, 10 * n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
  hipMemcpy(o.data(), dout, 10 * n * 8, hipMemcpyDeviceToHost);
  const char* names[] = {"seed v_rcp_f64", "1 Newton step", "2 Newton steps", "3 Newton steps", "cubic step", "cubic + Newton", "seed v_rsq_f64", "rsq 1 step", "rsq 2 steps", "rsq 3 steps"};
  for (int s = 0; s < 10; ++s) {
    long double worst = 0; long ne = 0; long gt1 = 0;
    for (int i = 0; i < n; ++i) {
      long double ex = s >= 6 ? 1.0L / sqrtl(fabsl((long double)x[i])) : 1.0L / (long double)x[i];
      double exd = (double)ex;
      long double ulp = fabsl((long double)(nextafter(fabs(exd), INFINITY) - fabs(exd)));
      long double err = fabsl((long double)o[s * n + i] - ex) / ulp;
      if (err > worst) worst = err;
      if (o[s * n + i] != exd) ++ne;
      if (err > 1.0L) ++gt1;
    }
    printf("%-16s max error %.3Lg ulp (2^%.1Lf), not correctly rounded %.4f %%, > 1 ulp %.4f %%\n", names[s], worst, log2l(worst > 0 ? worst : 1e-30L), 100.0 * ne / n, 100.0 * gt1 / n);
  }
  long d23 = 0, dc3 = 0, dq = 0;
  for (int i = 0; i < n; ++i) { d23 += o[2 * n + i] != o[3 * n + i]; dc3 += o[4 * n + i] != o[3 * n + i]; dq += o[8 * n + i] != o[9 * n + i]; }
  printf("rcp: 2 vs 3 Newton steps differ on %ld of %d; cubic vs 3 Newton steps differ on %ld; rsq 2 vs 3 steps differ on %ld\n", d23, n, dc3, dq);
  return 0;
}
