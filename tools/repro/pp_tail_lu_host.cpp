// Planar-push tail LU under sanitizers: the solver of od_solver.h + gen/planar_push.h compiled for the
// host with -fsanitize=undefined,address (and -ftrivial-auto-var-init=pattern: an uninitialised read would change the
// iteration counts), in the four exchange / guard variants that round 1 reported as erratic on the device.
//   tools/repro/run_pp_tail_lu.sh     prints one line per variant: non-converged knots, mean iterations, sanitizer reports
#include <cstdio>
#include <cmath>
#include <random>
#include "od_solver.h"
#include "gen/planar_push.h"
using namespace od;
using M = Model_planar_push;
struct Sink { static constexpr bool DEFER_GRAD = false, FULL_STATE = false; double g = 0; void grad(int, int, double v) { g += v; } void defer(const double*, double) {} };
int main() {
  std::mt19937_64 rng(11); std::normal_distribution<double> N(0, 1); std::uniform_real_distribution<double> Uf(0, 1);
  Opts<double> o{1e-8, 1e-4, 1e-2, 0.25, 1e-3, 0.1, 0.0, 100, 25, 0};
  double fric[4] = {0, 0, 0, 0};
  long fails = 0, iters = 0; double acc = 0;
  const int B = 512;
  for (int b = 0; b < B; ++b) {
    double x[10] = {0, 0, 0, -0.1 - 1e-8 - std::fabs(0.01 * N(rng)), -0.01 + 0.02 * N(rng)};
    for (int i = 0; i < 5; ++i) x[5 + i] = x[i];
    x[8] += std::fabs(0.003 * N(rng));
    double u[2] = {1.5 * Uf(rng), 0.2 * N(rng)};
    double th[M::NTH], z[M::NZ];
    mech_setup<M>(x, x + 5, u, fric, 0.1, th, z);
    Sink s; int it[2];
    const int st = ip_step_grad<M, double, Sink>(o, th, z, true, true, s, it);
    fails += (st & 3) != 3; iters += it[0]; acc += s.g + z[0];
  }
  std::printf("non-converged %ld / %d, mean iterations %.3f, checksum %.12e\n", fails, B, (double)iters / B, acc);
  return 0;
}
