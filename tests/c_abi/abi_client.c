/* A caller of the drop-in boundary written in plain C99 against include/od_mi355x.h only -- no HIP header, no C++, no Python: what a
 * cgo / ccall / JNI binding of a maintainer sees.  It asks the model table, creates the hopper's ImplicitDynamics
 * (reference: src/dynamics.jl:51-79), evaluates f, fx and fu at one (x, u) through the host-pointer entry points (the reference's
 * callbacks f(d, model, x, u, w), fx, fu: src/dynamics.jl:81-128) and prints the numbers with 17 digits; tests/test_abi.py compiles it
 * with gcc -std=c99 -pedantic -Wall -Werror, runs it, and compares the output with the Python mirror and the oracle.
 * Exit code 0: computed; 77: od_create reported that there is no device (the product has no CPU path); anything else: a failure. */
#include <stdio.h>
#include <string.h>
#include "od_mi355x.h"

static int die(const char* what, int rc) {
  fprintf(stderr, "%s: rc = %d: %s\n", what, rc, od_last_error());
  return 1;
}

int main(void) {
  int nq = 0, nu = 0, nz = 0, nth = 0, nfric = 0, i, rc;
  od_options o;
  od_handle h = 0;
  double x[8] = {0.0, 0.55, 0.0, 0.5, 0.0, 0.55, 0.0, 0.5};
  double u[2] = {0.0, 0.73575};            /* half the weight of the body over one step of h = 0.05 (examples/hopper.jl:270) */
  double d[8], dx[64], du[16], d2[8];
  int dev = -1;

  if (od_version() < 100) return die("od_version", od_version());
  if (od_model_id("hopper") != OD_HOPPER || strcmp(od_model_name(OD_HOPPER), "hopper") != 0) return die("model table", -1);
  if ((rc = od_model_dims(OD_HOPPER, &nq, &nu, &nz, &nth, &nfric)) != OD_OK) return die("od_model_dims", rc);
  if (nq != 4 || nu != 2) return die("hopper dimensions", -1);
  if ((rc = od_default_options(OD_HOPPER, &o)) != OD_OK) return die("od_default_options", rc);
  printf("hopper nq %d nu %d nz %d ntheta %d r_tol %.3g kappa_eval %.3g kappa_grad %.3g\n", nq, nu, nz, nth, o.r_tol, o.kappa_eval_tol, o.kappa_grad_tol);
  printf("constraints %d hopper_foot %d\n", od_num_constraints(), od_constraint_id("hopper_foot"));

  rc = od_create(OD_HOPPER, OD_F64, &o, 0.05, &h);
  if (rc == OD_ERR_NO_DEVICE) {
    printf("no device: %s\n", od_last_error());
    return 77;
  }
  if (rc != OD_OK) return die("od_create", rc);
  if ((rc = od_get_device(h, &dev)) != OD_OK) return die("od_get_device", rc);
  if ((rc = od_ffxfu_host(h, x, u, d, dx, du)) != OD_OK) return die("od_ffxfu_host", rc);
  if ((rc = od_f_host(h, x, u, d2)) != OD_OK) return die("od_f_host", rc);
  for (i = 0; i < 8; ++i) if (d[i] != d2[i]) return die("od_f_host differs from od_ffxfu_host", -1);
  printf("device %d\n", dev);
  printf("d");
  for (i = 0; i < 8; ++i) printf(" %.17g", d[i]);
  printf("\ndx");
  for (i = 0; i < 64; ++i) printf(" %.17g", dx[i]);
  printf("\ndu");
  for (i = 0; i < 16; ++i) printf(" %.17g", du[i]);
  printf("\n");
  /* an error is a code and a message, never a crash */
  if (od_f_host(h, 0, u, d) != OD_ERR_INVALID) return die("null argument was not refused", -1);
  if ((rc = od_destroy(h)) != OD_OK) return die("od_destroy", rc);
  return 0;
}
