/* A host that is neither Python nor C++: include/drs.h must compile as C, and the step-level entry points must link and answer
 * from plain C.  No GPU is touched: drs_net_create only builds the net tables and the buffer / variable lists.
 * Built and run by tests/test_c_abi_client.py:  gcc -std=c99 -Wall -Werror -Iinclude abi_client.c -L<pkg> -ldrs_hip */
#include <stdio.h>
#include <string.h>
#include "drs.h"

int main(int argc, char** argv) {
  const char* net_type = argc > 1 ? argv[1] : "dilated_grsl_rate8";
  drs_net_t* net = NULL;
  int rc = drs_net_create(net_type, 5, 6, 0.005f, 128, 85, 1, 0.5f, &net);
  if (rc != DRS_OK) { printf("create rc=%d\n", rc); return 2; }
  size_t n_params = 0, n_decay = 0, n_bn = 0, total = 0;
  int n_layers = 0, ld = 0, halo = 0;
  drs_net_layout(net, &n_params, &n_decay, &n_bn, &n_layers, &ld, &halo);
  for (int i = 0; i < drs_net_num_buffers(net); ++i) {
    char name[64]; size_t bytes; int dtype;
    if (drs_net_buffer_info(net, i, name, (int)sizeof name, &bytes, &dtype) != DRS_OK) return 3;
    total += bytes;
  }
  int found = 0;
  for (int i = 0; i < drs_net_num_variables(net); ++i) {
    char name[96]; size_t off, cnt; int shape[4], in_bn;
    drs_net_variable_info(net, i, name, (int)sizeof name, &off, &cnt, shape, &in_bn);
    if (strcmp(name, "conv8/weights") == 0 && shape[0] == 3 && shape[2] == 256 && shape[3] == 256 && !in_bn) found = 1;
  }
  /* nothing is bound: a step must be refused with DRS_ERR_ARG, not run */
  rc = drs_train_step(net, 128, 64, 0.01f, DRS_USE_ACC_MASK, 0.0, NULL);
  printf("layers=%d params=%zu decay=%zu bn=%zu x0_ld=%d x0_halo=%d buffers=%d bytes=%zu conv8=%d unbound_step_rc=%d lr=%.6f\n", n_layers, n_params,
         n_decay, n_bn, ld, halo, drs_net_num_buffers(net), total, found, rc, (double)drs_net_learning_rate(net, 0.01f));
  drs_net_destroy(net);
  return 0;
}
