/* The C ABI from plain C: include/tsm_hip.h must compile as C99 and libtsm_hip.so must be usable without
 * Python.  Built by tests/test_abi.py with gcc; executed only where a GPU exists (tests/test_engine_gpu.py).
 * Usage: abi_c_smoke            -> checks the no-GPU / bad-argument error paths only (exit 0)
 *        abi_c_smoke run       -> additionally creates an engine with all-zero weights and runs one clip */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "tsm_hip.h"

static int fill_zero_weights(tsm_engine *e) {
  /* every tensor of TSM.state_dict(): zeros, BatchNorm variance ones -> logits must equal fc.bias */
  static const int blocks[4] = {3, 4, 6, 3}, planes[4] = {64, 128, 256, 512};
  char name[128];
  int cin = 64, rc;
  int64_t shp[4];
#define SET(nm, n0, n1, n2, n3, nd, val)                                              \
  do {                                                                                \
    size_t cnt = (size_t)(n0) * (n1) * (n2) * (n3);                                   \
    float *buf = (float *)malloc(cnt * sizeof(float));                                \
    for (size_t i = 0; i < cnt; ++i) buf[i] = (val);                                  \
    shp[0] = (n0); shp[1] = (n1); shp[2] = (n2); shp[3] = (n3);                        \
    rc = tsm_set_tensor(e, nm, buf, shp, nd);                                         \
    free(buf);                                                                        \
    if (rc) { fprintf(stderr, "%s: %s\n", nm, tsm_last_error(e)); return rc; }       \
  } while (0)
#define BN(prefix, c)                                                   \
  do {                                                                  \
    snprintf(name, sizeof name, "%s.weight", prefix); SET(name, c, 1, 1, 1, 1, 1.0f);       \
    snprintf(name, sizeof name, "%s.bias", prefix); SET(name, c, 1, 1, 1, 1, 0.0f);         \
    snprintf(name, sizeof name, "%s.running_mean", prefix); SET(name, c, 1, 1, 1, 1, 0.0f); \
    snprintf(name, sizeof name, "%s.running_var", prefix); SET(name, c, 1, 1, 1, 1, 1.0f);  \
  } while (0)
  SET("base_model.conv1.weight", 64, 3, 7, 7, 4, 0.0f);
  BN("base_model.bn1", 64);
  for (int l = 0; l < 4; ++l)
    for (int b = 0; b < blocks[l]; ++b) {
      char p[64], q[96];
      snprintf(p, sizeof p, "base_model.layer%d.%d", l + 1, b);
      snprintf(name, sizeof name, "%s.conv1.net.weight", p); SET(name, planes[l], cin, 1, 1, 4, 0.0f);
      snprintf(q, sizeof q, "%s.bn1", p); BN(q, planes[l]);
      snprintf(name, sizeof name, "%s.conv2.weight", p); SET(name, planes[l], planes[l], 3, 3, 4, 0.0f);
      snprintf(q, sizeof q, "%s.bn2", p); BN(q, planes[l]);
      snprintf(name, sizeof name, "%s.conv3.weight", p); SET(name, planes[l] * 4, planes[l], 1, 1, 4, 0.0f);
      snprintf(q, sizeof q, "%s.bn3", p); BN(q, planes[l] * 4);
      if (b == 0) {
        snprintf(name, sizeof name, "%s.downsample.0.weight", p); SET(name, planes[l] * 4, cin, 1, 1, 4, 0.0f);
        snprintf(q, sizeof q, "%s.downsample.1", p); BN(q, planes[l] * 4);
      }
      cin = planes[l] * 4;
    }
  SET("fc.weight", 12, 2048, 1, 1, 2, 0.0f);
  {
    float bias[12];
    for (int i = 0; i < 12; ++i) bias[i] = 0.5f * (float)i - 1.0f;
    shp[0] = 12;
    rc = tsm_set_tensor(e, "fc.bias", bias, shp, 1);
    if (rc) return rc;
  }
  return 0;
}

int main(int argc, char **argv) {
  tsm_config cfg;
  tsm_engine *e = NULL;
  if (tsm_abi_version() != TSM_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 2; }
  memset(&cfg, 0, sizeof cfg);
  cfg.struct_size = 4;  /* wrong on purpose */
  if (tsm_create(&cfg, &e) != TSM_ERR_INVALID_ARG || e != NULL) { fprintf(stderr, "struct_size not checked\n"); return 3; }
  if (strstr(tsm_last_error(NULL), "struct_size") == NULL) { fprintf(stderr, "no message\n"); return 4; }
  if (tsm_forward(NULL, NULL, TSM_MEM_HOST, TSM_LAYOUT_NTCHW, 1, NULL, NULL) != TSM_ERR_INVALID_ARG) return 5;
  tsm_destroy(NULL);
  if (argc < 2 || strcmp(argv[1], "run") != 0) { printf("abi_c_smoke: error paths ok\n"); return 0; }

  cfg.struct_size = (int32_t)sizeof cfg; cfg.num_class = 12; cfg.num_segments = 8; cfg.height = 64; cfg.width = 64;
  cfg.shift_div = 8; cfg.is_shift = 1; cfg.max_clips = 1; cfg.device_id = 0; cfg.dtype = TSM_DTYPE_F32;
  if (tsm_create(&cfg, &e)) { fprintf(stderr, "create: %s\n", tsm_last_error(NULL)); return 6; }
  if (fill_zero_weights(e)) return 7;
  if (tsm_finalize(e)) { fprintf(stderr, "finalize: %s\n", tsm_last_error(e)); return 8; }
  {
    const size_t n = (size_t)8 * 3 * 64 * 64;
    float *clip = (float *)malloc(n * sizeof(float)), logits[12];
    for (size_t i = 0; i < n; ++i) clip[i] = (float)(i % 17) - 8.0f;
    if (tsm_forward(e, clip, TSM_MEM_HOST, TSM_LAYOUT_NTCHW, 1, logits, NULL)) { fprintf(stderr, "forward: %s\n", tsm_last_error(e)); return 9; }
    for (int i = 0; i < 12; ++i)
      if (logits[i] != 0.5f * (float)i - 1.0f) { fprintf(stderr, "logit %d = %g\n", i, logits[i]); return 10; }
    free(clip);
  }
  tsm_destroy(e);
  printf("abi_c_smoke: forward ok\n");
  return 0;
}
