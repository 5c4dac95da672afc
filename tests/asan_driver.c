/* Host-side paths of libodevio's C ABI under AddressSanitizer (no GPU needed): argument validation, error strings, the
 * config ABI guard, plan creation up to the point where a device is required, and the resize coefficient tables.
 * Built and run by tests/test_asan_host.py against libodevio_asan.so (make -C odevio_amd/csrc ASAN=1). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/odevio.h"

#define EXPECT(cond)                                                     \
  do {                                                                   \
    if (!(cond)) { fprintf(stderr, "FAILED: %s (line %d): %s\n", #cond, __LINE__, odevio_last_error()); return 1; } \
  } while (0)

int main(void) {
  EXPECT(odevio_version() == ODEVIO_VERSION);
  odevio_plan* plan = NULL;
  EXPECT(odevio_plan_create(NULL, NULL, 0, NULL, &plan) == ODEVIO_ERR_BAD_ARG);
  odevio_config cfg;
  memset(&cfg, 0, sizeof(cfg));
  odevio_tensor w[2] = {{"Image_net.conv1.0.weight", NULL, 1}, {NULL, NULL, 0}};
  cfg.struct_size = 4;
  EXPECT(odevio_plan_create(&cfg, w, 2, NULL, &plan) == ODEVIO_ERR_BAD_ARG && strstr(odevio_last_error(), "size mismatch"));
  cfg.struct_size = (int32_t)sizeof(cfg);
  cfg.model_type = 7;
  EXPECT(odevio_plan_create(&cfg, w, 2, NULL, &plan) == ODEVIO_ERR_UNSUPPORTED);
  cfg.model_type = ODEVIO_MODEL_ODE_RNN; cfg.img_h = 8; cfg.img_w = 8;
  EXPECT(odevio_plan_create(&cfg, w, 2, NULL, &plan) == ODEVIO_ERR_BAD_ARG);
  cfg.img_h = 256; cfg.img_w = 512; cfg.v_f_len = 512; cfg.i_f_len = 256; cfg.ode_hidden_dim = 512; cfg.ode_fn_num_layers = 3;
  cfg.rnn_num_layers = 2; cfg.ode_substeps = 1; cfg.ode_solver = 99;
  EXPECT(odevio_plan_create(&cfg, w, 2, NULL, &plan) == ODEVIO_ERR_BAD_ARG && strstr(odevio_last_error(), "Solver"));
  cfg.ode_solver = ODEVIO_RK4; cfg.rnn_num_layers = 99;
  EXPECT(odevio_plan_create(&cfg, w, 2, NULL, &plan) == ODEVIO_ERR_UNSUPPORTED);
  cfg.rnn_num_layers = 2;
  /* a valid config: without a GPU this ends with NO_DEVICE (or a HIP error), never with a memory error */
  const int rc = odevio_plan_create(&cfg, w, 2, NULL, &plan);
  EXPECT(rc != ODEVIO_OK && plan == NULL);
  /* every entry point refuses a null plan / null tensors */
  float x[6] = {0};
  EXPECT(odevio_forward(NULL, x, x, 101, x, NULL, 1, 11, x, x, NULL, NULL) == ODEVIO_ERR_BAD_ARG);
  EXPECT(odevio_forward_u8(NULL, (const uint8_t*)x, x, 101, x, NULL, 1, 11, x, x, NULL, NULL) == ODEVIO_ERR_BAD_ARG);
  EXPECT(odevio_ode_rnn_fwd(NULL, x, x, NULL, 1, 1, x, x, NULL, NULL) == ODEVIO_ERR_BAD_ARG);
  EXPECT(odevio_ode_rnn_bwd(NULL, x, x, NULL, 1, 1, x, NULL, NULL, NULL, NULL, 0, NULL) == ODEVIO_ERR_BAD_ARG);
  EXPECT(odevio_cde_fwd(NULL, x, 1, 2, NULL, 1, NULL, x, x, NULL, NULL) == ODEVIO_ERR_BAD_ARG);
  EXPECT(odevio_check(NULL, NULL) == ODEVIO_ERR_BAD_ARG);
  EXPECT(odevio_fuse_bwd(NULL, x, x, 1, x, x, x, NULL, 0, NULL) == ODEVIO_ERR_BAD_ARG);
  EXPECT(odevio_imu_encoder_bwd(NULL, x, 1, 11, x, NULL, 0, NULL) == ODEVIO_ERR_BAD_ARG);
  EXPECT(odevio_grad_clip(NULL, NULL, 0, 5.0f, x, NULL) == ODEVIO_ERR_BAD_ARG);
  EXPECT(odevio_adam_step(x, x, x, x, 4, 1e-4f, 0.9f, 0.999f, 1e-8f, 0.0f, 0, NULL, NULL) == ODEVIO_ERR_BAD_ARG);   /* step counts from 1 */
  EXPECT(odevio_plan_update(NULL, NULL, 0, NULL) == ODEVIO_ERR_BAD_ARG);
  EXPECT(odevio_fuse_hard_bwd(NULL, x, x, 1, 0, 0, x, x, x, NULL, 0, NULL) == ODEVIO_ERR_BAD_ARG);
  EXPECT(odevio_rng_state(NULL, NULL, NULL) == ODEVIO_ERR_BAD_ARG);
  EXPECT(odevio_debug_gumbel(0, 0, 0, NULL, NULL) == ODEVIO_ERR_BAD_ARG);
  EXPECT(odevio_set_seed(NULL, 1) == ODEVIO_ERR_BAD_ARG);
  EXPECT(odevio_set_rng_state(NULL, 1, 2) == ODEVIO_ERR_BAD_ARG);
  EXPECT(odevio_image_encoder_fwd_train(NULL, x, 1, 2, x, 512, NULL, 0, 0, NULL) == ODEVIO_ERR_BAD_ARG);
  EXPECT(odevio_imu_encoder_fwd_train(NULL, x, 1, 11, 0.0f, NULL, 0, x, 256, NULL) == ODEVIO_ERR_BAD_ARG);
  EXPECT(odevio_imu_encoder_bwd_train(NULL, x, 1, 11, 0.0f, 0, 0, x, NULL, 0, NULL) == ODEVIO_ERR_BAD_ARG);
  EXPECT(odevio_image_encoder_bwd(NULL, x, 1, 2, x, 512, NULL, 0, NULL) == ODEVIO_ERR_BAD_ARG);
  EXPECT(odevio_cde_bwd(NULL, x, 1, 2, NULL, 1, NULL, x, NULL, x, NULL, NULL, 0, NULL, NULL) == ODEVIO_ERR_BAD_ARG);
  EXPECT(odevio_debug_dropout(0, 0, 1.5f, 4, x, NULL) == ODEVIO_ERR_BAD_ARG);     /* p must be < 1 */
  EXPECT(odevio_sgd_step(x, x, NULL, 4, 1e-4f, 0.9f, 0.0f, 1, NULL, NULL) == ODEVIO_ERR_BAD_ARG);   /* momentum needs its buffer */
  EXPECT(odevio_optimizer_step(2, NULL, NULL, NULL, NULL, NULL, 1, 0.9f, 0.999f, 1e-8f, 0.0f, 1, NULL, NULL) == ODEVIO_ERR_BAD_ARG);
  {
    int64_t n = -1;
    EXPECT(odevio_ode_rnn_tape_floats(NULL, 1, 1, &n) == ODEVIO_ERR_BAD_ARG);
    EXPECT(odevio_ode_rnn_fwd_taped(NULL, x, x, NULL, 1, 1, x, x, x, 4, NULL) == ODEVIO_ERR_BAD_ARG);
    EXPECT(odevio_ode_rnn_bwd_taped(NULL, x, x, NULL, 1, 1, x, NULL, NULL, NULL, NULL, 0, NULL, 0, NULL) == ODEVIO_ERR_BAD_ARG);
  }
  EXPECT(odevio_resize_u8(NULL, 1, 4, 4, NULL, 2, 2, NULL, NULL) == ODEVIO_ERR_BAD_ARG);
  odevio_plan_destroy(NULL);
  /* resize tables: KITTI width and height, an upscale, a degenerate 1-pixel axis; capacity checked */
  const int cases[4][2] = {{1241, 512}, {376, 256}, {5, 17}, {1, 3}};
  for (int c = 0; c < 4; ++c) {
    const int in = cases[c][0], out = cases[c][1];
    int ksize = 0;
    int* bounds = (int*)malloc(sizeof(int) * 2 * out);
    const int cap = out * 9;
    int* kk = (int*)malloc(sizeof(int) * cap);
    EXPECT(odevio_resize_table(in, out, &ksize, bounds, kk, cap) == 0);
    EXPECT(ksize >= 3 && ksize * out <= cap);
    for (int i = 0; i < out; ++i) {
      long sum = 0;
      EXPECT(bounds[2 * i] >= 0 && bounds[2 * i + 1] >= 1 && bounds[2 * i] + bounds[2 * i + 1] <= in && bounds[2 * i + 1] <= ksize);
      for (int k = 0; k < bounds[2 * i + 1]; ++k) sum += kk[i * ksize + k];
      EXPECT(labs(sum - (1L << 22)) <= ksize);   /* normalised weights: 1.0 in 22-bit fixed point up to rounding */
    }
    EXPECT(odevio_resize_table(in, out, &ksize, bounds, kk, 1) == ODEVIO_ERR_BAD_ARG);   /* too small: refused, nothing written past it */
    free(bounds);
    free(kk);
  }
  printf("asan driver: ok\n");
  return 0;
}
