/* Plain-C user of libcnerf_hip.so (include/cnerf.h): no PyTorch, no C++ — what a maintainer binding the library from
 * another host language would write.  Reads a little-endian float32/int32 blob prepared by the test (two networks'
 * parameter tensors in the C-ABI order, a ray batch, the random streams, upstream gradients), runs
 *   cnerf_pack_weights x2 -> cnerf_render_fwd (train) -> cnerf_render_bwd
 * and writes every output map and parameter gradient back as one blob.  tests/test_gpu_parity.py compares that blob
 * bit for bit with the same calls made through ctypes.
 * usage: render_smoke <in.bin> <out.bin>      build: gcc -std=c99 (see the test) */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cnerf.h"

#define CHECK(x)                                                               \
  do {                                                                         \
    int rc__ = (int)(x);                                                       \
    if (rc__ != 0) { fprintf(stderr, "%s -> %d (%s)\n", #x, rc__, rc__ < 0 ? cnerf_strerror(rc__) : "hip error"); return 2; } \
  } while (0)

static float* to_device(const float* h, size_t n) {
  float* d = NULL;
  if (hipMalloc((void**)&d, (n ? n : 1) * sizeof(float)) != hipSuccess) return NULL;
  if (n && hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return NULL;
  return d;
}
static float* device_floats(size_t n) {
  float* d = NULL;
  return hipMalloc((void**)&d, (n ? n : 1) * sizeof(float)) == hipSuccess ? d : NULL;
}

int main(int argc, char** argv) {
  if (argc != 3) return 1;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 1;
  int32_t hdr[12];   /* D W multires multires_views use_viewdirs output_ch skip | B Nc Nf white ray_stride */
  if (fread(hdr, 4, 12, f) != 12) return 1;
  cnerf_net net = {hdr[0], hdr[1], hdr[2], hdr[3], hdr[4], hdr[5], hdr[6]};
  const int64_t B = hdr[7];
  cnerf_render_cfg cfg = {hdr[8], hdr[9], 0, hdr[10], hdr[11], 1};
  const int S = cfg.Nc + cfg.Nf, C = net.use_viewdirs ? 4 : net.output_ch;
  const int nt = cnerf_num_tensors(&net);
  if (nt <= 0) return 1;

  cnerf_ptrs params[2], grads[2];
  size_t numel[CNERF_MAX_TENSORS];
  float* packed[2];
  memset(params, 0, sizeof params); memset(grads, 0, sizeof grads);
  for (int k = 0; k < 2; ++k) {
    for (int i = 0; i < nt; ++i) {
      int64_t r, c;
      CHECK(cnerf_tensor_shape(&net, i, &r, &c));
      numel[i] = (size_t)(r * c);
      float* h = (float*)malloc(numel[i] * 4);
      if (fread(h, 4, numel[i], f) != numel[i]) return 1;
      params[k].p[i] = to_device(h, numel[i]);
      grads[k].p[i] = device_floats(numel[i]);
      free(h);
      if (!params[k].p[i] || !grads[k].p[i]) return 3;
    }
    packed[k] = device_floats((size_t)cnerf_packed_floats(&net));
    CHECK(cnerf_pack_weights(&net, &params[k], packed[k], NULL));
  }
  /* rays[B,rs], t_vals[Nc], t_rand[B,Nc], u[B,Nf], then the 8 upstream gradients */
  const size_t n_in[] = {(size_t)B * cfg.ray_stride, (size_t)cfg.Nc, (size_t)B * cfg.Nc, (size_t)B * cfg.Nf,
                         (size_t)B * 3, (size_t)B, (size_t)B, (size_t)B, (size_t)B * 3, (size_t)B, (size_t)B, (size_t)B};
  float* in[12];
  for (int i = 0; i < 12; ++i) {
    float* h = (float*)malloc((n_in[i] ? n_in[i] : 1) * 4);
    if (fread(h, 4, n_in[i], f) != n_in[i]) return 1;
    in[i] = to_device(h, n_in[i]);
    free(h);
    if (!in[i]) return 3;
  }
  fclose(f);

  const int64_t wsn = cnerf_render_ws_floats(&net, &net, &cfg, B);
  if (wsn < 0) return 4;
  float* ws = device_floats((size_t)wsn);
  const size_t n_out[] = {(size_t)B * 3, (size_t)B, (size_t)B, (size_t)B, (size_t)B * 3, (size_t)B, (size_t)B, (size_t)B,
                          (size_t)B, (size_t)B * S * C, (size_t)B * S, (size_t)B * S};
  float* o[12];
  for (int i = 0; i < 12; ++i)
    if (!(o[i] = device_floats(n_out[i]))) return 3;
  cnerf_render_out out = {o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7], o[8], o[9], o[10], o[11]};
  CHECK(cnerf_render_fwd(&net, packed[0], &net, packed[1], in[0], B, &cfg, in[1], in[2], in[3], cfg.Nf, NULL, NULL, &out,
                         ws, NULL));
  cnerf_render_grads g = {in[4], in[5], in[6], in[7], in[8], in[9], in[10], in[11]};
  CHECK(cnerf_render_bwd(&net, packed[0], &net, packed[1], in[0], B, &cfg, NULL, NULL, &g, ws, &grads[0], &grads[1], 0,
                         NULL));
  CHECK(hipDeviceSynchronize());

  FILE* w = fopen(argv[2], "wb");
  if (!w) return 1;
  for (int i = 0; i < 12; ++i) {
    float* h = (float*)malloc((n_out[i] ? n_out[i] : 1) * 4);
    CHECK(hipMemcpy(h, o[i], n_out[i] * 4, hipMemcpyDeviceToHost));
    fwrite(h, 4, n_out[i], w);
    free(h);
  }
  for (int k = 0; k < 2; ++k)
    for (int i = 0; i < nt; ++i) {
      float* h = (float*)malloc(numel[i] * 4);
      CHECK(hipMemcpy(h, grads[k].p[i], numel[i] * 4, hipMemcpyDeviceToHost));
      fwrite(h, 4, numel[i], w);
      free(h);
    }
  fclose(w);
  printf("render_smoke ok: B=%lld S=%d tensors=%d workspace=%lld floats\n", (long long)B, S, nt, (long long)wsn);
  return 0;
}
