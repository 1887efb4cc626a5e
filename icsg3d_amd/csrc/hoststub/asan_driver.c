/* Host driver of the `make asan` build: walks the C ABI of include/icsg3d.h end to end on the stubbed runtime
 * (hoststub/hip_host_stub.cpp) so that AddressSanitizer / UBSan / LeakSanitizer see every host-side code path:
 * engine construction (tensor tables, BN slab, workspace planners), named tensor I/O, train / test / predict
 * call chains (launch planning, split-K and bucket bookkeeping, profiler rows), the data-parallel entry points,
 * the fused inference tail, optimizer-state export and destruction.  Kernels do not run; results are not checked. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "icsg3d.h"

#define OK(x)                                                              \
  do {                                                                     \
    if ((x) != 0) {                                                        \
      fprintf(stderr, "FAILED %s: %s\n", #x, ics_last_error());            \
      return 1;                                                            \
    }                                                                      \
  } while (0)

static int tensor_io(ics_net* net) {
  int n = 0;
  OK(ics_net_num_tensors(net, &n));
  for (int i = 0; i < n; ++i) {
    const char* name; int nd, tr; int64_t dims[5];
    OK(ics_net_tensor_info(net, i, &name, &nd, dims, &tr));
    size_t cnt = 1;
    for (int k = 0; k < nd; ++k) cnt *= (size_t)dims[k];
    float* buf = (float*)malloc(cnt * sizeof(float));      /* exact size: an over-read/-write is an ASan report */
    for (size_t k = 0; k < cnt; ++k) buf[k] = 0.01f * (float)(k % 7);
    OK(ics_net_set_tensor(net, name, buf, cnt));
    OK(ics_net_get_tensor(net, name, buf, cnt));
    if (tr) OK(ics_net_get_grad(net, name, buf, cnt));
    if (ics_net_set_tensor(net, name, buf, cnt + 1) == 0) { fprintf(stderr, "size check missing for %s\n", name); return 1; }
    free(buf);
  }
  return 0;
}

static int run(int d, int C, int B, int with_comm) {
  const size_t M = (size_t)B * d * d * d;
  ics_unet_config uc = {C, 95, d, B, 1e-3f, 0.f, 1, 1, with_comm == 2};
  ics_vae_config vc = {C, 10, 256, {16, 32, 64, 128}, d, B, 5e-4f, 0.5f, 3e-4f, {1.f, 1.f, 1.f, 1.f}, 1};
  ics_net *unet = NULL, *vae = NULL;
  OK(ics_unet_create(&uc, &unet));
  OK(ics_vae_create(&vc, unet, &vae));
  if (tensor_io(unet) || tensor_io(vae)) return 1;
  if (with_comm) {
    char uid[128];
    OK(ics_comm_unique_id(uid));
    OK(ics_net_comm_init(unet, 0, 1, uid));
    OK(ics_net_comm_init(vae, 0, 1, uid));
    OK(ics_net_comm_broadcast_state(unet, 0));
    OK(ics_net_comm_broadcast_state(vae, 0));
    OK(ics_net_set_sync_bn(unet, with_comm > 1));
    OK(ics_net_set_sync_bn(vae, with_comm > 1));
    double v = 1.0;
    OK(ics_net_comm_allreduce_max(unet, &v));
  }
  float* x = (float*)calloc(M * C, sizeof(float));
  uint8_t* lab = (uint8_t*)calloc(M, 1);
  float* soft = (float*)malloc(M * 95 * sizeof(float));
  float* sig = (float*)malloc(M * sizeof(float));
  uint8_t* sp = (uint8_t*)malloc(M);
  uint8_t* mk = (uint8_t*)malloc(M);
  float* cond = (float*)calloc((size_t)B * 10, sizeof(float));
  float* eps = (float*)calloc((size_t)B * 256, sizeof(float));
  float* z = (float*)calloc((size_t)B * 256 * 3, sizeof(float));
  float* rec = (float*)malloc(M * C * sizeof(float));
  float mm[5], mv[4], minmax[6 * 8];
  { ics_net* v2 = NULL; OK(ics_vae_create(&vc, unet, &v2)); OK(ics_net_share_stream(v2, unet)); OK(ics_net_destroy(v2)); }
  OK(ics_net_profile_enable(unet, 1));
  OK(ics_net_profile_enable(vae, 1));
  for (int b = 1; b <= B; ++b) {   /* every batch size up to max_batch: the split plans depend on it */
    OK(ics_unet_train_step(unet, x, lab, b, mm));
    OK(ics_unet_test_step(unet, x, lab, b, mm));
    { double ms[7]; OK(ics_unet_metric_sums(unet, ms)); }
    OK(ics_unet_upload_batch(unet, x, lab, b));
    OK(ics_net_timer_start(unet));
    OK(ics_unet_predict_resident(unet, 0, 0.8f));
    OK(ics_unet_predict_resident(unet, 1, 0.8f));
    { double tms; OK(ics_net_timer_stop(unet, &tms)); }
    OK(ics_unet_predict(unet, x, b, soft, sig));
    OK(ics_unet_predict_labels(unet, x, b, 0.8f, sp, mk));
    OK(ics_vae_train_step(vae, x, cond, eps, b, mv));
    OK(ics_vae_test_step(vae, x, cond, eps, b, mv));
    OK(ics_vae_encode(vae, x, cond, eps, b, z, z + (size_t)B * 256, z + (size_t)B * 512));
    OK(ics_vae_decode(vae, z, cond, b, rec));
    OK(ics_vae_decode_to_unet_labels(vae, unet, z, cond, b, 0.8f, sp, mk, rec, minmax));
    {   /* ... continued through connected components / region statistics */
      int32_t* regions = (int32_t*)malloc(M * sizeof(int32_t));
      int32_t counts[2 * 8];
      int32_t* stats = (int32_t*)malloc((size_t)b * 64 * 11 * sizeof(int32_t));
      { int64_t* bnd = calloc((size_t)b * 64 * 8, 8); OK(ics_vae_decode_to_unet_atoms(vae, unet, z, cond, b, 0.8f, 3, 64, sp, mk, rec, minmax, regions, counts, stats, bnd)); free(bnd); }
      OK(ics_vae_decode_to_unet_atoms(vae, unet, z, cond, b, 0.8f, 3, 64, NULL, NULL, NULL, NULL, NULL, counts, stats, NULL));
      free(regions); free(stats);
    }
  }
  OK(ics_unet_upload_batch(unet, x, lab, B));
  OK(ics_unet_train_step_resident(unet, NULL));
  OK(ics_vae_upload_batch(vae, x, cond, eps, B));
  OK(ics_vae_train_step_resident(vae, mv));
  if (ics_unet_train_step(unet, x, lab, B + 1, mm) == 0) { fprintf(stderr, "max_batch check missing\n"); return 1; }
  int rows = 0;
  OK(ics_net_profile_count(unet, &rows));
  for (int r = 0; r < rows; ++r) {
    const char* label; int64_t launches; double ms, fl, by;
    OK(ics_net_profile_row(unet, r, &label, &launches, &ms, &fl, &by));
  }
  size_t np = 0;
  OK(ics_net_num_params(unet, &np));
  float* m1 = (float*)malloc(np * sizeof(float));
  float* v1 = (float*)malloc(np * sizeof(float));
  int step = 0;
  OK(ics_net_get_optimizer_state(unet, m1, v1, np, &step));
  OK(ics_net_set_optimizer_state(unet, m1, v1, np, step));
  OK(ics_net_reset_optimizer(unet));
  { int dirty = -1; OK(ics_net_check_canaries(unet, &dirty)); OK(ics_net_check_canaries(vae, &dirty)); }
  {   /* the graph probe needs the profiler off and a resident batch */
    double em, gm; int nodes, rk, nr, nb;
    OK(ics_net_profile_enable(unet, 0)); OK(ics_net_profile_enable(vae, 0));
    OK(ics_net_comm_info(unet, &rk, &nr, &nb));
    if (nr == 0) {
      OK(ics_net_graph_probe(unet, 2, &em, &gm, &nodes));
      OK(ics_net_graph_probe(vae, 2, &em, &gm, &nodes));
    } else if (ics_net_graph_probe(unet, 2, &em, &gm, &nodes) == 0) {
      fprintf(stderr, "graph probe must refuse an engine with a communicator\n"); return 1;
    }
    OK(ics_net_profile_enable(unet, 1)); OK(ics_net_profile_enable(vae, 1));
  }
  {
    size_t cnt = (size_t)B * (d / 8) * (d / 8) * (d / 8) * 512;      /* c10 activation */
    float* act = (float*)malloc(cnt * sizeof(float));
    float sc[512], sh[512];
    OK(ics_net_get_activation(unet, "c10", act, cnt));
    OK(ics_net_get_bn_affine(unet, "c10", sc, sh, 512));
    free(act);
  }
  int rk, nr, nb;
  OK(ics_net_comm_info(unet, &rk, &nr, &nb));
  free(m1); free(v1); free(x); free(lab); free(soft); free(sig); free(sp); free(mk); free(cond); free(eps); free(z); free(rec);
  OK(ics_net_destroy(vae));
  OK(ics_net_destroy(unet));
  return 0;
}

int main(void) {
  char name[256]; int cus; size_t hbm; int ndev = 0;
  OK(ics_device_count(&ndev));
  OK(ics_set_device(0));
  OK(ics_device_info(name, &cus, &hbm));
  printf("asan driver on %s (%s)\n", name, ics_version());
  setenv("ICSG3D_DGRAD_BNFUSE_MIN", "0", 1);   /* walk the BatchNorm-backward-in-backward-data fusion at these small sizes too */
  if (run(16, 1, 3, 0)) return 1;
  if (run(16, 4, 2, 1)) return 1;      /* 4 channels (Cin 44 -> padded loaders), single-rank communicator */
  if (run(32, 1, 2, 2)) return 1;      /* d = 32 plans, SyncBN */
  if (run(32, 1, 5, 0)) return 1;      /* plans that are not monotone in the batch: 3 grids on a 5-grid handle need more
                                          BatchNorm-backward blocks than 5 do (the engine's capacity checks fire if a
                                          workspace was sized for max_batch alone, round 6) */
  {   /* single-op entry points */
    const int B = 1, S = 8, Cin = 32, Cout = 64;
    size_t nx = (size_t)B * S * S * S * Cin, ny = (size_t)B * S * S * S * Cout, nw = (size_t)27 * Cin * Cout;
    float *x = calloc(nx, 4), *y = calloc(ny, 4), *w = calloc(nw, 4), *dx = calloc(nx, 4), *dw = calloc(nw, 4);
    OK(ics_op_conv3d_forward(x, w, NULL, B, S, Cin, Cout, 27, 1, y));
    OK(ics_op_conv3d_backward(x, w, y, B, S, Cin, Cout, 27, dx, dw));
    free(x); free(y); free(w); free(dx); free(dw);
  }
  {   /* the op-level head: both launch sequences */
    const size_t M = 64;
    float *x = calloc(M * 128, 4), *ws = calloc(128 * 95, 4), *bs = calloc(95, 4), *wg = calloc(128, 4), bg[1] = {0.f};
    float *out = calloc(M * 96, 4), met[5];
    double sums[7];
    uint8_t* lab = calloc(M, 1);
    for (int f = 0; f < 2; ++f)
      for (int mode = 0; mode < 3; ++mode) OK(ics_op_unet_head(x, ws, bs, wg, bg, lab, M, 95, 0.f, mode, f, out, met, sums));
    free(x); free(ws); free(bs); free(wg); free(out); free(lab);
  }
  {   /* connected components on host arrays */
    const int B = 2, d = 16;
    const size_t M = (size_t)B * d * d * d;
    uint8_t *mask = calloc(M, 1), *species = calloc(M, 1);
    int32_t* regions = malloc(M * sizeof(int32_t));
    int32_t counts[4];
    int32_t* stats = malloc((size_t)B * 32 * 11 * sizeof(int32_t));
    { int64_t* bnd = calloc((size_t)B * 32 * 8, 8); OK(ics_op_segment_atoms(mask, species, B, d, 3, 32, 95, regions, counts, stats, bnd)); free(bnd); }
    OK(ics_op_segment_atoms(mask, species, B, d, 3, 32, 95, regions, counts, stats, NULL));
    if (ics_op_segment_atoms(mask, species, B, 24, 3, 32, 95, regions, counts, stats, NULL) == 0) { fprintf(stderr, "grid check missing\n"); return 1; }
    free(mask); free(species); free(regions); free(stats);
  }
  {  /* segment_nuclei's box operations (round 4): host-side batching, descriptor tables, scratch sizing */
    int32_t dims[6] = {3, 4, 5, 2, 2, 7};
    int32_t vols[60 + 28], labels[60 + 28], wss[60 + 28], nlab[2], cls[2] = {1, 5};
    int32_t bstats[2 * 8 * 7], rstats[4 * 11];
    uint8_t sp[60];
    memset(vols, 0, sizeof vols); memset(sp, 0, sizeof sp);
    vols[7] = 1; vols[8] = 1; vols[61] = 5;
    OK(ics_op_label_boxes(vols, dims, 2, 1, 8, labels, nlab, bstats));
    OK(ics_op_label_boxes(vols, dims, 2, 3, 8, labels, nlab, NULL));
    OK(ics_op_watershed_split(vols, dims, cls, 2, 0, wss));
    OK(ics_op_watershed_split(vols, dims, cls, 2, 1, wss));
    OK(ics_op_region_stats(vols, sp, 3, 4, 5, 4, 95, rstats));
    dims[0] = 65;
    if (ics_op_label_boxes(vols, dims, 2, 1, 8, labels, nlab, bstats) == 0) { fprintf(stderr, "box extent check missing\n"); return 1; }
    if (ics_op_label_boxes(vols, dims + 3, 1, 2, 8, labels, nlab, bstats) == 0) { fprintf(stderr, "connectivity check missing\n"); return 1; }
    OK(ics_release_caches());     /* the per-thread stream and scratch of the box-level entry points (LeakSanitizer sees the rest) */
  }
  {  /* round 6: the HOST forms of the box operations are real code under this build (threads, heap flood, integer gift
        wrapping): a ragged pseudo-random volume, labelled here by a plain flood fill, through ics_op_component_bounds (with
        and without the exact hull count) and ics_op_watershed_split (many boxes: the thread pool; both tie rules) */
    enum { D = 20, H = 18, W = 16, V = D * H * W, ML = 512 };
    int32_t* vol = (int32_t*)calloc(V, 4);
    int32_t* lab = (int32_t*)calloc(V, 4);
    int32_t* stack = (int32_t*)malloc(V * 4);
    int32_t* st7 = (int32_t*)calloc(ML * 7, 4);
    unsigned lcg = 12345u;
    for (int i = 0; i < V; ++i) { lcg = lcg * 1664525u + 1013904223u; vol[i] = ((lcg >> 16) % 100) < 38; }
    for (int z = 6; z < 14; ++z) for (int y = 5; y < 13; ++y) for (int x = 4; x < 12; ++x) vol[(z * H + y) * W + x] = 1;   /* a solid core */
    int n = 0;
    for (int s0 = 0; s0 < V && n < ML; ++s0) {
      if (!vol[s0] || lab[s0]) continue;
      ++n;
      int top = 0; stack[top++] = s0; lab[s0] = n;
      int32_t* r = st7 + (n - 1) * 7;
      r[0] = 0; r[1] = D; r[2] = H; r[3] = W; r[4] = r[5] = r[6] = 0;
      while (top) {
        const int i = stack[--top], x = i % W, y = (i / W) % H, z = i / (W * H);
        r[0] += 1;
        if (z < r[1]) r[1] = z; if (y < r[2]) r[2] = y; if (x < r[3]) r[3] = x;
        if (z + 1 > r[4]) r[4] = z + 1; if (y + 1 > r[5]) r[5] = y + 1; if (x + 1 > r[6]) r[6] = x + 1;
        const int nb[6] = {z > 0 ? i - H * W : -1, z < D - 1 ? i + H * W : -1, y > 0 ? i - W : -1, y < H - 1 ? i + W : -1,
                           x > 0 ? i - 1 : -1, x < W - 1 ? i + 1 : -1};
        for (int k = 0; k < 6; ++k)
          if (nb[k] >= 0 && vol[nb[k]] && !lab[nb[k]]) { lab[nb[k]] = n; stack[top++] = nb[k]; }
      }
    }
    int32_t dims3[3] = {D, H, W}, nl[1] = {n};
    int64_t* bnd = (int64_t*)calloc((size_t)ML * 5, 8);
    OK(ics_op_component_bounds(lab, dims3, 1, nl, st7, ML, 3, 0.0, bnd));
    OK(ics_op_component_bounds(lab, dims3, 1, nl, st7, ML, 0, 0.3, bnd));
    OK(ics_op_component_bounds(lab, dims3, 1, nl, st7, ML, 3, 0.8, bnd));
    int exact = 0;
    for (int c = 0; c < n; ++c) {
      if (st7[c * 7] > 3 && !(bnd[c * 5 + 1] >= bnd[c * 5 + 2] && bnd[c * 5 + 2] >= bnd[c * 5])) { fprintf(stderr, "bounds out of order\n"); return 1; }
      if (bnd[c * 5 + 4]) { ++exact; if (bnd[c * 5 + 4] > bnd[c * 5 + 1] || bnd[c * 5 + 4] < bnd[c * 5 + 2]) { fprintf(stderr, "hull count outside its bounds\n"); return 1; } }
    }
    if (exact == 0) { fprintf(stderr, "the exact hull path was not exercised\n"); return 1; }
    /* every component as its own box {0, 1}, plus the whole volume: dozens of boxes through the thread pool */
    size_t tot = V; int nb = 1;
    for (int c = 0; c < n; ++c) tot += (size_t)(st7[c * 7 + 4] - st7[c * 7 + 1]) * (st7[c * 7 + 5] - st7[c * 7 + 2]) * (st7[c * 7 + 6] - st7[c * 7 + 3]);
    int32_t* boxes = (int32_t*)malloc(tot * 4);
    int32_t* out = (int32_t*)malloc(tot * 4);
    int32_t* bd = (int32_t*)malloc((size_t)(n + 1) * 3 * 4);
    int32_t* bc = (int32_t*)malloc((size_t)(n + 1) * 4);
    memcpy(boxes, vol, V * 4); bd[0] = D; bd[1] = H; bd[2] = W; bc[0] = 1;
    size_t off = V;
    for (int c = 0; c < n; ++c) {
      const int32_t* r = st7 + c * 7;
      const int bz = r[4] - r[1], by = r[5] - r[2], bx = r[6] - r[3];
      for (int z = 0; z < bz; ++z) for (int y = 0; y < by; ++y) for (int x = 0; x < bx; ++x)
        boxes[off + (z * by + y) * bx + x] = lab[((r[1] + z) * H + r[2] + y) * W + r[3] + x] == c + 1;
      bd[nb * 3] = bz; bd[nb * 3 + 1] = by; bd[nb * 3 + 2] = bx; bc[nb] = 1; ++nb;
      off += (size_t)bz * by * bx;
    }
    OK(ics_op_watershed_split(boxes, bd, bc, nb, 0, out));
    OK(ics_op_watershed_split(boxes, bd, bc, nb, 1, out));
    printf("asan driver: %d components, %d exact hull counts, %d boxes split on host threads\n", n, exact, nb);
    free(vol); free(lab); free(stack); free(st7); free(bnd); free(boxes); free(out); free(bd); free(bc);
  }
  printf("asan driver: all entry points walked, no sanitizer report\n");
  return 0;
}
