// Probe: do fp32 MFMA and fp32 VALU FMA issue concurrently on one CU (gfx950)?
// Build: hipcc --offload-arch=gfx950 -O3 -o pipe_probe pipe_probe.hip ; run on the GPU box.
// mode 0: every wave runs MFMA; 1: every wave runs VALU FMA; 2: even waves MFMA, odd waves VALU;
// 3: every wave interleaves both.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE, int PK>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
  const int wave = threadIdx.x >> 6;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = (float)threadIdx.x * 1e-3f, b = 1.0001f;
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = (float)i + a;
  f32x2 pv[8];
  for (int i = 0; i < 8; ++i) pv[i] = f32x2{(float)i, a};
  const bool do_mfma = MODE == 0 || MODE == 3 || (MODE == 2 && (wave & 1) == 0);
  const bool do_valu = MODE == 1 || MODE == 3 || (MODE == 2 && (wave & 1) == 1);
  for (int it = 0; it < iters; ++it) {
    if (do_mfma) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    if (do_valu) {
      if (PK) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
          for (int i = 0; i < 8; ++i) pv[i] = __builtin_elementwise_fma(pv[i], f32x2{b, b}, f32x2{a, a});
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = fmaf(v[i], b, a);
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 16; ++i) s += v[i];
  for (int i = 0; i < 8; ++i) s += pv[i].x + pv[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int PK>
static void run(const char* name, int threads, int blocks, int iters) {
  float* out;
  hipMalloc(&out, (size_t)blocks * threads * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<MODE, PK>), dim3(blocks), dim3(threads), 0, 0, out, iters / 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<MODE, PK>), dim3(blocks), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double waves = (double)blocks * threads / 64;
  double mf_w = (MODE == 0 || MODE == 3) ? waves : (MODE == 2 ? waves / 2 : 0);
  double va_w = (MODE == 1 || MODE == 3) ? waves : (MODE == 2 ? waves / 2 : 0);
  const double mfma_flop = mf_w * iters * 16.0 * (32 * 32 * 2 * 2);
  const double valu_flop = va_w * iters * (PK ? 64.0 * 2 : 128.0) * 64 * 2;
  printf("%-34s thr=%d blocks=%d  %.3f ms  MFMA %.1f TF/s  VALU %.1f TF/s\n", name, threads, blocks, ms,
         mfma_flop / ms / 1e9, valu_flop / ms / 1e9);
  hipFree(out);
}

int main() {
  const int it = 20000;
  for (int thr : {256, 512}) {
    const int blocks = 256 * (thr == 256 ? 2 : 1);   // 8 waves per CU either way
    run<0, 0>("mfma only", thr, blocks, it);
    run<1, 0>("valu fma only", thr, blocks, it);
    run<1, 1>("valu pk_fma only", thr, blocks, it);
    run<2, 0>("even waves mfma / odd waves fma", thr, blocks, it);
    run<2, 1>("even waves mfma / odd waves pk_fma", thr, blocks, it);
    run<3, 0>("every wave both (fma)", thr, blocks, it);
    run<3, 1>("every wave both (pk_fma)", thr, blocks, it);
  }
  return 0;
}
