// Probe: what does an instruction issued between two v_mfma_f32_32x32x2_f32 cost, with one or two waves per SIMD?
// hipcc --offload-arch=gfx950 -O3 -o mfma_filler mfma_filler.hip && ./mfma_filler
// Each wave runs ITER x 16 MFMAs on 16 independent accumulators with NF fillers of kind K after every MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f16v __attribute__((ext_vector_type(16)));

template <int KIND, int NF, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(float* out, const float* in, int iters) {
  __shared__ float lds[8192];
  const int tid = threadIdx.x;
  for (int i = tid; i < 8192; i += WAVES * 64) lds[i] = in[i & 1023];
  __syncthreads();
  constexpr int NA = WAVES == 4 ? 16 : 8;      // accumulators per wave (256 / 128 AGPRs)
  f16v acc[NA];
#pragma unroll
  for (int f = 0; f < NA; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;
  float a = in[tid], b = in[tid + 64];
  float v0 = a, v1 = b, v2 = a + b, v3 = a - b;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {a, b}, p1 = {b, a}, p2 = {a + b, a - b}, p3 = {a - b, a + b};
  float l0 = 0, l1 = 0;
  const float* gp = in + tid;
  float g0 = 0;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int f = 0; f < NA; ++f) {
      acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[f], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NF; ++j) {
        if (KIND == 1) {          // independent VALU fma chain elements
          if (j & 1) v0 = v0 * v1 + v2; else v3 = v3 * v1 + v2;
        } else if (KIND == 4) {   // packed fp32 fma
          if (j & 1) p0 = p0 * p1 + p2; else p3 = p3 * p1 + p2;
        } else if (KIND == 2) {   // LDS read b64
          float2 t = *reinterpret_cast<const float2*>(&lds[((tid * 2 + (f * NF + j) * 128 + it) & 8190)]);
          l0 += t.x; l1 += t.y;
        } else if (KIND == 3) {   // global load (L1/L2 hit)
          g0 += gp[((f * NF + j) * 64 + it) & 1023];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = v0 + v3 + l0 + l1 + g0 + p0.x + p0.y + p3.x + p3.y;
#pragma unroll
  for (int f = 0; f < NA; ++f) s += acc[f][0] + acc[f][7];
  out[blockIdx.x * WAVES * 64 + tid] = s;
  if (tid == 0 && blockIdx.x == 0) out[1 << 20] = (float)(t1 - t0);
}

template <int KIND, int NF, int WAVES>
static void run(const char* name, float* dout, float* din) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<KIND, NF, WAVES>), dim3(256), dim3(WAVES * 64), 0, 0, dout, din, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<KIND, NF, WAVES>), dim3(256), dim3(WAVES * 64), 0, 0, dout, din, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  constexpr int NA = WAVES == 4 ? 16 : 8;
  // per SIMD: WAVES/4 waves x iters x NA MFMAs
  double mf = (double)iters * NA * (WAVES / 4);
  double cyc = ms * 1e-3 * 2.4e9 / mf;
  printf("%-10s NF=%d waves/SIMD=%d: %.3f ms  -> %.1f cycles per MFMA slot (at 2.4 GHz; 64 = peak)\n", name, NF, WAVES / 4, ms, cyc);
}

int main() {
  float *dout, *din;
  hipMalloc(&dout, ((1 << 20) + 16) * 4); hipMalloc(&din, 4096 * 4);
  float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = (i % 17) * 0.01f - 0.05f;
  hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
  run<0, 0, 4>("none", dout, din);
  run<1, 1, 4>("valu", dout, din);  run<1, 2, 4>("valu", dout, din);  run<1, 4, 4>("valu", dout, din);  run<1, 8, 4>("valu", dout, din);
  run<4, 1, 4>("pkfma", dout, din);  run<4, 2, 4>("pkfma", dout, din);  run<4, 4, 4>("pkfma", dout, din);  run<4, 8, 4>("pkfma", dout, din);
  run<0, 0, 8>("none", dout, din);
  run<1, 1, 8>("valu", dout, din);  run<1, 2, 8>("valu", dout, din);  run<1, 4, 8>("valu", dout, din);  run<1, 8, 8>("valu", dout, din);
  run<4, 1, 8>("pkfma", dout, din);  run<4, 2, 8>("pkfma", dout, din);  run<4, 4, 8>("pkfma", dout, din);  run<4, 8, 8>("pkfma", dout, din);
  return 0;
}
