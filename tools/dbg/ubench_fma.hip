// Micro-benchmark (diagnostic, not shipped): issue cost of dependent vs independent fp64 FMAs for
// one wave per SIMD on gfx950.  K independent chains of v_fma_f64 are interleaved; cycles per
// instruction come from s_memtime and from the wall clock.
//   hipcc --offload-arch=gfx950 -O3 tools/dbg/ubench_fma.hip -o /tmp/ubench_fma && /tmp/ubench_fma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int K, typename T>
__global__ void chains(T* out, unsigned long long* ticks, int iters, T a, T b) {
  T x[K];
#pragma unroll
  for (int k = 0; k < K; ++k) x[k] = T(threadIdx.x + k) * T(1e-3);
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int k = 0; k < K; ++k) x[k] = __builtin_fma(x[k], a, b);
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  T s = 0;
#pragma unroll
  for (int k = 0; k < K; ++k) s += x[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int K, typename T>
void run(const char* name, int waves_per_simd) {
  const int iters = 2000;
  const int blocks = 256 * 4 * waves_per_simd;   // 64-thread blocks
  T* out; unsigned long long* ticks;
  hipMalloc(&out, blocks * 64 * sizeof(T));
  hipMalloc(&ticks, blocks * sizeof(unsigned long long));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  chains<K, T><<<blocks, 64>>>(out, ticks, iters, T(0.999), T(1e-3));
  hipEventRecord(e0);
  chains<K, T><<<blocks, 64>>>(out, ticks, iters, T(0.999), T(1e-3));
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), ticks, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : h) avg += v; avg /= blocks;
  const double ninstr = double(iters) * 16 * K;
  printf("%-6s K=%d waves/SIMD=%d: %.2f ticks/instr (s_memtime), %.2f ns/instr/wave-slot (wall %.3f ms)\n", name, K,
         waves_per_simd, avg / ninstr, ms * 1e6 / ninstr / waves_per_simd, ms);
  hipFree(out); hipFree(ticks);
}

int main() {
  run<1, double>("f64", 1); run<2, double>("f64", 1); run<4, double>("f64", 1); run<8, double>("f64", 1);
  run<1, double>("f64", 2); run<2, double>("f64", 2); run<4, double>("f64", 2);
  run<1, float>("f32", 1); run<2, float>("f32", 1); run<4, float>("f32", 1); run<8, float>("f32", 1);
  return 0;
}
