// Micro-benchmark (diagnostic, not shipped): what handing an fp64 value to a neighbouring lane costs a wave that
// owns its SIMD (gfx950) -- the price list behind "several lanes per environment" (small batches).
// A double is two 32-bit registers; DPP, v_permlane and ds_swizzle / ds_bpermute move 32 bits per instruction.
//   hipcc --offload-arch=gfx950 -O3 tools/dbg/ubench_shuffle.hip -o /tmp/ubench_shuffle && /tmp/ubench_shuffle
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define BODY_BEGIN                                                             \
  double x0 = threadIdx.x * 1e-3, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;       \
  unsigned long long t0 = __builtin_readcyclecounter();                        \
  for (int it = 0; it < iters; ++it) {
#define BODY_END                                                               \
  }                                                                            \
  unsigned long long t1 = __builtin_readcyclecounter();                        \
  out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3;                      \
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;

// baseline: 16 fp64 FMAs
__global__ void k_fma(double* out, unsigned long long* ticks, int iters, double a, double b) {
  BODY_BEGIN
#pragma unroll
  for (int r = 0; r < 4; ++r)
    asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5"
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));
  BODY_END
}
// quad_perm:[1,2,3,0] = 0x39: every lane takes the value of the next lane of its group of four
__device__ __forceinline__ double dpp_next(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x39, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x39, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// the same permutation through the LDS crossbar: ds_swizzle_b32 in quad-permute mode (0x8000 | perm)
__device__ __forceinline__ double swz_next(double v) {
  const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), 0x8039);
  const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), 0x8039);
  return __hiloint2double(hi, lo);
}
// 16 FMAs, each fed by a neighbour's double through DPP (two v_mov_b32_dpp per double)
__global__ void k_dpp(double* out, unsigned long long* ticks, int iters, double a, double b) {
  BODY_BEGIN
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    x0 = __builtin_fma(dpp_next(x0), a, b); x1 = __builtin_fma(dpp_next(x1), a, b);
    x2 = __builtin_fma(dpp_next(x2), a, b); x3 = __builtin_fma(dpp_next(x3), a, b);
    asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
  }
  BODY_END
}
// the same through ds_swizzle_b32 (two per double), four requests in flight before the wait
__global__ void k_swz(double* out, unsigned long long* ticks, int iters, double a, double b) {
  BODY_BEGIN
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const double n0 = swz_next(x0), n1 = swz_next(x1), n2 = swz_next(x2), n3 = swz_next(x3);
    x0 = __builtin_fma(n0, a, b); x1 = __builtin_fma(n1, a, b); x2 = __builtin_fma(n2, a, b); x3 = __builtin_fma(n3, a, b);
    asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
  }
  BODY_END
}

template <typename K>
static double run(K k, const char* name, int per_iter, int lanes = 64) {
  const int blocks = 256 * 4, iters = 2000;
  double* out; unsigned long long* ticks;
  hipMalloc(&out, blocks * 64 * sizeof(double)); hipMalloc(&ticks, blocks * sizeof(unsigned long long));
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k, dim3(blocks), dim3(lanes), 0, 0, out, ticks, iters, 0.999, 1e-3);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), ticks, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double s = 0; for (auto v : h) s += v;
  const double per = s / blocks / iters / per_iter;
  printf("%-34s %7.2f ticks per fp64 FMA (incl. what feeds it)\n", name, per);
  hipFree(out); hipFree(ticks);
  return per;
}

int main() {
  const double f = run(k_fma, "16 v_fma_f64", 16);
  const double d = run(k_dpp, "16 x (2 v_mov_b32_dpp + v_fma_f64)", 16);
  const double s = run(k_swz, "16 x (2 ds_swizzle_b32 + v_fma_f64)", 16);
  // does a wave with half or a quarter of its lanes active issue fp64 faster?  (it does not on GCN-lineage hardware)
  run(k_fma, "16 v_fma_f64, 32 of 64 lanes active", 16, 32);
  run(k_fma, "16 v_fma_f64, 16 of 64 lanes active", 16, 16);
  printf("handing one double to a neighbouring lane costs %.1f ticks by DPP, %.1f by ds_swizzle; an fp64 FMA %.1f\n", d - f, s - f, f);
  return 0;
}
