// Micro-benchmark (diagnostic, not shipped): what the non-VALU instructions around an fp64 FMA
// stream cost one wave that owns its SIMD (gfx950).  Ticks are s_memtime ticks per loop body.
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

// 16 independent-ish FMAs (4 chains)
__global__ void k_valu(double* out, unsigned long long* ticks, int iters, double a, double b) {
  BODY_BEGIN
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5"
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));
  }
  BODY_END
}
// the same with two s_mov_b32 (a 64-bit literal) before every FMA
__global__ void k_salu(double* out, unsigned long long* ticks, int iters, double a, double b) {
  BODY_BEGIN
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    asm volatile("s_mov_b32 s20, 0x54442d18\n s_mov_b32 s21, 0x3fef9db2\n v_fma_f64 %0, %0, s[20:21], %5\n"
                 "s_mov_b32 s22, 0x54442d18\n s_mov_b32 s23, 0x3fef9db2\n v_fma_f64 %1, %1, s[22:23], %5\n"
                 "s_mov_b32 s20, 0x54442d19\n s_mov_b32 s21, 0x3fef9db2\n v_fma_f64 %2, %2, s[20:21], %5\n"
                 "s_mov_b32 s22, 0x54442d1a\n s_mov_b32 s23, 0x3fef9db2\n v_fma_f64 %3, %3, s[22:23], %5"
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b) : "s20", "s21", "s22", "s23");
  }
  BODY_END
}
// the same with two v_readlane_b32 feeding the SGPR pair before every FMA
__global__ void k_lane(double* out, unsigned long long* ticks, int iters, double a, double b) {
  unsigned lo = __double2loint(a), hi = __double2hiint(a);
  BODY_BEGIN
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    asm volatile("v_readlane_b32 s20, %6, 3\n v_readlane_b32 s21, %7, 3\n s_nop 3\n v_fma_f64 %0, %0, s[20:21], %5\n"
                 "v_readlane_b32 s22, %6, 4\n v_readlane_b32 s23, %7, 4\n s_nop 3\n v_fma_f64 %1, %1, s[22:23], %5\n"
                 "v_readlane_b32 s20, %6, 5\n v_readlane_b32 s21, %7, 5\n s_nop 3\n v_fma_f64 %2, %2, s[20:21], %5\n"
                 "v_readlane_b32 s22, %6, 6\n v_readlane_b32 s23, %7, 6\n s_nop 3\n v_fma_f64 %3, %3, s[22:23], %5"
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b), "v"(lo), "v"(hi) : "s20", "s21", "s22", "s23");
  }
  BODY_END
}
// v_accvgpr round trip per FMA (AGPR spill traffic): write + read of both halves
__global__ void k_acc(double* out, unsigned long long* ticks, int iters, double a, double b) {
  unsigned u = threadIdx.x, w = threadIdx.x + 1;
  BODY_BEGIN
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    asm volatile("v_accvgpr_write_b32 a0, %6\n v_accvgpr_write_b32 a1, %7\n v_accvgpr_read_b32 %6, a0\n v_accvgpr_read_b32 %7, a1\n v_fma_f64 %0, %0, %4, %5\n"
                 "v_accvgpr_write_b32 a2, %6\n v_accvgpr_write_b32 a3, %7\n v_accvgpr_read_b32 %6, a2\n v_accvgpr_read_b32 %7, a3\n v_fma_f64 %1, %1, %4, %5\n"
                 "v_accvgpr_write_b32 a0, %6\n v_accvgpr_write_b32 a1, %7\n v_accvgpr_read_b32 %6, a0\n v_accvgpr_read_b32 %7, a1\n v_fma_f64 %2, %2, %4, %5\n"
                 "v_accvgpr_write_b32 a2, %6\n v_accvgpr_write_b32 a3, %7\n v_accvgpr_read_b32 %6, a2\n v_accvgpr_read_b32 %7, a3\n v_fma_f64 %3, %3, %4, %5"
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b), "v"(u), "v"(w) : "a0", "a1", "a2", "a3");
  }
  BODY_END
}
// LDS round trip: write, then read back and wait, once per 4 FMAs
__global__ void k_lds(double* out, unsigned long long* ticks, int iters, double a, double b) {
  __shared__ double sh[64 * 4];
  double* p = sh + threadIdx.x;
  BODY_BEGIN
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    p[0] = x0;
    asm volatile("" ::: "memory");
    x0 = p[0];
    asm volatile("s_waitcnt lgkmcnt(0)\n v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5"
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));
  }
  BODY_END
}
// dependent chain with mixed ops (fma -> max -> min -> add), as in a solver row
__global__ void k_dep(double* out, unsigned long long* ticks, int iters, double a, double b) {
  BODY_BEGIN
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    asm volatile("v_fma_f64 %0, %0, %4, %5\n v_max_f64 %0, %0, %1\n v_min_f64 %0, %0, %2\n v_add_f64 %0, %0, %3"
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));
  }
  BODY_END
}

template <typename K>
void run(const char* name, K kern, double per) {
  const int iters = 4000, blocks = 1024;
  double* out; unsigned long long* ticks;
  hipMalloc(&out, blocks * 64 * sizeof(double));
  hipMalloc(&ticks, blocks * sizeof(unsigned long long));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kern<<<blocks, 64>>>(out, ticks, iters, 0.999, 1e-3);
  hipEventRecord(e0);
  kern<<<blocks, 64>>>(out, ticks, iters, 0.999, 1e-3);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), ticks, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : h) avg += v; avg /= blocks;
  printf("%-34s %7.2f ticks per group of 4 FMAs  (%.2f ns wall)\n", name, avg / (iters * 4.0), ms * 1e6 / (iters * 4.0));
  hipFree(out); hipFree(ticks);
}

int main() {
  run("4 FMA", k_valu, 4);
  run("4 x (2 s_mov + FMA)", k_salu, 4);
  run("4 x (2 v_readlane + s_nop + FMA)", k_lane, 4);
  run("4 x (2 acc write + 2 acc read + FMA)", k_acc, 4);
  run("LDS write/read/wait + 4 FMA", k_lds, 4);
  run("fma->max->min->add dependent", k_dep, 4);
  return 0;
}
