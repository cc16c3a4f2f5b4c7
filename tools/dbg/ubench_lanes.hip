// Diagnostic (round 5): does a wave whose upper lanes are switched off for the whole kernel issue faster?  The same fp64 code -- the
// row pass of the exact solve, 36 instructions a row with fifteen independent accumulators, and a dependent chain (a Horner
// polynomial) -- with 64, 32, 16 and 1 lanes alive (the others `return` at the top of the kernel, so the exec mask never holds them),
// one wave per SIMD (1024 workgroups of 64); s_memtime cycles per repetition, median over the waves.
//   hipcc --offload-arch=gfx950 -O3 tools/dbg/ubench_lanes.hip -o /tmp/ubench_lanes && /tmp/ubench_lanes
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
constexpr int NQ = 5, NT = 15, R = 8;
__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
// a time stamp the compiler cannot move code across (asm volatile with a memory clobber, scheduling barriers on both sides)
__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t_;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t_;
}
__global__ __launch_bounds__(64) void k(const double* __restrict__ in, double* __restrict__ out, unsigned long long* __restrict__ ticks, int reps, int alive) {
  const int lane = threadIdx.x;
  if (lane >= alive) return;
  double g[R][NQ], c[R], t[R], y[NQ];
  const double* p = in + (size_t)(blockIdx.x * 64 + lane) * (R * 7 + NQ);
#pragma unroll
  for (int r = 0; r < R; ++r) {
#pragma unroll
    for (int i = 0; i < NQ; ++i) g[r][i] = p[r * 7 + i];
    c[r] = p[r * 7 + 5]; t[r] = p[r * 7 + 6];
  }
#pragma unroll
  for (int i = 0; i < NQ; ++i) y[i] = p[R * 7 + i];
  unsigned long long t_ilp = 0, t_chain = 0;
  double sink = 0.0;
  for (int rep = 0; rep < reps; ++rep) {
    double S[NT], h[NQ];
    asm volatile("" : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]));
#pragma unroll
    for (int r = 0; r < R; ++r) {   // (opaque rows: nothing of the block is loop-invariant)
#pragma unroll
      for (int i = 0; i < NQ; ++i) asm volatile("" : "+v"(g[r][i]));
      asm volatile("" : "+v"(c[r]), "+v"(t[r]));
    }
    unsigned long long a0 = stamp();
#pragma unroll
    for (int k_ = 0; k_ < NT; ++k_) S[k_] = 0.0;
#pragma unroll
    for (int i = 0; i < NQ; ++i) h[i] = 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      double w = -t[r];
#pragma unroll
      for (int i = 0; i < NQ; ++i) w = fma_(g[r][i], y[i], w);
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        const double gs = c[r] * g[r][i];
        h[i] = fma_(-gs, w, h[i]);
#pragma unroll
        for (int j = 0; j <= i; ++j) S[i * (i + 1) / 2 + j] = fma_(gs, g[r][j], S[i * (i + 1) / 2 + j]);
      }
    }
#pragma unroll
    for (int k_ = 0; k_ < NT; ++k_) sink += S[k_];
#pragma unroll
    for (int i = 0; i < NQ; ++i) sink += h[i];
    asm volatile("" : "+v"(sink));
    unsigned long long a1 = stamp();
    t_ilp += a1 - a0;
    y[0] += 1e-9 * sink;
    // a dependent chain of 64 FMAs
    double x = y[1];
    asm volatile("" : "+v"(x));
    unsigned long long b0 = stamp();
#pragma unroll
    for (int q = 0; q < 64; ++q) x = fma_(x, y[2], c[q & 7]);
    asm volatile("" : "+v"(x));
    unsigned long long b1 = stamp();
    t_chain += b1 - b0;
    sink += x;
    y[1] += 1e-9 * sink;
  }
  if (lane == 0) { out[blockIdx.x] = sink; ticks[2 * blockIdx.x] = t_ilp; ticks[2 * blockIdx.x + 1] = t_chain; }
}
int main() {
  const int blocks = 1024, reps = 200;
  const size_t per = R * 7 + NQ, n = (size_t)blocks * 64 * per;
  std::vector<double> h(n);
  unsigned long long s = 88172645463325252ull;
  for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (double)(s >> 11) / 9007199254740992.0 - 0.5; }
  double *din, *dout; unsigned long long* dt;
  (void)hipMalloc(&din, n * 8); (void)hipMalloc(&dout, blocks * 8); (void)hipMalloc(&dt, blocks * 16);
  (void)hipMemcpy(din, h.data(), n * 8, hipMemcpyHostToDevice);
  printf("fp64 on a wave that owns its SIMD: %d-instruction block with 15 independent accumulators / chain of 64 dependent FMAs; cycles, median over 1024 waves\n", R * 36 + 20);
  for (int alive : {64, 48, 32, 16, 1}) {
    k<<<blocks, 64>>>(din, dout, dt, reps, alive);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> t(2 * blocks);
    (void)hipMemcpy(t.data(), dt, blocks * 16, hipMemcpyDeviceToHost);
    std::vector<double> a(blocks), b(blocks);
    for (int i = 0; i < blocks; ++i) { a[i] = (double)t[2 * i] / reps; b[i] = (double)t[2 * i + 1] / reps; }
    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
    printf("%2d lanes alive: block %6.0f cycles (%.2f per instruction); chain %5.0f cycles (%.2f per FMA)\n", alive, a[blocks / 2], a[blocks / 2] / (R * 36 + 20), b[blocks / 2], b[blocks / 2] / 64);
  }
  return 0;
}
