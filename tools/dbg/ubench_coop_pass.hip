// Diagnostic (round 5): what would spreading a lone lane's exact solve over the idle lanes of its wave buy?
// Only the two row passes of the regularised solve parallelise over rows (the 5 x 5 factorisation and the proximal solves are a
// serial chain, step length and apply are short); this times pass 1 -- S = sum c_r g_r g_r^T (15 numbers), h = -sum c_r g_r w_r (5),
// w_r = g_r . y - t_r -- for R free rows of ONE live lane of a wave that owns its SIMD, two ways:
//   serial       the kernel's form: the live lane walks its rows (registers), 36 instructions a row, every other lane masked off;
//   cooperative  the live lane writes its rows to LDS (7 doubles a row: five coefficients, weight, target), the first R lanes take
//                a row each and compute w_r, twenty lanes accumulate one number of S / h each over the rows IN ROW ORDER (the same
//                FMA chain as the serial form: a lane's result must not depend on which path its wave took), the results go back
//                through LDS to the live lane.
// Cycles by s_memtime around each form, one wave per SIMD (1024 workgroups of 64 on an MI355X), medians over the waves.
//   hipcc --offload-arch=gfx950 -O3 tools/dbg/ubench_coop_pass.hip -o /tmp/ubench_coop && /tmp/ubench_coop
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

constexpr int NQ = 5, NT = 15, RMAX = 12;
__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ int tri(int i, int j) { return i * (i + 1) / 2 + j; }

template <int R>
__global__ __launch_bounds__(64) void k(const double* __restrict__ in, double* __restrict__ out, unsigned long long* __restrict__ ticks, int reps) {
  __shared__ double lds[RMAX * 8 + 64];
  const int lane = threadIdx.x;
  // every lane holds "its" problem in registers as the step kernel does; only lane 0 is live
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
  double S[NT], h[NQ];
  unsigned long long t_serial = 0, t_coop = 0;
  double sink = 0.0;
  for (int rep = 0; rep < reps; ++rep) {
    // (the rows are opaque to the optimiser in every repetition: otherwise the products that do not depend on y are hoisted out of
    // the loop and the serial form is timed at a third of its instructions)
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
      for (int i = 0; i < NQ; ++i) asm volatile("" : "+v"(g[r][i]));
      asm volatile("" : "+v"(c[r]), "+v"(t[r]));
    }
    // ---- serial: the live lane alone ----
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long a0 = __builtin_amdgcn_s_memtime();
    if (lane == 0) {
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
          for (int j = 0; j <= i; ++j) S[tri(i, j)] = fma_(gs, g[r][j], S[tri(i, j)]);
        }
        asm volatile("" ::: "memory");   // (the kernel's rows sit behind scalar branches: no interleaving across rows)
      }
#pragma unroll
      for (int k_ = 0; k_ < NT; ++k_) sink += S[k_];
#pragma unroll
      for (int i = 0; i < NQ; ++i) sink += h[i];
    }
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long a1 = __builtin_amdgcn_s_memtime();
    t_serial += a1 - a0;
    y[0] += 1e-9 * sink;   // (a dependence from one repetition to the next)
    // ---- cooperative: rows through LDS, twenty lanes accumulate in row order, results back through LDS ----
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long b0 = __builtin_amdgcn_s_memtime();
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) lds[r * 8 + i] = g[r][i];
        lds[r * 8 + 5] = c[r]; lds[r * 8 + 6] = t[r];
      }
#pragma unroll
      for (int i = 0; i < NQ; ++i) lds[RMAX * 8 + 32 + i] = y[i];
    }
    __syncthreads();
    if (lane < R) {     // a row per lane: its residual, weighted
      double w = -lds[lane * 8 + 6];
#pragma unroll
      for (int i = 0; i < NQ; ++i) w = fma_(lds[lane * 8 + i], lds[RMAX * 8 + 32 + i], w);
      lds[lane * 8 + 7] = w;
    }
    __syncthreads();
    double acc = 0.0;
    if (lane < NT + NQ) {
      // lane k < 15: S entry (i, j); lane 15 + i: h[i]
      int i = 0, j = 0;
      if (lane < NT) { while ((i + 1) * (i + 2) / 2 <= lane) ++i; j = lane - i * (i + 1) / 2; } else { i = lane - NT; }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const double gs = lds[r * 8 + 5] * lds[r * 8 + i];
        const double other = lane < NT ? lds[r * 8 + j] : -lds[r * 8 + 7];
        acc = fma_(gs, other, acc);
      }
      lds[RMAX * 8 + lane] = acc;
    }
    __syncthreads();
    if (lane == 0) {
#pragma unroll
      for (int k_ = 0; k_ < NT; ++k_) S[k_] = lds[RMAX * 8 + k_];
#pragma unroll
      for (int i = 0; i < NQ; ++i) h[i] = lds[RMAX * 8 + NT + i];
#pragma unroll
      for (int k_ = 0; k_ < NT; ++k_) sink += S[k_];
#pragma unroll
      for (int i = 0; i < NQ; ++i) sink += h[i];
    }
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long b1 = __builtin_amdgcn_s_memtime();
    t_coop += b1 - b0;
    y[1] += 1e-9 * sink;
  }
  if (lane == 0) {
    out[blockIdx.x] = sink;
    ticks[2 * blockIdx.x] = t_serial;
    ticks[2 * blockIdx.x + 1] = t_coop;
  }
}

template <int R>
void run(int blocks, int reps) {
  const size_t per = R * 7 + NQ, n = (size_t)blocks * 64 * per;
  std::vector<double> h(n);
  unsigned long long s = 88172645463325252ull;
  for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (double)(s >> 11) / 9007199254740992.0 - 0.5; }
  double *din, *dout; unsigned long long* dt;
  (void)hipMalloc(&din, n * 8); (void)hipMalloc(&dout, blocks * 8); (void)hipMalloc(&dt, blocks * 16);
  (void)hipMemcpy(din, h.data(), n * 8, hipMemcpyHostToDevice);
  k<R><<<blocks, 64>>>(din, dout, dt, reps);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> t(2 * blocks);
  (void)hipMemcpy(t.data(), dt, blocks * 16, hipMemcpyDeviceToHost);
  std::vector<double> a(blocks), b(blocks);
  for (int i = 0; i < blocks; ++i) { a[i] = (double)t[2 * i] / reps; b[i] = (double)t[2 * i + 1] / reps; }
  std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
  printf("%2d free rows: pass 1 serial %7.0f cycles (%.0f a row), cooperative through LDS %7.0f cycles  -> %+6.0f cycles (%.2f x)\n",
         R, a[blocks / 2], a[blocks / 2] / R, b[blocks / 2], b[blocks / 2] - a[blocks / 2], b[blocks / 2] / a[blocks / 2]);
  (void)hipFree(din); (void)hipFree(dout); (void)hipFree(dt);
}

int main() {
  printf("one live lane of a wave that owns its SIMD (1024 workgroups of 64), s_memtime cycles per repetition, median over the waves\n");
  run<2>(1024, 200);
  run<4>(1024, 200);
  run<8>(1024, 200);
  run<12>(1024, 200);
  return 0;
}
