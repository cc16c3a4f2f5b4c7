// Diagnostic: relative error of v_rcp_f64 / v_rsq_f64 and of 1 or 2 Newton steps on top (gfx950).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double* x, double* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = x[i];
  double r0 = __builtin_amdgcn_rcp(v);
  double r1 = __builtin_fma(__builtin_fma(-v, r0, 1.0), r0, r0);
  double r2 = __builtin_fma(__builtin_fma(-v, r1, 1.0), r1, r1);
  double s0 = __builtin_amdgcn_rsq(v);
  double s1 = s0 * __builtin_fma(-0.5 * v * s0, s0, 1.5);
  double s2 = s1 * __builtin_fma(-0.5 * v * s1, s1, 1.5);
  out[6 * i + 0] = r0; out[6 * i + 1] = r1; out[6 * i + 2] = r2; out[6 * i + 3] = s0; out[6 * i + 4] = s1; out[6 * i + 5] = s2;
}
int main() {
  const int n = 1 << 20;
  std::vector<double> h(n), o(6 * n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = std::ldexp(1.0 + (s >> 11) * (1.0 / 9007199254740992.0), int(s % 41) - 20); }
  double *dx, *dout;
  (void)hipMalloc(&dx, n * 8); (void)hipMalloc(&dout, 6 * n * 8);
  (void)hipMemcpy(dx, h.data(), n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, dout, n);
  (void)hipMemcpy(o.data(), dout, 6 * n * 8, hipMemcpyDeviceToHost);
  double e[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const long double ir = 1.0L / h[i], is = 1.0L / sqrtl((long double)h[i]);
    for (int j = 0; j < 3; ++j) { e[j] = fmax(e[j], (double)fabsl((o[6 * i + j] - ir) / ir)); e[3 + j] = fmax(e[3 + j], (double)fabsl((o[6 * i + 3 + j] - is) / is)); }
  }
  printf("rcp: raw %.3e, 1 Newton step %.3e, 2 steps %.3e (ulp = 1.1e-16)\nrsq: raw %.3e, 1 step %.3e, 2 steps %.3e\n", e[0], e[1], e[2], e[3], e[4], e[5]);
  return 0;
}
