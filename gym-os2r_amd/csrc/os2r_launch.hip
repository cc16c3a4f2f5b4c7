// os2r_launch.hip — dtype-level dispatch over the instantiation units + tiny utility kernels.
#include "os2r_kernels.hpp"

namespace os2r {

template <typename R, int UNIT>
int step_unit(bool contact, bool dr, const StepArgs<R>& a, hipStream_t s);
template <typename R, int N_>
int reset_unit(bool dr, const StepArgs<R>& a, hipStream_t s);

#define OS2R_DECL_STEP(R, U) template <> int step_unit<R, U>(bool, bool, const StepArgs<R>&, hipStream_t);
#define OS2R_DECL_RESET(R, N_) template <> int reset_unit<R, N_>(bool, const StepArgs<R>&, hipStream_t);
#define OS2R_DECL(R)                                                                              \
  OS2R_DECL_STEP(R, 0) OS2R_DECL_STEP(R, 1) OS2R_DECL_STEP(R, 2) OS2R_DECL_STEP(R, 3)               \
  OS2R_DECL_STEP(R, 12) OS2R_DECL_STEP(R, 13) OS2R_DECL_STEP(R, 14) OS2R_DECL_STEP(R, 15)           \
  OS2R_DECL_RESET(R, 2) OS2R_DECL_RESET(R, 3) OS2R_DECL_RESET(R, 4) OS2R_DECL_RESET(R, 5)
OS2R_DECL(float)
OS2R_DECL(double)

template <typename T>
int Launcher<T>::step(int nq, int model_id, bool contact, bool dr, const StepArgs<T>& a, hipStream_t s) {
  switch (model_id) {
    case 0: return step_unit<T, 0>(contact, dr, a, s);
    case 1: return step_unit<T, 1>(contact, dr, a, s);
    case 2: return step_unit<T, 2>(contact, dr, a, s);
    case 3: return step_unit<T, 3>(contact, dr, a, s);
    default: break;
  }
  switch (nq) {
    case 2: return step_unit<T, 12>(contact, dr, a, s);
    case 3: return step_unit<T, 13>(contact, dr, a, s);
    case 4: return step_unit<T, 14>(contact, dr, a, s);
    case 5: return step_unit<T, 15>(contact, dr, a, s);
    default: return 1;
  }
}
template <typename T>
int Launcher<T>::reset(int nq, bool dr, const StepArgs<T>& a, hipStream_t s) {
  switch (nq) {
    case 2: return reset_unit<T, 2>(dr, a, s);
    case 3: return reset_unit<T, 3>(dr, a, s);
    case 4: return reset_unit<T, 4>(dr, a, s);
    case 5: return reset_unit<T, 5>(dr, a, s);
    default: return 1;
  }
}
template <typename T>
void Launcher<T>::gravity(T* g, long long N, long long off, unsigned long long seed, double mean, double std_,
                          hipStream_t s) {
  hipLaunchKernelGGL((gravity_kernel<T>), dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, g, N, off, seed, mean, std_);
}
template <typename T>
void Launcher<T>::fill(T* dst, long long n, T value, hipStream_t s) {
  hipLaunchKernelGGL((fill_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dst, n, value);
}

template struct Launcher<float>;
template struct Launcher<double>;

// Which constexpr table (if any) equals this model bit for bit (gravity excluded: it is a
// per-handle config value passed as a kernel argument).
template <int ID>
static bool matches(const Os2rModel& m) {
  using Tb = gen::Tables<ID>;
  if (m.nq != Tb::nq || m.ncand != Tb::ncand) return false;
  for (int i = 0; i < Tb::nq; ++i) {
    if (m.axis[i] != Tb::axis[i] || m.mass[i] != Tb::mass[i] || m.damping[i] != Tb::damping[i] ||
        m.friction[i] != Tb::friction[i] || m.mu[i] != Tb::mu[i]) return false;
    for (int k = 0; k < 9; ++k) if (m.rfix[i][k] != Tb::rfix[i][k]) return false;
    for (int k = 0; k < 3; ++k) if (m.rpos[i][k] != Tb::rpos[i][k] || m.com[i][k] != Tb::com[i][k]) return false;
    for (int k = 0; k < 6; ++k) if (m.icom[i][k] != Tb::icom[i][k]) return false;
  }
  for (int k = 0; k < 2; ++k) if (m.act_dof[k] != Tb::act_dof[k] || m.max_torque[k] != Tb::max_torque[k]) return false;
  int b = 0;
  for (int c = 0; c < Tb::ncand; ++c) {
    while (c >= Tb::cand_begin[b + 1]) ++b;
    if (m.cand_body[c] != b) return false;
    for (int j = 0; j < 3; ++j) if (m.cand_p[c][j] != Tb::cand_p[c][j]) return false;
  }
  for (int i = 0; i < Tb::nq; ++i) {
    if (m.cand_radius[i] != Tb::cand_radius[i]) return false;
    for (int j = 0; j < 3; ++j) if (m.cand_center[i][j] != Tb::cand_center[i][j]) return false;
  }
  return true;
}

// bit-for-bit equality of two robots (gravity excluded, padding ignored)
bool same_model(const Os2rModel& a, const Os2rModel& b) {
  if (a.nq != b.nq || a.ncand != b.ncand) return false;
  for (int i = 0; i < a.nq; ++i) {
    if (a.axis[i] != b.axis[i] || a.mass[i] != b.mass[i] || a.damping[i] != b.damping[i] ||
        a.friction[i] != b.friction[i] || a.mu[i] != b.mu[i] || a.cand_radius[i] != b.cand_radius[i]) return false;
    for (int k = 0; k < 9; ++k) if (a.rfix[i][k] != b.rfix[i][k]) return false;
    for (int k = 0; k < 3; ++k)
      if (a.rpos[i][k] != b.rpos[i][k] || a.com[i][k] != b.com[i][k] || a.cand_center[i][k] != b.cand_center[i][k]) return false;
    for (int k = 0; k < 6; ++k) if (a.icom[i][k] != b.icom[i][k]) return false;
  }
  for (int k = 0; k < 2; ++k) if (a.act_dof[k] != b.act_dof[k] || a.max_torque[k] != b.max_torque[k]) return false;
  for (int c = 0; c < a.ncand; ++c) {
    if (a.cand_body[c] != b.cand_body[c]) return false;
    for (int j = 0; j < 3; ++j) if (a.cand_p[c][j] != b.cand_p[c][j]) return false;
  }
  return true;
}

int static_model_id(const Os2rModel& m) {
  if (matches<0>(m)) return 0;
  if (matches<1>(m)) return 1;
  if (matches<2>(m)) return 2;
  if (matches<3>(m)) return 3;
  return -1;
}

}  // namespace os2r
