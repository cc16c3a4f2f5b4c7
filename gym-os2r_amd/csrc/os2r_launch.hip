// os2r_launch.hip — dtype-level dispatch over the per-NQ instantiation units + tiny utility kernels.
#include "os2r_kernels.hpp"

namespace os2r {

template <typename R, int N_>
int step_unit(unsigned cmask, bool dr, bool std_axes, const StepArgs<R>& a, hipStream_t s);
template <typename R, int N_>
int reset_unit(bool dr, const StepArgs<R>& a, hipStream_t s);

#define OS2R_DECL(R)                                                                        \
  template <> int step_unit<R, 2>(unsigned, bool, bool, const StepArgs<R>&, hipStream_t);          \
  template <> int step_unit<R, 3>(unsigned, bool, bool, const StepArgs<R>&, hipStream_t);          \
  template <> int step_unit<R, 4>(unsigned, bool, bool, const StepArgs<R>&, hipStream_t);          \
  template <> int step_unit<R, 5>(unsigned, bool, bool, const StepArgs<R>&, hipStream_t);          \
  template <> int reset_unit<R, 2>(bool, const StepArgs<R>&, hipStream_t);                   \
  template <> int reset_unit<R, 3>(bool, const StepArgs<R>&, hipStream_t);                   \
  template <> int reset_unit<R, 4>(bool, const StepArgs<R>&, hipStream_t);                   \
  template <> int reset_unit<R, 5>(bool, const StepArgs<R>&, hipStream_t);
OS2R_DECL(float)
OS2R_DECL(double)

template <typename T>
int Launcher<T>::step(int nq, unsigned cmask, bool dr, int ax0, const StepArgs<T>& a, hipStream_t s) {
  switch (nq) {
    case 2: return step_unit<T, 2>(cmask, dr, ax0 == (2 >= 4 ? 2 : 0), a, s);
    case 3: return step_unit<T, 3>(cmask, dr, ax0 == (3 >= 4 ? 2 : 0), a, s);
    case 4: return step_unit<T, 4>(cmask, dr, ax0 == (4 >= 4 ? 2 : 0), a, s);
    case 5: return step_unit<T, 5>(cmask, dr, ax0 == (5 >= 4 ? 2 : 0), a, s);
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

}  // namespace os2r
