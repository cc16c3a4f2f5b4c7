// capi_rollout.cpp -- the C-ABI of include/os2r.h driven from plain C++: no Python, no torch.
//
// Reads an Os2rConfig (the bytes of the struct, written by `python -m gym_os2r_amd.dump_config cfg.bin ...`),
// creates a simulator handle, runs K env-steps with on-device random actions (actions = NULL) writing
// observation / reward / done into buffers it allocated with hipMalloc, and prints the throughput plus a
// checksum of the last outputs (tests/test_gpu_runtime.py compares it with the Python binding).
//
//   hipcc -O2 -I include examples/capi_rollout.cpp -L gym-os2r_amd -los2r -Wl,-rpath,$PWD/gym-os2r_amd -o capi_rollout
//   ./capi_rollout cfg.bin 1000
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "os2r.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_OS2R(sim, x) do { int rc_ = (x); if (rc_ != OS2R_OK) { std::fprintf(stderr, "%s -> %d: %s\n", #x, rc_, os2r_last_error(sim)); return 3; } } while (0)

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: %s config.bin steps\n", argv[0]); return 1; }
  Os2rConfig cfg;
  FILE* f = std::fopen(argv[1], "rb");
  if (!f || std::fread(&cfg, sizeof(cfg), 1, f) != 1) { std::fprintf(stderr, "cannot read an Os2rConfig (%zu bytes) from %s\n", sizeof(cfg), argv[1]); return 1; }
  std::fclose(f);
  const int steps = std::atoi(argv[2]);
  if (cfg.abi_version != os2r_abi_version()) { std::fprintf(stderr, "config is for ABI %d, library is %d\n", cfg.abi_version, os2r_abi_version()); return 1; }

  Os2rSim* sim = nullptr;
  CHECK_OS2R(nullptr, os2r_create(&cfg, &sim));
  CHECK_HIP(hipSetDevice(cfg.device));
  const size_t N = (size_t)cfg.num_envs, D = (size_t)cfg.task.obs_dim, esz = cfg.dtype == OS2R_F64 ? 8 : 4;
  void *obs = nullptr, *rew = nullptr; uint8_t* done = nullptr;
  CHECK_HIP(hipMalloc(&obs, N * D * esz));
  CHECK_HIP(hipMalloc(&rew, N * esz));
  CHECK_HIP(hipMalloc((void**)&done, N));
  hipStream_t stream;
  CHECK_HIP(hipStreamCreate(&stream));

  CHECK_OS2R(sim, os2r_reset(sim, nullptr, obs, stream));
  CHECK_HIP(hipStreamSynchronize(stream));
  const auto t0 = std::chrono::steady_clock::now();
  for (int k = 0; k < steps; ++k) CHECK_OS2R(sim, os2r_step(sim, nullptr, obs, rew, done, nullptr, stream));
  CHECK_HIP(hipStreamSynchronize(stream));
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

  std::vector<unsigned char> h_obs(N * D * esz), h_rew(N * esz), h_done(N);
  CHECK_HIP(hipMemcpy(h_obs.data(), obs, h_obs.size(), hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(h_rew.data(), rew, h_rew.size(), hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(h_done.data(), done, h_done.size(), hipMemcpyDeviceToHost));
  unsigned long long sum = 1469598103934665603ull;          // FNV-1a over the raw output bytes
  for (auto* v : {&h_obs, &h_rew, &h_done}) for (unsigned char c : *v) { sum ^= c; sum *= 1099511628211ull; }
  uint64_t count = 0;
  CHECK_OS2R(sim, os2r_get_step_count(sim, &count));
  std::printf("{\"envs\": %zu, \"steps\": %d, \"step_count\": %llu, \"env_steps_per_s\": %.6g, \"checksum\": \"%016llx\"}\n", N, steps,
              (unsigned long long)count, N * (double)steps / dt, sum);
  CHECK_OS2R(sim, os2r_destroy(sim));
  (void)hipFree(obs); (void)hipFree(rew); (void)hipFree(done); (void)hipStreamDestroy(stream);
  return 0;
}
