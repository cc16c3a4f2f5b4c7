// os2r_pybind.cpp — thin pybind11 module over the C-ABI of include/os2r.h.
//
// One function per entry point, integer addresses in (tensor.data_ptr(), ctypes.addressof of
// the config struct, the raw hipStream_t), status codes out; no torch types, no logic.  The GIL is
// released around every call.  gym_os2r_amd.sim uses it when OS2R_BINDING=pybind11 (default: ctypes).
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <cstdint>

#include "../../include/os2r.h"

namespace py = pybind11;
using addr = std::uintptr_t;

static Os2rSim* H(addr h) { return reinterpret_cast<Os2rSim*>(h); }
static void* P(addr a) { return reinterpret_cast<void*>(a); }

PYBIND11_MODULE(_os2r_py, m) {
  m.doc() = "pybind11 binding of libos2r.so (MI355X batched monopod stepper)";
  const auto nogil = py::call_guard<py::gil_scoped_release>();
  m.def("abi_version", &os2r_abi_version);
  m.def("create", [](addr cfg) {
    Os2rSim* s = nullptr;
    int rc;
    { py::gil_scoped_release rel; rc = os2r_create(reinterpret_cast<const Os2rConfig*>(cfg), &s); }
    return py::make_tuple(rc, reinterpret_cast<addr>(s));
  });
  m.def("destroy", [](addr h) { return os2r_destroy(H(h)); }, nogil);
  m.def("reset", [](addr h, addr mask, addr obs, addr st) { return os2r_reset(H(h), (const uint8_t*)P(mask), P(obs), P(st)); }, nogil);
  m.def("step", [](addr h, addr act, addr obs, addr rew, addr done, addr term, addr st) {
    return os2r_step(H(h), P(act), P(obs), P(rew), (uint8_t*)P(done), P(term), P(st)); }, nogil);
  m.def("rollout", [](addr h, int n, addr act, addr obs, addr rew, addr done, addr term, addr why, addr st) {
    return os2r_rollout(H(h), n, P(act), P(obs), P(rew), (uint8_t*)P(done), P(term), (uint16_t*)P(why), P(st)); }, nogil);
  m.def("get_solver_state", [](addr h, addr l, addr f, addr st) { return os2r_get_solver_state(H(h), P(l), (uint32_t*)P(f), P(st)); }, nogil);
  m.def("set_solver_state", [](addr h, addr l, addr f, addr st) { return os2r_set_solver_state(H(h), P(l), (const uint32_t*)P(f), P(st)); }, nogil);
  m.def("get_state", [](addr h, addr q, addr qd, addr st) { return os2r_get_state(H(h), P(q), P(qd), P(st)); }, nogil);
  m.def("set_state", [](addr h, addr q, addr qd, addr st) { return os2r_set_state(H(h), P(q), P(qd), P(st)); }, nogil);
  m.def("get_action_history", [](addr h, int w, addr o, addr st) { return os2r_get_action_history(H(h), w, P(o), P(st)); }, nogil);
  m.def("set_action_history", [](addr h, int w, addr i, addr st) { return os2r_set_action_history(H(h), w, P(i), P(st)); }, nogil);
  m.def("set_params", [](addr h, int f, addr s, addr st) { return os2r_set_params(H(h), f, P(s), P(st)); }, nogil);
  m.def("get_params", [](addr h, int f, addr d, addr st) { return os2r_get_params(H(h), f, P(d), P(st)); }, nogil);
  m.def("get_episode_info", [](addr h, addr s, addr e, addr p, addr st) {
    return os2r_get_episode_info(H(h), (int32_t*)P(s), (uint32_t*)P(e), (uint8_t*)P(p), P(st)); }, nogil);
  m.def("set_episode_info", [](addr h, addr s, addr e, addr p, addr st) {
    return os2r_set_episode_info(H(h), (const int32_t*)P(s), (const uint32_t*)P(e), (const uint8_t*)P(p), P(st)); }, nogil);
  m.def("get_action_violations", [](addr h, addr d, int clear, addr st) {
    return os2r_get_action_violations(H(h), (uint32_t*)P(d), clear, P(st)); }, nogil);
  m.def("get_step_count", [](addr h) { uint64_t v = 0; int rc = os2r_get_step_count(H(h), &v); return py::make_tuple(rc, v); });
  m.def("set_step_count", [](addr h, uint64_t v) { return os2r_set_step_count(H(h), v); });
  m.def("bench_steps", [](addr h, int n, addr st) {
    float ms = 0.f;
    int rc;
    { py::gil_scoped_release rel; rc = os2r_bench_steps(H(h), n, P(st), &ms); }
    return py::make_tuple(rc, ms);
  });
  m.def("bench_enqueue", [](addr h, int n, addr st) {
    py::gil_scoped_release rel;
    return os2r_bench_steps(H(h), n, P(st), nullptr);
  });
  m.def("bench_steps_multi", [](std::vector<addr> hs, std::vector<addr> sts, int n) {
    std::vector<Os2rSim*> sims; std::vector<void*> streams;
    for (addr h : hs) sims.push_back(H(h));
    for (addr s : sts) streams.push_back(P(s));
    if (sims.size() != streams.size()) return (int)OS2R_ERR_INVALID;
    py::gil_scoped_release rel;
    return os2r_bench_steps_multi(sims.data(), streams.data(), (int)sims.size(), n);
  });
  m.def("set_work_counters", [](addr h, addr buf) { return os2r_set_work_counters(H(h), (uint64_t*)P(buf)); });
  m.def("set_done_reasons", [](addr h, addr buf) { return os2r_set_done_reasons(H(h), (uint16_t*)P(buf)); });
  m.def("set_done_mask", [](addr h, addr buf) { return os2r_set_done_mask(H(h), (uint8_t*)P(buf)); });
  m.def("get_violation_mirror", [](addr h) { const volatile uint32_t* w = nullptr; int rc = os2r_get_violation_mirror(H(h), &w); return py::make_tuple(rc, (addr)w); });
  m.def("model_is_compiled_in", [](addr model) { return os2r_model_is_compiled_in((const Os2rModel*)P(model)); });
  m.def("register_model_kernels", [](addr model, int dtype, int device, const std::string& path) {
    return os2r_register_model_kernels((const Os2rModel*)P(model), dtype, device, path.c_str()); });
  m.def("last_error", [](addr h) { return std::string(os2r_last_error(H(h))); });
}
