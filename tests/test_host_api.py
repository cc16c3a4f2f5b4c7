"""Host-side logic that needs no GPU: task layout vs the reference's, rewards, thresholds,
registry, error behaviour, and the C-ABI library's exported symbols."""
import ctypes
import json
import os
import re
import sys
import warnings

import numpy as np
import pytest

import gym_os2r_amd as g
from conftest import GOLDEN, ROOT
from helpers import MODES, make_task, model_for
from gym_os2r_amd import abi, rewards


@pytest.mark.parametrize("normalized", [True, False])
@pytest.mark.parametrize("mode", MODES)
def test_task_layout_matches_reference(mode, normalized):
    with open(os.path.join(GOLDEN, "task_layout.json")) as f:
        combo = json.load(f)["combos"][f"{mode}__{'norm' if normalized else 'nonorm'}"]
    reward = "StraightV1" if mode == "simple" else "BalancingV1"
    t = make_task(mode, reward, normalized)
    assert t.observation_index == combo["observation_index"]
    assert t.observation_mask == combo["observation_mask"]
    assert t.periodic_joints == combo["periodic_joints"]
    assert t.joint_names == combo["joint_names"] and t.action_names == combo["action_names"]
    assert list(t.max_torques) == combo["max_torques"]
    assert np.array_equal(t.observation_space.low, combo["obs_space_low"])
    assert np.array_equal(t.observation_space.high, combo["obs_space_high"])
    assert np.array_equal(t.reset_space.low, combo["reset_space_low"])
    assert np.array_equal(t.reset_space.high, combo["reset_space_high"])


@pytest.mark.parametrize("normalized", [True, False])
@pytest.mark.parametrize("mode", MODES)
def test_host_rewards_done_and_thresholds_match_reference(mode, normalized):
    """numpy reward classes, get_state_info, and the kernel's done thresholds (bisection) against
    the reference's outputs on the same observations."""
    z = np.load(os.path.join(GOLDEN, "task_epilogue.npz"))
    with open(os.path.join(GOLDEN, "task_layout.json")) as f:
        layout = json.load(f)
    key = f"{mode}__{'norm' if normalized else 'nonorm'}"
    combo = layout["combos"][key]
    gold_obs, gold_done = z[key + "__obs"], z[key + "__done"].astype(bool)
    model = model_for(mode)
    for rname in combo["rewards"]:
        t = make_task(mode, rname, normalized)
        spec = t.kernel_spec(model)
        mt = np.array(model["max_torque"])
        for i in range(len(gold_obs)):
            acts = [(mt * z["a"][i]) / mt, (mt * z["a_prev"][i]) / mt]
            r, d = t.get_state_info(gold_obs[i], acts)
            assert d == bool(gold_done[i])
            assert abs(r - z[f"{key}__reward__{rname}"][i]) <= 1e-15
            assert t.normalize_observation is not None
    # thresholds: done  <=>  some pre-map value outside [done_lo, done_hi]
    t = make_task(mode, combo["rewards"][0], normalized)
    spec = t.kernel_spec(model)
    dof = model["dof_names"]
    for i in range(len(gold_obs)):
        q = np.zeros(5); qd = np.zeros(5)
        for j, name in enumerate(layout["joint_order"]):
            if name in dof:
                q[dof.index(name)] = z["q"][i, j]; qd[dof.index(name)] = z["qd"][i, j]
        a_prev = (np.array(model["max_torque"]) * z["a_prev"][i]) / np.array(model["max_torque"])
        done = False
        for d_ in range(spec["obs_dim"]):
            kind, src = spec["obs_kind"][d_], spec["obs_src"][d_]
            if kind in (abi.OBS_TORQUE_NORM, abi.OBS_TORQUE_RAW):
                y = a_prev[src]
            elif kind in (abi.OBS_VEL_TANH, abi.OBS_VEL_RAW):
                y = qd[src]
            else:
                y = q[src]
            if kind in (abi.OBS_POS_PERIODIC_NORM, abi.OBS_POS_PERIODIC_RAW):
                y = np.mod(y + np.pi, 2 * np.pi) - np.pi
            if not (spec["done_lo"][d_] <= y <= spec["done_hi"][d_]):
                done = True
        assert done == bool(gold_done[i]), (key, i)


def test_host_tolerance_matches_reference():
    z = np.load(os.path.join(GOLDEN, "tolerance.npz"))
    names = ["gaussian", "hyperbolic", "long_tail", "reciprocal", "cosine", "linear", "quadratic", "tanh_squared"]
    for si, s in enumerate(names):
        for pi, (lo, up, mg, vam) in enumerate(z["params"]):
            gold = z[f"sigmoid_{si}"][pi]
            if np.isnan(gold).all():
                with pytest.raises(ValueError):
                    rewards.tolerance(z["x"], (lo, up), mg, s, vam)
                continue
            with np.errstate(all="ignore"):
                got = rewards.tolerance(z["x"], (lo, up), mg, s, vam)
            assert np.array_equal(got, gold, equal_nan=True)
    with pytest.raises(ValueError):
        rewards.tolerance(0.0, (1.0, 0.0))
    with pytest.raises(ValueError):
        rewards.tolerance(0.0, (0.0, 1.0), margin=-1)


def test_reset_ik_matches_reference():
    with open(os.path.join(GOLDEN, "reset_ik.json")) as f:
        gold = json.load(f)
    from gym_os2r_amd.utils.reset import leg_joint_angles
    for mode, entry in gold["poses"].items():
        for pose, (pitch, hip, knee) in entry["angles"].items():
            d = dict(entry["definition"]); d["planarizer_pitch_joint"] = pitch
            got = leg_joint_angles(d)
            assert float(got[0]) == hip and float(got[1]) == knee
    with pytest.raises(RuntimeError):
        leg_joint_angles({"bogus": 1})


def test_task_constructor_errors_like_the_reference():
    from gym_os2r_amd.tasks.monopod import MonopodTask
    with pytest.raises(RuntimeError, match="Missing required kwarg"):
        MonopodTask(1000, task_mode="fixed_hip", reward_class=rewards.BalancingV1)
    with pytest.raises(RuntimeError, match="reset positions"):
        MonopodTask(1000, task_mode="fixed_hip", reward_class=rewards.BalancingV1, reset_positions=["nope"])
    with pytest.raises(RuntimeError, match="not supported"):
        MonopodTask(1000, task_mode="hovering", reward_class=rewards.BalancingV1, reset_positions=["stand"])
    t = MonopodTask(1000, task_mode="simple", reward_class=rewards.BalancingV1, reset_positions=["stand"])
    with pytest.raises(AssertionError, match="not supported by reward"):
        t.create_spaces()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        MonopodTask(1000, task_mode="fixed_hip", reward_class=rewards.BalancingV1, reset_positions=["stand"], extra=1)
        assert any(issubclass(x.category, SyntaxWarning) for x in w)


def test_settings_config_xpath():
    cfg = g.config.SettingsConfig()
    assert cfg.get_config("task_modes/free_hip/model") == "monopod"
    assert cfg.get_config("/resets/stand/planarizer_pitch_joint") == 0.15
    node = cfg.get_config("resets/stand")
    node["planarizer_pitch_joint"] = 9.0                      # shallow copy: the tree is untouched
    assert cfg.get_config("resets/stand/planarizer_pitch_joint") == 0.15
    cfg.set_config(0.3, "resets/custom/planarizer_pitch_joint")
    assert cfg.get_config("resets/custom") == {"planarizer_pitch_joint": 0.3}


def test_registry_is_the_reference_table():
    # gym_os2r/__init__.py:16-128
    want = {"Monopod-stand-v1": ("fixed_hip", "StandingV1", ["ground"], 100_000, True),
            "Monopod-balance-v1": ("fixed_hip_simple", "BalancingV1", ["stand"], 100_000, True),
            "Monopod-balance-v2": ("fixed_hip_simple", "BalancingV2", ["stand"], 100_000, True),
            "Monopod-balance-v3": ("fixed_hip_simple", "BalancingV2", ["stand", "half_stand", "ground", "lay", "float"], 10_000, True),
            "Monopod-nonorm-balance-v1": ("fixed_hip_simple", "BalancingV1", ["stand"], 100_000, False),
            "Monopod-nonorm-balance-v2": ("fixed_hip_simple", "BalancingV2", ["stand"], 100_000, False),
            "Monopod-nonorm-balance-v3": ("fixed_hip_simple", "BalancingV2", ["stand", "half_stand", "ground", "lay", "float"], 10_000, False),
            "Monopod-hop-v1": ("free_hip", "HoppingV1", ["stand"], 100_000, True),
            "Monopod-simple-v1": ("simple", "StraightV1", ["stand"], 100_000, True)}
    assert set(g.REGISTRY) == set(want)
    for env_id, (mode, rew, resets, steps, norm) in want.items():
        s = g.REGISTRY[env_id]
        kw = s["kwargs"]
        assert (kw["task_mode"], kw["reward_class"].__name__, kw["reset_positions"], s["max_episode_steps"]) == (mode, rew, resets, steps)
        assert kw["agent_rate"] == 1000 and kw["physics_rate"] == 10000 and kw["task_cls"].normalized is norm
        env = g.make(env_id, num_envs=3)
        assert env.num_of_steps_per_run == 10 and env.num_envs == 3
        assert env.observation_space.shape == (len(env.task.observation_index),)
    with pytest.raises(KeyError):
        g.make("Monopod-nope-v0")


def test_randomizer_wrappers_configure_the_reset_mode():
    e = g.randomizers.monopod.MonopodEnvRandomizer(env=lambda: g.make("Monopod-balance-v1", num_envs=2))
    assert e.unwrapped._reset_mode == abi.RESET_RANDOM and e.unwrapped._randomize_params
    e = g.randomizers.monopod_no_rand.MonopodEnvNoRandomizer(env=lambda: g.make("Monopod-balance-v1", num_envs=2))
    assert e.unwrapped._reset_mode == abi.RESET_FIXED and e.unwrapped.model["gravity_z"] == -9.80665
    v = g.common.make_mp_envs("Monopod-balance-v2", 6, 11, g.randomizers.monopod.MonopodEnvRandomizer, start_idx=4)
    assert v.num_envs == 6 and v.unwrapped._opts["seed"] == 11 and v.unwrapped._opts["env_offset"] == 4


def test_device_path_fails_loudly_without_gpu():
    """No CPU fallback: creating the simulator without a GPU raises instead of computing on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from gym_os2r_amd.sim import Os2rError
    env = g.make("Monopod-balance-v1", num_envs=2)
    with pytest.raises(Os2rError, match="no GPU|no CPU fallback"):
        env.reset()


def test_capi_library_exports_every_declared_symbol():
    """include/os2r.h <-> libos2r.so: every declared entry point is exported (no compute calls)."""
    from gym_os2r_amd import _lib
    with open(os.path.join(ROOT, "include", "os2r.h")) as f:
        header = f.read()
    declared = set(re.findall(r"\b(os2r_[a-z_0-9]+)\s*\(", header))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), name
    # ... and nothing else is: the dynamic symbol table of the library is the header (-fvisibility=hidden + csrc/libos2r.map)
    import shutil
    import subprocess
    if shutil.which("nm"):
        out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
        exported = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
        assert exported == declared, sorted(exported ^ declared)[:10]
    assert lib.os2r_abi_version() == abi.ABI_VERSION
    lib.os2r_last_error.restype = ctypes.c_char_p
    # struct layout agreement between the header (as compiled) and the ctypes mirror
    assert ctypes.sizeof(abi.Os2rConfig) == ctypes.sizeof(abi.Os2rModel) + ctypes.sizeof(abi.Os2rTaskSpec) + 104
    # a null config is rejected with an error code and a message, without touching a device
    out = ctypes.c_void_p()
    assert _lib.load().os2r_create(None, ctypes.byref(out)) == abi.ERR_INVALID
    assert b"null config" in _lib.load().os2r_last_error(None)


def test_the_checker_oracle_has_no_experimental_switch(oracle):
    """The oracle that the parity tests, smoke() and the CPU baseline load is the checker build: the switches of the solver studies
    (orc_set_experimental_*, the solve traces, the debug counters) exist in the laboratory build only (oracle/Makefile: `make lab`,
    -DORC_EXPERIMENTS), so no test can run against a specification that an earlier test or study left switched.
    And the product's device code carries no timing-only experiment of the solver either."""
    import shutil
    import subprocess
    so = os.path.join(ROOT, "oracle", "libos2r_oracle.so")
    assert os.path.exists(so)
    if shutil.which("nm"):
        out = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
        names = [ln.split()[-1] for ln in out.splitlines() if ln.strip()]
        assert names and not [n for n in names if "experimental" in n or "debug_counter" in n], names
    assert not hasattr(oracle.lib(), "orc_set_experimental_warm")
    with oracle.laboratory() as lab:                  # the laboratory is a second library; the checker stays what lib() returns
        assert hasattr(lab, "orc_set_experimental_warm")
    assert not hasattr(oracle.lib(), "orc_set_experimental_warm")
    for name in ("os2r_device.hpp", "os2r_kernels.hpp"):
        with open(os.path.join(ROOT, "gym-os2r_amd", "csrc", name)) as f:
            src = f.read()
        assert "OS2R_PAIR_FIX" not in src and "OS2R_WARM_FIRST" not in src, name


def test_model_compiler_on_a_synthetic_urdf(tmp_path):
    """Fixed-joint lumping, literal rpy, composite inertia: checked on a hand-computable chain."""
    urdf = tmp_path / "toy.urdf"
    urdf.write_text("""<robot name="toy">
      <link name="world"/>
      <link name="a"><inertial><origin xyz="0 0 0.5" rpy="0 0 0"/><mass value="2"/>
        <inertia ixx="0.1" ixy="0" ixz="0" iyy="0.2" iyz="0" izz="0.3"/></inertial></link>
      <link name="b"><inertial><origin xyz="0 0 0" rpy="0 0 0"/><mass value="1"/>
        <inertia ixx="0.01" ixy="0" ixz="0" iyy="0.01" iyz="0" izz="0.01"/></inertial></link>
      <link name="c"><inertial><origin xyz="0.1 0 0" rpy="0 0 0"/><mass value="0.5"/>
        <inertia ixx="0.001" ixy="0" ixz="0" iyy="0.002" iyz="0" izz="0.003"/></inertial></link>
      <joint name="j1" type="continuous"><origin xyz="0 0 1" rpy="0 0 0"/><parent link="world"/><child link="a"/>
        <axis xyz="0 0 1"/><dynamics damping="0.1" friction="0.2"/></joint>
      <joint name="weld" type="fixed"><origin xyz="0 0 1" rpy="0 0 0"/><parent link="a"/><child link="b"/></joint>
      <joint name="j2" type="continuous"><origin xyz="0 0.2 0" rpy="1.57 0 0"/><parent link="b"/><child link="c"/>
        <axis xyz="1 0 0"/><dynamics damping="0.3" friction="0.4"/></joint>
    </robot>""")
    from gym_os2r_amd.model_compiler import compile_urdf, rpy_to_matrix
    m = compile_urdf(str(urdf), actuated=("j1", "j2"), with_meshes=False)
    assert m["nq"] == 2 and m["dof_names"] == ["j1", "j2"] and m["axis"] == [2, 0]
    assert m["body_links"] == [["a", "b"], ["c"]]
    assert m["mass"] == [3.0, 0.5] and m["damping"] == [0.1, 0.3] and m["friction"] == [0.2, 0.4]
    np.testing.assert_allclose(m["com"][0], [0, 0, (2 * 0.5 + 1 * 1.0) / 3])
    # parallel axis: Ixx = 0.1 + 0.01 + 2*(1/6)^2 + 1*(1/3)^2
    np.testing.assert_allclose(m["icom"][0][0], 0.1 + 0.01 + 2 * (1 / 6) ** 2 + 1 * (1 / 3) ** 2)
    np.testing.assert_allclose(m["rpos"][1], [0, 0.2, 1.0])          # through the lumped fixed joint
    np.testing.assert_allclose(np.array(m["rfix"][1]).reshape(3, 3), rpy_to_matrix(1.57, 0, 0))
    assert abs(m["rfix"][1][4] - np.cos(1.57)) < 1e-16 and m["rfix"][1][4] != 0.0   # literal 1.57, not pi/2
    s = abi.model_struct(m)
    assert s.nq == 2 and s.ncand == 0


def test_pybind11_module_mirrors_the_c_abi():
    """The thin pybind11 module exposes one function per C-ABI entry point (no compute here)."""
    import importlib
    from gym_os2r_amd import _lib
    m = importlib.import_module("gym_os2r_amd._os2r_py")
    assert m.abi_version() == abi.ABI_VERSION
    for sym in _lib.SYMBOLS:
        assert hasattr(m, sym[len("os2r_"):]), sym
    rc, handle = m.create(0)                       # null config: rejected, no device touched
    assert rc == abi.ERR_INVALID and handle == 0 and "null config" in m.last_error(0)


def test_jit_builds_a_code_object_for_a_custom_robot(monkeypatch):
    """gym_os2r_amd/jit.py: the robot's constexpr table and the hipcc --genco build (cross-compiles without a
    GPU).  The code object must export the two kernels of the requested contact flag; a second request is a
    cache hit.  (Loading and running it is a GPU test.)"""
    from conftest import KERNEL_CACHE
    from helpers import perturbed_model
    from gym_os2r_amd import jit
    if jit.hipcc_path() is None:
        pytest.skip("no hipcc on this machine")
    monkeypatch.setenv("OS2R_KERNEL_CACHE", KERNEL_CACHE)
    model = perturbed_model("free_hip", np.random.default_rng(77))
    ms = abi.model_struct(model)
    src = jit.table_source(ms)
    assert "struct Tables<100>" in src and f"static constexpr int nq = {model['nq']};" in src
    assert float(model["mass"][2]).hex() in src                      # exact hex-float literals
    path = jit.build(ms, abi.F64, True)
    assert os.path.getsize(jit.build(ms, abi.F32, True)) > 100_000          # the f32 kernels of the same robot
    assert os.path.getsize(path) > 100_000
    blob = open(path, "rb").read()
    assert b"os2r_jit_step_c1_d0" in blob and b"os2r_jit_step_c1_d1" in blob and b"os2r_jit_step_c0_d0" not in blob
    mtime = os.path.getmtime(path)
    assert jit.build(ms, abi.F64, True) == path and os.path.getmtime(path) == mtime
    assert jit.code_object_path(ms, abi.F32, True) != path and jit.code_object_path(ms, abi.F64, False) != path
    ms.mass[0] = np.nextafter(ms.mass[0], 1.0)
    assert jit.code_object_path(ms, abi.F64, True) != path           # any constant of the robot is part of the key


def test_jit_code_object_with_the_task_layout_folded_in(monkeypatch):
    """With the layout of a task (`jit.task_layout`) the code object also exports the contact kernels that have
    the observation layout folded in, and the data symbol that announces which layout that is."""
    from conftest import KERNEL_CACHE
    from helpers import make_config, perturbed_model
    from gym_os2r_amd import jit
    if jit.hipcc_path() is None:
        pytest.skip("no hipcc on this machine")
    monkeypatch.setenv("OS2R_KERNEL_CACHE", KERNEL_CACHE)
    model = perturbed_model("fixed_hip", np.random.default_rng(79))       # the robot of the GPU test: cached for it
    cfg, _, _ = make_config("fixed_hip_simple", "BalancingV1", True, num_envs=8, contact=True, model_overrides=model)
    lay = jit.task_layout(cfg.task)
    assert lay == (0x22010, 0x32132, 5)                                    # kinds / source dofs, slot 0 lowest
    path = jit.build(cfg.model, abi.F64, True, layout=lay)
    blob = open(path, "rb").read()
    assert b"os2r_jit_step_c1_d0_l" in blob and b"os2r_jit_step_c1_d1_l" in blob and b"os2r_jit_layout" in blob
    assert path != jit.code_object_path(cfg.model, abi.F64, True, None)
    assert jit.code_object_path(cfg.model, abi.F64, False, lay) == jit.code_object_path(cfg.model, abi.F64, False, None)
    cfg.task.obs_kind[0] = 99                                              # does not fit four bits: no folded layout
    assert jit.task_layout(cfg.task) is None


def test_ctypes_mirror_matches_the_header_as_compiled(tmp_path):
    """include/os2r.h compiled by gcc <-> gym_os2r_amd/abi.py: size of every struct and offset of every
    Os2rConfig / Os2rTaskSpec field (the ABI changed in round 2: pgs_tol, gravity_rollouts)."""
    import subprocess
    cfg_fields = [f[0] for f in abi.Os2rConfig._fields_]
    task_fields = [f[0] for f in abi.Os2rTaskSpec._fields_]
    model_fields = [f[0] for f in abi.Os2rModel._fields_]
    src = tmp_path / "layout.c"
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{os.path.join(ROOT, "include", "os2r.h")}"', 'int main(void) {',
             '  printf("%zu %zu %zu\\n", sizeof(Os2rConfig), sizeof(Os2rTaskSpec), sizeof(Os2rModel));']
    for st, fields in (("Os2rConfig", cfg_fields), ("Os2rTaskSpec", task_fields), ("Os2rModel", model_fields)):
        for f in fields:
            lines.append(f'  printf("{st}.{f} %zu\\n", offsetof({st}, {f}));')
    lines += ['  return 0;', '}']
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c11", "-o", str(exe), str(src)])
    out = subprocess.check_output([str(exe)], text=True).splitlines()
    sizes = [int(x) for x in out[0].split()]
    assert sizes == [ctypes.sizeof(abi.Os2rConfig), ctypes.sizeof(abi.Os2rTaskSpec), ctypes.sizeof(abi.Os2rModel)]
    for line in out[1:]:
        name, off = line.split()
        st, f = name.split(".")
        assert getattr(getattr(abi, st), f).offset == int(off), name


def test_vec_env_indices_and_seed_semantics():
    """HipVecEnv.get_attr / set_attr / env_method with `indices` (common/vec_env/subproc_vec_env.py:125-214), on stand-in
    runtimes: no GPU involved.  An attribute or a method belongs to the batch: a strict subset of the environments is
    refused (a value recorded for a subset and reported back would not be what the simulation uses, and
    env_method('reset', indices=[0]) would reset everybody), the whole batch -- also spelled out as indices -- is served."""
    from gym_os2r_amd.common.vec_env import HipVecEnv

    class Runtime:
        observation_space = action_space = None
        color = "red"
        calls = 0

        def __init__(self, n):
            self.num_envs = n

        @property
        def unwrapped(self):
            return self

        def seed(self, seed=None):
            return [seed]

        def ping(self, x):
            self.calls += 1
            return x + 1

        def close(self):
            pass

    rt = Runtime(6)
    vec = HipVecEnv(rt)
    assert vec.get_attr("color") == ["red"] * 6
    assert vec.get_attr("color", indices=[1, 4]) == ["red", "red"] and vec.get_attr("color", indices=2) == ["red"]
    with pytest.raises(NotImplementedError):
        vec.set_attr("color", "blue", indices=[1, 4])                 # a subset: refused, nothing recorded
    assert vec.get_attr("color") == ["red"] * 6 and rt.color == "red"
    vec.set_attr("color", "green")                                    # everybody: the shared runtime's attribute
    assert vec.get_attr("color") == ["green"] * 6 and rt.color == "green"
    vec.set_attr("color", "teal", indices=range(6))                   # everybody, spelled out
    assert rt.color == "teal"
    with pytest.raises(NotImplementedError):
        vec.env_method("ping", 1, indices=[0, 5])
    assert rt.calls == 0
    assert vec.env_method("ping", 1) == [2] * 6 and rt.calls == 1
    assert vec.seed(7) == [7] * 6
    with pytest.raises(IndexError):
        vec.get_attr("color", indices=[6])
    with pytest.raises(RuntimeError):
        vec.step_wait()                                               # nothing in flight
    # several shards of one batch: sizes, slices, one call per shard
    a, b = Runtime(4), Runtime(2)
    vec2 = HipVecEnv(a, b)
    assert vec2.num_envs == 6 and vec2.num_splits == 2 and vec2.split_slices == [slice(0, 4), slice(4, 6)]
    assert vec2.env_method("ping", 3) == [4] * 6 and a.calls == 1 and b.calls == 1
    vec2.set_attr("color", "grey")
    assert a.color == b.color == "grey" and vec2.seed(3) == [3] * 6
    vec2.close()


def test_make_mp_envs_cuts_the_batch_into_contiguous_shards():
    """common.make_mp_envs(..., num_splits=k): shard r gets the balanced contiguous range of distributed.shard_range and
    its global offset (the key of every random stream): what makes the shards reproduce the single batch."""
    from gym_os2r_amd import common, randomizers
    vec = common.make_mp_envs("Monopod-balance-v1", 10, 7, randomizers.monopod.MonopodEnvRandomizer, start_idx=100, num_splits=3)
    assert [e.num_envs for e in vec.envs] == [4, 3, 3] and vec.num_envs == 10
    assert [e.unwrapped._opts["env_offset"] for e in vec.envs] == [100, 104, 107]
    assert all(e.unwrapped._opts["seed"] == 7 for e in vec.envs)
    assert vec.split_slices == [slice(0, 4), slice(4, 7), slice(7, 10)]


def test_gravity_is_drawn_anew_after_num_physics_rollouts(oracle):
    """MonopodEnvRandomizer(num_physics_rollouts=k): the reference re-creates its simulator -- and with it draws
    gravity (randomizers/monopod.py:36-41,56-61,371) -- after every k rollouts; here per environment, in the reset."""
    from helpers import make_config
    for k in (0, 2):
        cfg, task, model = make_config("free_hip", num_envs=32, reset_mode=abi.RESET_RANDOM, randomize_params=True,
                                       max_episode_steps=3, seed=5, contact=True)
        cfg.task.gravity_rollouts = k
        o = oracle.OracleSim(cfg, threads=4)
        g0 = o.get_params(abi.PARAM_GRAVITY)[0].copy()
        hist = [g0]
        for _ in range(13):                                           # TimeLimit 3: a rollout ends every third step
            o.step(None)
            hist.append(o.get_params(abi.PARAM_GRAVITY)[0].copy())
        _, epi, _ = o.episode_info()
        assert (epi >= 5).all()                                       # create + four finished rollouts at least
        h = np.array(hist)
        if k == 0:
            assert (h == g0).all()
        else:
            changes = (np.diff(h, axis=0) != 0).sum(axis=0)
            assert (changes >= 2).all() and (changes <= 4).all()      # after rollouts 2 and 4 (early `done`s add some)
            assert abs(h[-1].mean() + 9.8) < 0.2 and 0.05 < h[-1].std() < 0.4
        o.close()
    from gym_os2r_amd.randomizers.monopod import MonopodEnvRandomizer
    with pytest.raises(ValueError):
        MonopodEnvRandomizer(env=lambda: None, num_physics_rollouts=-1)


def test_code_objects_of_the_built_library_have_no_scratch_and_no_runtime_tables():
    """Guards two regressions that cost a factor of two each before they were seen in the ISA:
    (1) scratch: every step kernel (fp64 and fp32, compiled-in robots and the generic run-time-model ones) runs out of
        registers and LDS only; the fp32 ones fit two waves per SIMD (<= 256 unified registers);
    (2) the robots' constexpr tables are folded into the instruction stream: a code object that still carries a
        `gen::Tables<..>` data symbol reads a table at run time (as `cand_group_axis` did for the 40-point link of
        monopod-fixed_hip: 58 % of an env-step of Monopod-balance-v1).
    Read from the gfx950 code objects inside libos2r.so (tools/kernel_meta.py); no GPU needed."""
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_meta
    from gym_os2r_amd import _lib
    if not os.path.exists(os.path.join(kernel_meta.LLVM, "llvm-readelf")) or not os.path.exists(_lib.LIB_PATH):
        pytest.skip("needs ROCm's llvm-readelf and the built libos2r.so")
    meta = kernel_meta.kernel_meta(_lib.LIB_PATH)
    steps = {k: v for k, v in meta.items() if "step_kernel<" in k}
    assert sum("StModel<" in k for k in steps) >= 48 and sum("RtModel<" in k for k in steps) >= 32   # compiled-in robots; generic kernels
    # One phase-2 solver per kernel (the exact finish or the sweeps-only solver, chosen at launch) keeps even the generic
    # 5-dof fp64 contact kernels, which carry the rows of five bodies, inside the 512 registers: no scratch anywhere.
    rollouts = 0
    for name, m in steps.items():
        # the last template argument: the fused K-step variants of os2r_rollout (the step loop around the same body)
        rollout = re.search(r", true>\(os2r::StepArgs<", name) is not None and re.search(r", (true|false), \d, true>\(", name) is not None
        rollouts += rollout
        if m["private_segment_fixed_size"] != 0:
            # a few bytes reserved for the register allocator's SGPR spill bookkeeping are no scratch traffic: the kernel may
            # then hold no vector spill and no scratch instruction (read from its disassembly)
            import isa_histogram
            needle = name[name.index("step_kernel<"):name.index(">(os2r::StepArgs") + 2]
            _, _, insts = isa_histogram.disassemble(_lib.LIB_PATH, needle, counting=", true, " in needle[-20:])
            assert m["vgpr_spill_count"] == 0 and not [i for i in insts if i[1].startswith("scratch_")], (name, m)
            assert m["private_segment_fixed_size"] <= 68, (name, m)
        assert m["vgpr_spill_count"] <= 8, (name, m)                # AGPR spill slots of the register allocator, a handful at most
        if "step_kernel<float" in name:
            assert m["vgpr_count"] <= 256, (name, m["vgpr_count"])
        if "StModel<" in name and "os2r::StLayout" in name and not rollout:
            assert m["vgpr_spill_count"] <= 2, (name, m)             # the kernels of the reference's task modes
    assert rollouts >= 20, rollouts
    tables = []
    for co in kernel_meta.code_objects(_lib.LIB_PATH):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            out = subprocess.run([os.path.join(kernel_meta.LLVM, "llvm-readelf"), "--symbols", f.name], capture_output=True, text=True).stdout
        tables += [ln.split()[-1] for ln in out.splitlines() if ("Tables" in ln or "CandMeta" in ln) and "OBJECT" in ln]
    assert not tables, sorted(set(tables))[:5]


def test_flop_model_prices_the_sweep_units_from_the_source():
    """tools/flop_model.py: the two sweep units of profiles/flop_model.json are not fitted but counted --
    a row with nz non-zeros is nz FMA + 1 FMA + 1 ADD + nz FMA, a joint row starts its residual with a MUL -- times 64 lanes;
    and the committed model prices no solver unit at zero."""
    import json
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import flop_model
    joint = 64 * sum(4 * (j + 1) + 2 for j in range(5))
    assert flop_model.sweep_prices("C4", 3) == (joint, 64 * (3 * (15 + 19 + 23)) / 3)
    assert flop_model.sweep_prices("V1", 3)[0] == 64 * sum(4 * (j + 1) + 2 for j in range(4))
    model = json.load(open(os.path.join(ROOT, "profiles", "flop_model.json")))
    for wl in ("C4_f64", "C3_f64", "V1_f64"):
        k = model[wl]["flops_per_unit"]
        assert k["sweep"] > 0 and k["body_sweep"] > 0 and k["exact_solve"] > 0, (wl, k)
    assert model["C4_f64"]["flops_per_unit"]["sweep"] == joint


def test_python_defaults_are_the_compiled_in_standard_solver():
    """The kernels built for the default solver settings (compile-time loop bounds: StdSolver::kNormalIters, sweep_cap_base +
    kExactRounds in csrc/os2r_device.hpp) serve a handle only if its configuration says exactly those numbers: the defaults of
    abi.config_struct must be the header's, or every default handle silently runs the general kernels."""
    from gym_os2r_amd import abi
    src = open(os.path.join(ROOT, "gym-os2r_amd", "csrc", "os2r_device.hpp")).read()
    normal = {int(m) for m in re.findall(r"struct StdSolver[^;]*kNormalIters = (\d+);", src)}
    assert normal == {abi.DEFAULT_PGS_NORMAL_ITERS}
    base = re.search(r"constexpr int sweep_cap_base\(int nq\) \{ return nq >= 5 \? (\d+) : (\d+); \}", src)
    rounds = re.search(r"constexpr int kExactRounds = (\d+);", src)
    if base and rounds:
        assert abi.default_pgs_iters_exact(5) == int(base.group(1)) + int(rounds.group(1))
        assert abi.default_pgs_iters_exact(3) == int(base.group(2)) + int(rounds.group(1))
