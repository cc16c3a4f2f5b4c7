#!/usr/bin/env python3
"""Talking to the simulator the way the reference's examples/ignition_interaction.py does -- a simulator object, a
world, a model inserted by name, joints addressed by name, one `run()` per physics iteration -- through the N = 1
ScenarIO-shaped facade over the HIP stepper (`gym_os2r_amd.scenario`).

  python examples/scenario_interaction.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gym_os2r_amd as g

sc = g.scenario
gazebo = sc.GazeboSimulator(step_size=1e-4, rtf=1e9, steps_per_run=1)
assert gazebo.initialize()
world = gazebo.get_world()
world.to_gazebo().set_gravity((0, 0, -9.8))
world.set_physics_engine(sc.PhysicsEngine_dart)          # accepted for signature parity: there is one engine here

monopod = g.models.monopod.Monopod(world=world, monopod_version="monopod-fixed_hip")
print("models in the world:", world.model_names())
legs = ["hip_joint", "knee_joint"]
monopod.set_joint_control_mode(sc.JointControlMode_force, legs)
for name in legs:
    monopod.get_joint(name).set_joint_max_generalized_force([2.5])

everything = ["hip_joint", "knee_joint", "planarizer_pitch_joint", "planarizer_yaw_joint"]
monopod.to_gazebo().reset_joint_positions([0.2861059725058098, -0.587730986632999, 0.15, 0.0], everything)
monopod.to_gazebo().reset_joint_velocities([0.0] * 4, everything)
gazebo.run(paused=True)

for k in range(2000):                                    # 0.2 s: the foot drops 4.9 cm, lands, the leg pushes
    monopod.set_joint_generalized_force_targets([0.6, -0.3], legs)     # consumed by every run(), like the reference
    gazebo.run()
    if k % 400 == 399:
        q = monopod.joint_positions(everything)
        print(f"t = {(k + 1) * 1e-4:.3f} s  hip {q[0]:+.4f}  knee {q[1]:+.4f}  boom pitch {q[2]:+.4f}")
gazebo.close()
