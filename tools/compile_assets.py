#!/usr/bin/env python3
"""Build-side tool: compile the reference's robot descriptions into the package assets.

Reads (never copies) the URDF / STL / settings.yaml data files of a gym-os2r
checkout and writes the numbers the stepper needs:

  gym-os2r_amd/assets/models.json    one compiled serial chain per URDF variant
  gym-os2r_amd/assets/settings.json  the settings tree (task modes, spaces, resets)

Run in the build container only (the GPU box has no reference checkout):

  python tools/compile_assets.py --reference /root/reference
"""
import argparse
import importlib.util
import json
import os
import sys

import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "gym-os2r_amd")


def _load_compiler():
    spec = importlib.util.spec_from_file_location("os2r_model_compiler",
                                                  os.path.join(PKG, "model_compiler.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--max-cand-per-link", type=int, default=32)
    args = ap.parse_args()
    mc = _load_compiler()

    models_dir = os.path.join(args.reference, "gym_os2r", "models", "models")
    settings_path = os.path.join(args.reference, "gym_os2r", "models", "config", "default",
                                 "settings.yaml")
    with open(settings_path) as f:
        settings = yaml.load(f, Loader=yaml.FullLoader)

    variants = sorted({tm["model"] for tm in settings["task_modes"].values()})
    models = {}
    for name in variants:
        urdf = os.path.join(models_dir, name, name + ".urdf")
        if not os.path.exists(urdf):
            print(f"skip {name}: no URDF in the reference (settings entry without a model)")
            continue
        # torque limits are identical for every task mode that uses this model
        tms = [tm for tm in settings["task_modes"].values() if tm["model"] == name]
        act = tms[0]["spaces"]["action"]
        models[name] = mc.compile_urdf(urdf, actuated=list(act.keys()),
                                       max_torque=list(act.values()),
                                       max_cand_per_link=args.max_cand_per_link)
        m = models[name]
        print(f"{name}: nq={m['nq']} dofs={m['dof_names']} ncand={m['ncand']} "
              f"mass={[round(x, 4) for x in m['mass']]}")

    os.makedirs(os.path.join(PKG, "assets"), exist_ok=True)
    with open(os.path.join(PKG, "assets", "models.json"), "w") as f:
        json.dump({"max_cand_per_link": args.max_cand_per_link, "models": models}, f, indent=1)
    with open(os.path.join(PKG, "assets", "settings.json"), "w") as f:
        json.dump(settings, f, indent=1)
    print("wrote assets to", os.path.join(PKG, "assets"))


if __name__ == "__main__":
    sys.exit(main())
