"""Session helper (runs on the GPU box): takes the workloads that gpurun_out/flopmodel/flop_model.json holds into
profiles/flop_model.json, so that the bench runs that follow in the same session price their counted work with the fresh
fit; the merged file comes back as gpurun_out/flop_model_merged.json."""
import json

a = json.load(open("profiles/flop_model.json"))
a.update(json.load(open("gpurun_out/flopmodel/flop_model.json")))
json.dump(a, open("profiles/flop_model.json", "w"), indent=1)
json.dump(a, open("gpurun_out/flop_model_merged.json", "w"), indent=1)
