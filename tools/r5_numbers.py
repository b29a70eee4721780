"""Print the figures DESIGN.md section 6 quotes from a finished series (gpurun_out/r5f/*.json): python tools/r5_numbers.py [dir]"""
import json
import os
import sys

d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r5f"


def load(n):
    p = os.path.join(d, n + ".json")
    if not os.path.exists(p):
        return None
    lines = [l for l in open(p) if l.startswith("{")]
    return json.loads(lines[-1]) if lines else None


for n in ("default", "fwd", "default_b", "no_centre_first", "nodefer", "ddp1", "ddp1_nooverlap", "ddp1_torch", "spnet", "dense2m",
          "dense2m_bf16", "cylinder", "multi", "multi5", "default_bf16", "default_f32mlp"):
    j = load(n)
    if j is None:
        continue
    r = j.get("roofline") or {}
    a = j.get("attention_roofline") or {}
    f = j.get("fwd_only") or {}
    p = j.get("parity") or {}
    print(f"{n:16s} {j['ms_per_step']:8.3f} ms  {j['value'] / 1e6:6.3f} M/s  fwd {f.get('ms_per_step')}  conv us/launch {r.get('us_per_launch')} frac {r.get('frac')} "
          f"traffic {r.get('traffic')} stale {r.get('traffic_stale')} by_counters {r.get('frac_by_counters')} footprint {r.get('frac_of_hbm_by_footprint')}  attn {a.get('frac')} "
          f"{a.get('ms_per_forward')}  mem {j.get('peak_memory_gb')}  parity {p.get('max_abs_logit_diff')} / |logit| {p.get('max_abs_logit')} rel {p.get('max_rel_logit_diff')} "
          f"after {(p.get('after_training') or {}).get('max_abs_logit_diff')} steps {(p.get('after_training') or {}).get('steps')}")
    if j.get("train_storage"):
        t = j["train_storage"]
        print("   train_storage", t["ms_per_step_fp32_copies"], "->", t["ms_per_step"], "ms;", t["peak_memory_gb_fp32_copies"], "->", t["peak_memory_gb"], "GB")
j = load("default")
if j:
    print("idle", {k: v for k, v in (j.get("idle") or {}).items() if k != "note"})
    print("fp32_exact", j.get("fp32_exact"))
    print("cpu_baseline", j.get("cpu_baseline"))
    print("vs_fp64", (j.get("parity") or {}).get("vs_fp64_oracle"))
    print("voxel/aux", j["parity"].get("max_abs_voxel_logit_diff"), j["parity"].get("max_abs_aux_logit_diff"))
    print("conv_layers us", [x["us"] for x in j["conv_layers"]])
    print("lib", (j.get("roofline") or {}).get("lib_sha16"))
for n in ("attn.txt",):
    p = os.path.join(d, n)
    if os.path.exists(p):
        print(open(p).read().strip().splitlines()[-1])
