#!/usr/bin/env python3
"""profiles/traffic.json from the traffic_raw.json of tools/profile.sh runs (drop-in and fused):
HBM bytes per launch (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, KiB units) keyed by bench.py stage."""
import json
import sys

STAGE = {"cull": "cull", "shade1": "shade1", "shade2": "shade2", "finish": "finish", "render_bwd": "backward",
         "loss_bwd_fused": "loss_bwd_fused", "ray_loss": "ray_loss", "collect_valid": "collect", "refit": "refit"}


def load(path, fused):
    raw = json.load(open(path))
    tag = "<true>" if fused else "<false>"
    out = {}
    for k, v in raw.items():
        base = k.split("<")[0]
        if "<" in k and base != "trace" and not k.endswith(tag):
            continue                      # the other mode's instantiation
        if base in STAGE:
            out[STAGE[base]] = v["hbm_bytes_per_launch"]
        elif k == "trace<false>":         # closest-hit launches: trace1 and trace2 share the kernel name
            out["trace1"] = out["trace2"] = v["hbm_bytes_per_launch"]
        elif k == "trace<true>":
            out["trace3"] = v["hbm_bytes_per_launch"]
    return out


dropin, fused, dst = sys.argv[1], sys.argv[2], sys.argv[3]
res = {"dropin": load(dropin, False), "fused": load(fused, True),
       "note": "HBM bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --steps 1 "
               "--warmup 0 --no-extras --random-targets`; FETCH_SIZE x 1024 x 2 (gfx950 counts 128-B requests as 64 B; "
               "calibrated on k_cull, which reads exactly 48 B/ray), WRITE_SIZE x 1024 as is. trace1/trace2 are the "
               "per-launch mean over both closest-hit k_trace launches (one kernel name in rocprofv3)."}
json.dump(res, open(dst, "w"), indent=1)
print(json.dumps(res, indent=1)[:1400])
