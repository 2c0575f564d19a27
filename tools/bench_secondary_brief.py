import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
s = d.get("secondary", {})
def walk(p, v):
    if isinstance(v, dict):
        if "ms_per_step" in v or "value" in v:
            print(f"  {p}: " + ", ".join(f"{k}={v[k]:.4g}" if isinstance(v[k], float) else f"{k}={v[k]}" for k in v if k in ("value", "ms_per_step", "wall_ms_per_step", "solves_per_s", "ms", "wall_ms")))
        for k, x in v.items(): walk(p + "/" + k, x)
print(sys.argv[1], "headline", round(d["value"] / 1e6, 3), "M/s", round(d["ms_per_step"], 4), "ms")
walk("", s)
