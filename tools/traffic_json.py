"""profiles/<tag>_pmc.txt (tools/prof.sh) -> profiles/rNN_msda_traffic.json: HBM bytes per launch of the MSDA forward.
    python tools/traffic_json.py profiles/r02_bench_pmc.txt profiles/r02_msda_traffic.json [frames_per_launch]
HBM bytes = 2 * FETCH_SIZE + WRITE_SIZE (KB counters): on gfx950 FETCH_SIZE counts 64 B per 128-byte request for the
16 B/lane reads this kernel issues (MI355X_MICROARCH.md, HBM / rocprofv3 section); WRITE_SIZE is used as reported."""
import json
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 30
lines = open(src).read().split("\n")
# pass 1: the MSDA forward kernel with the most dispatches (the fused fp32 form of the pixel decoder)
name, disp = None, 0
for line in lines:
    m = re.match(r"(msda_fwd_tile\w*<.*>)\s+dispatches=(\d+)", line)
    if m and int(m.group(2)) > disp:
        name, disp = m.group(1), int(m.group(2))
# pass 2: its counters (one block per PMC group)
vals, on = {}, False
for line in lines:
    if not line.startswith(" "):
        on = name is not None and line.startswith(name)
        continue
    if on:
        m = re.match(r"\s+(\S+)\s+mean=(\S+)", line)
        if m:
            vals[m.group(1)] = float(m.group(2))
fetch, write = vals["FETCH_SIZE"], vals["WRITE_SIZE"]
out = {
    "kernel": name, "source": f"{src} (rocprofv3 --pmc, separate passes for FETCH_SIZE and WRITE_SIZE)",
    "dispatches": disp, "frames_per_launch": frames, "FETCH_SIZE_KB_mean": fetch, "WRITE_SIZE_KB_mean": write,
    "fetch_correction": 2.0,
    "note": "gfx950: FETCH_SIZE counts 64 B per 128-B request for wide (16 B/lane) reads -> x2 "
            "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE uncorrected",
    "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0,
    "l2_hit_rate": vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"]) if "TCC_HIT_sum" in vals else None,
    "ta_busy_frac": (vals["TA_TA_BUSY_sum"] / 256.0) / (vals["GRBM_GUI_ACTIVE"] / 8.0)
    if "TA_TA_BUSY_sum" in vals and "GRBM_GUI_ACTIVE" in vals else None,
}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
