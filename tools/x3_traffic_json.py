"""profiles/<tag>_pmc.txt (tools/prof.sh summaries of rocprofv3 --pmc passes over the BENCH command) -> profiles/rNN_x3_traffic.json:
HBM bytes per launch of the split-f16 kernel families, averaged over exactly the launch mix bench.py's roofline_conv_x3 /
roofline_ffn entries average over (every launch of the family in the timed clips).
    python tools/x3_traffic_json.py gpurun_out/prof/r06_benchpmc_pmc.txt profiles/r06_x3_traffic.json
HBM bytes = 2 * FETCH_SIZE + WRITE_SIZE (KB counters): gfx950's FETCH_SIZE counts 64 B per 128-byte request of the 16 B / lane reads
these kernels issue (MI355X_MICROARCH.md, HBM / rocprofv3 section; same correction as tools/traffic_json.py); WRITE_SIZE as reported."""
import json
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
FAMILIES = {"conv1x1_x3_kernel": r"conv1x1_x3_kernel<", "x3_ffn_kernel": r"x3_ffn_kernel", "x3_linear": r"x3_linear(_stream)?_kernel<",
            "x3_tile_kernel": r"x3_tile_kernel<", "msda_fwd": r"msda_fwd_tile", "mask_gemm_kernel": r"mask_gemm_kernel<",
            "attn_keysplit_kernel": r"attn_keysplit_kernel", "bneck_chain_kernel": r"bneck_chain_kernel<"}
blocks, cur = [], None
for line in open(src).read().split("\n"):
    m = re.match(r"(\S.*?)\s+dispatches=(\d+)", line)
    if m:
        cur = {"name": m.group(1), "n": int(m.group(2)), "c": {}}
        blocks.append(cur)
        continue
    m = re.match(r"\s+(\S+)\s+mean=(\S+)\s+sum=(\S+)", line)
    if m and cur is not None:
        cur["c"][m.group(1)] = float(m.group(3))
out = {}
for fam, pat in FAMILIES.items():
    f = w = 0.0
    nf = nw = 0
    names = set()
    for b in blocks:
        if re.match(pat, b["name"]):
            names.add(b["name"])
            if "FETCH_SIZE" in b["c"]:
                f += b["c"]["FETCH_SIZE"]
                nf += b["n"]
            if "WRITE_SIZE" in b["c"]:
                w += b["c"]["WRITE_SIZE"]
                nw += b["n"]
    if nf and nw:
        out[fam] = {"kernels": sorted(names), "dispatches": nf, "FETCH_SIZE_KB_sum": f, "WRITE_SIZE_KB_sum": w, "fetch_correction": 2.0,
                    "hbm_bytes_per_launch": round((2.0 * f / nf + w / nw) * 1024.0),
                    "source": f"{dst} <- {src}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over the bench command; "
                              "2 * FETCH_SIZE + WRITE_SIZE, mean over every launch of the family"}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
