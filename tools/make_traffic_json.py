"""Turn the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE summaries (tools/rocprof_summary.py pmc ...) into
profiles/pmc_traffic.json: HBM bytes per launch of the first-layer kernels, keyed by bench.py's kernel names.

FETCH_SIZE is doubled (MI355X_MICROARCH.md: on gfx950 it reports half the bytes of wide coalesced 16 B/lane reads);
WRITE_SIZE is taken as reported.  Usage: make_traffic_json.py fetch.txt write.txt chunk act out.json
"""
import json
import os
import re
import sys


def parse(path):
    out = {}
    for ln in open(path).read().split("\n")[2:]:
        m = re.match(r"(k_\w+(?:<[^>]*>|\([^)]*\)))\s+(\d+)\s+([\d.]+)\s+([\d.]+)", ln.strip())
        if m:
            out[m.group(1)] = float(m.group(4)) * 1024.0      # KiB per call -> bytes
    return out


def bench_name(sym):
    if sym.startswith("k_gather(") or sym.startswith("k_gather_tile("):
        return "gather"
    if sym.startswith("k_fc1_fwd_spec<3,"):
        return "layer1_fwd"
    if sym.startswith("k_fc1_bwd_fused<"):
        return "layer1_bwd"
    if sym.startswith("k_fc2_fwd_bf<"):
        return "layer2_fwd"
    if re.match(r"k_\w+<0,\d,", sym):      # value-only / value-tile kernels of bench.py's inference side figure, not the training step
        return None
    m = re.match(r"k_layer_coop<(\d+),(\d+),(\d+),(\d+),(\d+),(-?\d+),(\d+)(?:,(\w+))?(?:,[\w,]+)?>", sym)
    if m and (m.group(8) in (None, "false") or BF16):
        pro, epi = int(m.group(4)), int(m.group(5))
        if pro == 2 and epi == 0:
            return "layer1_fwd"
        if epi == 2:
            return "layer1_dgrad"
        return "layer2_fwd" if epi == 0 else "layer2_dgrad"
    m = re.match(r"k_layer_coop2<(\d+),(\d+),(\d+),(\d+),(-?\d+)>", sym)
    if m:
        return {2: "layer1_dgrad", 1: "layer2_dgrad"}.get(int(m.group(4)))
    m = re.match(r"k_wgrad_coop<(\d+),(\d+),(\d+),(-?\d+),(\d+),(\w+?)(?:,(\w+?))?(?:,[\d,]+)?>", sym)
    if m and int(m.group(3)) == 1 and m.group(6) == "false" and (m.group(7) in (None, "false") or BF16):
        return "layer1_wgrad"
    if m and int(m.group(3)) == 0 and int(m.group(5)) == 4 and m.group(6) == "false" and (m.group(7) in (None, "false") or BF16):
        return "layer2_wgrad"
    return None


BF16 = False     # True (6th argument "bf16"): accept the bf16-operand instantiations (configs[3] profile)


if __name__ == "__main__":
    BF16 = len(sys.argv) > 6 and sys.argv[6] == "bf16"
    fetch, write = parse(sys.argv[1]), parse(sys.argv[2])
    kernels = {}
    for sym in set(fetch) | set(write):
        name = bench_name(sym)
        if name:
            f, w = 2.0 * fetch.get(sym, 0.0), write.get(sym, 0.0)
            kernels[name] = dict(symbol=sym, fetch_bytes_corrected=f, write_bytes=w, hbm_bytes_per_launch=f + w)
    json.dump(dict(chunk=int(sys.argv[3]), act=sys.argv[4], kernels=kernels,
                   source="rocprofv3 --pmc FETCH_SIZE (x2, gfx950 correction) + --pmc WRITE_SIZE, separate passes, profiles/%s / %s"
                          % (os.path.basename(sys.argv[1]).replace("FETCH_SIZE", "fetch_size"),
                             os.path.basename(sys.argv[2]).replace("WRITE_SIZE", "write_size"))), open(sys.argv[5], "w"), indent=1)
