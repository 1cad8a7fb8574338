"""Per-loop instruction mix and s_waitcnt pattern of the kernels of one translation unit (device ISA of a .hip file):
where a loop waits with lgkmcnt(0) / vmcnt(0) between every few MFMAs, the wave exposes one LDS / memory latency per wait.

    python tools/micro/isa_waits.py space_time_pde_amd/csrc/jet_wgrad_s31.hip k_wgrad_quad [min_mfma]
"""
import os, re, subprocess, sys, tempfile

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics"]


def main():
    src, pat = sys.argv[1], sys.argv[2]
    min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    out = os.path.join(tempfile.gettempdir(), os.path.basename(src) + ".s")
    subprocess.run(["hipcc"] + FLAGS + ["-S", "--cuda-device-only", "-o", out, src], check=True, stderr=subprocess.DEVNULL)
    text = open(out).read().split("\n")
    name, start = None, 0
    for i, l in enumerate(text + ["_Zend:"]):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            if name and pat in name:
                report(name, text[start:i], min_mfma)
            name, start = m.group(1), i


def kind(op):
    if "mfma" in op:
        return "mfma"
    if re.match(r"v_(exp|log|rcp|rsq|sqrt)", op):
        return "trans"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "scratch_")):
        return "vmem"
    if op == "s_waitcnt":
        return "wait"
    return "other"


def report(name, body, min_mfma):
    demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    # basic blocks that are loop bodies: from a label with "Loop Header" to the branch back; approximated by label-to-label
    labels = [i for i, l in enumerate(body) if re.match(r"^\.LBB", l)] + [len(body)]
    printed = False
    for a, b in zip(labels[:-1], labels[1:]):
        ops = [re.match(r"\s+([a-z_0-9]+)(.*)", l) for l in body[a:b]]
        ops = [(m.group(1), m.group(2)) for m in ops if m]
        cnt = {}
        for op, _ in ops:
            cnt[kind(op)] = cnt.get(kind(op), 0) + 1
        if cnt.get("mfma", 0) < min_mfma:
            continue
        if not printed:
            print("==", demangled[:150])
            printed = True
        waits = [arg.strip() for op, arg in ops if op == "s_waitcnt"]
        hard = sum(1 for w in waits if "lgkmcnt(0)" in w or "vmcnt(0)" in w)
        print("   block %s: %s   waits: %d (%d of them to zero)" % (body[a].split(":")[0], cnt, len(waits), hard))


main()
