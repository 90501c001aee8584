#!/usr/bin/env python
"""Static audit of the compiled kernels for loads that wait one at a time (runs in the build container: hipcc cross-compiles, no GPU).

    python tools/isa_audit.py [file.hip ...]          default: every .hip under vision-zephyr_amd/csrc

For each kernel of each file: vector-memory loads, `s_waitcnt vmcnt(0)` (full drains), branches, MFMAs.  A kernel whose full drains are
about as many as its loads issues one request, waits for it, and only then issues the next - the pattern that cost the norm kernels
2 us per call, the argmax tail 6 us per token and the training transposes 37 ms per step before round 2's third session (a per-chunk
bounds test compiles to an exec branch per chunk with its own wait; a `for (k = lane; k < n; k += 64) use(x[k])` loop to one load per
trip).  The listing is a pointer to read the ISA (`hipcc -S --cuda-device-only`), not a verdict: hand-scheduled kernels drain on
purpose at the end of a burst."""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "vision-zephyr_amd", "csrc")


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout
        return out.strip().split("\n")
    except Exception:      # noqa: BLE001
        return names


def audit(path):
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "k.s")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-S", "--cuda-device-only", path, "-o", asm], check=True,
                       stderr=subprocess.DEVNULL, cwd=os.path.dirname(path))
        txt = open(asm).read()
    rows = []
    for fn in re.split(r"\n(?=_Z\w+:)", txt):
        m = re.match(r"(_Z\w+):", fn)
        if not m or "s_endpgm" not in fn:
            continue
        lines = [ln.strip() for ln in fn.split("s_endpgm")[0].split("\n")]
        loads = sum(ln.startswith(("global_load", "buffer_load", "flat_load")) and "lds" not in ln.split()[0] for ln in lines)
        dma = sum(ln.startswith(("global_load_lds", "buffer_load")) and "lds" in ln for ln in lines)
        drains = sum(ln.startswith("s_waitcnt") and "vmcnt(0)" in ln for ln in lines)
        rows.append((m.group(1), loads, dma, drains, sum(ln.startswith("s_cbranch") for ln in lines), sum(ln.startswith("v_mfma") for ln in lines)))
    names = demangle([r[0] for r in rows])
    return [(n,) + r[1:] for n, r in zip(names, rows)]


def main():
    files = sys.argv[1:] or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    for f in files:
        print(f"== {os.path.relpath(f, REPO)}")
        for name, loads, dma, drains, br, mfma in audit(os.path.abspath(f)):
            name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0][:70]
            flag = "  <-- one load per wait?" if drains >= 4 and 2 * drains >= loads + dma and mfma == 0 else ""
            print(f"   {name:70s} loads {loads:4d}  lds-dma {dma:4d}  vmcnt(0) {drains:4d}  branches {br:4d}  mfma {mfma:4d}{flag}")


if __name__ == "__main__":
    main()
