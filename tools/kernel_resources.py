"""Per-kernel register / scratch / LDS / occupancy table of one csrc/*.hip file, from hipcc's
-Rpass-analysis=kernel-resource-usage remarks (cross-compiles, no GPU needed).

    python tools/kernel_resources.py merkle.hip [name-filter]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "tap-stark_amd", "csrc")


def resources(src):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-x", "hip", "-O3", "-std=c++17", "-fPIC", "-I", CSRC,
           "-c", os.path.join(CSRC, src), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    out, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            name = t.split(":", 1)[1].strip()
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            cur = {"name": re.sub(r"\(.*", "", dem) or name}
            out.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    return out


if __name__ == "__main__":
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'LDS':>7s} {'occ':>4s}")
    for r in resources(sys.argv[1]):
        if flt in r["name"]:
            print(f"{r['name'][:70]:70s} {r.get('VGPRs', '?'):>5s} {r.get('AGPRs', '?'):>5s} {r.get('SGPRs', '?'):>5s} "
                  f"{r.get('ScratchSize [bytes/lane]', '?'):>8s} {r.get('LDS Size [bytes/block]', '?'):>7s} "
                  f"{r.get('Occupancy [waves/SIMD]', '?'):>4s}")
