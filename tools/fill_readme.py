"""One-shot: fills the @...@ placeholders of README.md's first-screen table from the round's bench records
under profiles/ (python tools/fill_readme.py r06)."""
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda name: json.load(open(os.path.join(root, "profiles", f"{tag}_{name}")))
d = P("bench_config3_k20.json")
cells = float((1 << 20) * 64)
w = d["windows_ms_per_step"]
ps = d["clocks"]["prover_sustained"]
rw = d["roofline_whole"]
lat = d["single_proof_latency_ms"]
h = d["h2d_inclusive"]
cb = d["cpu_baseline"]
cpu_s = sorted(cb["runs_s"])[1]
sub = {
    "W1": f"{w[0]:.3f}", "WINS": " / ".join(f"{x:.3f}" for x in w), "SUST": f"{ps['ms_per_step']:.3f}",
    "PPS": f"{1e3 / w[0]:.0f}", "CPS": f"{cells / (w[0] * 1e-3):.3g}".replace("e+10", "×10¹⁰"),
    "WHOLE": f"{rw['alg_bytes_per_proof'] / (w[0] * 1e-3) / 8e12:.2f}", "DOM": f"{d['roofline']['frac']:.2f}",
    "TRAF": f"{d['roofline']['traffic'] / d['roofline']['alg_bytes_per_launch']:.3f}" if d["roofline"].get("traffic") else "n/a",
    "NTT": f"{d['roofline_ntt']['frac']:.2f}" if d.get("roofline_ntt") else "n/a",
    "VALU": f"{d['valu_issue']['measured_clock']['ms_per_proof_at_ceiling'] / w[0]:.2f}",
    "MHZ": f"{ps['gfxclk_mhz_median']:.0f}", "WATT": f"{ps['socket_power_w_median']}",
    "LAT": f"{lat:.2f}", "LATPPS": f"{1e3 / lat:.0f}", "LATWHOLE": f"{rw['alg_bytes_per_proof'] / (lat * 1e-3) / 8e12:.2f}",
    "H2D": f"{h['ms_per_step']:.2f}", "H2DGB": f"{h['h2d_GB_per_s']}", "H2DPPS": f"{1e3 / h['ms_per_step']:.0f}",
    "H2DCPS": f"{cells / (h['ms_per_step'] * 1e-3):.3g}".replace("e+10", "×10¹⁰"),
    "CPU": f"{cpu_s:.2f}", "CPUPPS": f"{1 / cpu_s:.3f}", "CPUCPS": f"{cells / cpu_s:.2g}".replace("e+06", "×10⁶"),
    "C2": f"{P('bench_config2.json')['ms_per_step']:.2f}", "C4": f"{P('bench_config4.json')['ms_per_step']:.1f}",
    "C5": f"{P('bench_config5.json')['ms_per_step']:.1f}",
}
path = os.path.join(root, "README.md")
s = open(path).read()
for k, v in sub.items():
    s = s.replace(f"@{k}@", v)
left = [t for t in s.split("@")[1::2] if t.isupper() or t.replace("2", "").isalnum() and t.isupper()]
open(path, "w").write(s)
print(sub)
