"""Does HBM traffic cost VALU clock?  The library's register-resident butterfly loop (ts_bench_alu kind 0:
no memory traffic, the whole chip) is timed alone and beside a device-to-device copy stream of varying
intensity running on another HIP stream.  A copy kernel issues few VALU instructions, so if the ALU
loop slows by much more than the copy's share of the issue slots, the rest is clock (DVFS)."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tapstark_amd as ts

ctx = ts.default_context()
n = 1 << 28  # 1 GiB of int32
a = torch.empty(n, dtype=torch.int32, device="cuda:0")
b = torch.empty(n, dtype=torch.int32, device="cuda:0")
side = torch.cuda.Stream()


def alu(reps=6):
    return sorted(ctx.alu_ceiling(0) for _ in range(reps))[reps // 2]


base = alu()
print(f"ALU loop alone: {base:.4g} butterflies/s")
for frac in (1.0, 0.5, 0.25, 0.1):
    stop = threading.Event()
    moved = [0, 0.0]

    def copier():
        m = int(n * frac) if frac < 1 else n
        t0 = time.perf_counter()
        with torch.cuda.stream(side):
            while not stop.is_set():
                b[:m].copy_(a[:m], non_blocking=True)
                if frac < 1:  # duty cycle: idle for the rest of the period
                    torch.cuda._sleep(int(2.0e9 * (2 * 4 * m / 5e12) * (1 / frac - 1) * 0.5))
                moved[0] += 2 * 4 * m
                side.synchronize()
        moved[1] = time.perf_counter() - t0

    th = threading.Thread(target=copier)
    th.start()
    time.sleep(0.3)
    r = alu()
    stop.set()
    th.join()
    print(f"copy duty {frac:4.2f}: copy stream {moved[0] / moved[1] / 1e12:5.2f} TB/s (read+write)  "
          f"ALU loop {r:.4g} butterflies/s = {r / base:.3f} of alone")
