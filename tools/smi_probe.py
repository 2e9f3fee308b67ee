"""What the GPU box lets an ordinary user read about clocks and power (amdsmi / sysfs)."""
import glob, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    import amdsmi
    amdsmi.amdsmi_init()
    hs = amdsmi.amdsmi_get_processor_handles()
    print("amdsmi handles:", len(hs))
    h = hs[0]
    for name, fn in (("clock gfx", lambda: amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX)),
                     ("clock mem", lambda: amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.MEM)),
                     ("power", lambda: amdsmi.amdsmi_get_power_info(h)),
                     ("activity", lambda: amdsmi.amdsmi_get_gpu_activity(h)),
                     ("metrics", lambda: {k: v for k, v in amdsmi.amdsmi_get_gpu_metrics_info(h).items()
                                          if any(x in k for x in ("gfxclk", "socket_power", "temperature_hotspot", "throttle", "uclk"))}),
                     ("power cap", lambda: amdsmi.amdsmi_get_power_cap_info(h))):
        try:
            t0 = time.perf_counter()
            v = fn()
            print(f"{name}: {v}  ({1e3 * (time.perf_counter() - t0):.2f} ms)")
        except Exception as e:
            print(f"{name}: FAILED {e!r}")
except Exception as e:
    print("amdsmi unusable:", repr(e))
for f in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))[:2]:
    try:
        print(f, open(f).read().replace("\n", " | "))
    except Exception as e:
        print(f, "unreadable", e)
for f in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average"))[:2]:
    try:
        print(f, open(f).read().strip())
    except Exception as e:
        print(f, "unreadable", e)
