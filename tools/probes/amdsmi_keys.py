"""What amdsmi reports on this box (keys of the clock / power dictionaries), for bench.py's clock sampler."""
import amdsmi
amdsmi.amdsmi_init()
hs = amdsmi.amdsmi_get_processor_handles()
print("handles", len(hs))
h = hs[0]
for t in ("GFX", "SYS", "MEM"):
    try:
        print(t, amdsmi.amdsmi_get_clock_info(h, getattr(amdsmi.AmdSmiClkType, t)))
    except Exception as e:  # noqa: BLE001
        print(t, "failed:", e)
try:
    print("power", amdsmi.amdsmi_get_power_info(h))
except Exception as e:  # noqa: BLE001
    print("power failed:", e)
try:
    print("cap", amdsmi.amdsmi_get_power_cap_info(h))
except Exception as e:  # noqa: BLE001
    print("cap failed:", e)
amdsmi.amdsmi_shut_down()
