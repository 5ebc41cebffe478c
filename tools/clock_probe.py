#!/usr/bin/env python3
"""What this box offers for reading the shader clock the chip actually holds (dev tool): sysfs pp_dpm_sclk, rocm-smi, amdsmi."""
import glob
import os
import subprocess
import sys

import torch

p = torch.cuda.get_device_properties(0)
print("torch props:", {k: getattr(p, k) for k in dir(p) if k.startswith("pci") or k in ("name", "clock_rate", "gcnArchName", "multi_processor_count")})
for f in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")):
    try:
        ue = open(os.path.join(os.path.dirname(f), "uevent")).read()
        slot = [l for l in ue.splitlines() if l.startswith("PCI_SLOT_NAME")]
        print(f, slot, repr(open(f).read()))
    except Exception as e:
        print(f, "unreadable:", e)
for cmd in (["rocm-smi", "--showclocks"], ["amd-smi", "metric", "--clock"]):
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=20)
        print(" ".join(cmd), "rc", out.returncode, out.stdout[-1500:], out.stderr[-300:])
    except Exception as e:
        print(cmd, "failed:", e)
try:
    import amdsmi
    print("amdsmi importable", amdsmi.__file__)
except Exception as e:
    print("amdsmi:", e)
try:
    print("torch.cuda.clock_rate:", torch.cuda.clock_rate(0))
except Exception as e:
    print("torch.cuda.clock_rate:", e)
