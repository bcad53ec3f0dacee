"""Per-kernel register / LDS / occupancy table of the device code object (hipcc -Rpass-analysis=kernel-resource-usage).
    python tools/resource_usage.py [-DFLAG ...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = os.path.join(ROOT, "zopfli_amd", "csrc", "device")
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
       "-I" + os.path.join(ROOT, "zopfli_amd", "csrc", "host"), "-I" + dev, "--cuda-device-only", "-c",
       os.path.join(dev, "zmx_hip.hip"), "-o", "/tmp/zmx_ru.o", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[1:]
txt = subprocess.run(cmd, capture_output=True, text=True).stderr
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split("\n")[0].strip()

    def g(k):
        m = re.search(k + r": (\d+)", b)
        return m.group(1) if m else "?"
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dn = re.sub(r"\(.*", "", dn).replace("void ", "")[:44]
    print("%-44s VGPR %4s AGPR %3s SGPR %3s spill %3s scratch %4s occ %2s LDS %6s" % (
        dn, g("VGPRs"), g("AGPRs"), g("SGPRs"), g("VGPRs Spill"), g(r"ScratchSize \[bytes/lane\]"),
        g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
