"""Builds a variant of libzopfli_amd.so with extra -D flags into tools/_build/ (A/B of compile-time switches on the GPU box):
    python tools/build_variant.py NAME -DFOO=1 ...   ->   tools/_build/libzopfli_amd_NAME.so
A script selects it with ZOPFLI_AMD_LIB=<path> (zopfli_amd/api.py)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zopfli_amd import _build  # noqa: E402

name, flags = sys.argv[1], sys.argv[2:]
hip, cc, hdr = _build._sources()
csrc = _build.CSRC
out = os.path.join(ROOT, "tools", "_build", f"libzopfli_amd_{name}.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fvisibility=hidden",
                       "-Wl,--version-script=" + os.path.join(csrc, "libzopfli_amd.map"), "-I" + os.path.join(ROOT, "include"),
                       "-I" + os.path.join(csrc, "host"), "-I" + os.path.join(csrc, "device")] + flags + [hip] + cc +
                      ["-o", out, "-lpthread", "-ldl"], stderr=subprocess.DEVNULL)
print(out)
