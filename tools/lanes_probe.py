"""How much host/device overlap several contexts on ONE device buy: ZopfliCompress of the 100 MB class-T workload with
ZOPFLI_AMD_DEVICES = 0 / 0,0 / 0,0,0 (one process each; includes the H2D copy and the CRC, unlike bench.py)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = (
    "import sys, time, hashlib\n"
    "sys.path.insert(0, %r)\n"
    "from zopfli_amd import ZopfliOptions, api, generate\n"
    "data = generate('T', 100000000)\n"
    "opt = ZopfliOptions(15, 0, 15)\n"
    "api.compress(data[:8000000], 0, opt)\n"
    "best = 1e9\n"
    "for i in range(3):\n"
    "    t = time.perf_counter(); out = api.compress(data, 0, opt); best = min(best, time.perf_counter() - t)\n"
    "print('%%.1f ms  %%.1f MB/s  %%s' %% (best * 1e3, 100.0 / best, hashlib.sha256(out).hexdigest()[:16]))\n" % ROOT)
for devs in sys.argv[1:] or ["0", "0,0", "0,0,0"]:
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ZOPFLI_AMD_DEVICES=devs), capture_output=True, text=True)
    print(devs, r.stdout.strip(), r.stderr.strip()[-300:] if r.returncode else "")
