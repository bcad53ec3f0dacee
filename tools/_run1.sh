cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stream_golden or squeeze_runs" 2>&1 | tail -2
for kt in 0 1; do
echo "== kernel timing $kt"
ZOPFLI_AMD_KERNEL_TIMING=$kt python - <<'PY'
import os, time, sys, concurrent.futures as cf
sys.path.insert(0, '.')
from zopfli_amd import ZopfliOptions, api, generate
lib = api.library()
lib.zmx_set_kernel_timing(int(os.environ["ZOPFLI_AMD_KERNEL_TIMING"]))
opts = ZopfliOptions(15, 1, 15)
for size, count in ((65536, 600), (1000000, 120)):
    files = [generate("TX"[i & 1], size, seed=1000 + i) for i in range(count)]
    one = lambda d: api.compress(d, api.FORMAT_GZIP, opts, lib=lib)
    for f in files[:3]: one(f)
    for k in (1, 16):
        with cf.ThreadPoolExecutor(k) as ex:
            list(ex.map(one, files[:2 * k]))
            t0 = time.perf_counter(); n = count if k > 1 else count // 6
            list(ex.map(one, files[:n])); dt = time.perf_counter() - t0
        print(size, "callers", k, round(n * size / 1e6 / dt, 2), "MB/s", round(dt / n * 1e3, 2), "ms per file")
PY
done
