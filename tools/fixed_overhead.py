"""What the timed region of bench.py costs besides its K steps: elapsed(K) = a + b K over K = 5 ... 320, host waiting by spinning and
by yielding.  `a` is what the driver's 20-step form pays on top of 20 x (K1 + K2): the first launch on an idle queue + the wake-up
after the last kernel.   python tools/fixed_overhead.py"""
import json, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
Ks = (5, 10, 20, 40, 80, 160, 320)
for wait in ("spin", "yield"):
    pts = []
    for K in Ks:
        best = None
        for rep in range(3):
            out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", str(K), "--warmup", "5", "--no-cpu-baseline",
                                  "--no-e2e", "--no-events", "--host-wait", wait], capture_output=True, text=True, cwd=REPO).stdout
            d = json.loads(out.strip().split("\n")[-1])
            el = d["ms_per_step"] * K * 1e3
            best = el if best is None else min(best, el)
        pts.append((K, best))
    n = len(pts); sx = sum(k for k, _ in pts); sy = sum(e for _, e in pts); sxx = sum(k * k for k, _ in pts); sxy = sum(k * e for k, e in pts)
    b = (n * sxy - sx * sy) / (n * sxx - sx * sx); a = (sy - b * sx) / n
    print(f"host wait {wait:5s}: elapsed(K) = {a:6.1f} us + {b:6.2f} us x K   " + "  ".join(f"K={k}: {e:.0f} us ({e / k:.2f}/step)" for k, e in pts), flush=True)
