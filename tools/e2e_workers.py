"""e2e leg of bench.py (AtariVecEnv.step with the native runner, resets inside the timed steps) for several emulator worker
counts: which thread count a job with a CPU quota should use.  usage: python tools/e2e_workers.py [16 32 64 ...]"""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "active-gym_amd")]
import torch
import bench
hc = bench.host_cores()
print("host cores", hc, flush=True)
for w in [int(a) for a in sys.argv[1:]] or [hc["usable"], 2 * hc["usable"], 4 * hc["usable"]]:
    e = bench.run_e2e(torch.device("cuda:0"), 1024, dict(hc, usable=w, present=max(w, hc["present"])))
    print(w, json.dumps({k: ({kk: round(vv, 4) if isinstance(vv, float) else vv for kk, vv in v.items()} if isinstance(v, dict) else v)
                         for k, v in e.items() if k in ("rgb", "gray", "workers", "h2d_GBps")}), flush=True)
