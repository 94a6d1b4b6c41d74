#!/usr/bin/env python3
"""tools/isa.py <kernel-substring> [--dump] - compile agx_api.hip with -save-temps (in /tmp/agx_isa) and print the
instruction histogram, register use and (with --dump) the ISA of every kernel whose mangled name contains the substring.
    python tools/isa.py k_ingest_full12
"""
import collections, os, re, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = "/tmp/agx_isa"
def main():
    pat = sys.argv[1]; dump = "--dump" in sys.argv; table = "--table" in sys.argv      # --table: one line per kernel (pat "" = all)
    extra = [a for a in sys.argv[2:] if a.startswith("-D")]
    os.makedirs(OUT, exist_ok=True)
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
                    "-I", REPO + "/include", "-I", REPO + "/active-gym_amd/csrc", "-DAGX_BUILD", *extra, "-save-temps",
                    REPO + "/active-gym_amd/csrc/agx_api.hip", "-o", OUT + "/x.so"], cwd=OUT, check=True,
                   stderr=subprocess.DEVNULL)
    s = open(OUT + "/agx_api-hip-amdgcn-amd-amdhsa-gfx950.s").read()
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n", s, flags=re.M):
        name = m.group(1)
        if pat not in name: continue
        k = s.find(".amdhsa_kernel " + name + "\n", m.end())
        if k < 0: continue
        body = s[m.end():s.rfind(".section", m.end(), k)]
        tail = s[k:s.index(".end_amdhsa_kernel", k)]
        ops = collections.Counter()
        for ln in body.splitlines():
            ln = ln.strip()
            if not ln or ln.startswith((";", ".", "//")) or ln.endswith(":"): continue
            ops[ln.split()[0]] += 1
        v = sum(c for o, c in ops.items() if o.startswith("v_")); sa = sum(c for o, c in ops.items() if o.startswith("s_"))
        lds = sum(c for o, c in ops.items() if o.startswith("ds_")); mem = sum(c for o, c in ops.items() if o.startswith(("global_", "buffer_", "flat_")))
        regs = {k: re.search(r"\.amdhsa_" + k + r"\s+(\d+)", tail) for k in ("next_free_vgpr", "next_free_sgpr", "group_segment_fixed_size", "private_segment_fixed_size")}
        if table:
            sp = re.search(r"\.amdhsa_private_segment_fixed_size\s+(\d+)", tail)
            spill = re.search(r"; SGPRSpill: (\d+)|sgpr_spill_count:\s*(\d+)", s[k:k + 4000])
            short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0][:92]
            print(f"| `{short}` | {regs['next_free_vgpr'].group(1)} | {regs['next_free_sgpr'].group(1)} | {sp.group(1) if sp else '?'} | {v} | {sa} | {lds} |")
            continue
        print(f"== {name}\n   static: VALU {v}  SALU {sa}  LDS {lds}  VMEM {mem}   " + "  ".join(f"{k}={r.group(1)}" for k, r in regs.items() if r))
        print("   " + "  ".join(f"{o}:{c}" for o, c in ops.most_common(45)))
        if dump: print(body)
if __name__ == "__main__":
    main()
