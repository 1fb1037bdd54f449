"""Per-kernel ISA census of drn_amd/csrc/*.hip for gfx950 (cross-compiled, no GPU needed): instructions, VGPRs, scratch bytes and
the signatures of two accidents round 3 found: a run-time index into a register array (v_cmp_eq + v_cndmask chains: 3000 such
instructions = 10 us in the query attention kernels) and wave-uniform operands fetched by dozens of scalar loads that the compiler
serialises behind branches and waits (the heads backward kernel: 12 of its 37 us).  usage: python scripts/isa_scan.py [file.hip ...]"""
import collections
import glob
import os
import re
import subprocess
import sys
import tempfile

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
files = [os.path.abspath(a) for a in sys.argv[1:]] or sorted(glob.glob(os.path.join(root, "drn_amd", "csrc", "*.hip")))
print("%-86s %6s %5s %7s %7s %7s %6s %6s %6s" % ("kernel", "instr", "vgpr", "scratch", "cndmask", "cmp_eq", "s_load", "waits", "branch"))
for f in files:
    with tempfile.TemporaryDirectory() as td:
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", f, "-o", os.path.join(td, "o.o"),
                        "-save-temps"], cwd=td, stderr=subprocess.DEVNULL, check=True)
        asm = open(glob.glob(os.path.join(td, "*gfx950.s"))[0]).read()
    meta = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", asm, re.S):
        body = m.group(2)
        g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, body).group(1)) if re.search(r"\.%s:\s+(\d+)" % k, body) else -1
        meta[m.group(1)] = (g("vgpr_count"), g("private_segment_fixed_size"))
    for name in meta:
        m = re.search(r"^%s:[^\n]*\n(.*?)s_endpgm" % re.escape(name), asm, re.S | re.M)
        if not m:
            continue
        ops = collections.Counter(l.split()[0] for l in m.group(1).splitlines() if l.strip() and not l.strip().startswith((";", ".")) and not l.strip().endswith(":"))
        try:
            dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
        except OSError:
            dem = name
        dem = re.sub(r"\(.*", "", dem)
        cnd = sum(v for k, v in ops.items() if k.startswith("v_cndmask"))
        ceq = sum(v for k, v in ops.items() if k.startswith("v_cmp_eq_u32") or k.startswith("v_cmp_ne_u32"))
        flag = "  <-- select chain?" if ceq > 200 and cnd > 200 else ("  <-- scratch" if meta[name][1] > 0 else "")
        sl = sum(v for k, v in ops.items() if k.startswith("s_load") or k.startswith("s_buffer_load"))
        wt = ops.get("s_waitcnt", 0)
        br = sum(v for k, v in ops.items() if k.startswith("s_cbranch"))
        if not flag and sl > 60 and br > 60:
            flag = "  <-- serial scalar loads?"       # dozens of s_load + wait + branch: serial trips through a cold scalar cache
        print("%-86s %6d %5d %7d %7d %7d %6d %6d %6d%s" % ((os.path.basename(f) + ":" + dem)[:86], sum(ops.values()), meta[name][0], meta[name][1], cnd,
                                                       ceq, sl, wt, br, flag))
