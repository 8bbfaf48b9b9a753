"""Instruction mix per chunk step of the fused SDF kernel, from the disassembly of the library that ships (the steps are the stretches
between two s_barrier): what profiles/r05_sdf_step_issue_table.md quotes.
usage: step_mix.py [mangled-name substring, default the mode-1 inference kernel]"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LLVM = "/opt/rocm/lib/llvm/bin"
key = sys.argv[1] if len(sys.argv) > 1 else "sdf_fwd2_kernelILi1ELb0E"
tmp = tempfile.mkdtemp()
lib = os.path.join(tmp, "lib.so")
import shutil
shutil.copy(os.path.join(ROOT, "vdn-nerf_amd", "vdn_hip", "libvdn_render.so"), lib)
subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", lib], check=True, capture_output=True)
for f in sorted(os.listdir(tmp)):
    if "amdgcn" not in f:
        continue
    syms = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-s", "-W", os.path.join(tmp, f)], capture_output=True, text=True).stdout
    names = [l.split()[7] for l in syms.splitlines() if len(l.split()) >= 8 and l.split()[3] == "FUNC" and key in l.split()[7]]
    if not names:
        continue
    dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--disassemble-symbols=" + names[0], os.path.join(tmp, f)], capture_output=True, text=True).stdout
    lines = [l.split("//")[0].strip() for l in dis.splitlines() if l.startswith("\t")]
    bars = [i for i, l in enumerate(lines) if l.startswith("s_barrier")]
    kinds, agg = collections.Counter(), {}
    for a, b in zip(bars, bars[1:]):
        h = collections.Counter(l.split()[0] for l in lines[a:b])
        k = "hidden layer" if h["v_exp_f32_e32"] >= 16 and h["v_rcp_f32_e32"] >= 16 else ("gradient sweep" if any("ubyte" in x for x in h) else "other")
        kinds[k] += 1
        agg.setdefault(k, collections.Counter()).update(h)
    print(names[0][:80], ":", len(lines), "instructions,", len(bars), "chunk steps")
    for k, n in kinds.items():
        c = agg[k]
        trans = sum(v for x, v in c.items() if re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_f32", x))
        mfma = sum(v for x, v in c.items() if x.startswith("v_mfma"))
        valu = sum(v for x, v in c.items() if x.startswith("v_")) - trans - mfma
        salu = sum(v for x, v in c.items() if x.startswith("s_"))
        ds = sum(v for x, v in c.items() if x.startswith("ds_"))
        vm = sum(v for x, v in c.items() if x.startswith(("global_", "buffer_")))
        print("  %-15s %3d steps: per step %6.1f instructions = MFMA %.1f, transcendental %.1f, other VALU %.1f, LDS %.1f, vector memory %.1f, scalar %.1f (s_waitcnt %.1f, s_nop %.1f)"
              % (k, n, sum(c.values()) / n, mfma / n, trans / n, valu / n, ds / n, vm / n, salu / n, c["s_waitcnt"] / n, c["s_nop"] / n))
    break
