#!/usr/bin/env python3
"""Diagnostics: skeleton of one kernel's gfx950 ISA — per basic block the counts of MFMA / LDS / VMEM / LDS-DMA instructions and
every s_waitcnt / s_barrier in order (spots a compiler-inserted `s_waitcnt vmcnt(0)` inside a counted-wait pipeline).
    hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only -Iinclude -I<csrc> <file>.hip -o /tmp/k.s
    python tools/isa_skeleton.py /tmp/k.s wn_layer_dgrad_kernel [min_mfma_per_block]"""
import re, sys
path, kern = sys.argv[1], sys.argv[2]
min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 0
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + kern + r"\w*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.end_amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))
blocks, cur = [], ["entry", []]
for l in lines[start + 1:end]:
    if re.match(r"^\.LBB\w+:", l):
        blocks.append(cur); cur = [l.split(":")[0] + " " + l.split(";")[-1].strip() if ";" in l else l, []]
    elif l.startswith("\t") and not l.startswith("\t."):
        cur[1].append(l.strip())
blocks.append(cur)
for name, ins in blocks:
    n_mfma = sum(i.startswith("v_mfma") for i in ins)
    if n_mfma < min_mfma:
        continue
    out, cnt = [], {}
    def flush():
        if cnt:
            out.append("  [" + " ".join(f"{k}×{v}" for k, v in cnt.items()) + "]"); cnt.clear()
    for i in ins:
        op = i.split()[0]
        if op in ("s_waitcnt", "s_barrier") or op.startswith("s_cbranch") or op == "s_branch":
            flush(); out.append("  " + i.split(";")[0].strip())
        else:
            key = ("mfma" if op.startswith("v_mfma") else "dma" if ("lds" in i and op.startswith("global_load")) else
                   "ds_read" if op.startswith("ds_read") else "ds_write" if op.startswith("ds_write") else
                   "vmem_ld" if op.startswith(("global_load", "buffer_load")) else
                   "vmem_st" if op.startswith(("global_store", "buffer_store", "global_atomic")) else
                   "scratch" if op.startswith("scratch") else "smem" if op.startswith("s_load") else None)
            if key:
                cnt[key] = cnt.get(key, 0) + 1
    flush()
    print(f"{name}: {len(ins)} instr, {n_mfma} mfma")
    print("\n".join(out))
