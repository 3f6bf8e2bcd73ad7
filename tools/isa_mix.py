"""Instruction mix of one kernel's basic blocks from a hipcc -save-temps .s file (experiments aid).

    python tools/isa_mix.py file.s <kernel-name-substring> [min_block_len]
"""
import collections, re, sys

lines = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 80
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and key in l and l.rstrip().split(":")[0].endswith(l.split(":")[0]) and ":" in l)
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = [l.strip() for l in lines[start + 1:end] if l.strip() and not l.strip().startswith((";", ".s", ".p", ".c", ".a"))]
blocks, cur = collections.OrderedDict(), "entry"
blocks[cur] = []
for l in body:
    if re.match(r"^\.?LBB\d+_\d+:", l):
        cur = l
        blocks[cur] = []
    else:
        blocks[cur].append(l)
print(key, "instructions:", sum(len(b) for b in blocks.values()))
for b, ins in blocks.items():
    if len(ins) < minlen:
        continue
    c = collections.Counter()
    for i in ins:
        op = i.split()[0]
        if op.startswith("v_mfma"): c["mfma"] += 1
        elif op.startswith("ds_"): c["lds"] += 1; c["lds:" + op] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): c["vmem"] += 1; c["vmem:" + op] += 1
        elif op.startswith("v_"): c["valu"] += 1; c["valu:" + op] += 1
        elif op.startswith("s_nop"): c["s_nop"] += 1
        elif op.startswith("s_waitcnt"): c["s_waitcnt"] += 1
        elif op.startswith("s_"): c["salu"] += 1
    print(" block", b, len(ins), {k: v for k, v in c.items() if ":" not in k})
    print("   valu:", sorted([(v, k[5:]) for k, v in c.items() if k.startswith("valu:")], reverse=True)[:16])
    print("   lds :", sorted([(v, k[4:]) for k, v in c.items() if k.startswith("lds:")], reverse=True))
    print("   vmem:", sorted([(v, k[5:]) for k, v in c.items() if k.startswith("vmem:")], reverse=True))
for i in range(end, min(end + 200, len(lines))):
    for k in ("next_free_vgpr", "group_segment_fixed_size", "private_segment_fixed_size"):
        if ".amdhsa_" + k in lines[i]:
            print("  ", lines[i].strip())
