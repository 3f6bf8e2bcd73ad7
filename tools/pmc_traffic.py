"""Turn two rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE) of `python bench.py` into per-kernel HBM traffic.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcF -- python bench.py --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmcW -- python bench.py --no-cpu-baseline
    python tools/pmc_traffic.py gpurun_out/pmcF gpurun_out/pmcW > profiles/rN_traffic.json

Per MI355X_MICROARCH.md (HBM / rocprofv3): counters are in KiB; on gfx950 FETCH_SIZE reports exactly half
of the bytes of a wide coalesced streaming read, so reads are doubled; WRITE_SIZE is exact for 16-B streaming
stores.  Output: mean bytes per launch for each kernel, keyed by the names bench.py prints."""
import collections, csv, glob, json, os, re, sys


def short(n):
    """rocprofv3 kernel name -> the name bench.py prints (e.g. gcnx_bwd_kernel<3>, pgemm_tn_kernel<5>)."""
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", n)
    if m:                                   # Itanium mangling: <len><name>[I Li<k>E ... E]
        ln = int(m.group(1))
        rest = n[m.end():]
        name, tail = rest[:ln], rest[ln:]
        t = re.match(r"I((?:L[ib]\d+E)+)E", tail)      # <int..., bool X3>: bench.py prints X3 = false as ",f16"
        if t:
            args = re.findall(r"Li(\d+)E", t.group(1))
            if name.startswith("gcngi"):                        # <NT, X3, IO, NG, NM, R>: the launcher prints NT (and ",f16") only
                args = args[:1]
            flags = re.findall(r"Lb(\d)E", t.group(1))        # first bool = X3 (false: bench.py prints ",f16"); later
            if flags and flags[0] == "0":                       # flags (e.g. the TN GEMM's two-source A) are not printed
                args.append("f16")
            elif name.startswith("pgemm_nt") and len(flags) >= 2 and flags[1] == "0":
                args.append("x2")                               # <T, X3, ALO, OUT16>: ALO false = single-plane A operand, two passes
            elif name.startswith("pgemm_tn") and len(flags) >= 3 and flags[2] == "0":
                args.append("x2")                               # <T, X3, A2, ALO>
            return name + ("<" + ",".join(args) + ">" if args else "")
        return name
    n = n.split("(")[0]
    return re.sub(r"\s+", "", n)


def collect(d, counter):
    vals = collections.defaultdict(list)
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                vals[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return vals


METHOD = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; KiB -> bytes; reads x2 (gfx950 "
          "FETCH_SIZE half-count for wide streaming reads, MI355X_MICROARCH.md)")


def traffic(fetch_dir, write_dir):
    """{kernel: {read_bytes, write_bytes, bytes_per_launch, launches_seen}} from the two pass directories."""
    fetch, write = collect(fetch_dir, "FETCH_SIZE"), collect(write_dir, "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f = sum(fetch[k]) / len(fetch[k]) if fetch.get(k) else 0.0
        w = sum(write[k]) / len(write[k]) if write.get(k) else 0.0
        out[k] = {"read_bytes": 2.0 * f * 1024, "write_bytes": w * 1024, "bytes_per_launch": (2.0 * f + w) * 1024,
                  "launches_seen": max(len(fetch.get(k, [])), len(write.get(k, [])))}
    return out


if __name__ == "__main__":
    json.dump({"method": METHOD, "kernels": traffic(sys.argv[1], sys.argv[2])}, sys.stdout, indent=1)
