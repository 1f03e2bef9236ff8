"""Static resources of every kernel in libbreach_hip.so, as hipcc reports them for gfx950 (no GPU needed).

Compiles each source of breaching_amd/csrc with the library's own flags plus `-Rpass-analysis=kernel-resource-usage` and
tabulates, per kernel instantiation: VGPRs, AGPRs, SGPRs, scratch bytes per lane, spills, LDS bytes per block and the occupancy
(waves per SIMD) the register budget allows.  DESIGN.md section 3's occupancy notes are read off this table, and tests/test_abi.py
holds the library to them.

`--isa` adds an instruction census of the generated gfx950 assembly per kernel: global loads / stores by width, how many of
them carry the non-temporal bit, LDS and cross-lane instructions, MFMA and scratch instructions (both expected to be zero:
nothing on this path is GEMM-shaped).

`--loops` lists every loop of the generated code that loads from global memory: loads and stores per trip and the
`s_waitcnt vmcnt(0)` in it -- a trip with one or two loads and a full wait is a serial memory round trip per trip, harmless
in a kernel that runs thousands of wavefronts, the whole cost of one that runs a wavefront per SIMD (how kernels E and F
were found in round 4).

    python scripts/kernel_resources.py [--out profiles/kernel_resources.txt] [--isa --isa-out profiles/kernel_isa_census.txt]
                                       [--loops --loops-out profiles/kernel_loop_census.txt]
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FIELDS = [("VGPRs", "vgpr"), ("AGPRs", "agpr"), ("TotalSGPRs", "sgpr"), ("ScratchSize [bytes/lane]", "scratch"),
          ("Occupancy [waves/SIMD]", "waves"), ("SGPRs Spill", "sgpr_spill"), ("VGPRs Spill", "vgpr_spill"),
          ("LDS Size [bytes/block]", "lds")]


def compiler_id():
    """One line naming the hipcc the tables came from: instruction scheduling and register allocation differ between ROCm
    releases, so the committed tables are compared byte for byte only under the compiler that wrote them (tests/test_abi.py)."""
    from breaching_amd.build import _hipcc

    out = subprocess.run([_hipcc(), "--version"], capture_output=True, text=True).stdout
    lines = [ln.strip() for ln in out.splitlines() if "HIP version" in ln or "clang version" in ln]
    return "# compiler: " + "; ".join(lines)


def demangle(names):
    import shutil

    tool = next((t for t in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", shutil.which("c++filt")) if t and os.path.exists(t)), None)
    if tool is None:
        return {n: n for n in names}
    out = subprocess.run([tool, *names], capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out))


def short(name):
    """`(anonymous namespace)::gm_fwd_kernel<1, true>(float const*, ...)` -> `gm_fwd_kernel<1, true>`."""
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    depth, out = 0, []
    for ch in name:  # cut the argument list: the first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out).strip()


def collect():
    from breaching_amd import build

    rows = []
    flags = [f for f in build.FLAGS if f != "-shared"]
    with tempfile.TemporaryDirectory() as tmp:
        for src in build.SOURCES:
            cmd = [build._hipcc(), *flags, f"-I{build.INCLUDE}", f"-I{build.CSRC}", "-Rpass-analysis=kernel-resource-usage", "-c",
                   os.path.join(build.CSRC, src), "-o", os.path.join(tmp, src + ".o")]
            proc = subprocess.run(cmd, capture_output=True, text=True)
            if proc.returncode != 0:
                raise RuntimeError(proc.stderr[-2000:])
            current = None
            for line in proc.stderr.splitlines():
                m = re.search(r"remark: Function Name: (\S+)", line)
                if m:
                    current = dict(source=src, mangled=m.group(1))
                    rows.append(current)
                    continue
                m = re.search(r"remark:\s+([^:]+): (\S+) \[-Rpass-analysis", line)
                if m and current is not None:
                    for label, key in FIELDS:
                        if m.group(1).strip() == label:
                            current[key] = int(m.group(2))
    names = demangle([r["mangled"] for r in rows])
    for r in rows:
        r["kernel"] = short(names[r["mangled"]])
    return rows


ISA_COUNTS = [("ld128", r"global_load_dwordx4"), ("ld64", r"global_load_dwordx2"), ("ld32", r"global_load_(dword|ubyte|sbyte|ushort|sshort)\b"),
              ("st128", r"global_store_dwordx4"), ("st64", r"global_store_dwordx2"), ("st32", r"global_store_(dword|byte|short)\b"),
              ("atomics", r"global_atomic|flat_atomic"), ("lds", r"\bds_(read|write|load|store)"), ("xlane", r"ds_bpermute|ds_swizzle|v_readlane|v_permlane|_dpp|row_shr|row_bcast"),
              ("mfma", r"v_mfma|v_smfmac"), ("scratch", r"scratch_(load|store)|buffer_(load|store)[a-z0-9_]* .*offen")]


_RAW_BODIES = None


def _raw_bodies():
    """[(source, demangled short name, [raw assembly lines])] from `hipcc -S --cuda-device-only` of every source; compiled once per process."""
    global _RAW_BODIES
    if _RAW_BODIES is not None:
        return _RAW_BODIES
    from breaching_amd import build

    flags = [f for f in build.FLAGS if f not in ("-shared", "-fPIC")]
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        for src in build.SOURCES:
            asm = os.path.join(tmp, src + ".s")
            cmd = [build._hipcc(), *flags, f"-I{build.INCLUDE}", f"-I{build.CSRC}", "-S", "--cuda-device-only", "-Wno-unused-command-line-argument",
                   os.path.join(build.CSRC, src), "-o", asm]
            proc = subprocess.run(cmd, capture_output=True, text=True)
            if proc.returncode != 0:
                raise RuntimeError(proc.stderr[-2000:])
            current, bodies = None, {}
            with open(asm) as f:
                for line in f:
                    m = re.match(r"^(_Z\w+):", line)
                    if m:
                        current = m.group(1)
                        bodies[current] = []
                        continue
                    if line.startswith(".Lfunc_end"):
                        current = None
                    elif current is not None and line.strip():
                        bodies[current].append(line.rstrip())
            names = demangle(list(bodies))
            out.extend((src, short(names[mangled]), body) for mangled, body in bodies.items())
    _RAW_BODIES = out
    return out


def kernel_bodies(keep_labels=False):
    """[(source, kernel, lines)]: the raw lines (labels, block annotations) or only the instructions."""
    if keep_labels:
        return _raw_bodies()
    out = []
    for src, name, lines in _raw_bodies():
        code = [c for c in (line.split(";")[0].strip() for line in lines) if c and not c.startswith(".")]
        out.append((src, name, code))
    return out


def loop_census():
    """One row per loop that loads from global memory: kernel, loop header, depth, loads / stores / full waits / partial waits per trip.
    Loop membership comes from the compiler's own block annotations (`Loop Header` / `in Loop: Header=BBn_m`): the blocks of a
    rotated loop lie on both sides of its header."""
    rows = []
    for src, name, lines in kernel_bodies(keep_labels=True):
        blocks, current = [], None  # [label, annotation, [code]]
        for line in lines:
            m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", line) or re.match(r"^; %bb\.(\d+):\s*(;.*)?$", line)
            if m:
                current = [m.group(1), m.group(2) or "", []]
                blocks.append(current)
            elif current is not None:
                code = line.split(";")[0].strip()
                if code and not code.startswith("."):
                    current[2].append(code)
        for label, note, _ in blocks:
            m = re.search(r"Loop Header: Depth=(\d+)", note)
            if not m:
                continue
            key = label.replace(".L", "")
            trip = [c for lab, n, code in blocks if lab == label or re.search(r"in Loop: Header=" + re.escape(key) + r"\b", n) for c in code]
            loads = sum("global_load" in x for x in trip)
            if not loads:
                continue
            rows.append(dict(source=src, kernel=name, loop=label, depth=int(m.group(1)), loads=loads, stores=sum("global_store" in x for x in trip),
                             full_waits=sum(bool(re.search(r"s_waitcnt.*vmcnt\(0\)", x)) for x in trip),
                             partial_waits=sum(bool(re.search(r"s_waitcnt.*vmcnt\([1-9]", x)) for x in trip), instructions=len(trip)))
    return rows


def render_loops(rows):
    lines = [compiler_id(), "# loops of the generated gfx950 code that load from global memory (scripts/kernel_resources.py --loops)",
             "# loads / stores / s_waitcnt vmcnt(0) / s_waitcnt vmcnt(n > 0) per trip; few loads with a full wait per trip = one serial round trip per trip",
             f"{'kernel':50s} {'loop':>10s} {'depth':>5s} {'loads':>6s} {'stores':>6s} {'wait0':>6s} {'waitN':>6s} {'instr':>6s}"]
    for r in rows:
        lines.append(f"{r['kernel'][:50]:50s} {r['loop']:>10s} {r['depth']:5d} {r['loads']:6d} {r['stores']:6d} {r['full_waits']:6d} {r['partial_waits']:6d} {r['instructions']:6d}")
    return "\n".join(lines) + "\n"


def isa_census():
    """{demangled short kernel name: counts} of the generated assembly of every source."""
    out = {}
    for src, name, body in kernel_bodies():
        row = dict(source=src, instructions=len(body))
        for key, pattern in ISA_COUNTS:
            row[key] = sum(1 for c in body if re.search(pattern, c))
        row["ld_nt"] = sum(1 for c in body if re.match(r"global_load", c) and re.search(r"\bnt\b", c))
        row["st_nt"] = sum(1 for c in body if re.match(r"global_store", c) and re.search(r"\bnt\b", c))
        out[name] = row
    return out


def render_isa(census):
    keys = ["instructions", "ld128", "ld64", "ld32", "ld_nt", "st128", "st64", "st32", "st_nt", "atomics", "lds", "xlane", "mfma", "scratch"]
    lines = [compiler_id(), "# instruction census of the gfx950 assembly (hipcc -O3 -S --cuda-device-only), one line per kernel instantiation",
             "# ld/st = global loads / stores by width in bits; *_nt = of those, with the non-temporal bit; xlane = DPP / permute / readlane",
             f"{'kernel':58s} " + " ".join(f"{k:>8s}" for k in keys)]
    for name, row in census.items():
        lines.append(f"{name[:58]:58s} " + " ".join(f"{row[k]:8d}" for k in keys))
    lines.append(f"# {len(census)} kernels; MFMA instructions {sum(r['mfma'] for r in census.values())}, scratch accesses {sum(r['scratch'] for r in census.values())}")
    return "\n".join(lines) + "\n"


def render(rows):
    head = f"{'source':24s} {'kernel':58s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'spills':>7s} {'LDS B':>7s} {'waves/SIMD':>10s}"
    lines = [compiler_id(), "# hipcc --offload-arch=gfx950 -O3 -Rpass-analysis=kernel-resource-usage, one line per kernel instantiation",
             "# (scripts/kernel_resources.py; 512 VGPRs per SIMD lane -> 8 waves/SIMD needs <= 64)", head]
    for r in rows:
        lines.append(f"{r['source']:24s} {r['kernel'][:58]:58s} {r['vgpr']:5d} {r['agpr']:5d} {r['sgpr']:5d} {r['scratch']:8d} "
                     f"{r['sgpr_spill'] + r['vgpr_spill']:7d} {r['lds']:7d} {r['waves']:10d}")
    lines.append(f"# {len(rows)} kernels; max VGPRs {max(r['vgpr'] for r in rows)}, max scratch {max(r['scratch'] for r in rows)} B/lane, "
                 f"min occupancy {min(r['waves'] for r in rows)} waves/SIMD, max LDS {max(r['lds'] for r in rows)} B/block")
    return "\n".join(lines) + "\n"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--isa", action="store_true")
    ap.add_argument("--isa-out", default=None)
    ap.add_argument("--loops", action="store_true")
    ap.add_argument("--loops-out", default=None)
    args = ap.parse_args()
    text = render(collect())
    if args.out:
        with open(args.out, "w") as f:
            f.write(text)
    sys.stdout.write(text)
    if args.isa or args.isa_out:
        text = render_isa(isa_census())
        if args.isa_out:
            with open(args.isa_out, "w") as f:
                f.write(text)
        sys.stdout.write(text)
    if args.loops or args.loops_out:
        text = render_loops(loop_census())
        if args.loops_out:
            with open(args.loops_out, "w") as f:
                f.write(text)
        sys.stdout.write(text)


if __name__ == "__main__":
    main()
