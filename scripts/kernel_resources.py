"""Static resources of every kernel in libbreach_hip.so, as hipcc reports them for gfx950 (no GPU needed).

Compiles each source of breaching_amd/csrc with the library's own flags plus `-Rpass-analysis=kernel-resource-usage` and
tabulates, per kernel instantiation: VGPRs, AGPRs, SGPRs, scratch bytes per lane, spills, LDS bytes per block and the occupancy
(waves per SIMD) the register budget allows.  DESIGN.md section 3's claim -- "<= 64 VGPRs, no scratch, 8 waves/SIMD for every
kernel" -- is read off this table, and tests/test_abi.py holds the library to it.

    python scripts/kernel_resources.py [--out profiles/r4_kernel_resources.txt]
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


def render(rows):
    head = f"{'source':24s} {'kernel':58s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'spills':>7s} {'LDS B':>7s} {'waves/SIMD':>10s}"
    lines = ["# hipcc --offload-arch=gfx950 -O3 -Rpass-analysis=kernel-resource-usage, one line per kernel instantiation",
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
    args = ap.parse_args()
    text = render(collect())
    if args.out:
        with open(args.out, "w") as f:
            f.write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
