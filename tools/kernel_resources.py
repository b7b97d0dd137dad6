"""Register and occupancy table of the kernels in one HIP source file.

usage: python tools/kernel_resources.py raft_amd/csrc/rq_kernels.hip [substring]

Compiles the file for gfx950 with -Rpass-analysis=kernel-resource-usage (no GPU
needed) and prints, per kernel whose demangled name contains `substring`:
VGPRs, SGPRs, waves per SIMD, scratch bytes per lane.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def resources(src, extra=()):
    with tempfile.TemporaryDirectory() as tmp:
        r = subprocess.run(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
                            "-mllvm", "-amdgpu-kernarg-preload-count=16",        # as the Makefile builds
                            "-Rpass-analysis=kernel-resource-usage", *extra, "-c", src, "-o", os.path.join(tmp, "k.o")],
                           cwd=ROOT, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stderr[-4000:])
            raise SystemExit(r.returncode)
    rows = []
    for b in re.split(r"remark: Function Name: ", r.stderr)[1:]:
        name = b.split()[0]
        f = lambda k: int(re.search(re.escape(k) + r": (\d+)", b).group(1))
        rows.append((name, f("VGPRs"), f("AGPRs"), f("TotalSGPRs"), f("Occupancy [waves/SIMD]"), f("ScratchSize [bytes/lane]")))
    names = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True).stdout.split("\n")
    out = []
    for n, r in zip(names, rows):
        n = n.replace("(anonymous namespace)::", "").replace("void ", "")
        n = re.sub(r"\((rq_|unsigned|const|int|long|char).*$", "", n)
        out.append((n,) + r[1:])
    return out


if __name__ == "__main__":
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    print("%-70s %5s %5s %5s %5s %7s" % ("kernel", "VGPR", "AGPR", "SGPR", "waves", "scratch"))
    for row in resources(sys.argv[1], sys.argv[3:]):
        if want in row[0]:
            print("%-70s %5d %5d %5d %5d %7d" % row)
