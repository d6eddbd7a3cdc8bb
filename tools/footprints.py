"""Register / LDS / occupancy footprint of every kernel of the library (hipcc -Rpass-analysis=kernel-resource-usage).
Usage: python tools/footprints.py [file.hip ...] > profiles/rNN_footprints.txt   (no GPU needed)"""
import os
import re
import subprocess
import sys
import tempfile

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vag-nmt_amd", "csrc")


def footprint(src):
    with tempfile.TemporaryDirectory() as d:
        r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function",
                            "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", os.path.join(d, "x.o")],
                           capture_output=True, text=True)
    out = []
    for b in re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]:
        name = b.split("\n")[0].strip()

        def g(k):
            m = re.search(k + r": (\S+)", b)
            return m.group(1) if m else "?"
        out.append((name, g("VGPRs"), g("AGPRs"), g("SGPRs"), g("VGPRs Spill"), g(r"Occupancy \[waves/SIMD\]"),
                    g(r"LDS Size \[bytes/block\]"), g(r"ScratchSize \[bytes/lane\]")))
    return out


if __name__ == "__main__":
    files = sys.argv[1:] or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    print("%-100s %5s %5s %5s %6s %4s %8s %8s" % ("kernel (demangled by c++filt below)", "vgpr", "agpr", "sgpr", "spill", "occ", "lds", "scratch"))
    for f in files:
        rows = footprint(os.path.join(CSRC, f) if not os.path.isabs(f) else f)
        names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
        print("# " + f)
        for r, n in zip(rows, names):
            n = re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", ""))
            print("%-100s %5s %5s %5s %6s %4s %8s %8s" % ((n[:100],) + r[1:]))
