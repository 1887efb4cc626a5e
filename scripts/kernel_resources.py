"""Parse `hipcc -Rpass-analysis=kernel-resource-usage` remarks -> {kernel: {VGPRs, AGPRs, Occupancy, Scratch, LDS}}
and diff two builds.  usage: kernel_resources.py new.txt [old.txt] [substring]"""
import re
import subprocess
import sys


def parse(path):
    out, cur = {}, None
    for line in open(path):
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = m.group(1); out[cur] = {}
            continue
        m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", line)
        if m and cur:
            out[cur][m.group(1).split(" [")[0]] = int(m.group(2))
    return out


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return dict(zip(names, p.stdout.splitlines()))


if __name__ == "__main__":
    new = parse(sys.argv[1])
    old = parse(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] != "-" else None
    filt = sys.argv[3] if len(sys.argv) > 3 else ""
    dm = demangle(list(new))
    for k, v in new.items():
        name = dm[k].replace("void ics::", "").split("(")[0]
        if filt not in name:
            continue
        if old is None:
            print("%-100s %s" % (name[:100], v))
        elif k in old and old[k] != v:
            print("%-100s\n    old %s\n    new %s" % (name[:100], old[k], v))
