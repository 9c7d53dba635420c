"""Static instruction histogram of kernels whose mangled name contains a substring.
    python tools/isa_kernel.py <substring> [top]    (compiles visualslam_amd/csrc/vslam_hip.hip to /tmp/v.s first)"""
import collections, os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only",
                "-o", "/tmp/v.s", os.path.join(root, "visualslam_amd/csrc/vslam_hip.hip")], check=True, stderr=subprocess.DEVNULL)
s = open("/tmp/v.s").read()
top = int(sys.argv[2]) if len(sys.argv) > 2 else 50
for f in re.split(r"\n(?=_Z\w+:)", s):
    name = f.split(":", 1)[0]
    if sys.argv[1] not in name:
        continue
    c = collections.Counter()
    for l in f.splitlines():
        m = re.match(r"\s+([vsdg]\w+|buffer\w+|flat\w+)\s", l)
        if m:
            c[m.group(1)] += 1
    blk = s[s.find(".name:           " + name):][:3000]
    g = lambda k: (re.search(r"\." + k + r":\s+(\d+)", blk) or [None, "?"])[1]
    print(name[:90], "VALU", sum(v for k, v in c.items() if k.startswith("v_")), "vgpr", g("vgpr_count"), "sgpr", g("sgpr_count"),
          "spill", g("vgpr_spill_count"), "lds", g("group_segment_fixed_size"))
    print(sorted(c.items(), key=lambda x: -x[1])[:top])
