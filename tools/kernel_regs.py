"""Register / LDS use of every kernel of the library, from the device assembly (cross-compiles without a GPU).
    python tools/kernel_regs.py [filter]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tinyslam_amd import build
os.makedirs("/tmp/isa", exist_ok=True)
flags = [f for f in build.HIPCC_FLAGS if f not in ("-shared", "-fPIC")] + os.environ.get("TINYORB_BUILD_EXTRA", "").split()
s = ""
pat = sys.argv[1] if len(sys.argv) > 1 else ""
for src, extra, obj in build.UNITS:  # every translation unit with device code (k_front's instances: one unit per arithmetic form)
    if src == "orb_node.hip" or (pat and (pat.startswith("k_front<") or pat.startswith("k_front_pair")) != (src == "orb_front_inst.hip")):  # a filter names the units it needs
        continue
    out = "/tmp/isa/" + obj.replace(".o", ".s")
    subprocess.check_call([build._hipcc()] + flags + extra + ["--cuda-device-only", "-S", "-o", out, os.path.join(build.CSRC, src)],
                          stderr=subprocess.DEVNULL)
    s += open(out).read()
for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.vgpr_count:\s+(\d+)", s, re.S):
    name, body, vg = m.group(1), m.group(2), m.group(3)
    sg = re.search(r"\.sgpr_count:\s+(\d+)", body)
    lds = re.search(r"\.group_segment_fixed_size:\s+(\d+)", body)
    scr = re.search(r"\.private_segment_fixed_size:\s+(\d+)", body)
    d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    d = re.sub(r"\(.*", "", d).replace("void orb::", "")
    if pat in d:
        print("%-44s vgpr %3s sgpr %3s lds %6s scratch %s" % (d, vg, sg.group(1) if sg else "?", lds.group(1) if lds else "?", scr.group(1) if scr else "?"))
