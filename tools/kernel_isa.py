#!/usr/bin/env python3
"""Static instruction mix and resources of the kernels in a gfx950 assembly file:
    hipcc --offload-arch=gfx950 ... -S --cuda-device-only -o x.s csrc/aisx_lib.hip
    python tools/kernel_isa.py x.s [name-substring]"""
import collections
import re
import sys


def main(path, sub=""):
    txt = open(path).read()
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", txt, flags=re.S | re.M):
        name, body = m.group(1), m.group(2)
        if sub not in name:
            continue
        c = collections.Counter()
        for line in body.split("\n"):
            line = line.strip()
            if not line or line[0] in ";." or line.endswith(":"):
                continue
            i = line.split()[0]
            if i.startswith("v_pk_"):
                c["valu_pk"] += 1
            elif i.startswith("v_"):
                c["valu"] += 1
            elif i.startswith("ds_"):
                c["lds"] += 1
            elif i.startswith(("buffer_", "global_", "flat_", "scratch_")):
                c["vmem"] += 1
            elif i.startswith("s_barrier"):
                c["barrier"] += 1
            elif i.startswith("s_waitcnt"):
                c["waitcnt"] += 1
            elif i.startswith(("s_cbranch", "s_branch")):
                c["branch"] += 1
            elif i.startswith("s_"):
                c["salu"] += 1
        res = {}
        # the kernel's own entry of amdhsa.kernels (entries start with "  - .agpr_count" or "  - .args")
        meta = ""
        for entry in re.split(r"\n  - \.a", txt[txt.find("amdhsa.kernels"):]):
            if re.search(r"\.name:\s+%s\n" % re.escape(name), entry):
                meta = entry
                break
        for key in ("vgpr_count", "agpr_count", "sgpr_count", "group_segment_fixed_size", "vgpr_spill_count", "private_segment_fixed_size"):
            mm = re.search(r"\.%s:\s*(\d+)" % key, meta)
            if mm:
                res[key] = int(mm.group(1))
        print(name, dict(c), res)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
