#!/usr/bin/env python3
"""Per-kernel resource table of the built library: VGPR / AGPR / SGPR counts, spills, scratch and LDS of every gfx950 kernel in
libhelix_vec_gfx950.so (the .hip_fatbin section holds one offload bundle per translation unit).
usage: kernel_meta.py [lib.so] [out.json]"""
import json, os, re, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def kernels_of(lib):
    tmp = tempfile.mkdtemp()
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    out = {}
    for i, s in enumerate(starts):
        part = os.path.join(tmp, f"b{i}.bin")
        open(part, "wb").write(blob[s: starts[i + 1] if i + 1 < len(starts) else len(blob)])
        co = os.path.join(tmp, f"b{i}.co")
        r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            f"--output={co}"], capture_output=True, text=True)
        if r.returncode or not os.path.exists(co) or os.path.getsize(co) == 0:
            continue
        txt = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
        for blk in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk).group(1)
            g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", blk).group(1))
            out[name] = dict(vgpr=g("vgpr_count"), agpr=int(re.match(r"\s*(\d+)", blk).group(1)), sgpr=g("sgpr_count"),
                             vgpr_spill=g("vgpr_spill_count"), sgpr_spill=g("sgpr_spill_count"), scratch=g("private_segment_fixed_size"),
                             lds=g("group_segment_fixed_size"))
    return out


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "helix-db_amd", "libhelix_vec_gfx950.so")
    k = kernels_of(lib)
    if len(sys.argv) > 2:
        json.dump(k, open(sys.argv[2], "w"), indent=0, sort_keys=True)
    bad = {n: v for n, v in k.items() if v["scratch"] or v["sgpr_spill"] or v["vgpr_spill"]}
    print(f"{len(k)} kernels; {len(bad)} with spills / scratch")
    for n, v in sorted(bad.items()):
        dem = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
        print(f"  {dem[:140]}  {v}")
