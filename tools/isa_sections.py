"""tools/isa_sections.py [-DFLAG ...] -- compile the super-k-mer query kernel with section markers (-DMC_SK_MARK: s_nop 8..15 between
the phases of a step) and count the instructions between them, per class, in program order.  Static counts of straight-line
code: loops and skipped branches are not weighted.  Runs in the build container (no GPU needed)."""
import collections, os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "jn_cuclark_amd", "csrc")
t = tempfile.mkdtemp()
kern = os.environ.get("KERNEL", "_ZN2mc2sk15sk_query_kernelILi0ELi31EEEvNS0_6SkArgsE")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "--offload-device-only",
                       "-DMC_SK_MARK"] + sys.argv[1:] + ["-c", os.path.join(src, "mc_api.hip"), "-o", t + "/dev.o"])
subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + t + "/dev.o",
                       "--targets=hip-amdgcn-amd-amdhsa--gfx950", "--output=" + t + "/dev.elf"])
asm = subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--disassemble-symbols=" + kern, t + "/dev.elf"], text=True)
open(t + "/k.s", "w").write(asm)
names = {9: "front: cut, keys, ends", 10: "window minima, runs", 11: "publish", 12: "descriptors", 13: "fetch", 14: "match", 15: "fold", 8: "after"}
cur, sec = None, []
for ln in asm.split("\n"):
    m = re.match(r"\s+([a-z_0-9]+)\s*(\S*)", ln)
    if not m:
        continue
    op, arg = m.group(1), m.group(2)
    if op == "s_nop" and arg.isdigit() and int(arg) in names:
        cur = collections.Counter(); sec.append((int(arg), cur)); continue
    if cur is None:
        continue
    cls = ("valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_")) else
           "branch" if op.startswith(("s_cbranch", "s_branch")) else "wait" if op.startswith(("s_waitcnt", "s_nop")) else "salu")
    cur[cls] += 1
for n, c in sec:
    print("%-26s %s" % (names[n], dict(c)))
print("listing:", t + "/k.s")
