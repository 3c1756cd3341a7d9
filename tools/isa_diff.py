"""Compare the gfx950 code of two libvb2.so builds kernel by kernel (a refactoring aid; runs where there is no GPU).

    python tools/isa_diff.py OLD.so NEW.so ['old kernel name substring=new kernel name substring' ...]

Without pairs: lists both builds' kernels with instruction counts, VGPR/SGPR/scratch/LDS figures.  With pairs: the
instruction streams of each pair (addresses, symbol names and branch targets stripped) are diffed -- identical streams
mean identical kernels.  Uses llvm-objdump / llvm-readelf from /opt/rocm/lib/llvm/bin.
"""
import difflib
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin/"


def code_object(so, tmp):
    d = os.path.join(tmp, os.path.basename(so) + ".d")
    os.makedirs(d, exist_ok=True)
    local = os.path.join(d, "lib.so")
    subprocess.check_call(["cp", so, local])
    subprocess.check_call([LLVM + "llvm-objdump", "--offloading", local], cwd=d, stdout=subprocess.DEVNULL)
    co = [f for f in os.listdir(d) if "amdgcn" in f]
    return os.path.join(d, co[0])


def kernels(co):
    dis = subprocess.run([LLVM + "llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout
    names, cur, out = {}, None, {}
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        if cur is None or not line.strip():
            continue
        ins = line.split("//")[0].strip()
        ins = re.sub(r"^[0-9a-f]+:\s*", "", ins)
        if ins:
            out[cur].append(ins)
    dem = subprocess.run(["c++filt"], input="\n".join(out.keys()), capture_output=True, text=True).stdout.splitlines()
    return {d: out[k] for k, d in zip(out.keys(), dem)}


def normalise(ins):
    ins = re.sub(r"<[^>]*>", "<sym>", ins)
    ins = re.sub(r"\b(s_c?branch\w*|s_call\w*)\s+\S+", r"\1 <target>", ins)
    return ins


def main():
    old, new = sys.argv[1], sys.argv[2]
    pairs = [p.split("=", 1) for p in sys.argv[3:]]
    with tempfile.TemporaryDirectory() as tmp:
        ko, kn = kernels(code_object(old, tmp)), kernels(code_object(new, tmp))
    if not pairs:
        for tag, ks in (("OLD", ko), ("NEW", kn)):
            print("%s: %d functions, %d instructions" % (tag, len(ks), sum(len(v) for v in ks.values())))
            for k, v in ks.items():
                print("  %6d  %s" % (len(v), k[:150]))
        return 0
    bad = 0
    for a, b in pairs:
        ca = [k for k in ko if a in k]
        cb = [k for k in kn if b in k]
        if len(ca) != 1 or len(cb) != 1:
            print("ambiguous pair %r (%d) = %r (%d)" % (a, len(ca), b, len(cb)))
            bad += 1
            continue
        ia, ib = [normalise(x) for x in ko[ca[0]]], [normalise(x) for x in kn[cb[0]]]
        if ia == ib:
            print("IDENTICAL  %d instructions  %s" % (len(ia), b))
        else:
            d = list(difflib.unified_diff(ia, ib, lineterm="", n=0))
            print("DIFFERENT  %d -> %d instructions, %d diff lines  %s" % (len(ia), len(ib), len(d), b))
            for line in d[:40]:
                print("    " + line)
            bad += 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
