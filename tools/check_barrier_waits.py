"""Static check of gfx950 assembly: every s_barrier must be reached with NO LDS write still in flight.

Round 5 found hipcc (ROCm 7.2, gfx950) dropping the `s_waitcnt lgkmcnt(0)` of `__syncthreads()`'s workgroup release fence at a loop
header whose BACK EDGE carries pending ds_write instructions: a wave then arrives at the barrier with its LDS stores still queued, the
other waves pass the barrier and read the old values whenever that wave's LDS queue is backed up by co-resident LDS-heavy waves (the
multi-workgroup tridiagonalisation beside rocBLAS's dsymm kernel: DESIGN.md section 6).  This walks the control-flow graph of every
kernel in the given .s files (hipcc -S --cuda-device-only) with a forward "an LDS write may be pending" dataflow and reports the
barriers reached with one.

    python tools/check_barrier_waits.py [--with-lab] [file.s ...]   (no files: compiles mesheditor_amd/csrc/*.hip to a temp dir first;
                                                                     the lab library's soak kernels reproduce the defect on purpose)"""
import glob
import os
import re
import subprocess
import sys
import tempfile

LDS_WRITE = re.compile(r"^\s*(ds_write|ds_add|ds_sub|ds_min|ds_max|ds_and|ds_or|ds_xor|ds_inc|ds_dec|ds_cmpst|ds_wrxchg|ds_pk_add|ds_store|ds_wrap|ds_mskor|ds_rsub)")
WAITS_ALL_LGKM = re.compile(r"^\s*s_waitcnt\b.*lgkmcnt\(0\)")
BRANCH = re.compile(r"^\s*(s_branch|s_cbranch_\w+)\s+(\.L\w+)")
LABEL = re.compile(r"^(\.L\w+):")


def kernels(text):
    """(name, lines) of every function body in an assembly file"""
    out = []
    for m in re.finditer(r"^(\w+):\s*(?:;.*)?\n", text, re.M):
        name = m.group(1)
        if name.startswith(".") or name.startswith("__hip"):
            continue
        end = text.find(".Lfunc_end", m.end())
        if end < 0:
            continue
        body = text[m.end():end]
        if "s_endpgm" not in body and "s_setpc" not in body:
            continue
        out.append((name, body.split("\n")))
    return out


def check(lines):
    # basic blocks: split at labels and after branches / s_endpgm
    blocks, cur, label_of = [], [], {}
    def flush():
        nonlocal cur
        if cur:
            blocks.append(cur)
            cur = []
    for ln in lines:
        m = LABEL.match(ln)
        if m:
            flush()
            label_of[m.group(1)] = len(blocks)
            cur = [ln]
            continue
        s = ln.strip()
        if not s or s.startswith(";") or s.startswith("."):
            continue
        cur.append(ln)
        if BRANCH.match(ln) or s.startswith("s_endpgm") or s.startswith("s_setpc"):
            flush()
    flush()
    succ = [[] for _ in blocks]
    for i, b in enumerate(blocks):
        last = b[-1].strip()
        m = BRANCH.match(b[-1])
        if m:
            if m.group(2) in label_of:
                succ[i].append(label_of[m.group(2)])
            if not last.startswith("s_branch") and i + 1 < len(blocks):
                succ[i].append(i + 1)
        elif not (last.startswith("s_endpgm") or last.startswith("s_setpc")) and i + 1 < len(blocks):
            succ[i].append(i + 1)
    pending_in = [False] * len(blocks)
    findings = set()
    work = list(range(len(blocks)))
    visited_once = [False] * len(blocks)
    while work:
        i = work.pop(0)
        p = pending_in[i]
        for ln in blocks[i]:
            if LDS_WRITE.match(ln):
                p = True
            elif WAITS_ALL_LGKM.match(ln):
                p = False
            elif ln.strip().startswith("s_barrier") and p:
                findings.add((i, ln.strip()))
        first = not visited_once[i]
        visited_once[i] = True
        for j in succ[i]:
            if (p and not pending_in[j]) or first:
                if p and not pending_in[j]:
                    pending_in[j] = True
                    if j not in work:
                        work.append(j)
    return sorted(findings), len(blocks)


def check_text(text):
    """[(kernel name, number of unprotected barriers)] of an assembly text"""
    out = []
    for name, lines in kernels(text):
        found, _ = check(lines)
        out.append((name, len(found)))
    return out


def main():
    with_lab = "--with-lab" in sys.argv
    files = [a for a in sys.argv[1:] if a != "--with-lab"]
    tmp = None
    if not files:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        tmp = tempfile.mkdtemp(prefix="mh_asm_")
        sources = sorted(glob.glob(os.path.join(root, "mesheditor_amd", "csrc", "*.hip")))
        if with_lab:
            sources += sorted(glob.glob(os.path.join(root, "mesheditor_amd", "csrc", "lab", "*.hip")))
        for src in sources:
            out = os.path.join(tmp, os.path.basename(src)[:-4] + ".s")
            # the flags of mesheditor_amd/csrc/Makefile (CXXFLAGS incl. $(EXTRA) from the environment, STRICT for the two bit-exact files): the
            # assembly checked is the assembly shipped
            strict = ["-ffp-contract=off"] if os.path.basename(src) in ("mh_pipeline.hip", "mh_bank.hip") else []
            extra = os.environ.get("EXTRA", "").split()
            p = subprocess.run(["/opt/rocm/bin/hipcc", *extra, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "--cuda-device-only", "-S", *strict, "-o", out, src],
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            if p.returncode != 0:
                sys.stderr.write(p.stdout)
                sys.stderr.write(f"check_barrier_waits: hipcc failed on {src} (exit {p.returncode})\n")
                return 2
            files.append(out)
    total, bad = 0, 0
    for f in files:
        for name, lines in kernels(open(f).read()):
            total += 1
            found, _ = check(lines)
            if found:
                bad += 1
                print(f"{os.path.basename(f)}: {name[:110]}: {len(found)} barrier(s) reachable with an LDS write in flight")
    print(f"{total} kernels checked, {bad} with an unprotected barrier")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
