#!/usr/bin/env python3
"""check_isa.py -- build-time gate for the hand-counted loads of the direct-B ping-pong kernels (`make check-isa`).

k_pairwise_pp<..., BD = 1> (metagenome_vector_sketches_amd/csrc/mvs_pairwise.hip) loads the B operand's fragments of the
NEXT k-slice with `asm volatile("global_load_dwordx4 ...")` and counts `vmcnt` by hand.  The compiler believes an asm's
result is there at once, so the scheme is correct only while the generated code leaves the loaded registers alone between
the load's issue and the `s_waitcnt vmcnt` that retires it, and feeds exactly those registers to the matrix cores.  A
register copy inserted in between would read stale data: wrong cells, no fault.  Nothing in the language enforces that, so
this script checks the machine code of every such kernel in the library that ships:

  R1  every `v_mfma_i32_16x16x64_i8` of the k-loop takes its B operand (src1) straight from the destination registers of a
      plain `global_load_dwordx4 v[..], v[..], off` of the k-loop (the asm loads; `global_load_lds_*` copies have no register
      destination);
  R2  walking the code in order -- the k-loop's body twice, so that a load issued in one iteration meets its consumer in the
      next --, from a load's issue to the first MFMA that reads its destination: (a) no other instruction mentions any of the
      destination registers before the load has been retired, (b) it HAS been retired by then.  Retired = an `s_waitcnt vmcnt(N)`
      with at most N memory instructions issued after the load (vmcnt counts in order); where scalar code selects between
      several waits (the kernel waits for vmcnt(6) / (4) / (0) depending on what the phase issued) the weakest one counts.

Exit code 0 = every kernel passes.  `--self-test` also proves the checker bites: it must reject three mutations of the real
code (a `v_mov` of a fragment register right after its load, an MFMA whose B operand comes from another register, the
retiring wait weakened).

Usage: check_isa.py [--lib libmvs_hip.so | --obj mvs_pairwise.o | --dis file.dis] [--self-test] [-v]
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get("MVS_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
KERNEL_RE = re.compile(r"k_pairwise_ppILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi[12]EE")    # BD = 1 or 2


def run(cmd):
    return subprocess.run(cmd, check=True, capture_output=True)


def disassemble(path, is_dis=False):
    """-> disassembly text of the gfx950 code object(s) inside `path` that hold k_pairwise_pp kernels"""
    if is_dis:
        with open(path) as f:
            return f.read()
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        # (with an explicit output file: objcopy otherwise rewrites its INPUT in place -- the library that ships came out with
        # a fresh mtime and `make -q` saw every host tool stale against it)
        run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, path, os.path.join(tmp, "copy.so")])
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        if not starts:
            raise SystemExit("check_isa: no offload bundle in %s (compressed fat binary?)" % path)
        for k, s in enumerate(starts):
            piece = os.path.join(tmp, "bundle%d" % k)
            with open(piece, "wb") as f:
                f.write(blob[s:starts[k + 1] if k + 1 < len(starts) else len(blob)])
            co = os.path.join(tmp, "co%d" % k)
            r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + piece,
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], capture_output=True)
            if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            syms = run([os.path.join(LLVM, "llvm-readelf"), "-s", "-W", co]).stdout.decode()
            if "k_pairwise_pp" not in syms:
                continue
            out.append(run([os.path.join(LLVM, "llvm-objdump"), "-d", co]).stdout.decode())
    if not out:
        raise SystemExit("check_isa: no gfx950 code object with k_pairwise_pp kernels in %s" % path)
    return "\n".join(out)


class Insn:
    __slots__ = ("addr", "mnem", "ops", "text")

    def __init__(self, addr, mnem, ops, text):
        self.addr, self.mnem, self.ops, self.text = addr, mnem, ops, text


INSN_RE = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")


def split_kernels(text):
    """-> {symbol: [Insn]} for the direct-B ping-pong kernels"""
    kernels, cur = {}, None
    for line in text.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = m.group(1) if KERNEL_RE.search(m.group(1)) and ".kd" not in m.group(1) else None
            if cur:
                kernels[cur] = []
            continue
        if cur is None:
            continue
        m = INSN_RE.match(line)
        if m:
            kernels[cur].append(Insn(int(m.group(3), 16), m.group(1), m.group(2), line.strip()))
    return kernels


def vregs(operand):
    """VGPR numbers an operand names: v12, v[12:15]"""
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", operand)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", operand)
    return {int(m.group(1))} if m else set()


def operands(ins):
    return [o.strip() for o in re.split(r",(?![^\[]*\])", ins.ops)] if ins.ops else []


def all_vregs(ins):
    out = set()
    for o in operands(ins):
        out |= vregs(o.split(" ")[0])
    return out


def is_vm(ins):
    return ins.mnem.startswith(("global_load", "global_store", "global_atomic", "buffer_", "flat_", "scratch_"))


def is_bload(ins):
    return ins.mnem == "global_load_dwordx4" and ins.ops.rstrip().endswith("off") and len(operands(ins)) == 3


def is_mfma(ins):
    return ins.mnem.startswith("v_mfma_i32_16x16x64_i8")


def vmcnt_of(ins):
    if ins.mnem != "s_waitcnt":
        return None
    m = re.search(r"vmcnt\((\d+)\)", ins.ops)
    return int(m.group(1)) if m else None


def branch_target(ins):
    """address a branch goes to (objdump prints the offset in dwords after the instruction)"""
    if not ins.mnem.startswith(("s_cbranch", "s_branch")):
        return None
    m = re.match(r"^(-?\d+)", ins.ops)
    if not m:
        return None
    off = int(m.group(1))
    if off >= 0x8000:
        off -= 0x10000
    return ins.addr + 4 + 4 * off


def check_kernel(name, insns, verbose=False):
    """-> list of violations (strings)"""
    bad = []
    mf = [i for i, x in enumerate(insns) if is_mfma(x)]
    if not mf:
        return ["%s: no v_mfma_i32_16x16x64_i8 at all" % name]
    first_mf, last_mf = mf[0], mf[-1]
    # the k-loop: the tightest backward branch around at least half of the MFMAs (two phases of the loop body; the odd tail
    # phase sits behind it; wider backward branches are jumps to shared exit code)
    addr_index = {x.addr: i for i, x in enumerate(insns)}
    best = None
    for i, x in enumerate(insns):
        t = branch_target(x)
        if t is not None and t <= x.addr and t in addr_index:
            j = addr_index[t]
            n_in = sum(1 for k in mf if j <= k <= i)
            if 2 * n_in >= len(mf) and (best is None or i - j < best[1] - best[0]):
                best = (j, i, n_in)
    if best is None:
        return ["%s: no loop around the MFMAs" % name]
    loop_a, loop_b = best[0], best[1]
    # the fragment registers = destinations of the register loads inside the loop; the region starts at the first load in
    # front of the loop that writes one of them (the prologue's loads of slice 0) and ends at the last MFMA
    loop_loads = [i for i in range(loop_a, loop_b + 1) if is_bload(insns[i])]
    if not loop_loads:
        return ["%s: no register load inside the k-loop" % name]
    dest_set = set()
    for i in loop_loads:
        d = vregs(operands(insns[i])[0])
        if len(d) != 4:
            bad.append("%s: unexpected destination in `%s`" % (name, insns[i].text))
        dest_set.add(frozenset(d))
    pro = [i for i in range(0, loop_a) if is_bload(insns[i]) and frozenset(vregs(operands(insns[i])[0])) in dest_set]
    if not pro:
        return ["%s: no load of the first slice's fragments in front of the k-loop" % name]
    start = pro[0]
    region_loads = pro + loop_loads
    # R1
    for i in mf:
        ops = operands(insns[i])
        b = frozenset(vregs(ops[2]))
        if b not in dest_set:
            bad.append("%s: R1 the B operand %s of `%s` is not the destination of a k-loop register load" % (name, ops[2], insns[i].text[:90]))
    # R2: prologue + loop body twice + tail, in order
    # (unconditional forward branches are followed -- the `nk == 0` arm that zero-fills the fragment registers sits in the
    # text right behind the prologue's loads and is jumped over; conditional ones fall through)
    seq, i, again = [], start, True
    while i <= last_mf and len(seq) < 4 * len(insns):
        seq.append(i)
        t = branch_target(insns[i])
        if insns[i].mnem == "s_branch" and t in addr_index and addr_index[t] > i:
            i = addr_index[t]
        elif i == loop_b and again:
            again, i = False, loop_a
        else:
            i += 1
    n = len(seq)
    checked = 0
    for p, idx in enumerate(seq):
        ins = insns[idx]
        if not is_bload(ins) or idx < start:
            continue
        R = vregs(operands(ins)[0])
        issued_after, retired, consumer = 0, False, None
        q = p + 1
        while q < n:
            x = insns[seq[q]]
            if x.mnem.startswith("s_"):
                # a run of scalar instructions: the weakest vmcnt wait in it is what every path through it guarantees
                waits = []
                while q < n and insns[seq[q]].mnem.startswith("s_"):
                    w = vmcnt_of(insns[seq[q]])
                    if w is not None:
                        waits.append(w)
                    q += 1
                if waits and max(waits) <= issued_after:
                    retired = True
                continue
            touched = all_vregs(x) & R
            if is_mfma(x) and frozenset(vregs(operands(x)[2])) == frozenset(R):
                consumer = x
                break
            if touched and not retired:
                if is_bload(x) and vregs(operands(x)[0]) == R:
                    bad.append("%s: R2 `%s` is issued again before its earlier issue was consumed" % (name, x.text[:80]))
                else:
                    bad.append("%s: R2a `%s` touches v%s while `%s` is still in flight" %
                               (name, x.text[:80], sorted(touched), ins.text[:60]))
                break
            if is_vm(x):
                issued_after += 1
            q += 1
        if consumer is not None:
            checked += 1
            if not retired:
                bad.append("%s: R2b `%s` is consumed by `%s` without a vmcnt wait that retires it (%d memory instructions issued after it)" %
                           (name, ins.text[:60], consumer.text[:70], issued_after))
    if checked < 8:
        bad.append("%s: only %d load -> MFMA chains found (the walk lost the loop?)" % (name, checked))
    if verbose:
        print("  %s\n    %d MFMAs, %d register loads in the k-loop region, %d load -> wait -> MFMA chains checked, loop %#x..%#x" %
              (name, len(mf), len(region_loads), checked, insns[loop_a].addr, insns[loop_b].addr))
    return bad


def mutate(insns, kind):
    """a deliberately broken copy of a kernel's instruction list"""
    out = [Insn(x.addr, x.mnem, x.ops, x.text) for x in insns]
    mf = [i for i, x in enumerate(out) if is_mfma(x)]
    loads = [i for i, x in enumerate(out) if is_bload(x) and mf[0] < i < mf[-1]]
    if kind == "copy":        # a register copy of a fragment right after its load was issued
        i = loads[0]
        r = sorted(vregs(operands(out[i])[0]))[0]
        out.insert(i + 1, Insn(out[i].addr + 1, "v_mov_b32_e32", "v255, v%d" % r, "v_mov_b32_e32 v255, v%d  // inserted" % r))
    elif kind == "operand":   # an MFMA fed from a register no load wrote
        i = mf[len(mf) // 2]
        ops = operands(out[i])
        ops[2] = "v[250:253]"
        out[i].ops = ", ".join(ops)
        out[i].text = out[i].mnem + " " + out[i].ops + "  // mutated"
    elif kind == "wait":      # the retiring waits weakened
        for x in out:
            if mf[0] < out.index(x) < mf[-1] and vmcnt_of(x) is not None:
                x.ops = re.sub(r"vmcnt\(\d+\)", "vmcnt(15)", x.ops)
                x.text = "s_waitcnt " + x.ops + "  // mutated"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "metagenome_vector_sketches_amd", "libmvs_hip.so"))
    ap.add_argument("--obj")
    ap.add_argument("--dis")
    ap.add_argument("--self-test", action="store_true")
    ap.add_argument("-v", "--verbose", action="store_true")
    args = ap.parse_args()
    text = disassemble(args.dis, True) if args.dis else disassemble(args.obj or args.lib)
    kernels = split_kernels(text)
    if not kernels:
        print("check_isa: no k_pairwise_pp<..., BD = 1> kernel found", file=sys.stderr)
        return 2
    failures = []
    for name, insns in sorted(kernels.items()):
        failures += check_kernel(name, insns, args.verbose)
    if failures:
        print("check_isa: FAILED", file=sys.stderr)
        for f in failures:
            print("  " + f, file=sys.stderr)
        return 1
    print("check_isa: %d direct-B kernels pass (B operands come straight from the hand-counted loads; nothing touches them "
          "before the vmcnt wait that retires them)" % len(kernels))
    if args.self_test:
        for kind in ("copy", "operand", "wait"):
            caught = 0
            for name, insns in sorted(kernels.items()):
                caught += 1 if check_kernel(name, mutate(insns, kind)) else 0
            if caught != len(kernels):
                print("check_isa: self-test: mutation `%s` went unnoticed in %d of %d kernels" % (kind, len(kernels) - caught, len(kernels)),
                      file=sys.stderr)
                return 3
        print("check_isa: self-test: all 3 mutations (register copy after the load, foreign B operand, weakened wait) are rejected "
              "in every kernel")
    return 0


if __name__ == "__main__":
    sys.exit(main())
