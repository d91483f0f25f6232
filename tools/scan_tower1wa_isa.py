"""ISA scan of k_tower1wa (build container).  usage: scan_tower1wa_isa.py [FILE.s]   (without FILE.s: compiles
csrc/xq_tower.hip to assembly first: hipcc --offload-arch=gfx950 -O3 ... --cuda-device-only -S)
The kernel's HIP parts (input convolution, its epilogue) name their accumulator AGPRs a[0:191] literally and issue MFMAs as
asm statements; the residual tower is one asm statement that clobbers v0..v253, a0..a223 and its SGPRs.  hipcc does not
know that a[0:191] are live between those statements.  Checked here, on both instantiations (plain / stamped):
(1) no VALU instruction writes a source register of an asm MFMA within the two instructions in front of it (hipcc inserts
    no wait states for an asm MFMA);
(2) every AGPR the compiler uses by itself (spills around the big statement) while the accumulators are live - i.e. in
    front of the END of the tower statement - is a224 or higher;
(3) the kernel uses no scratch memory and at most 512 registers;
(4) the generated bodies write m0 (one s_mov / s_add per weight-DMA piece) and "m0" cannot be listed as a clobber (hipcc
    answers with its reserved-register warning): every compiler-issued LDS-DMA behind the END of a big statement must
    therefore be preceded - behind that END - by a compiler write of m0.  Checked for k_tower1wa and for
    k_policy_fc1w (csrc/xq_policy.hip, compiled here too)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    path = sys.argv[1]
else:
    path = os.path.join(tempfile.mkdtemp(), "xq_tower.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-Wno-unused-function",
                           "--cuda-device-only", "-S", "-o", path, os.path.join(ROOT, "chinesechessai_amd", "csrc", "xq_tower.hip")],
                          stderr=subprocess.DEVNULL)
text = open(path).read()


def regs(tok, cls="v"):
    mm = re.match(cls + r"\[(\d+):(\d+)\]", tok)
    if mm:
        return set(range(int(mm.group(1)), int(mm.group(2)) + 1))
    mm = re.match(cls + r"(\d+)$", tok)
    return {int(mm.group(1))} if mm else set()


rc = 0
for name in ("k_tower1waILb0ELi0E", "k_tower1waILb1ELi0E"):
    m = re.search(r"^_ZN\w*%s\w*:[^\n]*\n(.*?)\n\s*\.end_amdhsa_kernel" % name, text, re.S | re.M)
    body = m.group(1)
    lines = body.split("\n")
    # the big statement = the asm block with the most lines
    blocks, cur = [], None
    for i, l in enumerate(lines):
        if "#ASMSTART" in l:
            cur = i
        elif "#ASMEND" in l and cur is not None:
            blocks.append((cur, i))
            cur = None
    big = max(blocks, key=lambda b: b[1] - b[0])
    ins = [(i, l.strip()) for i, l in enumerate(lines) if l.strip() and not l.strip().startswith((";", ".")) and not l.strip().endswith(":")]
    bad = 0
    for k, (i, l) in enumerate(ins):
        if big[0] < i < big[1]:
            continue                                         # the generated body has its own checker
        if l.startswith("v_mfma") and l.split()[1].startswith("a["):
            ops = [t.strip() for t in l.split(None, 1)[1].split(",")]
            src = regs(ops[1]) | regs(ops[2])
            for back in (1, 2):
                pl = ins[k - back][1]
                if pl.startswith("s_nop"):
                    break
                if pl.startswith("v_") and not pl.startswith(("v_mfma", "v_accvgpr_write")):
                    if regs(pl.split(None, 1)[1].split(",")[0].strip()) & src:
                        bad += 1
                        print("HAZARD line", i + 1, ":", pl, "->", l)
    inasm, low_cc = False, []
    for i, l in enumerate(lines[:big[1]]):
        if "#ASMSTART" in l:
            inasm = True
        elif "#ASMEND" in l:
            inasm = False
        elif not inasm and not l.strip().startswith(";"):
            for tok in re.findall(r"a\[\d+:\d+\]|\ba\d+\b", l):
                if min(regs(tok, "a") or {999}) < 224:
                    low_cc.append((i + 1, l.strip()))
    scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", text[m.end() - 20:m.end() + 3000] if False else body + text[m.end():m.end() + 3000]).group(1))
    nfree = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body + text[m.end():m.end() + 3000]).group(1))
    print("%s: VALU -> asm MFMA hazards %d; compiler-generated uses of a0..a223 in front of the end of the tower statement: %d; "
          "scratch %d bytes; registers %d; tower statement %d lines" % (name, bad, len(low_cc), scratch, nfree, big[1] - big[0]))
    for ln, l in low_cc[:5]:
        print("   line %d: %s" % (ln, l))
    if bad or low_cc or scratch or nfree > 512:
        rc = 1


def m0_after_big_statement(text, name):
    """(number of LDS-DMA instructions behind the big asm statement, those among them not preceded by an m0 write)"""
    m = re.search(r"^_ZN?\w*%s\w*:[^\n]*\n(.*?)\n\s*\.end_amdhsa_kernel" % name, text, re.S | re.M)
    lines = m.group(1).split("\n")
    blocks, cur = [], None
    for i, l in enumerate(lines):
        if "#ASMSTART" in l:
            cur = i
        elif "#ASMEND" in l and cur is not None:
            blocks.append((cur, i))
            cur = None
    big = max(blocks, key=lambda b: b[1] - b[0])
    m0_set, n_dma, bad = False, 0, []
    for i in range(big[1] + 1, len(lines)):
        l = lines[i].strip()
        if not l or l.startswith((";", ".")) or l.endswith(":"):
            continue
        ops = l.split(None, 1)
        if len(ops) > 1 and ops[0].startswith("s_") and ops[1].split(",")[0].strip() == "m0":
            m0_set = True
        if (ops[0].startswith("global_load_lds") or (ops[0].startswith("buffer_load") and l.endswith(" lds"))):
            n_dma += 1
            if not m0_set:
                bad.append((i + 1, l))
    return n_dma, bad


for name in ("k_tower1waILb0ELi0E", "k_tower1waILb1ELi0E"):
    n_dma, badm = m0_after_big_statement(text, name)
    print("%s: %d compiler-issued LDS-DMA behind the tower statement, %d without an m0 write in front" % (name, n_dma, len(badm)))
    for ln, l in badm[:5]:
        print("   line %d: %s" % (ln, l))
    if badm or n_dma == 0:          # (the heads' weights arrive by LDS-DMA: none at all means this scan no longer sees them)
        rc = 1
if len(sys.argv) <= 1:
    ppath = os.path.join(os.path.dirname(path), "xq_policy.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-Wno-unused-function",
                           "--cuda-device-only", "-S", "-o", ppath, os.path.join(ROOT, "chinesechessai_amd", "csrc", "xq_policy.hip")],
                          stderr=subprocess.DEVNULL)
    ptext = open(ppath).read()
    for name in sorted(set(re.findall(r"^(_Z\w*k_policy_fc1w\w*):", ptext, re.M))):
        n_dma, badm = m0_after_big_statement(ptext, name.lstrip("_ZN"))
        print("%s: %d compiler-issued LDS-DMA behind the K-loop statement, %d without an m0 write in front" % (name, n_dma, len(badm)))
        if badm:
            rc = 1
sys.exit(rc)
