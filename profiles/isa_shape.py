"""Instruction-class 'shape' of one kernel in a hipcc -S listing (M=MFMA R/W=LDS read/write G=global S=scratch
v=VALU a=accvgpr move s=SALU w=waitcnt n=nop |=barrier B=branch).  usage: isa_shape.py file.s kernel-substring"""
import sys


def cls(l):
    l = l.strip()
    for p, c in (("v_mfma", "M"), ("ds_read", "R"), ("ds_load", "R"), ("ds_write", "W"), ("ds_store", "W"),
                 ("ds_bpermute", "P"), ("s_waitcnt", "w"), ("s_nop", "n"), ("v_accvgpr", "a"), ("v_", "v"),
                 ("s_cbranch", "B"), ("s_branch", "B"), ("s_barrier", "|"), ("s_", "s"), ("scratch_", "S"),
                 ("global_", "G"), ("buffer_", "G")):
        if l.startswith(p):
            return c
    if l.endswith(":") and l.startswith(".LBB"):
        return "\n" + l + " "
    return ""


lines = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(lines) if sys.argv[2] in l and l.split(":")[0].strip().startswith("_Z") and l.rstrip().endswith(tuple(": ; @" + w for w in [""])) or (sys.argv[2] in l and l.startswith("_Z") and ": " in l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
print("".join(cls(l) for l in lines[start:end]))
