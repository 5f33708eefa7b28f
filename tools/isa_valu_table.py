"""Vector-issue cycles of ONE key tile of an attention kernel, by class, read off the ISA hipcc emits.
    python3 tools/isa_valu_table.py [--defs "-DTRX_ATT_AUG=0"] [--kernel attention_fwd_mfma_kernelILi1ELb0] > table.json
Compiles textreact_amd/csrc/nn_ops.hip to assembly (device only), takes the named kernel, finds the key-tile loop body -- from
the block that issues the first product's MFMAs to the block after the second product's -- and prices the instructions on the
COMMON path (blocks reached only through a branch that is rarely taken -- hidden keys of a tail / causal tile, a moved
reference -- are listed apart) with the issue costs of /opt/skills/guides/MI355X_MICROARCH.md ("vector-instruction ISSUE cost,
one wave's stream on one SIMD"): transcendental 8, v_pk_*_f32 8 (two passes), other VALU 4, v_cvt_pk_bf16_f32 4.5, an MFMA holds
the vector issue for 8 of its cycles, s_nop N counts N + 1 and LDS / scalar instructions nothing (other ports)."""
import argparse, collections, json, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--defs", default="")
ap.add_argument("--kernel", default="attention_fwd_mfma_kernelILi1ELb0")
ap.add_argument("--asm", default=None, help="an assembly file made earlier instead of compiling")
args = ap.parse_args()
if args.asm:
    text = open(args.asm).read()
else:
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "nn.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-w"] + args.defs.split() +
                       ["-o", out, os.path.join(ROOT, "textreact_amd", "csrc", "nn_ops.hip")], check=True)
        text = open(out).read()
lines = text.splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % re.escape(args.kernel), l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
body = lines[start:end]
# basic blocks
blocks, cur, name = collections.OrderedDict(), [], "entry"
for l in body:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks[name] = cur; name, cur = m.group(1), []
        continue
    t = l.split(";")[0].strip()
    if t and not t.startswith("."):
        cur.append(t)
blocks[name] = cur
names = list(blocks)
mf = [n for n in names if sum(1 for i in blocks[n] if i.startswith("v_mfma")) >= 8]
first = names.index(mf[0]); last = names.index(mf[-1])
# the tile body: from the first-product block through the second-product block.  A block is "rare" when the block before it
# ends in a conditional branch that jumps OVER it (the compiler lays the unlikely side out in line)
def cost(ins):
    op = ins.split()[0]
    if op.startswith("v_mfma"): return "mfma (8 of its cycles hold the vector issue)", 8
    if op in ("v_exp_f32_e32", "v_exp_f32_e64", "v_log_f32_e32", "v_rcp_f32_e32"): return "transcendental (v_exp_f32)", 8
    if op.startswith("v_pk_fma_f32") or op.startswith("v_pk_mul_f32") or op.startswith("v_pk_add_f32"): return "packed f32 (v_pk_fma / v_pk_mul / v_pk_add: two passes)", 8
    if op.startswith("v_cvt_pk_bf16"): return "convert to bf16 pairs (v_cvt_pk_bf16_f32)", 4.5
    if op.startswith("v_dot2c"): return "row sum (v_dot2c_f32_bf16)", 4
    if op.startswith("v_max3") or op.startswith("v_max_f32") or op.startswith("v_permlane"): return "row maximum (v_max3 / v_max / permlane)", 4
    if op == "s_nop": return "s_nop (hazard padding)", int(ins.split()[1]) + 1
    if op.startswith("v_"): return "other VALU (reference bookkeeping, addresses, compares, moves)", 4
    if op.startswith("ds_"): return "LDS instructions (own port: not priced)", 0
    return "scalar / branch / wait (own port: not priced)", 0
common, rare = collections.OrderedDict(), collections.OrderedDict()
def add(tab, ins):
    cls, c = cost(ins)
    e = tab.setdefault(cls, {"instructions": 0, "issue_cycles": 0.0})
    e["instructions"] += 1; e["issue_cycles"] += c
# A forward conditional branch whose target still lies inside the tile jumps OVER the unlikely side (hipcc lays it out in line):
# what follows it, up to its target, is priced apart.  (The branch around the whole tile -- a key-split wave skipping a tile
# that is not its own -- targets a label past the tile and is not one of these.)
skip_to = None
for bi in range(first, last + 1):
    n = names[bi]
    if skip_to == n: skip_to = None
    for ins in blocks[n]:
        add(rare if skip_to else common, ins)
        m = re.match(r"s_cbranch_(\w+)\s+(\.LBB\d+_\d+)", ins)
        if m and skip_to is None and m.group(2) in names and bi < names.index(m.group(2)) <= last:
            skip_to = m.group(2)
tot = lambda t: sum(e["issue_cycles"] for e in t.values())
print(json.dumps({"kernel": args.kernel, "defs": args.defs, "what": __doc__.split("\n")[0],
                  "tile": "64 keys x 32 queries per wave = 32 score elements per lane", "blocks": names[first:last + 1],
                  "common_path": common, "common_path_vector_issue_cycles": tot(common),
                  "rarely_taken_blocks": rare, "rarely_taken_vector_issue_cycles": tot(rare)}, indent=1))
